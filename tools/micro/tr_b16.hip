// Micro-probe (tools only): what does ds_read_b64_tr_b16 return?
// RESULT (MI355X): inside every group of 16 lanes, with M[i][e] = halfword e of the 8 bytes at lane i's address,
//     out[l][j] = M[4 j + (l >> 2)][l & 3]
// -- i.e. to give lane (m = l & 15 (+16), k-group) the four k-consecutive elements X[p0 + j][m] of a row-major [pixel][32 channels]
// (64-B rows) LDS tile, fetch lane i reads row p0 + (i >> 2), 8-B chunk (i & 3) of the 32-B half row that holds channel m: the
// fragment of a pixel-contraction (weight-gradient) MFMA straight from the forward's activation layout.  LDS holds u16 element i at halfword i; lane l supplies byte address
// addr(l); every lane prints the four halfwords it received.  Patterns: (a) all lanes address 0 + lane-dependent row pitch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x];
    short4v v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int* da; unsigned short* dout;
    hipMalloc(&da, 256); hipMalloc(&dout, 512);
    std::vector<int> a(64); std::vector<unsigned short> o(256);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: lane l -> row (l & 15) of a 64-B-pitch matrix, 8-B column block (l >> 4)      [row-major [16 rows][32 halfwords]]
        // mode 1: lane l -> byte address 8 * l (consecutive 8-B pieces)
        // mode 2: lane l -> row (l & 15) pitch 64 B, column block 0 for all
        for (int l = 0; l < 64; ++l) a[l] = mode == 0 ? (l & 15) * 64 + (l >> 4) * 8 : mode == 1 ? 8 * l : (l & 15) * 64;
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("mode %d (halfword indices received; lane: addr/2 -> 4 values)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  l%2d a%4d: %4d %4d %4d %4d%s", l, a[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
