// Tools only: what a CU can pull from its XCD's L2 per clock, by path -- the number that bounds the small-grid convolution kernels
// (conv_halo8 / conv_halo10 stage 20 KB per 3-tap step through LDS-DMA: 0.45 us per step against 0.3 us of matrix work).
//   dma     buffer_load ... lds (16 B / lane, 1 KB per wave instruction), the path of every staged tile
//   vgpr    buffer_load_dwordx4 into registers (what a direct-to-fragment weight load would use)
//   both    half of the waves each
// Every workgroup (8 waves, one per CU) re-reads its own 96-KB slice (L2-resident: 3 MB per XCD); `inflight` loads per wave between
// waits.  Prints bytes per clock per CU from the shader clock (s_memtime) of the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o load_paths load_paths.hip && ./load_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT>   // MODE 0 dma, 1 vgpr, 2 both (even waves dma, odd waves vgpr)
__global__ __launch_bounds__(512) void pull_kernel(const char* src, int slice_bytes, int iters, unsigned long long* out, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * slice_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, slice_bytes, 0x00020000);
    const bool dma = MODE == 0 || (MODE == 2 && !(wave & 1));
    i32x4 acc = {0, 0, 0, 0};
    const int per_wave = slice_bytes / 8;                 // bytes of the slice this wave walks
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        for (int off = 0; off < per_wave; off += INFLIGHT * 1024) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                const unsigned o = (unsigned)(wave * per_wave + off + j * 1024 + lane * 16);
                if (dma) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + wave * 16384 + (j & 15) * 1024), 16, o, 0, 0, 0);
                } else {
                    const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
                    acc[0] ^= v[0]; acc[1] += v[1]; acc[2] ^= v[2]; acc[3] += v[3];
                }
            }
            if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[0] = acc[0];
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int MODE, int INFLIGHT>
static void run(const char* name, const char* src, unsigned long long* out, int* sink) {
    const int slice = 96 * 1024, iters = 40;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((pull_kernel<MODE, INFLIGHT>), dim3(256), dim3(512), 8 * 16384, 0, src, slice, iters, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < 256; ++i) { cyc += h[2 * i] / 256.0; wall += h[2 * i + 1] / 256.0; }
    const double bytes = (double)slice * iters;
    printf("%-34s in flight %2d: %6.1f B/clk/CU  (%.2f GHz, %.1f TB/s chip-wide)\n", name, INFLIGHT, bytes / cyc, cyc / (wall / 1e8) / 1e9, bytes * 256 / (wall / 1e8) / 1e12);
    fflush(stdout);
}

int main() {
    char* src;
    hipMalloc(&src, 256 * 96 * 1024);
    hipMemset(src, 1, 256 * 96 * 1024);
    unsigned long long* out;
    int* sink;
    hipMalloc(&out, 512 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(pull_kernel<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    run<0, 2>("LDS-DMA (buffer_load ... lds)", src, out, sink);
    run<0, 4>("LDS-DMA (buffer_load ... lds)", src, out, sink);
    run<0, 12>("LDS-DMA (buffer_load ... lds)", src, out, sink);
    run<1, 2>("VGPR (buffer_load_dwordx4)", src, out, sink);
    run<1, 4>("VGPR (buffer_load_dwordx4)", src, out, sink);
    run<1, 12>("VGPR (buffer_load_dwordx4)", src, out, sink);
    run<2, 4>("both (4 waves each)", src, out, sink);
    run<2, 12>("both (4 waves each)", src, out, sink);
    return 0;
}
