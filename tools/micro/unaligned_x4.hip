// Micro-probe (tools only): are 16-byte global loads from 4-byte-aligned addresses legal on gfx950, (a) into VGPRs, (b) as LDS-DMA?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k_vgpr(const float* src, float* dst, int shift) {
    const f4u v = *reinterpret_cast<const f4u*>(src + shift + threadIdx.x * 5);   // stride 20 B: every alignment class
    for (int j = 0; j < 4; ++j) dst[threadIdx.x * 4 + j] = v[j];
}
__global__ void k_dma(const float* src, float* dst, int shift) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __builtin_amdgcn_global_load_lds((gptr_t)(src + shift + threadIdx.x * 5), (lptr_t)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) dst[threadIdx.x * 4 + j] = reinterpret_cast<float*>(lds)[threadIdx.x * 4 + j];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, n * 4); hipMalloc(&d, 64 * 4 * 4);
    hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> o(256);
    for (int mode = 0; mode < 2; ++mode)
        for (int shift = 0; shift < 4; ++shift) {
            hipMemset(d, 0xff, 1024);
            if (mode == 0) hipLaunchKernelGGL(k_vgpr, dim3(1), dim3(64), 0, 0, s, d, shift);
            else hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 1024, 0, s, d, shift);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int t = 0; t < 64; ++t) for (int j = 0; j < 4; ++j) bad += o[t * 4 + j] != (float)(shift + t * 5 + j);
            printf("%s shift %d: err=%d mismatches=%d\n", mode ? "lds-dma x4" : "vgpr x4   ", shift, (int)e, bad);
        }
    return 0;
}
