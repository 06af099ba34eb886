// Debug-only (tools/): one-thread kernel that writes the 100 MHz wall clock into a slot -- timeline of a hipGraph replay
// without the profiler.  Not part of the product library.
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
extern "C" int stamp(unsigned long long* slot, void* stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    return (int)hipGetLastError();
}

// MFMA issue-rate probe: every wave runs `iters` x 8 independent v_mfma_f32_32x32x16_f16 (8 accumulators); waves_per_simd waves
// per SIMD on every CU.  out[0] = wall-clock ticks (100 MHz) of block 0, out[1] = shader cycles (s_memtime) of block 0.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void mfma_rate_kernel(unsigned long long* out, int iters, float* sink) {
    half8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x16_t c0, c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    const unsigned long long w0 = wall_clock64(), t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long w1 = wall_clock64(), t1 = clock64();
    const float s = c0[0] + c1[0] + c2[0] + c3[0];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = w1 - w0; out[1] = t1 - t0; }
}
extern "C" int mfma_rate(unsigned long long* out, int iters, int blocks, int threads, float* sink, void* stream) {
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters, sink);
    return (int)hipGetLastError();
}
