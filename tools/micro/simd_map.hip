// Which SIMD does wave k of a workgroup run on?  (tools only; round 6, K7's per-wave work balance)
// Launches many 256-thread workgroups with K7's LDS footprint and records HW_ID per wave: if wave k always lands on SIMD k, the per-wave
// instruction counts of a kernel whose phases use different numbers of waves ARE per-SIMD loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;      // keep the workgroup resident for a while so that CUs fill up
    if (v == 12345.f) smem[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main() {
    const int WG = 19200;
    unsigned* d;
    hipMalloc(&d, WG * 4 * 2 * 4);
    for (int lds : {0, 20000}) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(probe, dim3(WG), dim3(256), lds, 0, d, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(WG * 8);
        hipMemcpy(h.data(), d, WG * 32, hipMemcpyDeviceToHost);
        long hist[4][4] = {};
        long same_rot = 0;
        for (int b = 0; b < WG; ++b) {
            int s0 = (h[(b * 4) * 2] >> 4) & 3;
            bool rot = true;
            for (int w = 0; w < 4; ++w) {
                int s = (h[(b * 4 + w) * 2] >> 4) & 3;
                hist[w][s]++;
                if (s != ((s0 + w) & 3)) rot = false;
            }
            same_rot += rot;
        }
        printf("LDS %d B per workgroup, %d workgroups of 4 waves: rows = wave index in the workgroup, columns = SIMD id (HW_ID[5:4])\n", lds, WG);
        for (int w = 0; w < 4; ++w) printf("  wave %d: %6ld %6ld %6ld %6ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("  workgroups whose waves sit on SIMDs s0, s0+1, s0+2, s0+3 (mod 4): %ld of %d; first 8 workgroups' (simd of wave 0..3): ", same_rot, WG);
        for (int b = 0; b < 8; ++b) printf("[%u%u%u%u] ", (h[b * 8] >> 4) & 3, (h[b * 8 + 2] >> 4) & 3, (h[b * 8 + 4] >> 4) & 3, (h[b * 8 + 6] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
