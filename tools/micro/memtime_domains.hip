// Is s_memtime one counter per chip, per XCD, per shader engine or per CU?  (tools only; round 6: bflow_shader_clock_stamp must compare two
// readings of the SAME counter.)  2048 one-wave workgroups sample (HW_ID, XCC_ID, s_memtime, s_memrealtime) within a few microseconds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ void k(unsigned long long* out) {
    unsigned hw, xcc; unsigned long long cyc, rt;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc), "=s"(rt));
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = cyc; out[blockIdx.x * 4 + 3] = rt; }
}
int main() {
    const int WG = 2048;
    unsigned long long* d; hipMalloc(&d, WG * 32);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(WG), dim3(64), 0, 0, d); hipDeviceSynchronize();
        std::vector<unsigned long long> h(WG * 4); hipMemcpy(h.data(), d, WG * 32, hipMemcpyDeviceToHost);
        unsigned long long rt0 = ~0ull; for (int i = 0; i < WG; ++i) rt0 = std::min(rt0, h[i * 4 + 3]);
        // cycles minus 24 x ticks (2.4 GHz / 100 MHz): constant within a few thousand for one counter sampled within microseconds
        std::map<int, std::pair<long long, long long>> byx, byse, bycu;
        auto upd = [](std::map<int, std::pair<long long, long long>>& m, int key, long long v) { auto it = m.find(key); if (it == m.end()) m[key] = {v, v}; else { it->second.first = std::min(it->second.first, v); it->second.second = std::max(it->second.second, v); } };
        long long gmin = 0; bool first = true;
        for (int i = 0; i < WG; ++i) { long long v = (long long)h[i * 4 + 2] - 24 * (long long)(h[i * 4 + 3] - rt0); if (first || v < gmin) gmin = v; first = false; }
        for (int i = 0; i < WG; ++i) {
            unsigned hw = (unsigned)h[i * 4], xcc = (unsigned)h[i * 4 + 1] & 0xf;
            int se = (hw >> 13) & 7, cu = (hw >> 8) & 15;
            long long v = (long long)h[i * 4 + 2] - 24 * (long long)(h[i * 4 + 3] - rt0) - gmin;
            upd(byx, xcc, v); upd(byse, xcc * 8 + se, v); upd(bycu, (xcc * 8 + se) * 16 + cu, v);
        }
        printf("launch %d: s_memtime - 24 x s_memrealtime (relative to the smallest), [min, max] per group; the launch spans %llu ticks of 10 ns\n", rep, [&]{unsigned long long m=0; for (int i=0;i<WG;++i) m=std::max(m,h[i*4+3]-rt0); return m;}());
        printf("  per XCD:"); for (auto& e : byx) printf("  x%d [%lld, %lld]", e.first, e.second.first, e.second.second); printf("\n");
        printf("  per (XCD, SE), first 12:"); int c = 0; for (auto& e : byse) { if (c++ < 12) printf("  x%d.se%d [%lld, %lld]", e.first / 8, e.first % 8, e.second.first, e.second.second); } printf("\n");
        long long worst = 0; for (auto& e : bycu) worst = std::max(worst, e.second.second - e.second.first);
        printf("  %zu (XCD, SE, CU) groups; largest spread inside one CU %lld cycles\n", bycu.size(), worst);
    }
    return 0;
}
