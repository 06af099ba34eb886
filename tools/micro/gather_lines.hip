// Tools only: what HBM delivers to a gather of the look-up's shape -- the ceiling K7 (corr_lookup_tile_kernel) is measured against at the
// C4 shard (batch 8: 38 400 private planes of 60 x 80 + their pyramid, 3.1 GB; nothing is ever re-read, every line comes from DRAM).
// A "patch" = RUNS runs of RUNLEN consecutive 128-B lines (a 12 x 16 window on 4 x 8-element tiles touches 3-4 tile rows x 2-3 tiles =
// 3.75 x 2.5 lines on average: 3 runs of 3 lines, or 4 x 3), the runs one tile row (TW lines) apart, at a pseudo-random plane of the buffer.
// Every lane moves 16 B by LDS-DMA (global_load_lds_dwordx4: 8 lanes per line, as the kernel does); a 256-thread workgroup keeps
// PATCHES_WG patches (= 14 pairs of the kernel: 7 planes x 2 pixels) in flight, waits for them (s_waitcnt vmcnt(0) + barrier), and takes the next
// -- 8 workgroups per CU like the kernel (20 KB of LDS each).  Modes: stream (consecutive lines, the DRAM-friendly upper bound), runs 3x3, runs
// 4x3, single random lines.  Prints TB/s of LINE bytes.
//   hipcc --offload-arch=gfx950 -O3 -o gather_lines gather_lines.hip && ./gather_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// plane_lines: lines of one private plane (60 x 80 fp32 tiled = 150 lines); n_planes planes in the buffer; TW: lines per tile row (10)
template <int RUNS, int RUNLEN, bool STREAM>
__global__ __launch_bounds__(256) void gather_kernel(const char* src, unsigned n_planes, int plane_lines, int tw, int wg_iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int PATCHES_WG = 14;
    constexpr int UNITS = PATCHES_WG * RUNS * RUNLEN * 8;          // 16-B units per workgroup round
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int it = 0; it < wg_iters; ++it) {
        const unsigned round = (unsigned)it * gridDim.x + blockIdx.x;
        for (int u0 = wave * 64; u0 < UNITS; u0 += 256) {
            const int u = min(u0 + lane, UNITS - 1);
            const int patch = u / (RUNS * RUNLEN * 8), r = u - patch * (RUNS * RUNLEN * 8);
            const int run = r / (RUNLEN * 8), lu = r - run * (RUNLEN * 8);        // unit inside the run: line lu / 8, 16-B piece lu % 8
            size_t line;
            if (STREAM) {
                line = ((size_t)round * PATCHES_WG + patch) * (RUNS * RUNLEN) + run * RUNLEN + (lu >> 3);
                line %= (size_t)n_planes * plane_lines;
            } else {
                const unsigned h = hash32(round * PATCHES_WG + patch + 0x9e3779b9u);
                const unsigned plane = h % n_planes;
                const int rows = plane_lines / tw;
                const int ty = (int)((h >> 8) % (unsigned)max(rows - RUNS + 1, 1)), tx = (int)((h >> 20) % (unsigned)max(tw - RUNLEN + 1, 1));
                line = (size_t)plane * plane_lines + (size_t)(ty + run) * tw + tx + (lu >> 3);
            }
            const char* p = src + line * 128 + (lu & 7) * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + u0 * 16), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (lds[tid] == 0x7f && sink) sink[0] = 1;
}

template <int RUNS, int RUNLEN, bool STREAM>
static void run(const char* name, const char* src, unsigned n_planes, int* sink) {
    constexpr int PATCHES_WG = 14;
    const int lds = PATCHES_WG * RUNS * RUNLEN * 128 + 1024;
    const int grid = 256 * 8, wg_iters = 75;                       // ~ the kernel's 19 200 workgroups of 14 pairs at the C4 shard, 8 per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel<RUNS, RUNLEN, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((gather_kernel<RUNS, RUNLEN, STREAM>), dim3(grid), dim3(256), lds, 0, src, n_planes, 150, 10, wg_iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)grid * wg_iters * PATCHES_WG * RUNS * RUNLEN * 128;
    printf("%-46s %7.1f us for %6.1f MB of lines: %5.2f TB/s (best of 5; median %5.2f)\n", name, ts[0] * 1e3, bytes / 1e6, bytes / (ts[0] * 1e-3) / 1e12,
           bytes / (ts[2] * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const unsigned n_planes = 8u * 4800u * 4u;                     // level-0 planes of the C4 shard (batch 8, 4 targets): 2.95 GB
    const size_t bytes = (size_t)n_planes * 150 * 128;
    char* src; int* sink;
    if (hipMalloc(&src, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(src, 1, bytes);
    hipDeviceSynchronize();
    run<3, 3, true>("stream (consecutive lines, 9 per patch)", src, n_planes, sink);
    run<3, 3, false>("3 runs x 3 lines per patch, random planes", src, n_planes, sink);
    run<4, 3, false>("4 runs x 3 lines per patch, random planes", src, n_planes, sink);
    run<3, 2, false>("3 runs x 2 lines per patch, random planes", src, n_planes, sink);
    run<9, 1, false>("9 single lines per patch, one tile row apart", src, n_planes, sink);
    hipFree(src); hipFree(sink);
    return 0;
}
