// Tools only (not part of the product library): what bounds a pure STORE stream shaped like K5's (corr_stream.hip)?
// Writes the C2 volume (4 slabs of 4800 x 4800 fp32 = 368.6 MB) with no arithmetic, in several shapes:
//   linear      grid-stride fill, 4 / 16 B per lane                                        (the ceiling of the write path)
//   k5          256 persistent 8-wave workgroups; item = 32 rows x 1 KB (8 waves x one 128-B line), rows PS*4 bytes apart;
//               items handed out exactly as corr_stream_kernel does (XCD grid RX x PX, panel-major list, equal contiguous ranges)
//   lock        the same items in (slab, chunk, panel) order, workgroup w takes items w, w+256, ...: simultaneously running workgroups
//               write ADJACENT panels of the SAME rows (best-case DRAM-page locality for this tile shape)
// each with per-lane store width 4 B (one register = two full lines per wave instruction, what K5 issues today) or 16 B (lane = 16 B of a
// line, 8 full lines per wave instruction, what an LDS-transposed epilogue would issue), and aux = 0 / nt.
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int N = 4800, PS = 4800, SLABS = 4, JP = 19, CHUNKS = 150;

__global__ void fill_linear4(float* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0f;
}
__global__ void fill_linear16(float4* out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1, 2, 3, 4);
}

// MODE 0 = k5 order, 1 = lock order.  W = 4 | 16 bytes per lane.  AUX = buffer aux bits.
template <int MODE, int W, int AUX>
__global__ __launch_bounds__(512) void fill_k5(float* out, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    int n_items, first, stride;
    int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    // k5 order: XCD grid 2 x 4
    const int RX = 2, PX = 4, CH2 = 75, n_pan = SLABS * JP;
    const int rx = xcd % RX, px = xcd / RX;
    const int c_lo = 2 * (rx * CH2 / RX), c_hi = 2 * ((rx + 1) * CH2 / RX);
    const int p_lo = px * n_pan / PX, p_hi = (px + 1) * n_pan / PX;
    const int len = c_hi - c_lo;
    const long long pairs = (long long)(p_hi - p_lo) * (len >> 1);
    const int start = 2 * (int)(pairs * idx / 32), end = 2 * (int)(pairs * (idx + 1) / 32);
    if (MODE == 0) { n_items = end - start; first = start; stride = 1; }
    else { const int tot = SLABS * CHUNKS * JP; first = blockIdx.x; stride = gridDim.x; n_items = (tot - first + stride - 1) / stride; }
    for (int rep = 0; rep < reps; ++rep)
    for (int k = 0; k < n_items; ++k) {
        int slab, jp, c;
        if (MODE == 0) {
            const int it = first + k;
            const int pan = p_lo + it / len;
            c = c_lo + it % len;
            slab = pan / JP; jp = pan % JP;
        } else {
            const int it = first + k * stride;
            jp = it % JP; c = (it / JP) % CHUNKS; slab = it / (JP * CHUNKS);
        }
        float* sl = out + (size_t)slab * N * PS;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sl, 0, N * PS * 4, 0x00020000);
        if (W == 4) {
            const int col = jp * 256 + wave * 32 + l31;
            const unsigned base = col < PS ? (unsigned)(((c * 32 + 4 * kh) * PS + col) * 4) : 0x80000000u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(r + k)), rs, base + (unsigned)((r & 3) + 8 * (r >> 2)) * PS * 4, 0, AUX);
        } else {
            const int col = jp * 256 + wave * 32 + (lane & 7) * 4;
            const unsigned base = col < PS ? (unsigned)(((c * 32 + (lane >> 3)) * PS + col) * 4) : 0x80000000u;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u4 v = {(unsigned)k, (unsigned)r, 3u, 4u};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (unsigned)(8 * r) * PS * 4, 0, AUX);
            }
        }
    }
}

template <typename F>
static float time_it(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const size_t n = (size_t)SLABS * N * PS;
    float* out;
    hipMalloc(&out, n * 4 + 4096);
    const double mb = n * 4 / 1e6;
    auto rep = [&](const char* name, float ms) { printf("%-44s %8.1f us  %7.0f GB/s  (%.3f of 8 TB/s)\n", name, ms * 1e3, mb / ms, mb / ms / 8000.0); fflush(stdout); };
    for (int g : {1024, 2048, 8192}) {
        char nm[64];
        snprintf(nm, 64, "linear 4 B/lane, %d x 256", g);
        rep(nm, time_it([&] { hipLaunchKernelGGL(fill_linear4, dim3(g), dim3(256), 0, 0, out, n); }));
        snprintf(nm, 64, "linear 16 B/lane, %d x 256", g);
        rep(nm, time_it([&] { hipLaunchKernelGGL(fill_linear16, dim3(g), dim3(256), 0, 0, (float4*)out, n / 4); }));
    }
    rep("hipMemsetAsync", time_it([&] { hipMemsetAsync(out, 0, n * 4, 0); }));
#define RUN(M, W, A, NAME) rep(NAME, time_it([&] { hipLaunchKernelGGL((fill_k5<M, W, A>), dim3(256), dim3(512), 0, 0, out, 1); }));
    RUN(0, 4, 0, "k5 order,   4 B/lane (2 lines / instr)")
    RUN(0, 16, 0, "k5 order,  16 B/lane (8 lines / instr)")
    RUN(0, 4, 2, "k5 order,   4 B/lane, nt")
    RUN(0, 16, 2, "k5 order,  16 B/lane, nt")
    RUN(1, 4, 0, "lock order, 4 B/lane")
    RUN(1, 16, 0, "lock order, 16 B/lane")
    RUN(1, 4, 2, "lock order, 4 B/lane, nt")
    RUN(1, 16, 2, "lock order, 16 B/lane, nt")
    // more workgroups per CU (512 x 512 threads): does issue parallelism matter?
    rep("lock order, 16 B/lane, 512 wgs", time_it([&] { hipLaunchKernelGGL((fill_k5<1, 16, 0>), dim3(512), dim3(512), 0, 0, out, 1); }));
    rep("lock order, 4 B/lane, 512 wgs", time_it([&] { hipLaunchKernelGGL((fill_k5<1, 4, 0>), dim3(512), dim3(512), 0, 0, out, 1); }));
    hipFree(out);
    return 0;
}
