// K7 + K9a fused: correlation look-up and the 1x1 convolution that consumes it (BasicMotionEncoder.convc1 + ReLU) in ONE launch.
// Reference: CorrBlockParallelMultiTarget.__call__ (models/raft_utils/corr.py:307-351), bilinear_sampler (models/raft_utils/utils.py:5-21),
// BezierCurves.get_flow_from_reference + coords0 (models/raft_spline/raft.py:180-184), cor = relu(convc1(corr)) (models/raft_spline/update.py:88).
//
// Why.  At batch 1 an update iteration is a chain of dependent launches of 12-18 us on a 60 x 80 grid; look-up (13 us: a latency chain
// parameters -> gather -> interpolate -> store) and convc1 (14 us for 3 us of matrix work: 152 workgroups staging 128-pixel activation tiles
// and the weights through LDS) are two of them, with the (B, 576, N) look-up features written to memory and read back in between.  Here a
// workgroup owns TP <= 28 consecutive query pixels x ALL planes (<= 8) x ALL output channels (<= 256):
//   * phase A, gather, tap tables and interpolation are the tile look-up of corr_lookup_tile.hip (same arithmetic, same order: the features
//     are bit-identical); the planes are gathered in two passes of four (58 KB of patches at 19 pixels), and the interpolation writes the
//     split-fp16 features straight into an LDS operand tile [hi | lo][k-block][TP pixels][32 channels];
//   * the 1x1 convolution is D[channel][pixel] = W x features on the matrix cores: wave w owns output channels 32 w .. 32 w + 31, and its
//     WEIGHT fragments (private to the wave: 4 x 16 B per lane and 32-channel k-block) are requested straight into REGISTERS: the k-blocks
//     the first pass completes at the start of the kernel, those of the second pass while the MFMAs of the first run (refilling its
//     registers), under the second gather -- the matrix phases read only LDS;
//   * the epilogue is the conv engine's arithmetic (bias, activation, hi / lo split, memory-order stores through a per-wave LDS slab).
// One workgroup per CU (512 threads, 2 waves per SIMD, <= 256 VGPRs each), TP chosen so that the grid is a whole number of rounds.
// Status: parity-green; 26.0 us against 13.3 + 14.2 us for the two launches alone at DSEC size, but 2-5 us per
// iteration SLOWER inside the captured forward (it shares the chip with nothing; the two launches run next to the Bezier branch), so the
// model uses it only on request (BFLOW_LOOKUP_CONV=1).  Several notes below are about hipcc's wait-count bookkeeping around LDS-DMA.
#include "conv_engine.h"

namespace {

constexpr int R = BFLOW_LOOKUP_RADIUS;      // 4
constexpr int WIN = 2 * R + 1;              // 9
constexpr int NCH = WIN * WIN;              // 81
constexpr int PATCH = 12;                   // rows floor(c)-5 .. floor(c)+6 cover every bilinear corner incl. round-off flips
[[maybe_unused]] constexpr int TILE_H = 4, TILE_W = 8;       // plane tiling (elements)
[[maybe_unused]] constexpr int EPU = 4, UPR = 4, PCOLS = 16, UPP = PATCH * UPR, PELEMS = PATCH * PCOLS;   // fp32 planes: see corr_lookup_tile.hip
constexpr int LC_THREADS = 512, LC_WAVES = 8;
constexpr int LC_MAX_TP = 28;

struct LcPlane {
    const void* base;
    int h, w;
    float inv_scale;
    int target;
};

#ifdef LC_STAMPS   // tools/lookup_conv_stamps.sh build: s_memtime stamps of every wave at the phase boundaries (24 x u64 per wave)
static unsigned long long* g_lc_stamp_buf = nullptr;
extern "C" __attribute__((visibility("default"))) void bflow_lookup_conv_set_stamp_buffer(void* p) { g_lc_stamp_buf = (unsigned long long*)p; }
#define STAMP(i) \
    if (la.stamps && lane == 0) la.stamps[((blockIdx.y * gridDim.x + blockIdx.x) * LC_WAVES + wave) * 24 + (i)] = __builtin_readcyclecounter();
#define STAMP_RT(i) \
    if (la.stamps && lane == 0) la.stamps[((blockIdx.y * gridDim.x + blockIdx.x) * LC_WAVES + wave) * 24 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define STAMP(i)
#define STAMP_RT(i)
#endif

struct LcArgs {
    LcPlane planes[BFLOW_MAX_PLANES];
    float pcoef[BFLOW_MAX_PLANES * BFLOW_MAX_DEGREE];   // time coefficients of each plane's target: row p = coef[planes[p].target]
    const float* params;
    int P, T, deg, TP, h1, w1;
    unsigned long long* stamps;   // LC_STAMPS builds only
};

__device__ __forceinline__ float roundtrip(float x, int size) {   // utils.py:13-14 + grid_sample's un-normalisation (corr_lookup.hip)
    const float sm1 = (float)(size - 1);
    const float g = 2.0f * x / sm1 - 1.0f;
    return (g + 1.0f) * (sm1 / 2.0f);
}

__device__ __forceinline__ int tiled_index(int y, int x, int tw) {
    return (((y >> 2) * tw + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7);
}

// LDS bytes of one launch (host and device agree through these functions).  P1 = planes gathered in the first pass (<= 4).
constexpr int LC_PASS_PLANES = 4;
constexpr int LC_SLAB_BYTES = LC_WAVES * 32 * CONV_STG_STRIDE * 4;     // the epilogue's per-wave transposition slabs
// tables, per pair: gather record (32 B), 18 tap records of 16 B, sampling centre + plane size (16 B)
__host__ __device__ inline int lc_tables_bytes(int npair) { return npair * (32 + 18 * 16 + 16); }
__host__ __device__ inline int lc_tile_bytes(int nkb, int tp) { return 2 * nkb * tp * 64; }
__host__ __device__ inline int lc_work_bytes(int nkb, int tp, int npair1) {   // operand tile + patches of one pass (+ 1 KB: idle lanes of the last
    const int w = lc_tile_bytes(nkb, tp) + npair1 * PELEMS * 4 + 1024;        // DMA instruction); the epilogue's slabs overlay both
    return w > LC_SLAB_BYTES ? w : LC_SLAB_BYTES;
}

// LDS stores the compiler does not see as memory operations.  hipcc makes every LDS STORE that follows an LDS-DMA wait for the DMA (it may
// alias), and in front of a LOOP with such a store it drains the whole counter (vmcnt(0)): the tap tables would wait for the gather they are
// meant to overlap, and the interpolation for the weight fragments that fly behind the gather.  The stores below go to regions no DMA
// touches; their completion is covered by the explicit `s_waitcnt lgkmcnt(0)` in front of each barrier.
// (The same bookkeeping decides by the access TYPE whether a plain LDS load or store waits for an LDS-DMA in flight: accesses that carry
// type-based alias information -- scalars, ext_vector types -- do not; HIP's struct vectors (float2 / float4) and struct copies, which are
// lowered without it, DO.  Every LDS access of this kernel that can execute while a gather flies therefore uses the typedefs below.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_store_b128(void* p, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(lptr_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_store_b16(void* p, _Float16 v) {
    asm volatile("ds_write_b16 %0, %1" ::"v"((unsigned)(size_t)(lptr_t)p), "v"((unsigned)__builtin_bit_cast(unsigned short, v)) : "memory");
}

struct LcRec {          // gather record of one (plane, pixel) pair
    long long base;     // byte address of the pair's tiled plane
    int oy, ox;         // patch origin (x aligned down to a 16-B unit)
    int ylim, xlim;     // extent of the tile grid (pads included)
    int tw;             // tiles per tile row
    int pi;             // plane | pixel of the tile << 8
};

// NKB1 k-blocks (32 feature channels each) are complete after the first gather pass (planes 0 .. 3), NKB2 more after the second (planes 4 .. 7).
template <int NKB1, int NKB2>
__global__ __launch_bounds__(LC_THREADS) void lookup_conv_kernel(LcArgs la, ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NKB = NKB1 + NKB2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int N = la.h1 * la.w1, P = la.P, TP = la.TP;
    const int P1 = P < LC_PASS_PLANES ? P : LC_PASS_PLANES;
    const int b = blockIdx.y;
    const int np0 = blockIdx.x * TP;
    const int npair = P * TP;                 // pair = plane * TP + pixel of the tile
    LcRec* const s_rec = reinterpret_cast<LcRec*>(smem);                           // [npair]
    f32x4* const s_tap = reinterpret_cast<f32x4*>(smem + npair * 32);              // [npair][18]: (patch-relative near index, near weight, far
                                                                                   // weight, -); weights ZERO outside the plane / the patch
    f32x4* const s_c = reinterpret_cast<f32x4*>(smem + npair * (32 + 18 * 16));    // [npair]: sampling centre x, y, plane width, height (ints)
    char* const tile = smem + lc_tables_bytes(npair);                       // [hi | lo][NKB][TP pixels][64 B], 16-B chunks XOR-swizzled
    const int TKB = TP * 64;                                                                        // bytes of one k-block of one plane
    const int TPLANE = NKB * TKB;
    float* const patch = reinterpret_cast<float*>(tile + 2 * TPLANE);                                // [pairs of one pass][12][16]

    STAMP_RT(20) STAMP(0)
    // ---- phase A (corr_lookup_tile.hip phase A): thread = (plane, pixel) pair.  EVERY global load of the phase -- the pair's plane record and
    // time coefficients (kernel-argument memory, indexed per lane), the Bezier parameters, the epilogue's bias -- is issued up front and
    // UNCONDITIONALLY (clamped indices; only the LDS stores are predicated): vmcnt retires in order, so whatever is requested after the weight
    // fragments below would wait for all of them, and a load inside a branch stays "pending" in hipcc's counter bookkeeping on the path around
    // its use -- the first re-use of its register behind the gather loop would then become vmcnt(0).
    const int pa_t = min(tid, npair - 1);
    const int pa_p = pa_t / TP, pa_i = pa_t - pa_p * TP;
    const int pa_n = np0 + pa_i;
    const bool pa_on = pa_n < N;
    const int deg = la.deg;
    float px[4], py[4], pcf[4];
    const float* const pp = la.params + (long long)b * 2 * deg * N + min(pa_n, N - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = min(j, deg - 1);
        px[j] = pp[(long long)k * N];
        py[j] = pp[(long long)(deg + k) * N];
        pcf[j] = la.pcoef[pa_p * BFLOW_MAX_DEGREE + k];
    }
    const long long pl_base = reinterpret_cast<long long>(la.planes[pa_p].base);
    const int pl_h = la.planes[pa_p].h, pl_w = la.planes[pa_p].w;
    const float pl_inv = la.planes[pa_p].inv_scale;
    const int n0 = wave * 32;                     // this wave's output channels
    const int ech = (lane & 7) * 4;               // the epilogue's bias: channels n0 + 4 (lane & 7) .. + 3
    float ebias[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ebias[k] = a.shift ? a.shift[min(n0 + ech + k, a.Cout - 1)] : 0.f;
    // ---- weight fragments straight into registers.  Fragment of (k-block, 16-deep step ks): row n0 + l31 of the k-tile, halves
    // (2 ks + kh) * 8 .. + 7 (the packed layout of bflow_conv_pack_weights: (k-tile, cout_pad, 32))
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, NKB * wtile_b, 0x00020000);
    const rsrc_t r_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.wl, 0, NKB * wtile_b, 0x00020000);
    const unsigned wvo = (unsigned)((min(n0 + l31, a.cout_pad - 1) * 32 + kh * 8) * 2);
    half8 w1h[NKB1][2], w1l[NKB1][2];
    half8 w2h[NKB2 ? NKB2 : 1][2], w2l[NKB2 ? NKB2 : 1][2];
#define LC_LOAD_W(WH, WL, J, KB)                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                    \
        WH[J][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r_wh, wvo + ks * 32, (KB) * wtile_b, 0)); \
        WL[J][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r_wl, wvo + ks * 32, (KB) * wtile_b, 0)); \
    }
    // The weights of pass 1 are requested NOW: they depend on nothing, and their 40 x 1 KB per wave then stream in under phase A instead of
    // queueing behind the gather in the CU's one vector-memory pipe.
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NKB1>([&](auto u) __attribute__((always_inline)) { LC_LOAD_W(w1h, w1l, decltype(u)::value, decltype(u)::value) });
    __builtin_amdgcn_sched_barrier(0);
    STAMP(4)
    // The operand tile starts as zeros (the pad channels of the last k-block).  Before any gather: an LDS store issued while a gather is in
    // flight makes hipcc drain EVERY outstanding load (the DMA may alias it), the weights included.
    for (int it = tid; it < 2 * TPLANE / 16; it += LC_THREADS) *reinterpret_cast<f32x4*>(tile + it * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        float fx = 0.f, fy = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < deg) {
                fx = fmaf(px[j], pcf[j], fx);
                fy = fmaf(py[j], pcf[j], fy);
            }
        if (deg > 4) {                                   // higher degrees: the remaining control points, same accumulation order
            for (int k = 4; k < deg; ++k) {
                const float cfk = la.pcoef[pa_p * BFLOW_MAX_DEGREE + k];
                fx = fmaf(pp[(long long)k * N], cfk, fx);
                fy = fmaf(pp[(long long)(deg + k) * N], cfk, fy);
            }
        }
        const int y = pa_n / la.w1, x = pa_n - y * la.w1;
        const float cx = pa_on ? ((float)x + fx) * pl_inv : 0.f;
        const float cy = pa_on ? ((float)y + fy) * pl_inv : 0.f;
        // patch origin from the (clamped) centre; far-away centres see an all-zero neighbourhood, as zero padding demands
        const float ccx = fminf(fmaxf(cx, -64.f), (float)pl_w + 64.f);
        const float ccy = fminf(fmaxf(cy, -64.f), (float)pl_h + 64.f);
        const int th = (pl_h + TILE_H - 1) >> 2, tw = (pl_w + TILE_W - 1) >> 3;
        const long long base = pl_base + ((long long)b * N + min(pa_n, N - 1)) * (th * tw * 32) * 4;
        if (tid < npair) {
            i32x4* const rp = reinterpret_cast<i32x4*>(s_rec + tid);        // LcRec as two 16-B vectors
            rp[0] = i32x4{(int)(unsigned)base, (int)(base >> 32), (int)floorf(ccy) - (R + 1), ((int)floorf(ccx) - (R + 1)) & ~(EPU - 1)};
            rp[1] = i32x4{th * TILE_H, tw * TILE_W, tw, pa_p | (pa_i << 8)};
            s_c[tid] = f32x4{cx, cy, __builtin_bit_cast(float, pl_w), __builtin_bit_cast(float, pl_h)};
        }
    }
    STAMP(1)
    __syncthreads();
    STAMP(2)

    // ---- gather of the planes of one pass by LDS-DMA: unit u = (pair of the pass, row, unit of the row), 64 consecutive units per wave
    // instruction (pairs are plane-major, so pair = first pair of the pass + u / 48); the swizzle uses the pass-local pair index
    auto gather = [&](int pair0, int pairs) __attribute__((always_inline)) {
        const int units = pairs * UPP;
        for (int u0 = wave * 64; u0 < units; u0 += LC_THREADS) {
            const int u = min(u0 + lane, units - 1);
            const int lp = u / UPP, ru = u - lp * UPP;
            const int r = ru >> 2, k = ru & 3;
            const i32x4* const rp = reinterpret_cast<const i32x4*>(s_rec + pair0 + lp);
            const i32x4 r0 = rp[0], r1 = rp[1];             // (base lo, base hi, oy, ox), (ylim, xlim, tw, plane | pixel)
            const int ks = k ^ (((r >> 1) ^ lp) & 3);
            const int gy = r0[2] + r, gx = r0[3] + ks * EPU;
            const bool in = (unsigned)gy < (unsigned)r1[0] && (unsigned)gx < (unsigned)r1[1];
            const long long base = (long long)(((unsigned long long)(unsigned)r0[1] << 32) | (unsigned)r0[0]);
            const char* src = reinterpret_cast<const char*>(base) + (in ? tiled_index(gy, gx, r1[2]) * 4 : 0);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(reinterpret_cast<char*>(patch) + u0 * 16), 16, 0, 0);
        }
    };
    // ---- interpolation of the planes of one pass.  item = (pair of the pass, window row, third of the row): 3 samples, split and written
    // into the operand tile; two items per trip (every LDS read of both before the first store: the loads overlap)
    auto interpolate = [&](int pair0, int pairs) __attribute__((always_inline)) {
        const int items = pairs * 27;
        for (int it0 = tid; it0 < items; it0 += 2 * LC_THREADS) {
            float v[2][3];
            int o[2][3];
            bool on[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                on[q] = it0 + q * LC_THREADS < items;
                const int it = on[q] ? it0 + q * LC_THREADS : it0;
                const int lp = it / 27, r27 = it - lp * 27;
                const int ky = r27 / 3, kg = r27 - ky * 3;
                const int pi = s_rec[pair0 + lp].pi;
                const int p = pi & 255, i = pi >> 8;
                const f32x4* tp = s_tap + (pair0 + lp) * 18;
                const f32x4 ty = tp[WIN + ky];
                const float wn = ty[1], ws = ty[2];
                const int ay = __builtin_bit_cast(int, ty[0]);
                const float* row0 = patch + lp * PELEMS + ay * PCOLS;
                const float* row1 = row0 + PCOLS;
                const int z0 = (((ay >> 1) ^ lp) & 3) << 2, z1 = ((((ay + 1) >> 1) ^ lp) & 3) << 2;
                const int psw = (i >> 2) & 3;               // chunk swizzle of the operand tile's pixel row (the fragment read's `sw`)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int kx = kg * 3 + j;
                    const f32x4 tx = tp[kx];
                    const float ww = tx[1], we = tx[2];
                    const int c0 = __builtin_bit_cast(int, tx[0]), c1 = c0 + 1;
                    // utils.py:19 / grid_sample: nw*I_nw + ne*I_ne + sw*I_sw + se*I_se, weights as products of the axis weights
                    float s_ = row0[c0 ^ z0] * (ww * wn);
                    s_ += row0[c1 ^ z0] * (we * wn);
                    s_ += row1[c0 ^ z1] * (ww * ws);
                    s_ += row1[c1 ^ z1] * (we * ws);
                    v[q][j] = s_;
                    const int c = p * NCH + ky * WIN + kx;  // feature channel (corr.py:343-351: plane-major, window row-major)
                    o[q][j] = (c >> 5) * TKB + i * 64 + ((((c >> 3) & 3) ^ psw) << 4) + (c & 7) * 2;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (on[q]) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        _Float16 hv, lv;
                        split1(v[q][j], hv, lv);
                        lds_store_b16(tile + o[q][j], hv);
                        lds_store_b16(tile + TPLANE + o[q][j], lv);
                    }
                }
        }
    };
    f32x16 hh, xx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        hh[r] = 0.f;
        xx[r] = 0.f;
    }
    const bool active = n0 < a.cout_pad && n0 < ((a.Cout + 31) & ~31);
    // D[channel][pixel] += W[channel][k] x F[k][pixel] for one k-block; split product = hi*hi + (lo*hi + hi*lo) 2^-11.  Lanes of the pixel
    // columns TP .. 31 re-read row TP - 1 (their columns of D are never stored).
    const char* const frow = tile + min(l31, TP - 1) * 64;
    const int fsw = (min(l31, TP - 1) >> 2) & 3;
#define LC_MFMA(WH, WL, J, KB)                                                                                            \
    {                                                                                                                     \
        half8 fh[2], fl[2];                                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                \
            const int co = (KB) * TKB + (((ks * 2 + kh) ^ fsw) << 4);                                                      \
            fh[ks] = *reinterpret_cast<const half8*>(frow + co);                                                          \
            fl[ks] = *reinterpret_cast<const half8*>(frow + TPLANE + co);                                                 \
        }                                                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                \
            hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH[J][ks], fh[ks], hh, 0, 0, 0);                                  \
            xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(WL[J][ks], fh[ks], xx, 0, 0, 0);                                  \
            xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH[J][ks], fl[ks], xx, 0, 0, 0);                                  \
        }                                                                                                                 \
    }

    // ---- pass 1: gather, then the tap tables of ALL planes while it is in flight
    gather(0, P1 * TP);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3)
    for (int it = tid; it < npair * 18; it += LC_THREADS) {      // item = (pair, axis, tap)
        const int pair = it / 18, ax = it - pair * 18;
        const int rec_oy = s_rec[pair].oy, rec_ox = s_rec[pair].ox;
        const float* const cp = reinterpret_cast<const float*>(s_c + pair);     // scalar reads: see the note at lds_store_b128
        const bool isy = ax >= WIN;
        const int d = isy ? ax - WIN : ax;
        const int size = __builtin_bit_cast(int, isy ? cp[3] : cp[2]);
        const float ccx_ = cp[0], ccy_ = cp[1];
        float ic = roundtrip((isy ? ccy_ : ccx_) + (float)(d - R), size);
        ic = fminf(fmaxf(ic, -1.0e4f), 1.0e4f);
        const float f0 = floorf(ic);
        const float w1_ = ic - f0, w0_ = 1.f - w1_;
        const int g0 = (int)f0;
        const int ai = g0 - (isy ? rec_oy : rec_ox);
        const bool inpatch = ai >= 0 && ai + 1 < (isy ? PATCH : PCOLS);
        f32x4 rec4 = {__builtin_bit_cast(float, inpatch ? ai : 0), (inpatch && g0 >= 0 && g0 < size) ? w0_ : 0.f,
                      (inpatch && g0 + 1 >= 0 && g0 + 1 < size) ? w1_ : 0.f, 0.f};
        lds_store_b128(s_tap + it, rec4);
    }
    STAMP(5)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's gather has landed (and everything before it)
    __builtin_amdgcn_sched_barrier(0);
    STAMP(6)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(7)
    interpolate(0, P1 * TP);
    STAMP(8)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                 // k-blocks 0 .. NKB1 - 1 of the operand tile are complete; the patch buffer is free
    __builtin_amdgcn_sched_barrier(0);
    STAMP(9)

    if (NKB2 > 0) {
        // ---- pass 2: the gather of planes 4 .. P - 1 flies while the matrix cores work through the k-blocks of pass 1; the weight registers
        // of a finished k-block are refilled with a k-block of pass 2
        // (everything requested so far has landed -- the wait in front of the first interpolation -- but hipcc's own counter bookkeeping does not
        // read inline assembly: said again with the BUILTIN, or the first use of a weight register behind the gather loop, whose trip count
        // it cannot see, becomes vmcnt(0) = "wait for the gather")
        __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) only
        gather(P1 * TP, (P - P1) * TP);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(10)
        static_for<0, NKB1>([&](auto u) __attribute__((always_inline)) {
            constexpr int J = decltype(u)::value;
            if (active) LC_MFMA(w1h, w1l, J, J)
            if (J < NKB2) LC_LOAD_W(w2h, w2l, (J < NKB2 ? J : 0), NKB1 + J)
        });
        static_for<NKB1, (NKB2 > NKB1 ? NKB2 : NKB1)>([&](auto u) __attribute__((always_inline)) {
            constexpr int J = decltype(u)::value;
            LC_LOAD_W(w2h, w2l, (J < NKB2 ? J : 0), NKB1 + J)
        });
        STAMP(11)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NKB2) : "memory");      // the second gather has landed
        __builtin_amdgcn_sched_barrier(0);
        STAMP(12)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(13)
        interpolate(P1 * TP, (P - P1) * TP);
        STAMP(14)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(15)
        if (active) static_for<0, (NKB2 ? NKB2 : 1)>([&](auto u) __attribute__((always_inline)) {
            constexpr int J = decltype(u)::value;
            if (J < NKB2) LC_MFMA(w2h, w2l, J, NKB1 + J)
        });
    } else {
        if (active) static_for<0, NKB1>([&](auto u) __attribute__((always_inline)) { LC_MFMA(w1h, w1l, decltype(u)::value, decltype(u)::value) });
    }
    STAMP(16)
#undef LC_LOAD_W
#undef LC_MFMA
    // ---- epilogue (the arithmetic of conv_epilogue: bias, activation, hi / lo split; a.scale / addend / gates do not occur here).  The
    // accumulator tile is D[channel][pixel]: lane = pixel, registers 4j .. 4j+3 = channels 8j + 4 kh + 0..3.  Each wave transposes ITS tile
    // through a private LDS slab (row stride 36 floats) and stores in memory order: lane -> (pixel lane/8 + 8 it, channels 4 (lane & 7) .. + 3),
    // 512 B contiguous per wave instruction and plane.  The slabs overlay the operand tile and the patches: every wave must be done with both.
    __builtin_amdgcn_s_barrier();
    if (active && n0 < a.Cout) {
        constexpr int RS = CONV_STG_STRIDE;
        float* const stg = reinterpret_cast<float*>(tile) + wave * (32 * RS);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(stg + l31 * RS + 8 * j + 4 * kh) =
                f32x4{hh[4 * j] + xx[4 * j] * LO_INV, hh[4 * j + 1] + xx[4 * j + 1] * LO_INV, hh[4 * j + 2] + xx[4 * j + 2] * LO_INV,
                      hh[4 * j + 3] + xx[4 * j + 3] * LO_INV};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int rr = lane >> 3;
        f32x4 raws[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) raws[it] = *reinterpret_cast<const f32x4*>(stg + (it * 8 + rr) * RS + ech);
        const long long ob = (((long long)b * a.CBo + a.cb_off + wave) * a.P_out) * 32 + ech;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + rr;
            const int m = np0 + row;
            if (row >= TP || m >= N) continue;
            float v[4] = {raws[it][0] * 1.f + ebias[0], raws[it][1] * 1.f + ebias[1], raws[it][2] * 1.f + ebias[2], raws[it][3] * 1.f + ebias[3]};
            half4v h4, l4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (a.act == 1) v[k] = fmaxf(v[k], 0.f);
                else if (a.act == 2) v[k] = tanhf(v[k]);
                if (n0 + ech + k >= a.Cout) v[k] = 0.f;     // padded channels of the last block are written as zeros
                _Float16 x1, x2;
                split1(v[k], x1, x2);
                h4[k] = x1;
                l4[k] = x2;
            }
            *reinterpret_cast<half4v*>(a.oh + ob + (long long)m * 32) = h4;
            *reinterpret_cast<half4v*>(a.ol + ob + (long long)m * 32) = l4;
        }
    }
    STAMP(17) STAMP_RT(21)
#endif
}

}  // namespace

// C ABI: see include/bflow_hip.h
extern "C" int bflow_corr_lookup_conv1x1(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                         const void* w_hi, const void* w_lo, int Cout, int cout_pad, int k_blocks, const float* bias, int act,
                                         void* out_hi, void* out_lo, int out_channel_blocks, int out_block_offset, int out_rows_per_image,
                                         int B, int h1, int w1, bflow_stream_t stream) {
    using namespace bflow;
    BFLOW_REQUIRE(planes && P > 0 && T > 0, BFLOW_E_ARG, "corr_lookup_conv1x1: bad plane table");
    BFLOW_REQUIRE(P <= BFLOW_MAX_PLANES, BFLOW_E_LIMIT, "corr_lookup_conv1x1: %d planes > %d", P, BFLOW_MAX_PLANES);
    BFLOW_REQUIRE(T <= BFLOW_MAX_TARGETS, BFLOW_E_LIMIT, "corr_lookup_conv1x1: %d targets > %d", T, BFLOW_MAX_TARGETS);
    BFLOW_REQUIRE(deg >= 1 && deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "corr_lookup_conv1x1: degree %d", deg);
    BFLOW_REQUIRE(params && coef && w_hi && w_lo && out_hi && out_lo && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_conv1x1: bad arguments");
    BFLOW_REQUIRE(B <= 65535, BFLOW_E_LIMIT, "corr_lookup_conv1x1: batch %d", B);
    BFLOW_REQUIRE(k_blocks == (P * NCH + 31) / 32, BFLOW_E_ARG, "corr_lookup_conv1x1: weights packed for %d k-blocks, %d planes need %d", k_blocks, P,
                  (P * NCH + 31) / 32);
    BFLOW_REQUIRE(P <= 2 * LC_PASS_PLANES, BFLOW_E_LIMIT, "corr_lookup_conv1x1: %d planes (> 8: two gather passes of four planes; run "
                  "bflow_corr_lookup_bezier_split_tiled + bflow_conv_split)", P);
    BFLOW_REQUIRE(Cout > 0 && Cout <= 32 * LC_WAVES && cout_pad >= Cout && cout_pad % 32 == 0, BFLOW_E_LIMIT, "corr_lookup_conv1x1: Cout %d (pad %d)",
                  Cout, cout_pad);
    BFLOW_REQUIRE(act == 0 || act == 1 || act == 2, BFLOW_E_ARG, "corr_lookup_conv1x1: activation %d", act);
    BFLOW_REQUIRE(out_block_offset >= 0 && out_block_offset + (Cout + 31) / 32 <= out_channel_blocks && out_rows_per_image >= h1 * w1, BFLOW_E_ARG,
                  "corr_lookup_conv1x1: output too small");
    LcArgs la;
    la.P = P; la.T = T; la.deg = deg; la.h1 = h1; la.w1 = w1; la.params = params;
    for (int p = 0; p < P; ++p) {
        BFLOW_REQUIRE(planes[p].base && planes[p].h > 0 && planes[p].w > 0 && planes[p].level >= 0 && planes[p].level < 16 && planes[p].target >= 0 &&
                          planes[p].target < T,
                      BFLOW_E_ARG, "corr_lookup_conv1x1: bad descriptor for plane %d", p);
        la.planes[p].base = planes[p].base;
        la.planes[p].h = planes[p].h;
        la.planes[p].w = planes[p].w;
        la.planes[p].inv_scale = 1.0f / (float)(1 << planes[p].level);
        la.planes[p].target = planes[p].target;
    }
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < deg; ++i) la.pcoef[p * BFLOW_MAX_DEGREE + i] = coef[planes[p].target * deg + i];
    // pixels per workgroup: one workgroup per CU and round; the cost of a round grows with TP (gather, interpolation and stores are per
    // pixel, launch + weights + matrix phase are not): fewest rounds first, then the smallest tile that reaches them
    static const int tp_env = [] { const char* e = getenv("BFLOW_LOOKUP_CONV_TP"); return e ? atoi(e) : 0; }();
    const int N = h1 * w1;
    const int P1 = P < LC_PASS_PLANES ? P : LC_PASS_PLANES;
    auto lds_bytes = [&](int t) { return lc_tables_bytes(P * t) + lc_work_bytes(k_blocks, t, P1 * t); };
    int tp = 0;
    if (tp_env >= 1 && tp_env <= LC_MAX_TP && lds_bytes(tp_env) <= 160 * 1024) tp = tp_env;
    else {
        double best = 1e30;
        for (int t = 2; t <= LC_MAX_TP && P * t <= LC_THREADS; ++t) {
            if (lds_bytes(t) > 160 * 1024) break;
            const long long wgs = (long long)ceil_div(N, t) * B;
            const double cost = (double)ceil_div(wgs, 256) * (6.0 + 0.25 * t);
            if (cost < best) { best = cost; tp = t; }
        }
        BFLOW_REQUIRE(tp > 0, BFLOW_E_LIMIT, "corr_lookup_conv1x1: %d planes do not fit the LDS", P);
    }
    la.TP = tp;
#ifdef LC_STAMPS
    la.stamps = g_lc_stamp_buf;
#else
    la.stamps = nullptr;
#endif
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.wh = (const _Float16*)w_hi; a.wl = (const _Float16*)w_lo;
    a.Cout = Cout; a.cout_pad = cout_pad;
    a.oh = (_Float16*)out_hi; a.ol = (_Float16*)out_lo;
    a.CBo = out_channel_blocks; a.cb_off = out_block_offset; a.P_out = out_rows_per_image;
    a.shift = bias; a.act = act;
    a.stats_reps = 1;
    const int lds = lds_bytes(tp);
    dim3 grid(ceil_div(N, tp), B);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, grid, dim3(LC_THREADS), lds, (hipStream_t)stream, la, a);
    };
    switch (P) {       // k-blocks complete after the first four planes | the rest
        case 1: go(lookup_conv_kernel<3, 0>); break;
        case 2: go(lookup_conv_kernel<6, 0>); break;
        case 3: go(lookup_conv_kernel<8, 0>); break;
        case 4: go(lookup_conv_kernel<11, 0>); break;
        case 5: go(lookup_conv_kernel<10, 3>); break;
        case 6: go(lookup_conv_kernel<10, 6>); break;
        case 7: go(lookup_conv_kernel<10, 8>); break;
        case 8: go(lookup_conv_kernel<10, 11>); break;
        default: break;
    }
    return launch_status("corr_lookup_conv1x1");
}
