// K6 + K7 (+K8, K14) on TILED planes: the product path of inference.
// Reference: CorrData.get_downsampled (models/raft_utils/corr.py:108-125), CorrBlockParallelMultiTarget.__call__ (corr.py:307-351),
// bilinear_sampler (models/raft_utils/utils.py:5-21; F.grid_sample bilinear / zeros / align_corners=True),
// BezierCurves.get_flow_from_reference + coords0 (raft.py:180-181).
//
// Layout.  Every query pixel owns a PRIVATE h_L x w_L plane of the 4-D volume and a look-up reads one <= 12 x 12 neighbourhood of it.
// With row-major planes that is 12 row segments of 48 B, each pulling one or two 128-B lines: ~2.1 KB of lines for 400 B of taps, and
// the first look-up kernel (corr_lookup.hip) ran at 1.5-1.9 TB/s algorithmic at every size because HBM was busy moving those lines.
// Here a plane is stored as ceil(h/4) x ceil(w/8) TILES of 4 x 8 elements (tile-row-major, row-major inside a tile; one fp32 tile =
// one 128-B line), written directly by the correlation build (corr_stream.hip: a wave's 32 stationary columns are one tile) and by the
// pooling kernel below: the same neighbourhood is ~9 lines, every one of them used.
//
// Look-up kernel:
//   * a workgroup owns TP consecutive query pixels and ALL P planes; pair = (plane, pixel).  Phase A: Bezier evaluation -> sampling
//     centre and patch origin per pair;
//   * phase B: the gather is pure LDS-DMA: a patch is 12 rows x UPR aligned 16-B units (fp32: 4 units = 16 columns, fp16: 3 units = 24
//     columns; a unit never crosses a tile row), the flat unit list IS the LDS image, so a wave instruction moves 64 units = 1 KB with
//     no VGPR round trip, no LDS-write instruction and no masks: units outside the tile grid are redirected to the slab's first unit,
//     and zero padding is applied through the WEIGHTS (a corner outside the plane gets weight 0; every stored value is finite);
//   * the 18 taps of a pair (west / north index + two corner weights each; they sit behind the reference's normalise / un-normalise
//     round trip, a true fp32 division) are computed once while the gather is in flight;
//   * phase C: thread = (pair, window column) produces 9 samples into a channel-ordered staging tile; phase D: thread = (channel block,
//     pixel, 8-channel chunk) converts to the split format and stores 16 B of hi + 16 B of lo in the conv engine's blocked layout
//     (B, CB, rows, 32): 512 B contiguous per channel block and workgroup.
//   * the volume element type is a template parameter: fp32 planes, or fp16 planes (bflow_corr_build_f16: BASELINE configs[4]).
#include <cstdlib>
#include "common.h"

namespace {

constexpr int R = BFLOW_LOOKUP_RADIUS;      // 4
constexpr int WIN = 2 * R + 1;              // 9
constexpr int NCH = WIN * WIN;              // 81
constexpr int PATCH = 12;                   // rows floor(c)-5 .. floor(c)+6 cover every bilinear corner incl. round-off flips
constexpr int TILE_H = 4, TILE_W = 8;       // plane tiling (elements)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct TilePlane {
    const void* base;
    int h, w;
    float inv_scale;
    int target;
    float rcp_hm1, rcp_wm1;     // 1 / (h - 1), 1 / (w - 1), correctly rounded (host division): see exact_div
};

struct TileArgs {
    TilePlane planes[BFLOW_MAX_PLANES];
    float pcoef[BFLOW_MAX_PLANES * BFLOW_MAX_DEGREE];   // time coefficients of each plane's target: row p = coef[planes[p].target]
    int P, T, deg;
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// a / b as IEEE division gives it, from r = RN(1 / b): the refinement the compiler's own division expands to (v_div_scale / v_rcp / two
// Newton steps on the quotient / v_div_fmas / v_div_fixup: ~13 vector instructions) without the reciprocal's own refinement -- r is a
// per-plane constant, correctly rounded on the host -- and without the scaling / fix-up of operands near the ends of the exponent range
// (coordinates and plane sizes are far from them).  Checked against `a / b` on the host for every fp32 mantissa of a and every
// b = 1 .. 140, random mantissas up to b = 4096: no difference (tools/micro/exact_div_check.c).  b == 0 (a plane one element wide) has
// r = inf: a * inf is what IEEE a / 0 is (+-inf, NaN for a == 0).
__device__ __forceinline__ float exact_div(float a, float b, float r) {
    const float q0 = a * r;
    const float q1 = fmaf(fmaf(-b, q0, a), r, q0);
    const float q = fmaf(fmaf(-b, q1, a), r, q1);
    return b == 0.0f ? q0 : q;
}

__device__ __forceinline__ float roundtrip(float x, float sm1, float rcp) {   // see corr_lookup.hip: utils.py:13-14 + grid_sample's un-normalisation
    const float g = exact_div(2.0f * x, sm1, rcp) - 1.0f;
    return (g + 1.0f) * (sm1 * 0.5f);
}

// element offset of (y, x) inside a tiled plane with `tw` tiles per tile row
// (24-bit multiply: v_mul_lo_u32 issues at a quarter of the rate; callers pass non-negative coordinates inside the tile grid)
__device__ __forceinline__ int tiled_index(int y, int x, int tw) {
    return ((int)(__umul24(y >> 2, tw) + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7);
}

// small-index arithmetic of the phases on the full-rate 24-bit multiplier; i / 9 for i < 16 k
__device__ __forceinline__ int mul24(int a, int b) { return (int)__umul24(a, b); }
__device__ __forceinline__ int div9(int i) { return (int)(__umul24(i, 7282) >> 16); }

// LDS carve-up of the look-up kernel (byte offsets), shared by the kernel and its launcher.  `stage` and `patch` start on 128-B boundaries:
// phase D reads `stage` as 16-B vectors and `patch` is the destination of 16-B LDS-DMA units.
struct LookupLds {
    int hw, cxy, rec, at, wt, uni, stage, patch, total;
};
__host__ __device__ inline LookupLds lookup_lds(int P, int TP, int unit_bytes_per_pair) {
    const int np = P * TP, cstride = ((P * NCH + 31) >> 5) * 32;
    LookupLds L;
    L.hw = 0;                                  // [4][MAX_PLANES]: h, w (int), 1/(h-1), 1/(w-1) (float)
    L.cxy = 16 * BFLOW_MAX_PLANES;             // [2][np] sampling centres
    L.rec = L.cxy + 8 * np;                    // [np] gather record, 48 B (np is even: 16-B aligned)
    L.at = L.rec + 48 * np;                    // [np][18] patch-relative west index of the 9 columns, north index of the 9 rows
    L.wt = L.at + 72 * np;                     // [np][18][2] (west, east) / (north, south) weights
    L.uni = L.wt + 144 * np;                   // [np] 1: the 9 window rows are the patch rows 1 .. 10 in order
    L.stage = (L.uni + 4 * np + 127) & ~127;   // [TP][cstride] the tile's features in channel order
    L.patch = L.stage + TP * cstride * 4;      // [np][12][PCOLS]
    L.total = L.patch + np * unit_bytes_per_pair;
    return L;
}

// COLS: phase C as (pair, window column) threads -- fewer instructions, the form for grids of many rounds (instruction-bound: batch >= 4 at
// 60 x 80) -- instead of (pair, window row) threads, the form of the latency-bound single-round grids.
//
// Round 6: the kernel is bound by VALU ISSUE at batch 8 (75 workgroups per CU x ~1 500 vector instructions per workgroup x 4 cycles per
// wave64 instruction = the launch; the dispatcher rotates a workgroup's first wave over the SIMDs, tools/micro/simd_map, so it is the
// TOTAL that counts, not the per-wave split).  Four cuts of that total (profiles/r06_k7_valu_diet.txt):
//   * gather: ONE wave instruction per pair (48 / 36 active lanes, LDS slots 768 / 576 B apart), so a lane's (row, unit) is loop-invariant
//     and the pair's record -- plane address, origin, tile-grid limits, unpacked by the pair's thread of phase A -- is wave-uniform:
//     14 vector instructions per pair instead of 38 per 64 units;
//   * tap tables: one thread per (pair, tap) does both axes; the round trip's division is `exact_div` (5 instead of ~13 instructions);
//   * interpolation (COLS, regular pairs): horizontal pass over the ten patch rows, then the vertical pass -- 2 x (mul + fma) per sample on
//     packed fp32 instructions instead of four weight products and four multiply-adds (the weights' products are never formed; the
//     rounding differs from the reference's nw*I_nw + ne*I_ne + sw*I_sw + se*I_se by the order of four fp32 operations);
//   * output: the split as hardware conversions under FP16_OVFL + flushed fp16 results, two values per instruction (see phase D).
template <typename VT, int TP, int THREADS, bool COLS = false>
__global__ __launch_bounds__(THREADS) void corr_lookup_tile_kernel(TileArgs args, const float* __restrict__ params, _Float16* __restrict__ oh,
                                                                 _Float16* __restrict__ ol, int CBk, int Prow, int h1, int w1, int abl,
                                                                 bflow::Im2colArgs rider, int rider_blocks) {
    constexpr int EPU = 16 / (int)sizeof(VT);          // elements per 16-B unit: 4 (fp32) / 8 (fp16)
    constexpr int UPR = sizeof(VT) == 4 ? 4 : 3;       // units per patch row: columns [ox_al, ox_al + UPR*EPU) cover ox .. ox+11 for any alignment
    constexpr int PCOLS = UPR * EPU;                   // 16 / 24
    constexpr int UPP = PATCH * UPR;                   // units per pair: 48 / 36
    constexpr int PELEMS = PATCH * PCOLS;              // patch elements per pair: 192 / 288
    extern __shared__ __attribute__((aligned(128))) char smem[];   // the only shared object; carved by lookup_lds (sizes follow P)
    const int tid = threadIdx.x;
    const int N = h1 * w1;
    const int P = args.P;
    const int b = blockIdx.y;
    // Rider (bflow_corr_lookup_im2col): the first `rider_blocks` workgroups of every image expand the 7x7 windows of the SAME Bezier parameters
    // for the motion encoder's convf1 (update.py:91) -- the other first kernel of an update iteration, 5-8 us of its own on a second queue
    // before -- and leave; they are dispatched first and gone long before the gather tiles finish (as the LAST workgroups of the grid
    // instead: 3.425-3.439 vs 3.412-3.432 ms per frame, three alternating pairs).
    if ((int)blockIdx.x < rider_blocks) {
        const unsigned per_image = (unsigned)rider.CBk * (unsigned)rider.P * 4u;
        for (unsigned e = blockIdx.x * THREADS + tid; e < per_image; e += (unsigned)rider_blocks * THREADS) bflow::im2col_small_item(rider, b, e);
        return;
    }
    const int n0 = ((int)blockIdx.x - rider_blocks) * TP;
    const int npair = P * TP;                 // pair = plane * TP + pixel of the tile
    const bool swz_on = !(abl & 16);
    const int cstride = ((P * NCH + 31) >> 5) * 32;   // staged channels per pixel (whole channel blocks)
    const LookupLds L = lookup_lds(P, TP, UPP * 16);
    // plane table (indexed per lane: a runtime index into the kernel-argument struct would spill the struct to scratch)
    int* s_h = reinterpret_cast<int*>(smem + L.hw);                                // [MAX_PLANES]
    int* s_w = s_h + BFLOW_MAX_PLANES;
    float* s_rh = reinterpret_cast<float*>(s_w + BFLOW_MAX_PLANES);               // 1 / (h - 1), 1 / (w - 1)
    float* s_rw = s_rh + BFLOW_MAX_PLANES;
    float* s_cx = reinterpret_cast<float*>(smem + L.cxy);                          // [npair] sampling centres
    float* s_cy = s_cx + npair;
    // [npair] gather record, 48 B: plane address (64 bits) | origin y | origin x (aligned down to a unit) | tile-grid rows | tile-grid
    // columns (elements) | tiles per tile row | - | first patch row to fetch | rows - 1 | first unit of a row | units - 1 -- everything the
    // gather needs per pair, unpacked ONCE by the pair's thread of phase A
    int* s_rec = reinterpret_cast<int*>(smem + L.rec);
    int* s_at = reinterpret_cast<int*>(smem + L.at);
    float* s_wt = reinterpret_cast<float*>(smem + L.wt);       // ZERO where the corner lies outside the plane (= grid_sample's zero padding) or the patch
    int* s_uni = reinterpret_cast<int*>(smem + L.uni);
    float* stage = reinterpret_cast<float*>(smem + L.stage);
    VT* patch = reinterpret_cast<VT*>(smem + L.patch);

    // ---- phase A: thread = (plane, pixel) pair: Bezier evaluation -> sampling centre and patch origin ---------------------------
    // Every global load of a pair's thread is issued up front: the pair's plane record and time
    // coefficients sit in kernel-argument memory indexed per lane, i.e. they are global loads like the parameters, and as
    // "record -> coefficient row -> parameters" they were a chain of three dependent round trips at the head of a latency-bound kernel
    // (the phase was 3.7 of its 13.2 us at DSEC size).
    if (tid >= THREADS - BFLOW_MAX_PLANES && tid - (THREADS - BFLOW_MAX_PLANES) < P) {      // (last wave: the first one has the pairs)
        const int p = tid - (THREADS - BFLOW_MAX_PLANES);
        s_h[p] = args.planes[p].h;
        s_w[p] = args.planes[p].w;
        s_rh[p] = args.planes[p].rcp_hm1;
        s_rw[p] = args.planes[p].rcp_wm1;
    }
    if (tid < npair) {
        const int p = tid / TP, i = tid - p * TP;
        const int n = n0 + i;
        const int deg = args.deg;
        const float* const pp = params + (long long)b * 2 * deg * N + min(n, N - 1);
        float px[4], py[4], pcf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(j, deg - 1);
            px[j] = pp[(long long)k * N];
            py[j] = pp[(long long)(deg + k) * N];
            pcf[j] = args.pcoef[p * BFLOW_MAX_DEGREE + k];
        }
        const int pl_h = args.planes[p].h, pl_w = args.planes[p].w;
        const float pl_inv = args.planes[p].inv_scale;
        // coords = coords0 + sum_i coef[t][i] * P_i   (bezier.py:185, raft.py:181); params (B, 2*deg, N)
        float fx = 0.f, fy = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < deg) {
                fx = fmaf(px[j], pcf[j], fx);
                fy = fmaf(py[j], pcf[j], fy);
            }
        for (int k = 4; k < deg; ++k) {                  // higher degrees: same accumulation order
            const float cfk = args.pcoef[p * BFLOW_MAX_DEGREE + k];
            fx = fmaf(pp[(long long)k * N], cfk, fx);
            fy = fmaf(pp[(long long)(deg + k) * N], cfk, fy);
        }
        const int y = n / w1, x = n - y * w1;
        const float cx = n < N ? ((float)x + fx) * pl_inv : 0.f;   // corr.py:333 (division by 2^level == exact multiply)
        const float cy = n < N ? ((float)y + fy) * pl_inv : 0.f;
        // patch origin from the (clamped) centre; far-away centres see an all-zero neighbourhood, as zero padding demands
        const float ccx = fminf(fmaxf(cx, -64.f), (float)pl_w + 64.f);
        const float ccy = fminf(fmaxf(cy, -64.f), (float)pl_h + 64.f);
        const int ox = ((int)floorf(ccx) - (R + 1)) & ~(EPU - 1);   // floor to a unit boundary (two's complement: also for negative origins)
        const int oy = (int)floorf(ccy) - (R + 1);
        s_cx[tid] = cx;
        s_cy[tid] = cy;
        const int th = (pl_h + TILE_H - 1) >> 2, tw = (pl_w + TILE_W - 1) >> 3;
        const long long plane = (long long)b * N + min(n, N - 1);                        // pixels past the end re-read the last one (never stored)
        const char* pb = reinterpret_cast<const char*>(args.planes[p].base) + plane * (th * tw * 32) * (long long)sizeof(VT);
        const unsigned long long pbu = (unsigned long long)pb;
        typedef int i32x4_ __attribute__((ext_vector_type(4)));
        i32x4_* rec = reinterpret_cast<i32x4_*>(s_rec + 12 * tid);
        rec[0] = i32x4_{(int)(unsigned)pbu, (int)(unsigned)(pbu >> 32), oy, ox};
        rec[1] = i32x4_{th * TILE_H, tw * TILE_W, tw, 0};
        // Window row ky samples y = cy + ky - 4: north corner floor(cy) + ky - 4 = patch row 1 + ky -- unless the centre was clamped (far
        // outside: every weight is zero) or cy sits within the round trip's error of an integer, where `roundtrip` may move single rows
        // across it.  Those pairs take phase C's general form; EDGE >> the round trip's error (a few ulps of the coordinate).
        const float fry = cy - floorf(cy);
        const float edge = fmaxf(1.0f / 1024.0f, (float)pl_h * 1.0e-6f);
        const bool uni_y = cy == ccy && fry > edge && fry < 1.0f - edge;
        s_uni[tid] = uni_y ? 1 : 0;
        // The taps of a REGULAR pair touch the patch rows 1 .. 10 and the plane columns floor(cx) - 4 .. floor(cx) + 5 only: rows 0 / 11 and
        // the outer units exist for the round trip's integer flips.  What no tap can touch is not fetched (the kernel moves HBM lines at
        // the rate a pure gather of its shape reaches once its instruction count is cut: profiles/r06_k7_valu_diet.txt); the LDS positions
        // keep stale data that no regular pair reads.  Pairs near an integer or clamped fetch the full patch.
        const float frx = cx - floorf(cx);
        const float edgx = fmaxf(1.0f / 1024.0f, (float)pl_w * 1.0e-6f);
        const bool uni_x = cx == ccx && frx > edgx && frx < 1.0f - edgx && !(abl & 32);
        const int offx = ((int)floorf(ccx) - (R + 1)) & (EPU - 1);          // first tap column - 1, relative to the aligned origin
        const int klo = uni_x ? (offx + 1) / EPU : 0, khi = uni_x ? (offx + 2 * R + 2) / EPU : UPR - 1;
        rec[2] = (uni_y && !(abl & 32)) ? i32x4_{1, 2 * R + 1, klo, khi - klo} : i32x4_{0, PATCH - 1, klo, khi - klo};
    }
    __syncthreads();

    // ---- phase B: gather by LDS-DMA: ONE wave instruction per pair; lane = (patch row, unit of the row), UPP active lanes ----------------
    {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int r = lane / UPR, k = lane - r * UPR;
        // LDS position (pair, r, k) holds source unit k ^ swz(r, pair): spreads the interpolation's reads over the banks (a patch row is
        // 64 B and a patch 768 B, so without it rows alternate between two bank groups and all pairs share them)
        const int kr = (UPR == 4 && swz_on) ? (k ^ ((r >> 1) & 3)) : k;
        for (int pair = wave; pair < npair && !(abl & 1); pair += THREADS / 64) {
            // (inline assembly: the compiler puts a visible LDS read behind vmcnt(0) while an LDS-DMA is in flight -- the DMA writes LDS --,
            //  which would make every iteration wait for the previous one's gather; the records are not what the DMA writes)
            typedef int i32x4_ __attribute__((ext_vector_type(4)));
            i32x4_ ra, rb, rc;
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ra), "=&v"(rb), "=&v"(rc)
                         : "v"((unsigned)(size_t)(s_rec + 12 * pair)));
            const int ks = (UPR == 4 && swz_on) ? (kr ^ (pair & 3)) : kr;
            const int gy = ra[2] + r, gx = ra[3] + ks * EPU;
            const bool in = (unsigned)gy < (unsigned)rb[0] && (unsigned)gx < (unsigned)rb[1];   // inside the tile grid (pads included)
            const int idx = in ? tiled_index(gy, gx, rb[2]) : 0;     // units outside the grid re-read the slab's first unit (they only meet zero weights)
            const char* src = reinterpret_cast<const char*>(((unsigned long long)(unsigned)ra[1] << 32) | (unsigned)ra[0]) + (long long)idx * (int)sizeof(VT);
            if (lane < UPP && (unsigned)(r - rc[0]) <= (unsigned)rc[1] && (unsigned)(ks - rc[2]) <= (unsigned)rc[3])
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(reinterpret_cast<char*>(patch) + pair * (UPP * 16)), 16, 0, 0);
        }
    }
    // tap tables while the gather is in flight: item = (pair, tap), both axes
    for (int it = tid; it < npair * WIN && !(abl & 8); it += THREADS) {
        const int pair = div9(it), d = it - mul24(pair, WIN);
        const int p = pair / TP;
        const float off = (float)(d - R);
#pragma unroll
        for (int isy = 0; isy < 2; ++isy) {
            const int size = isy ? s_h[p] : s_w[p];
            float ic = roundtrip((isy ? s_cy[pair] : s_cx[pair]) + off, (float)(size - 1), isy ? s_rh[p] : s_rw[p]);
            ic = fminf(fmaxf(ic, -1.0e4f), 1.0e4f);
            const float f0 = floorf(ic);
            const float w1_ = ic - f0, w0_ = 1.f - w1_;     // far (east / south) and near (west / north) corner weights
            const int g0 = (int)f0;                         // plane coordinate of the near corner
            const int ai = g0 - s_rec[12 * pair + (isy ? 2 : 3)];
            const bool inpatch = (unsigned)ai < (unsigned)((isy ? PATCH : PCOLS) - 1);
            const int o = mul24(pair, 18) + isy * WIN + d;
            s_at[o] = inpatch ? ai : 0;
            typedef float f32x2w_ __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2w_*>(s_wt + 2 * o) = f32x2w_{(inpatch && (unsigned)g0 < (unsigned)size) ? w0_ : 0.f,
                                                               (inpatch && (unsigned)(g0 + 1) < (unsigned)size) ? w1_ : 0.f};
        }
    }
    // pad channels of the last channel block are written as zeros
    if (tid < TP * (cstride - P * NCH)) {            // (< 32 TP <= THREADS items)
        const int i = tid / (cstride - P * NCH), c = tid - i * (cstride - P * NCH);
        stage[i * cstride + P * NCH + c] = 0.f;
    }
    __syncthreads();   // (the compiler drains the DMA with vmcnt(0) in front of it)

    // ---- phase C: interpolation.  unit = (pair, window row): 9 samples -> stage[pixel][plane*81 + row*9 ..] -------------------------
    if constexpr (!COLS) {
        for (int it = tid; it < npair * WIN && !(abl & 2); it += THREADS) {
            const int pair = div9(it), ky = it - mul24(pair, WIN);
            const int p = pair / TP, i = pair - p * TP;
            const int* at = s_at + mul24(pair, 18);
            const float* wt = s_wt + mul24(pair, 36);
            const float wn = wt[2 * (WIN + ky)], ws = wt[2 * (WIN + ky) + 1];
            const int ay = at[WIN + ky];
            const VT* row0 = patch + mul24(pair, PELEMS) + ay * PCOLS;
            const VT* row1 = row0 + PCOLS;
            const int z0 = (UPR == 4 && swz_on) ? ((((ay >> 1) ^ pair) & 3) << 2) : 0, z1 = (UPR == 4 && swz_on) ? (((((ay + 1) >> 1) ^ pair) & 3) << 2) : 0;
            float* dst = stage + mul24(i, cstride) + mul24(p, NCH) + ky * WIN;
#pragma unroll
            for (int kx = 0; kx < WIN; ++kx) {
                const float ww = wt[2 * kx], we = wt[2 * kx + 1];
                const int c0 = at[kx], c1 = c0 + 1;
                // utils.py:19 / grid_sample: nw*I_nw + ne*I_ne + sw*I_sw + se*I_se with the weights the products of the axis weights --
                // evaluated as horizontal pass, then vertical pass (the SAME expression in every form of this phase: a sample does not
                // depend on the batch size that selects the form)
                const float hn = fmaf((float)row0[c1 ^ z0], we, (float)row0[c0 ^ z0] * ww);
                const float hs = fmaf((float)row1[c1 ^ z1], we, (float)row1[c0 ^ z1] * ww);
                dst[kx] = fmaf(hs, ws, hn * wn);
            }
        }
    } else {
    // (COLS: thread = (pair, window COLUMN).  In the regular case -- s_uni -- the 9 window rows of a pair are the patch rows 1 .. 10 in order,
    //  so a column thread reads its two patch columns of those ten rows ONCE, interpolates them horizontally (ten values) and then vertically.)
    for (int it = tid; it < npair * WIN && !(abl & 2); it += THREADS) {
        const int pair = div9(it), kx = it - mul24(pair, WIN);
        const int p = pair / TP, i = pair - p * TP;
        const int* at = s_at + mul24(pair, 18);
        const float* wt = s_wt + mul24(pair, 36);
        const float ww = wt[2 * kx], we = wt[2 * kx + 1];
        const int c0 = at[kx], c1 = c0 + 1;
        const VT* pb = patch + mul24(pair, PELEMS);
        float* dst = stage + mul24(i, cstride) + mul24(p, NCH) + kx;
        auto zof = [&](int row) -> int { return (UPR == 4 && swz_on) ? ((((row >> 1) ^ pair) & 3) << 2) : 0; };
        if (s_uni[pair] && at[WIN] == 1) {
            float hr[WIN + 1];
#pragma unroll
            for (int r = 0; r <= WIN; ++r) {
                const int z = zof(1 + r);
                hr[r] = fmaf((float)pb[(1 + r) * PCOLS + (c1 ^ z)], we, (float)pb[(1 + r) * PCOLS + (c0 ^ z)] * ww);
            }
#pragma unroll
            for (int ky = 0; ky < WIN; ++ky) {
                const float wn = wt[2 * (WIN + ky)], ws = wt[2 * (WIN + ky) + 1];
                dst[ky * WIN] = fmaf(hr[ky + 1], ws, hr[ky] * wn);
            }
        } else {
#pragma unroll
            for (int ky = 0; ky < WIN; ++ky) {
                const float wn = wt[2 * (WIN + ky)], ws = wt[2 * (WIN + ky) + 1];
                const int ay = at[WIN + ky];
                const VT* row0 = pb + ay * PCOLS;
                const VT* row1 = row0 + PCOLS;
                const int z0 = zof(ay), z1 = zof(ay + 1);
                // (the same two-pass expression as above: a pair's samples do not depend on which form its neighbours in the wave take)
                const float hn = fmaf((float)row0[c1 ^ z0], we, (float)row0[c0 ^ z0] * ww);
                const float hs = fmaf((float)row1[c1 ^ z1], we, (float)row1[c0 ^ z1] * ww);
                dst[ky * WIN] = fmaf(hs, ws, hn * wn);
            }
        }
    }
    }
    __syncthreads();

    // ---- phase D: output in memory order.  item = (channel block, pixel, 8-channel chunk), chunk fastest: 16 B of hi + 16 B of lo ----------
    // (round 5, instruction diet: the staged values are read as two 16-B vectors -- `stage` is 128-B aligned (lookup_lds) --, and the stores go
    //  through one buffer descriptor per plane and image with a 32-bit offset: the per-item 64-bit row arithmetic was a tenth of this phase)
    const int items = (cstride >> 5) * TP * 4;
    const long long img = (long long)b * CBk * Prow * 32;                         // elements of the images before this one
    const __amdgpu_buffer_rsrc_t r_oh = __builtin_amdgcn_make_buffer_rsrc((void*)(oh + img), 0, CBk * Prow * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ol = __builtin_amdgcn_make_buffer_rsrc((void*)(ol + img), 0, CBk * Prow * 64, 0x00020000);
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef int i32x4v_ __attribute__((ext_vector_type(4)));
#ifndef K7_SPLIT1
    // (round 6, instruction diet: the split of the output phase.  bflow::split1 per value is 11 vector instructions -- clamp, NaN select,
    //  convert, convert back, subnormal select, subtract, scale, convert, pack: 88 of the ~110 of an item, and this phase is a quarter of a
    //  kernel that is bound by VALU issue at batch 8.  Here the special cases are the HARDWARE's, set once per wave in the MODE register:
    //    FP16_OVFL = 1   a finite value beyond +-65504 converts to +-65504 instead of inf (saturation; split1's clamp),
    //    FP_DENORM (f16) = flush results, keep sources: a conversion whose result would be an fp16 subnormal gives 0 -- split1's
    //                    "hi = 0 below 2^-14" (the matrix cores flush subnormal fp16 INPUTS anyway, also those of the lo plane),
    //  and two values share v_cvt_pk_f16_f32 / v_pk_add_f32 / v_pk_mul_f32: 6 instructions per PAIR.  Normal values get the bits split1
    //  gives them; NaN stays NaN; +-inf (which split1 saturates) becomes hi = inf, lo = NaN -- a volume with infinities has no meaning.)
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // hwreg(HW_REG_MODE, 23, 1): FP16_OVFL
    __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 1);      // hwreg(HW_REG_MODE, 6, 2): f16 / f64 denormals: sources kept, results flushed
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
#endif
    for (int it = tid; it < items && !(abl & 4); it += THREADS) {
        const int chunk = it & 3, i = (it >> 2) % TP, cb = it / (4 * TP);
        const int n = n0 + i;
        if (n >= N) continue;
        const f32x4_* sp = reinterpret_cast<const f32x4_*>(stage + mul24(i, cstride) + cb * 32 + chunk * 8);
        const f32x4_ v0 = sp[0], v1 = sp[1];
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        half8 h8, l8;
#ifdef K7_SPLIT1
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 hh, ll;
            bflow::split1(v[j], hh, ll);
            h8[j] = hh;
            l8[j] = ll;
        }
#else
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2_ x = {v[j], v[j + 1]};
            const f16x2_ hh = __builtin_convertvector(x, f16x2_);
            const f32x2_ r = (x - __builtin_convertvector(hh, f32x2_)) * bflow::SPLIT_LO_SCALE;      // (exact: hh is x rounded to 11 bits)
            const f16x2_ ll = __builtin_convertvector(r, f16x2_);
            h8[j] = hh[0];
            h8[j + 1] = hh[1];
            l8[j] = ll[0];
            l8[j + 1] = ll[1];
        }
#endif
        const unsigned o = (unsigned)(((mul24(cb, Prow) + n) * 32 + chunk * 8) * 2);    // bytes inside the image's plane (< 2^31: checked by the launcher)
        // non-temporal: the consumer (convc1) is the next kernel and starts with a cold L2 anyway; the 13.5 MB of features then leave during
        // the kernel instead of in the write-back at its end (round 4: 12.8 -> 11.8 us at C2; LOOKUP_PLAIN_STORES for A/B)
#ifdef LOOKUP_PLAIN_STORES
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v_, h8), r_oh, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v_, l8), r_ol, o, 0, 0);
#else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v_, h8), r_oh, o, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v_, l8), r_ol, o, 0, 2);
#endif
    }
}

// ---- K6 on tiled planes: 2x2 mean, floor on odd sizes (corr.py:119), tiled in -> tiled out; pad positions of the output are zero -------
template <typename VT>
__global__ __launch_bounds__(256) void corr_pool2x2_tiled_kernel(const VT* __restrict__ in, VT* __restrict__ out, long long planes, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const int twi = (w + TILE_W - 1) >> 3, thi = (h + TILE_H - 1) >> 2;
    const int two = (wo + TILE_W - 1) >> 3, tho = (ho + TILE_H - 1) >> 2;
    const int psi = thi * twi * 32, pso = tho * two * 32;
    const long long total = planes * pso;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long pl = idx / pso;
        const int e = (int)(idx - pl * pso);
        const int tile = e >> 5, ty = tile / two, tx = tile - ty * two;
        const int y = ty * 4 + ((e >> 3) & 3), x = tx * 8 + (e & 7);
        float v = 0.f;
        if (y < ho && x < wo) {
            const VT* q = in + pl * psi;
            // F.avg_pool2d accumulates the window row-major and divides by the window size (4: exact)
            v = ((((float)q[tiled_index(2 * y, 2 * x, twi)] + (float)q[tiled_index(2 * y, 2 * x + 1, twi)]) +
                  (float)q[tiled_index(2 * y + 1, 2 * x, twi)]) + (float)q[tiled_index(2 * y + 1, 2 * x + 1, twi)]) * 0.25f;
        }
        out[idx] = (VT)v;
    }
}

}  // namespace

namespace bflow {

int lookup_tile_launch(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg, void* out_hi, void* out_lo,
                       int channel_blocks, int rows_per_image, int B, int h1, int w1, bool f16_planes, hipStream_t stream,
                       const Im2colArgs* rider) {
    BFLOW_REQUIRE(planes && P > 0 && T > 0, BFLOW_E_ARG, "corr_lookup: bad plane table");
    BFLOW_REQUIRE(P <= BFLOW_MAX_PLANES, BFLOW_E_LIMIT, "corr_lookup: %d planes > %d", P, BFLOW_MAX_PLANES);
    BFLOW_REQUIRE(T <= BFLOW_MAX_TARGETS, BFLOW_E_LIMIT, "corr_lookup: %d targets > %d", T, BFLOW_MAX_TARGETS);
    BFLOW_REQUIRE(deg >= 1 && deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "corr_lookup_bezier_split: degree %d", deg);
    BFLOW_REQUIRE(params && coef && out_hi && out_lo && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_bezier_split: bad arguments");
    BFLOW_REQUIRE(B <= 65535, BFLOW_E_LIMIT, "corr_lookup_bezier_split: batch %d", B);
    BFLOW_REQUIRE(channel_blocks * 32 >= P * NCH && rows_per_image >= h1 * w1, BFLOW_E_ARG, "corr_lookup_bezier_split: output too small");
    BFLOW_REQUIRE((long long)channel_blocks * rows_per_image * 64 < (1LL << 31), BFLOW_E_LIMIT, "corr_lookup_bezier_split: one image's feature plane exceeds 2 GiB");
    TileArgs a;
    a.P = P;
    a.T = T;
    a.deg = deg;
    for (int p = 0; p < P; ++p) {
        BFLOW_REQUIRE(planes[p].base && planes[p].h > 0 && planes[p].w > 0 && planes[p].level >= 0 && planes[p].level < 16 && planes[p].target >= 0 &&
                          planes[p].target < T,
                      BFLOW_E_ARG, "corr_lookup: bad descriptor for plane %d", p);
        a.planes[p].base = planes[p].base;
        a.planes[p].h = planes[p].h;
        a.planes[p].w = planes[p].w;
        a.planes[p].inv_scale = 1.0f / (float)(1 << planes[p].level);
        a.planes[p].target = planes[p].target;
        a.planes[p].rcp_hm1 = 1.0f / (float)(planes[p].h - 1);      // correctly rounded (IEEE host division); inf for a one-row plane
        a.planes[p].rcp_wm1 = 1.0f / (float)(planes[p].w - 1);
    }
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < deg; ++i) a.pcoef[p * BFLOW_MAX_DEGREE + i] = coef[planes[p].target * deg + i];
    // tile = 2 query pixels x all planes, 256 threads: 16-33 KB of LDS per workgroup -> 5-8 workgroups per CU (measured best of
    // {2, 4, 8} pixels x {128, 256} threads at every BASELINE shape; BFLOW_LOOKUP_TP / BFLOW_LOOKUP_ABL are timing knobs for tools/)
    static const int tp_env = [] { const char* e = getenv("BFLOW_LOOKUP_TP"); return e ? atoi(e) : 0; }();
    static const int abl = [] { const char* e = getenv("BFLOW_LOOKUP_ABL"); return e ? atoi(e) : 0; }();
    const int tp = (tp_env == 2 || tp_env == 4 || tp_env == 8) ? tp_env : 2;
    const int lds = lookup_lds(P, tp, PATCH * 16 * (f16_planes ? 3 : 4)).total;
    Im2colArgs m = {};
    int rider_blocks = 0;
    if (rider) {
        m = *rider;
        rider_blocks = (int)ceil_div((long long)m.CBk * m.P * 4, 256);      // one 8-channel group per thread
        if (rider_blocks > 1024) rider_blocks = 1024;
    }
    dim3 grid(ceil_div((long long)h1 * w1, tp) + rider_blocks, B);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, params, (_Float16*)out_hi, (_Float16*)out_lo, channel_blocks, rows_per_image, h1, w1, abl,
                           m, rider_blocks);
    };
    static const int cols_env = [] { const char* e = getenv("BFLOW_LOOKUP_COLS"); return e ? atoi(e) : -1; }();   // tools A/B: 0 / 1 force a form
    // (round 6: with the two-pass interpolation the column form is the faster one at every BASELINE shape -- C2 10.4 vs 10.5 us, C4 shard
    //  56.7 vs 59.3, C5 60.9 vs 63.7 us, profiles/r06_k7_tp_cols.txt -- and one form for every batch size keeps a sample's bits independent of it)
    const bool cols = cols_env >= 0 ? cols_env != 0 : true;
    if (cols) {
        if (f16_planes) {
            if (tp == 2) go(corr_lookup_tile_kernel<_Float16, 2, 256, true>);
            else if (tp == 4) go(corr_lookup_tile_kernel<_Float16, 4, 256, true>);
            else go(corr_lookup_tile_kernel<_Float16, 8, 256, true>);
        } else {
            if (tp == 2) go(corr_lookup_tile_kernel<float, 2, 256, true>);
            else if (tp == 4) go(corr_lookup_tile_kernel<float, 4, 256, true>);
            else go(corr_lookup_tile_kernel<float, 8, 256, true>);
        }
    } else if (f16_planes) {
        if (tp == 2) go(corr_lookup_tile_kernel<_Float16, 2, 256>);
        else if (tp == 4) go(corr_lookup_tile_kernel<_Float16, 4, 256>);
        else go(corr_lookup_tile_kernel<_Float16, 8, 256>);
    } else {
        if (tp == 2) go(corr_lookup_tile_kernel<float, 2, 256>);
        else if (tp == 4) go(corr_lookup_tile_kernel<float, 4, 256>);
        else go(corr_lookup_tile_kernel<float, 8, 256>);
    }
    return launch_status("corr_lookup_bezier_split");
}

int pool_tiled_launch(const void* in, void* out, long long planes, int h, int w, bool f16, hipStream_t stream) {
    BFLOW_REQUIRE(in && out && planes > 0 && h >= 2 && w >= 2, BFLOW_E_ARG, "corr_pool2x2_tiled: bad arguments");
    const long long total = planes * (ceil_div(h / 2, TILE_H) * ceil_div(w / 2, TILE_W) * 32);
    const int grid = stream_grid(total, 256) * 4;
    if (f16) hipLaunchKernelGGL(corr_pool2x2_tiled_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, (const _Float16*)in, (_Float16*)out, planes, h, w);
    else hipLaunchKernelGGL(corr_pool2x2_tiled_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)in, (float*)out, planes, h, w);
    return launch_status("corr_pool2x2_tiled");
}

}  // namespace bflow
