// K6: correlation pyramid (2x2 average pooling) and K7: 9x9 bilinear window look-up.
// Reference: CorrData.get_downsampled (corr.py:108-125), CorrBlockParallelMultiTarget.__call__ (corr.py:307-351),
// bilinear_sampler (raft_utils/utils.py:5-21; F.grid_sample bilinear / zeros / align_corners=True).
//
// Look-up design (HBM / gather bound):
//   every query pixel owns a PRIVATE h_L x w_L plane of the 4-D volume, so nothing is shared between pixels;
//   per (pixel, plane) the 81 samples touch one <= 12x12 patch.  A block takes 64 consecutive query pixels of
//   one plane:
//     phase 1  the 64 patches are fetched cooperatively, 16 lanes per patch row (48-B row segments instead of
//              64 scattered dwords per wave load) with zero fill outside the plane (= grid_sample's zero padding),
//              into LDS at an odd per-pixel stride (conflict-free for lane == pixel);
//     phase 2  lane == pixel: each wave produces a quarter of the 81 window positions from LDS and writes
//              out[b, p*81+k, pixel..pixel+63] as one coalesced 256-B row.
//   The Bezier evaluation + coords0 (raft.py:180-181) is optionally fused in front (FUSED = true), which removes
//   the `flows`/`coords1` tensors and one launch per GRU iteration.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int R = BFLOW_LOOKUP_RADIUS;   // 4
constexpr int WIN = 2 * R + 1;           // 9
constexpr int PATCH = 12;                // floor(c)-5 .. floor(c)+6 covers every bilinear corner incl. round-off flips
constexpr int PSTRIDE = PATCH * PATCH + 1;  // 145 floats (odd -> bank-conflict free for lane == pixel)
constexpr int TILE = 64;                 // query pixels per block (fp32 NCHW output)
constexpr int TILE_SPLIT = 16;           // ... of the split-output variant: 2x the workgroups (the gather is latency-bound)
constexpr int LTHREADS = 192;           // 3 waves: each produces 3 of the 9 window rows in phase 2

struct PlaneDev {
    const float* base;
    int h, w;
    float inv_scale;  // 1 / 2^level (exact)
    int target;
};

struct LookupArgs {
    PlaneDev planes[BFLOW_MAX_PLANES];
    float coef[BFLOW_MAX_TARGETS * BFLOW_MAX_DEGREE];
    int P, T, deg;
};

// pixel coordinate -> the coordinate grid_sample actually samples at.  bilinear_sampler normalises
// (utils.py:13-14: 2*x/(W-1) - 1) and grid_sample un-normalises ((g+1) * (W-1)/2, align_corners=True); the
// round trip is reproduced so that floor()/fraction see the same round-off as the reference.
__device__ __forceinline__ float roundtrip(float x, int size) {
    const float sm1 = (float)(size - 1);
    const float g = 2.0f * x / sm1 - 1.0f;
    return (g + 1.0f) * (sm1 / 2.0f);
}

// SPLIT = true: the features are written as the blocked split-fp16 tensor (2 planes (B, CBk, Prow, 32), value = hi + lo/2048) the
// conv engine consumes (conv_split.hip), instead of fp32 NCHW: the 81 values of a pixel are staged in LDS and written in
// memory order (lane = 8 consecutive channels of a pixel), which removes the NCHW -> blocked conversion launch per iteration.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <bool FUSED, int BATCH, bool SPLIT, int TL>
__global__ __launch_bounds__(LTHREADS) void corr_lookup_kernel(LookupArgs args, const float* __restrict__ src,
                                                              float* __restrict__ out, _Float16* __restrict__ oh,
                                                              _Float16* __restrict__ ol, int CBk, int Prow, int B, int h1, int w1) {
    __shared__ float patch[TL * PSTRIDE];
    __shared__ float stage[SPLIT ? TL * (WIN * WIN) : 1];
    __shared__ float s_cx[TL], s_cy[TL];
    __shared__ int s_ox[TL], s_oy[TL];

    const int tid = threadIdx.x;
    const int N = h1 * w1;
    const int p = blockIdx.y;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * TL;
    const PlaneDev pl = args.planes[p];

    if (tid < TL) {
        const int n = n0 + tid;
        float cx = 0.f, cy = 0.f;
        if (n < N) {
            if (FUSED) {
                // coords = coords0 + sum_i coef[t][i] * P_i   (bezier.py:185, raft.py:181); src = params (B, 2*deg, N)
                const int deg = args.deg;
                const float* pp = src + (long long)b * 2 * deg * N + n;
                const float* cf = args.coef + pl.target * deg;
                float fx = 0.f, fy = 0.f;
                for (int i = 0; i < deg; ++i) {
                    fx = fmaf(pp[(long long)i * N], cf[i], fx);
                    fy = fmaf(pp[(long long)(deg + i) * N], cf[i], fy);
                }
                const int y = n / w1, x = n - y * w1;
                cx = (float)x + fx;
                cy = (float)y + fy;
            } else {
                // src = coords (T, B, 2, N)
                const float* cc = src + ((long long)(pl.target * B + b) * 2) * N + n;
                cx = cc[0];
                cy = cc[N];
            }
            cx *= pl.inv_scale;  // corr.py:333  (division by 2^level == exact multiply)
            cy *= pl.inv_scale;
        }
        s_cx[tid] = cx;
        s_cy[tid] = cy;
        // patch origin from the (clamped) centre; far-away centres give an all-zero patch, as zero padding demands
        const float ccx = fminf(fmaxf(cx, -64.f), (float)pl.w + 64.f);
        const float ccy = fminf(fmaxf(cy, -64.f), (float)pl.h + 64.f);
        s_ox[tid] = (int)floorf(ccx) - (R + 1);
        s_oy[tid] = (int)floorf(ccy) - (R + 1);
    }
    __syncthreads();

    // ---- phase 1: gather 64 patches, 16 lanes per 12-wide patch row -------------------------------------
    // 48 row segments per thread, issued in batches of 12 independent loads (memory-level parallelism: the gather is
    // latency-bound, not bandwidth-bound, at batch 1) before any of them is written to LDS.
    {
        const int r = tid >> 4, c = tid & 15;                   // 12 row groups x 16 lanes: pass == pixel, group == patch row
        const int plane_sz = pl.h * pl.w;                       // < 2^24 (checked on the host): 32-bit offsets inside the slab
        const float* slab = pl.base + ((long long)b * N + n0) * plane_sz;
        constexpr int ROWS_PER_PASS = LTHREADS / 16;            // 12
        static_assert(ROWS_PER_PASS == PATCH && TL % BATCH == 0, "one pass gathers one 12x12 patch");
        const int npix = min(TL, N - n0);
#pragma unroll
        for (int p0 = 0; p0 < TL; p0 += BATCH) {
            float v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int pix = p0 + j;
                const int gy = s_oy[pix] + r, gx = s_ox[pix] + c;
                const bool ok = (c < PATCH) && (pix < npix) && gy >= 0 && gy < pl.h && gx >= 0 && gx < pl.w;
                // branch-free: out-of-plane lanes read element 0 of the slab (always mapped) and are zeroed afterwards, so
                // the BATCH loads issue back to back instead of one exec-masked branch + wait per load
                const int off = ok ? (pix * plane_sz + gy * pl.w + gx) : 0;
                const float ld = slab[off];
                v[j] = ok ? ld : 0.f;
            }
            if (c < PATCH) {
#pragma unroll
                for (int j = 0; j < BATCH; ++j) patch[(p0 + j) * PSTRIDE + r * PATCH + c] = v[j];
            }
        }
    }
    __syncthreads();

    // ---- phase 2: thread = (pixel, group g of LTHREADS/TL): window rows dy = g, g + G, ... (9 taps each) ----------------
    {
        constexpr int G = LTHREADS / TL;
        const int pix = tid % TL, wv = tid / TL;
        const int n = n0 + pix;
        const float cx = s_cx[pix], cy = s_cy[pix];
        const int ox = s_ox[pix], oy = s_oy[pix];
        // the 9 sample columns of the window: patch-relative west corner + east weight (shared by the 3 rows)
        int rx[WIN];
        float wx[WIN];
#pragma unroll
        for (int d = 0; d < WIN; ++d) {
            float ix = roundtrip(cx + (float)(d - R), pl.w);
            ix = fminf(fmaxf(ix, -1.0e4f), 1.0e4f);
            const float fx0 = floorf(ix);
            wx[d] = ix - fx0;
            const int ax = (int)fx0 - ox;
            rx[d] = (ax >= 0 && ax + 1 < PATCH) ? ax : -1;
        }
        const float* pp = patch + pix * PSTRIDE;
        float* o = out + ((long long)b * args.P * (WIN * WIN) + (long long)p * (WIN * WIN)) * N + n;
        if (n < N) {
#pragma unroll
        for (int jr = 0; jr < (WIN + G - 1) / G; ++jr) {
            const int ky = wv + jr * G;
            if (ky >= WIN) break;
            float iy = roundtrip(cy + (float)(ky - R), pl.h);
            iy = fminf(fmaxf(iy, -1.0e4f), 1.0e4f);
            const float fy0 = floorf(iy);
            const float ws = iy - fy0, wn = 1.f - ws;   // south (y0+1) / north (y0) row weights
            const int ay = (int)fy0 - oy;
            const bool yok = (ay >= 0 && ay + 1 < PATCH);
            const float* qrow = pp + (yok ? ay : 0) * PATCH;
#pragma unroll
            for (int kx = 0; kx < WIN; ++kx) {
                const float we = wx[kx], ww = 1.f - we;
                const bool ok = yok && rx[kx] >= 0;
                const float* q = qrow + (ok ? rx[kx] : 0);     // branch-free: clamped LDS address, select afterwards
                float v = q[0] * (ww * wn);
                v += q[1] * (we * wn);
                v += q[PATCH] * (ww * ws);
                v += q[PATCH + 1] * (we * ws);
                v = ok ? v : 0.f;
                if (SPLIT) stage[pix * (WIN * WIN) + ky * WIN + kx] = v;
                else o[(long long)(ky * WIN + kx) * N] = v;
            }
        }
        }
    }
    if (SPLIT) {
        __syncthreads();
        // channels [81 p, 81 p + 81) of the pixel tile: items = (channel block, pixel, 8-channel chunk), chunk fastest
        constexpr int NCH = WIN * WIN;
        const int c_first = p * NCH, cb_first = c_first >> 5, cb_last = (c_first + NCH - 1) >> 5;
        const int items = (cb_last - cb_first + 1) * TL * 4;
        for (int it = tid; it < items; it += LTHREADS) {
            const int chunk = it & 3, pix = (it >> 2) % TL, cb = cb_first + it / (4 * TL);
            const int n = n0 + pix;
            const int k0 = cb * 32 + chunk * 8 - c_first;          // window index of the chunk's first channel
            if (n >= N || k0 <= -8 || k0 >= NCH) continue;
            const long long o = (((long long)b * CBk + cb) * Prow + n) * 32 + chunk * 8;
            const float* sp = stage + pix * NCH;
            if (k0 >= 0 && k0 + 8 <= NCH) {
                half8 h8, l8;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    _Float16 hh, ll;
                    bflow::split1(sp[k0 + k], hh, ll);
                    h8[k] = hh;
                    l8[k] = ll;
                }
                *reinterpret_cast<half8*>(oh + o) = h8;
                *reinterpret_cast<half8*>(ol + o) = l8;
            } else {
                for (int k = 0; k < 8; ++k) {                   // ragged chunk shared with the neighbouring plane's workgroup
                    if (k0 + k < 0 || k0 + k >= NCH) continue;
                    _Float16 hh, ll;
                    bflow::split1(sp[k0 + k], hh, ll);
                    oh[o + k] = hh;
                    ol[o + k] = ll;
                }
            }
        }
    }
}

int fill_args(LookupArgs& a, const bflow_plane_t* planes, int P, int T) {
    BFLOW_REQUIRE(planes && P > 0 && T > 0, BFLOW_E_ARG, "corr_lookup: bad plane table");
    BFLOW_REQUIRE(P <= BFLOW_MAX_PLANES, BFLOW_E_LIMIT, "corr_lookup: %d planes > %d", P, BFLOW_MAX_PLANES);
    BFLOW_REQUIRE(T <= BFLOW_MAX_TARGETS, BFLOW_E_LIMIT, "corr_lookup: %d targets > %d", T, BFLOW_MAX_TARGETS);
    a.P = P;
    a.T = T;
    a.deg = 0;
    for (int p = 0; p < P; ++p) {
        BFLOW_REQUIRE(planes[p].base && planes[p].h > 0 && planes[p].w > 0 && planes[p].level >= 0 && planes[p].level < 16 &&
                          planes[p].target >= 0 && planes[p].target < T,
                      BFLOW_E_ARG, "corr_lookup: bad descriptor for plane %d", p);
        BFLOW_REQUIRE((long long)planes[p].h * planes[p].w * TILE < (1LL << 30), BFLOW_E_LIMIT, "corr_lookup: plane %d too large", p);
        a.planes[p].base = planes[p].base;
        a.planes[p].h = planes[p].h;
        a.planes[p].w = planes[p].w;
        a.planes[p].inv_scale = 1.0f / (float)(1 << planes[p].level);
        a.planes[p].target = planes[p].target;
    }
    return 0;
}

// Gather depth (independent loads in flight per thread) is a tunable: BFLOW_LOOKUP_BATCH in {16, 32, 64}.
int lookup_batch() {
    static int v = [] {
        const char* e = getenv("BFLOW_LOOKUP_BATCH");
        int b = e ? atoi(e) : 32;
        return (b == 16 || b == 32 || b == 64) ? b : 32;
    }();
    return v;
}

template <bool FUSED>
void launch_lookup(dim3 grid, hipStream_t s, const LookupArgs& a, const float* src, float* out, int B, int h1, int w1) {
    switch (lookup_batch()) {
        case 16: hipLaunchKernelGGL((corr_lookup_kernel<FUSED, 16, false, TILE>), grid, dim3(LTHREADS), 0, s, a, src, out, nullptr, nullptr, 0, 0, B, h1, w1); break;
        case 64: hipLaunchKernelGGL((corr_lookup_kernel<FUSED, 64, false, TILE>), grid, dim3(LTHREADS), 0, s, a, src, out, nullptr, nullptr, 0, 0, B, h1, w1); break;
        default: hipLaunchKernelGGL((corr_lookup_kernel<FUSED, 32, false, TILE>), grid, dim3(LTHREADS), 0, s, a, src, out, nullptr, nullptr, 0, 0, B, h1, w1); break;
    }
}

// ---- K6 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_pool2x2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           long long planes, int h, int w, int ho, int wo) {
    const long long total = planes * ho * wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long pl = idx / (ho * wo);
        const int rem = (int)(idx - pl * (ho * wo));
        const int y = rem / wo, x = rem - y * wo;
        const float* q = in + pl * h * w + (long long)(2 * y) * w + 2 * x;
        // F.avg_pool2d accumulates the window row-major and divides by the window size (4: exact)
        out[idx] = (((q[0] + q[1]) + q[w]) + q[w + 1]) * 0.25f;
    }
}

}  // namespace

extern "C" int bflow_corr_pool2x2(const float* in, float* out, long long planes, int h, int w, bflow_stream_t stream) {
    BFLOW_REQUIRE(in && out && planes > 0 && h >= 2 && w >= 2, BFLOW_E_ARG, "corr_pool2x2: bad arguments");
    const int ho = h / 2, wo = w / 2;
    const long long total = planes * ho * wo;
    hipLaunchKernelGGL(corr_pool2x2_kernel, dim3(bflow::stream_grid(total, 256) * 4), dim3(256), 0, (hipStream_t)stream, in, out,
                       planes, h, w, ho, wo);
    return bflow::launch_status("corr_pool2x2");
}

extern "C" int bflow_corr_lookup(const bflow_plane_t* planes, int P, const float* coords, float* out, int T, int B, int h1,
                                 int w1, bflow_stream_t stream) {
    LookupArgs a;
    int rc = fill_args(a, planes, P, T);
    if (rc) return rc;
    BFLOW_REQUIRE(coords && out && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup: bad arguments");
    dim3 grid(bflow::ceil_div((long long)h1 * w1, TILE), P, B);
    launch_lookup<false>(grid, (hipStream_t)stream, a, coords, out, B, h1, w1);
    return bflow::launch_status("corr_lookup");
}

extern "C" int bflow_corr_lookup_bezier(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T,
                                        int deg, float* out, int B, int h1, int w1, bflow_stream_t stream) {
    LookupArgs a;
    int rc = fill_args(a, planes, P, T);
    if (rc) return rc;
    BFLOW_REQUIRE(params && coef && out && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_bezier: bad arguments");
    BFLOW_REQUIRE(deg >= 1 && deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "corr_lookup_bezier: degree %d", deg);
    a.deg = deg;
    for (int i = 0; i < T * deg; ++i) a.coef[i] = coef[i];
    dim3 grid(bflow::ceil_div((long long)h1 * w1, TILE), P, B);
    launch_lookup<true>(grid, (hipStream_t)stream, a, params, out, B, h1, w1);
    return bflow::launch_status("corr_lookup_bezier");
}

namespace bflow {
int pool_tiled_launch(const void* in, void* out, long long planes, int h, int w, bool f16, hipStream_t stream);
int lookup_tile_launch(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg, void* out_hi, void* out_lo,
                       int channel_blocks, int rows_per_image, int B, int h1, int w1, bool f16_planes, hipStream_t stream,
                       const Im2colArgs* rider = nullptr);
}

extern "C" int bflow_corr_pool2x2_tiled(const float* in, float* out, long long planes, int h, int w, bflow_stream_t stream) {
    return bflow::pool_tiled_launch(in, out, planes, h, w, false, (hipStream_t)stream);
}

extern "C" int bflow_corr_pool2x2_tiled_f16(const void* in, void* out, long long planes, int h, int w, bflow_stream_t stream) {
    return bflow::pool_tiled_launch(in, out, planes, h, w, true, (hipStream_t)stream);
}

extern "C" int bflow_corr_lookup_bezier_split_tiled_f16(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                                        void* out_hi, void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1,
                                                        bflow_stream_t stream) {
    return bflow::lookup_tile_launch(planes, P, params, coef, T, deg, out_hi, out_lo, channel_blocks, rows_per_image, B, h1, w1, true,
                                     (hipStream_t)stream);
}

extern "C" int bflow_corr_lookup_bezier_split_tiled(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                                    void* out_hi, void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1,
                                                    bflow_stream_t stream) {
    return bflow::lookup_tile_launch(planes, P, params, coef, T, deg, out_hi, out_lo, channel_blocks, rows_per_image, B, h1, w1, false,
                                     (hipStream_t)stream);
}

// Look-up (tiled planes, fp32 or fp16) + bflow_im2col_small of the same Bezier parameters as ONE launch: the two first kernels of an update
// iteration (update.py:117 / corr.py look-up and the 7x7 `convf1` input of update.py:91) both read `params` and nothing of each other.
extern "C" int bflow_corr_lookup_im2col(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg, void* out_hi,
                                        void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1, int f16_planes, void* col_hi,
                                        void* col_lo, int KH, int KW, int pad_h, int pad_w, int col_rows_per_image, bflow_stream_t stream) {
    BFLOW_REQUIRE(col_hi && col_lo && KH > 0 && KW > 0 && deg >= 1 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_im2col: bad im2col arguments");
    const int Pc = col_rows_per_image > 0 ? col_rows_per_image : h1 * w1;
    BFLOW_REQUIRE(Pc >= h1 * w1, BFLOW_E_ARG, "corr_lookup_im2col: col_rows_per_image < h1*w1");
    BFLOW_REQUIRE((long long)((KH * KW * 2 * deg + 31) / 32) * Pc * 4 < (1LL << 31), BFLOW_E_LIMIT, "corr_lookup_im2col: image too large for 32-bit item indices");
    bflow::Im2colArgs m{params, (_Float16*)col_hi, (_Float16*)col_lo, 2 * deg, h1, w1, KH, KW, pad_h, pad_w, (KH * KW * 2 * deg + 31) / 32, Pc};
    return bflow::lookup_tile_launch(planes, P, params, coef, T, deg, out_hi, out_lo, channel_blocks, rows_per_image, B, h1, w1, f16_planes != 0,
                                     (hipStream_t)stream, &m);
}

extern "C" int bflow_corr_lookup_bezier_split(const bflow_plane_t* planes, int P, const float* params, const float* coef, int T, int deg,
                                              void* out_hi, void* out_lo, int channel_blocks, int rows_per_image, int B, int h1, int w1,
                                              bflow_stream_t stream) {
    LookupArgs a;
    int rc = fill_args(a, planes, P, T);
    if (rc) return rc;
    BFLOW_REQUIRE(params && coef && out_hi && out_lo && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_bezier_split: bad arguments");
    BFLOW_REQUIRE(deg >= 1 && deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "corr_lookup_bezier_split: degree %d", deg);
    BFLOW_REQUIRE(channel_blocks * 32 >= P * (2 * BFLOW_LOOKUP_RADIUS + 1) * (2 * BFLOW_LOOKUP_RADIUS + 1) && rows_per_image >= h1 * w1,
                  BFLOW_E_ARG, "corr_lookup_bezier_split: output too small");
    a.deg = deg;
    for (int i = 0; i < T * deg; ++i) a.coef[i] = coef[i];
    dim3 grid(bflow::ceil_div((long long)h1 * w1, TILE_SPLIT), P, B);
    hipLaunchKernelGGL((corr_lookup_kernel<true, TILE_SPLIT, true, TILE_SPLIT>), grid, dim3(LTHREADS), 0, (hipStream_t)stream, a, params, nullptr, (_Float16*)out_hi,
                       (_Float16*)out_lo, channel_blocks, rows_per_image, B, h1, w1);
    return bflow::launch_status("corr_lookup_bezier_split");
}
