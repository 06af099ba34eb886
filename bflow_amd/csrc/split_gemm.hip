// Split-fp16 MFMA engine: fp32-class accuracy at fp16 matrix-core rate.
//
// An fp32 value x is carried as two fp16 numbers   x ~= hi + lo * 2^-11,   hi = fp16(x), lo = fp16((x - hi) * 2^11)
// (the 2^11 pre-scale keeps `lo` in fp16's normal range, so the pair holds ~22 significant bits).  A dot product is
// evaluated with three v_mfma_f32_32x32x16_f16 chains accumulating in fp32:
//       sum a*b ~= sum(a_hi*b_hi)  +  2^-11 * ( sum(a_hi*b_lo) + sum(a_lo*b_hi) )            (a_lo*b_lo ~ 2^-22, dropped)
// i.e. 3 fp16 MFMAs (3 x 1/16 of the fp32-MFMA time) instead of one fp32 MFMA: 5.3x the fp32 matrix peak, with a
// relative error per product of ~2^-22 (fp32 itself: 2^-24).  On MI355X this moves the all-pairs correlation build
// (K5) from fp32-MFMA-bound (0.30 ms at peak for C2) to HBM-write-bound (0.05-0.06 ms), which is what the north star
// asks of the correlation kernel.
//
//   bflow_split_pack     : (R, D, N) fp32, pixel-contiguous  ->  hi/lo (R, Np, D) fp16, feature-contiguous (= MFMA operand
//                          order: lane holds 8 consecutive k), rows N..Np zero (Np = N rounded up to the 128 tile)
//   bflow_corr_build_split: out[t,b,i,j] = <f1[.,b,i,:], f2[t,b,j,:]> / sqrt(D) on the packed operands, 128x128 block tile,
//                          4 waves x (2x2 MFMA 32x32x16 tiles) x {hh, cross} accumulators, BK = 32, double-buffered LDS
//                          (80-B padded rows: conflict-free ds_read_b128 fragments), one barrier per k-tile.
#include <cstdlib>
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float LO_INV = 1.0f / 2048.0f;

using bflow::split1;   // common.h: saturating hi/lo split

// ---------------------------------------------------------------------------------------------------------------
// pack: transpose (D, N) -> (Np, D) and split.  Block = 64 pixels x 64 features.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ src, _Float16* __restrict__ hi,
                                                         _Float16* __restrict__ lo, int D, int N, int Np) {
    __shared__ float tile[64][65];
    const int r = blockIdx.z;
    const int n0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const float* s = src + (long long)r * D * N;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 rows of 64 lanes
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = d0 + ty + 4 * i, n = n0 + tx;
        tile[ty + 4 * i][tx] = (d < D && n < N) ? s[(long long)d * N + n] : 0.f;
    }
    __syncthreads();
    // thread -> (pixel = tid/4 + 64*?, 8 consecutive features): 64 pixels x 8 chunks of 8 features = 512 chunks, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = threadIdx.x + 256 * i;
        const int pn = chunk >> 3, dc = (chunk & 7) * 8;
        const int n = n0 + pn;
        if (n < Np && d0 + dc < D) {
            half8 h, l;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                _Float16 a, b;
                split1(tile[dc + k][pn], a, b);
                h[k] = a;
                l[k] = b;
            }
            // k-blocked layout (R, D/32, Np, 32): a 32-deep k-tile of consecutive rows is one contiguous run of 64-B rows
            const long long o = (((long long)r * (D / 32) + (d0 + dc) / 32) * Np + n) * 32 + ((d0 + dc) & 31);
            *reinterpret_cast<half8*>(hi + o) = h;
            *reinterpret_cast<half8*>(lo + o) = l;
        }
    }
}

constexpr int BK = 32;

// ---------------------------------------------------------------------------------------------------------------
// v2: 256x128 block tile, 8 waves (4x2, wave tile 64x64), direct-to-LDS loads (global_load_lds_dwordx4) into a 3-stage
// ring, XOR-swizzled 64-B LDS rows (no padding is possible with LDS-DMA: the image is lane-linear, so the swizzle is
// applied to the per-lane SOURCE address and again on the fragment read), counted vmcnt + one raw s_barrier per k-tile.
// Two k-tiles of loads stay in flight across every barrier; no VGPRs are spent on staging.
// ---------------------------------------------------------------------------------------------------------------
constexpr int V2_BM = 256, V2_BN = 128, V2_T = 512, V2_STAGES = 3;
constexpr int V2_AH = 0, V2_AL = V2_BM * 64, V2_BH = 2 * V2_BM * 64, V2_BL = 2 * V2_BM * 64 + V2_BN * 64;
constexpr int V2_STAGE = 2 * V2_BM * 64 + 2 * V2_BN * 64;   // 48 KB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(V2_T, 2) void corr_build_split_v2_kernel(const _Float16* __restrict__ f1h, const _Float16* __restrict__ f1l,
                                                                      const _Float16* __restrict__ f2h, const _Float16* __restrict__ f2l,
                                                                      float* __restrict__ out, int B, int D, int N, int Np,
                                                                      long long f1_tstride, float sqrt_d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 3 x 48 KB ring (the ONLY shared object in this kernel)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tb = blockIdx.y, t = tb / B, b = tb - t * B;
    int i0, j0;
    {
        const int nwg = gridDim.x, tj = (N + V2_BN - 1) / V2_BN, ti = (N + V2_BM - 1) / V2_BM;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = nwg >> 3, r = nwg & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int panel = wg / (8 * ti), rem = wg - panel * 8 * ti;
        const int pw = min(8, tj - panel * 8);
        i0 = (rem / pw) * V2_BM;
        j0 = (panel * 8 + rem % pw) * V2_BN;
    }
    // rows of A beyond Np do not exist (Np is a multiple of 128, the A tile is 256 rows): clamp the source row, the
    // corresponding outputs are never stored
    const int a_rows = Np - i0;   // >= 128

    // operands are k-blocked: (R, D/32, Np, 32) -> element (row, k) of matrix R lives at ((R*D/32 + k/32)*Np + row)*32 + k%32
    const long long aoff = t * f1_tstride + (long long)b * Np * D;
    const long long boff = (long long)tb * Np * D + (long long)j0 * 32;
    const _Float16 *Ah = f1h + aoff, *Al = f1l + aoff, *Bh = f2h + boff, *Bl = f2l + boff;
    const long long kstep = (long long)Np * 32;   // elements between consecutive 32-deep k-tiles

    // per-lane source coordinates inside a 16-row x 64-B unit: row = lane/4, logical chunk = slot ^ ((row>>2)&3)
    const int urow = lane >> 2;
    const int uchunk = (lane & 3) ^ ((lane >> 4) & 3);
    // A units of this wave: rows (wave*2 + j)*16 + urow, j = 0,1 ; B unit: rows wave*16 + urow
    int arow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 16 + urow;
        arow[j] = i0 + (r < a_rows ? r : a_rows - 1);
    }
    const int brow = wave * 16 + urow;
    const long long asrc0 = (long long)arow[0] * 32 + uchunk * 8, asrc1 = (long long)arow[1] * 32 + uchunk * 8;
    const long long bsrc = (long long)brow * 32 + uchunk * 8;

#define V2_ISSUE(SLOT, K0)                                                                                       \
    {                                                                                                            \
        char* sb = lds + (SLOT) * V2_STAGE;                                                                      \
        __builtin_amdgcn_global_load_lds((gptr_t)(Ah + asrc0 + (K0)), (lptr_t)(sb + V2_AH + (wave * 2 + 0) * 1024), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((gptr_t)(Ah + asrc1 + (K0)), (lptr_t)(sb + V2_AH + (wave * 2 + 1) * 1024), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((gptr_t)(Al + asrc0 + (K0)), (lptr_t)(sb + V2_AL + (wave * 2 + 0) * 1024), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((gptr_t)(Al + asrc1 + (K0)), (lptr_t)(sb + V2_AL + (wave * 2 + 1) * 1024), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((gptr_t)(Bh + bsrc + (K0)), (lptr_t)(sb + V2_BH + wave * 1024), 16, 0, 0);            \
        __builtin_amdgcn_global_load_lds((gptr_t)(Bl + bsrc + (K0)), (lptr_t)(sb + V2_BL + wave * 1024), 16, 0, 0);            \
    }

    f32x16 hh[2][2], xx[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hh[m][n][r] = 0.f;
                xx[m][n][r] = 0.f;
            }

    const int nk = D / BK;
    const int l31 = lane & 31, kh = lane >> 5;
    // fragment read offsets: row R, logical chunk c -> R*64 + ((c ^ ((R>>2)&3)) * 16); (R>>2)&3 == (l31>>2)&3 (tile bases are x32)
    const int sw = (l31 >> 2) & 3;
    int aro[2], bro[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) aro[m] = (wm * 64 + m * 32 + l31) * 64;
#pragma unroll
    for (int n = 0; n < 2; ++n) bro[n] = (wn * 64 + n * 32 + l31) * 64;

    V2_ISSUE(0, 0)
    V2_ISSUE(1, (1 < nk ? 1 : 0) * kstep)

    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // this wave's loads of k-tile kt have landed (kt+1 may be in flight)
        __builtin_amdgcn_s_barrier();                      // ... and everybody else's; everybody is also done reading slot (kt+2)%3
        __builtin_amdgcn_sched_barrier(0);
        {
            const int kn = (kt + 2 < nk) ? (kt + 2) : 0;   // tail: keep the instruction count per iteration constant
            const int slot = (kt + 2) % V2_STAGES;
            V2_ISSUE(slot, kn * kstep)
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* cur = lds + (kt % V2_STAGES) * V2_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
            const int co = ((ks * 2 + kh) ^ sw) * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = *reinterpret_cast<const half8*>(cur + V2_AH + aro[m] + co);
                al[m] = *reinterpret_cast<const half8*>(cur + V2_AL + aro[m] + co);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[n] = *reinterpret_cast<const half8*>(cur + V2_BH + bro[n] + co);
                bl[n] = *reinterpret_cast<const half8*>(cur + V2_BL + bro[n] + co);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) hh[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], hh[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) xx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], xx[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) xx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], xx[m][n], 0, 0, 0);
        }
    }
#undef V2_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail prefetches before the block may exit

    float* O = out + (long long)tb * N * N;
    const bool interior = (i0 + V2_BM <= N) && (j0 + V2_BN <= N);   // block-uniform
    if (interior) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    O[(long long)row * N + j0 + wn * 64 + n * 32 + l31] = (hh[m][n][r] + xx[m][n][r] * LO_INV) / sqrt_d;
            }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int col = j0 + wn * 64 + n * 32 + l31;
                    if (row < N && col < N) O[(long long)row * N + col] = (hh[m][n][r] + xx[m][n][r] * LO_INV) / sqrt_d;
                }
            }
    }
}

}  // namespace

namespace bflow {
bool corr_stream_supported(int T, int B, int D, int N, int Np);
int corr_stream_launch(const void* f1_hi, const void* f1_lo, const void* f2_hi, const void* f2_lo, void* out, int T, int B, int D, int N, int Np,
                       long long f1_target_stride, int plane_h, int plane_w, int arithmetic, bool out_fp16, void* pool_out, const int* pool_index,
                       hipStream_t stream);
}

extern "C" int bflow_split_pack(const float* src, void* hi, void* lo, int R, int D, int N, int Np, bflow_stream_t stream) {
    BFLOW_REQUIRE(src && hi && lo && R > 0 && D > 0 && N > 0, BFLOW_E_ARG, "split_pack: bad arguments");
    BFLOW_REQUIRE(D % 8 == 0 && Np >= N && Np % 64 == 0, BFLOW_E_ARG, "split_pack: D %% 8 and Np %% 64 must be 0 (D=%d Np=%d)", D, Np);
    dim3 grid(Np / 64, bflow::ceil_div(D, 64), R);
    hipLaunchKernelGGL(split_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)hi, (_Float16*)lo, D, N, Np);
    return bflow::launch_status("split_pack");
}

extern "C" int bflow_corr_build_split(const void* f1_hi, const void* f1_lo, const void* f2_hi, const void* f2_lo, float* out, int T, int B,
                                      int D, int N, int Np, long long f1_target_stride, bflow_stream_t stream) {
    BFLOW_REQUIRE(f1_hi && f1_lo && f2_hi && f2_lo && out, BFLOW_E_ARG, "corr_build_split: null pointer");
    BFLOW_REQUIRE(T > 0 && B > 0 && N > 0 && D > 0 && D % BK == 0 && Np >= N && Np % 128 == 0, BFLOW_E_ARG,
                  "corr_build_split: bad sizes T=%d B=%d D=%d N=%d Np=%d", T, B, D, N, Np);
    // D in {64, 128, 256}: the A-stationary streaming kernel (corr_stream.hip); anything else: the 256x128 tile kernel below
    static const bool force_tile = getenv("BFLOW_CORR_TILE_KERNEL") != nullptr;   // A/B timing only (tools/)
    if (!force_tile && bflow::corr_stream_supported(T, B, D, N, Np))
        return bflow::corr_stream_launch(f1_hi, f1_lo, f2_hi, f2_lo, out, T, B, D, N, Np, f1_target_stride, 0, 0, 0, false, nullptr, nullptr, (hipStream_t)stream);
    BFLOW_REQUIRE((long long)T * B <= 65535, BFLOW_E_LIMIT, "corr_build_split: T*B too large");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_build_split_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              V2_STAGES * V2_STAGE);   // 144 KB of dynamic LDS; idempotent, per device
    dim3 grid2(bflow::ceil_div(N, V2_BN) * bflow::ceil_div(N, V2_BM), T * B);
    hipLaunchKernelGGL(corr_build_split_v2_kernel, grid2, dim3(V2_T), V2_STAGES * V2_STAGE, (hipStream_t)stream, (const _Float16*)f1_hi,
                       (const _Float16*)f1_lo, (const _Float16*)f2_hi, (const _Float16*)f2_lo, out, B, D, N, Np, f1_target_stride,
                       sqrtf((float)D));
    return bflow::launch_status("corr_build_split");
}

// The same volume with TILED planes (the product path of inference): plane (h x w) of query pixel i is stored as ceil(h/4) x ceil(w/8)
// tiles of 4 x 8 elements (128 B = one cache line; tile-row-major, row-major inside a tile), so that the 12 x 12 neighbourhood the
// look-up gathers is ~9 full lines instead of 12 x 1.4 partial ones.  out: (T, B, N, tiles * 32) fp32; pad positions of edge tiles
// hold finite values.  Only the streaming kernel writes this layout: D in {64, 128, 256}, else BFLOW_E_ARG (use the row-major entry).
extern "C" int bflow_corr_build_split_tiled(const void* f1_hi, const void* f1_lo, const void* f2_hi, const void* f2_lo, float* out, int T, int B,
                                            int D, int h, int w, int Np, long long f1_target_stride, bflow_stream_t stream) {
    BFLOW_REQUIRE(f1_hi && f1_lo && f2_hi && f2_lo && out, BFLOW_E_ARG, "corr_build_split_tiled: null pointer");
    const int N = h * w;
    BFLOW_REQUIRE(T > 0 && B > 0 && h > 0 && w > 0 && Np >= N && Np % 128 == 0, BFLOW_E_ARG, "corr_build_split_tiled: bad sizes T=%d B=%d h=%d w=%d Np=%d", T,
                  B, h, w, Np);
    BFLOW_REQUIRE(bflow::corr_stream_supported(T, B, D, N, Np), BFLOW_E_ARG, "corr_build_split_tiled: needs D in {64, 128, 256} and < 2 GiB slabs (D=%d N=%d)",
                  D, N);
    return bflow::corr_stream_launch(f1_hi, f1_lo, f2_hi, f2_lo, out, T, B, D, N, Np, f1_target_stride, h, w, 0, false, nullptr, nullptr, (hipStream_t)stream);
}

// BASELINE configs[4] ("fp16 MFMA correlation ... HBM-bound 4D volume stress"): the volume from PLAIN fp16 operands (the hi planes of the
// split tensors: features rounded to fp16), one fp16 MFMA pass with fp32 accumulation, stored as fp16 tiled planes: a third of the
// matrix-core work and half of the bytes of bflow_corr_build_split_tiled, at fp16 accuracy (2^-11 per operand and per stored value).
// f1_hi (B | T*B, D/32, Np, 32) fp16, f2_hi (T*B, D/32, Np, 32) fp16, out (T, B, N, tiles*32) fp16.  D in {128, 256}.
extern "C" int bflow_corr_build_f16_tiled(const void* f1_hi, const void* f2_hi, void* out, int T, int B, int D, int h, int w, int Np,
                                          long long f1_target_stride, bflow_stream_t stream) {
    BFLOW_REQUIRE(f1_hi && f2_hi && out, BFLOW_E_ARG, "corr_build_f16_tiled: null pointer");
    const int N = h * w;
    BFLOW_REQUIRE(T > 0 && B > 0 && h > 0 && w > 0 && Np >= N && Np % 128 == 0, BFLOW_E_ARG, "corr_build_f16_tiled: bad sizes T=%d B=%d h=%d w=%d Np=%d", T, B,
                  h, w, Np);
    BFLOW_REQUIRE((D == 128 || D == 256) && bflow::corr_stream_supported(T, B, D, N, Np), BFLOW_E_ARG,
                  "corr_build_f16_tiled: needs D in {128, 256} and < 2 GiB slabs (D=%d N=%d)", D, N);
    return bflow::corr_stream_launch(f1_hi, nullptr, f2_hi, nullptr, out, T, B, D, N, Np, f1_target_stride, h, w, 1, true, nullptr, nullptr, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// x8 planes for the fp8 cross terms of bflow_corr_build_tiled (arithmetic = 2): per (row, 32-channel block) the 64 bytes
// [e4m3(hi) x 32 | e4m3(lo) x 32] of the split pair (OCP e4m3, round to nearest even, saturating).  One thread = 8 channels.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void split_to_x8_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, unsigned char* __restrict__ x8,
                                                           long long rows) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;          // (row, 8-channel group)
    if (i >= rows * 4) return;
    const long long row = i >> 2;
    const int g = (int)(i & 3);
    typedef _Float16 half8v __attribute__((ext_vector_type(8)));
    const half8v h = *reinterpret_cast<const half8v*>(hi + row * 32 + g * 8), l = *reinterpret_cast<const half8v*>(lo + row * 32 + g * 8);
    // v_cvt_pk_fp8_f32 does NOT saturate by itself (no clamp modifier, MODE.FP16_OVFL clear): |x| > 448 would become the e4m3 NaN 0x7F and
    // poison a whole row / column of the volume.  v_med3_f32 clamps to +-448 first; a NaN input stays NaN (med3 of (NaN, -448, 448) with
    // IEEE mode returns the NaN-propagating min/max result, and the split planes of a NaN feature are NaN in the fp16 term anyway).
    auto sat = [](float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); };
    auto pack4 = [&](float a, float b, float c, float d) {
        int v = __builtin_amdgcn_cvt_pk_fp8_f32(sat(a), sat(b), 0, false);
        return __builtin_amdgcn_cvt_pk_fp8_f32(sat(c), sat(d), v, true);
    };
    const int2 ph = make_int2(pack4((float)h[0], (float)h[1], (float)h[2], (float)h[3]), pack4((float)h[4], (float)h[5], (float)h[6], (float)h[7]));
    const int2 pl = make_int2(pack4((float)l[0], (float)l[1], (float)l[2], (float)l[3]), pack4((float)l[4], (float)l[5], (float)l[6], (float)l[7]));
    *reinterpret_cast<int2*>(x8 + row * 64 + g * 8) = ph;
    *reinterpret_cast<int2*>(x8 + row * 64 + 32 + g * 8) = pl;
}
}  // namespace

extern "C" int bflow_split_to_x8(const void* hi, const void* lo, void* x8, long long rows, bflow_stream_t stream) {
    BFLOW_REQUIRE(hi && lo && x8 && rows > 0, BFLOW_E_ARG, "split_to_x8: bad arguments");
    hipLaunchKernelGGL(split_to_x8_kernel, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)hi,
                       (const _Float16*)lo, (unsigned char*)x8, rows);
    return bflow::launch_status("split_to_x8");
}

// The tiled volume with every arithmetic / storage combination of the streaming kernel (csrc/corr_stream.hip):
//   arithmetic 0: split pairs, three fp16 MFMA passes (f*_second = lo planes)       = bflow_corr_build_split_tiled when out_fp16 = 0
//              1: plain fp16 operands, one pass (f*_second ignored, may be null)     = bflow_corr_build_f16_tiled when out_fp16 = 1
//              2: hi*hi on fp16 + both cross terms on the fp8 rate (f*_second = x8 planes written by bflow_split_to_x8)
//   out_fp16    : the volume is stored as fp16 tiled planes instead of fp32.
// D in {128, 256} (and 64 for arithmetic 0 with an fp32 volume).
//   pool_out    : optional (T1, B, N, tiles1*32) buffer of the volume's element type, pool_index (HOST, T ints): row of target t in it or -1.  The
//                 2 x 2 mean of the level-0 planes of those targets (K6, level 0 -> 1: corr.py:108-125,297-305; F.avg_pool2d's summation order,
//                 floor on odd sizes, pad positions zero) is written by the same launch -- bflow_corr_pool2x2_tiled no longer re-reads the
//                 level-0 planes it was just handed.  (split | split8, fp32) and (fp16, fp16) only; D in {128, 256}; T <= 8.
extern "C" int bflow_corr_build_tiled(const void* f1_hi, const void* f1_second, const void* f2_hi, const void* f2_second, void* out, int T, int B,
                                      int D, int h, int w, int Np, long long f1_target_stride, int arithmetic, int out_fp16, void* pool_out,
                                      const int* pool_index, bflow_stream_t stream) {
    BFLOW_REQUIRE(f1_hi && f2_hi && out && arithmetic >= 0 && arithmetic <= 2, BFLOW_E_ARG, "corr_build_tiled: bad arguments");
    BFLOW_REQUIRE(arithmetic == 1 || (f1_second && f2_second), BFLOW_E_ARG, "corr_build_tiled: arithmetic %d needs the second operand planes", arithmetic);
    const int N = h * w;
    BFLOW_REQUIRE(T > 0 && B > 0 && h > 0 && w > 0 && Np >= N && Np % 128 == 0, BFLOW_E_ARG, "corr_build_tiled: bad sizes T=%d B=%d h=%d w=%d Np=%d", T, B, h, w, Np);
    BFLOW_REQUIRE(bflow::corr_stream_supported(T, B, D, N, Np) && (D != 64 || (arithmetic == 0 && !out_fp16)), BFLOW_E_ARG,
                  "corr_build_tiled: needs D in {128, 256} (64: split / fp32 only) and < 2 GiB slabs (D=%d N=%d)", D, N);
    BFLOW_REQUIRE(!pool_out || (pool_index && T <= 8 && D != 64 && ((arithmetic != 1 && !out_fp16) || (arithmetic == 1 && out_fp16))), BFLOW_E_ARG,
                  "corr_build_tiled: the fused level-1 pooling needs (split | split8, fp32 volume) or (fp16, fp16 volume), D in {128, 256}, T <= 8");
    const int rc = bflow::corr_stream_launch(f1_hi, f1_second, f2_hi, f2_second, out, T, B, D, N, Np, f1_target_stride, h, w, arithmetic, out_fp16 != 0,
                                             pool_out, pool_index, (hipStream_t)stream);
    BFLOW_REQUIRE(rc != BFLOW_E_ARG, BFLOW_E_ARG, "corr_build_tiled: unsupported combination (D=%d, arithmetic=%d, out_fp16=%d, pool=%d)", D, arithmetic, out_fp16, pool_out != nullptr);
    return rc;
}
