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
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float LO_SCALE = 2048.0f;           // 2^11
constexpr float LO_INV = 1.0f / 2048.0f;

// The matrix cores flush fp16 subnormal INPUTS, so `hi` must never be subnormal: below 2^-14 the whole value goes into
// the (pre-scaled) `lo` term, which stays normal down to 2^-25 (anything smaller contributes < 3e-8 absolute).
__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    const float h = (fabsf(x) >= 6.103515625e-05f) ? (float)(_Float16)x : 0.0f;
    hi = (_Float16)h;
    lo = (_Float16)((x - h) * LO_SCALE);
}

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
            const long long o = ((long long)r * Np + n) * D + d0 + dc;
            *reinterpret_cast<half8*>(hi + o) = h;
            *reinterpret_cast<half8*>(lo + o) = l;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NT GEMM on split operands
// ---------------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                      // LDS row pitch in bytes (64 B of data + 16 B pad)
constexpr int ARR = BM * ROWB;                // bytes of one 128 x 32 half tile
constexpr int STAGE = 4 * ARR;                // A_hi, A_lo, B_hi, B_lo
constexpr int GT = 256;

// Register staging of one k-tile (4 arrays x 128 rows x 32 halves = 4 x 512 chunks of 16 B; thread -> chunks tid, tid+256).
// Named scalars (token pasting), not arrays: hipcc keeps indexed staging arrays in scratch.
#define BFLOW_G2R(P, K0)                                                                          \
    {                                                                                             \
        const long long off0 = (long long)(tid >> 2) * D + (K0) + (tid & 3) * 8;                  \
        const long long off1 = off0 + 64LL * D;                                                   \
        P##0 = *reinterpret_cast<const uint4*>(Ah + off0);                                        \
        P##1 = *reinterpret_cast<const uint4*>(Al + off0);                                        \
        P##2 = *reinterpret_cast<const uint4*>(Bh + off0);                                        \
        P##3 = *reinterpret_cast<const uint4*>(Bl + off0);                                        \
        P##4 = *reinterpret_cast<const uint4*>(Ah + off1);                                        \
        P##5 = *reinterpret_cast<const uint4*>(Al + off1);                                        \
        P##6 = *reinterpret_cast<const uint4*>(Bh + off1);                                        \
        P##7 = *reinterpret_cast<const uint4*>(Bl + off1);                                        \
    }
#define BFLOW_R2S(P, LDSBASE)                                                                     \
    {                                                                                             \
        char* w0 = (LDSBASE) + (tid >> 2) * ROWB + (tid & 3) * 16;                                \
        char* w1 = w0 + 64 * ROWB;                                                                \
        *reinterpret_cast<uint4*>(w0 + 0 * ARR) = P##0;                                           \
        *reinterpret_cast<uint4*>(w0 + 1 * ARR) = P##1;                                           \
        *reinterpret_cast<uint4*>(w0 + 2 * ARR) = P##2;                                           \
        *reinterpret_cast<uint4*>(w0 + 3 * ARR) = P##3;                                           \
        *reinterpret_cast<uint4*>(w1 + 0 * ARR) = P##4;                                           \
        *reinterpret_cast<uint4*>(w1 + 1 * ARR) = P##5;                                           \
        *reinterpret_cast<uint4*>(w1 + 2 * ARR) = P##6;                                           \
        *reinterpret_cast<uint4*>(w1 + 3 * ARR) = P##7;                                           \
    }

__global__ __launch_bounds__(GT, 2) void corr_build_split_kernel(const _Float16* __restrict__ f1h, const _Float16* __restrict__ f1l,
                                                              const _Float16* __restrict__ f2h, const _Float16* __restrict__ f2l,
                                                              float* __restrict__ out, int B, int D, int N, int Np,
                                                              long long f1_tstride, float sqrt_d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 stages x 40 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tb = blockIdx.y, t = tb / B, b = tb - t * B;
    // XCD-aware tile order (blocks are dealt round-robin to the 8 XCDs, each with a private 4 MB L2): give every XCD one
    // contiguous chunk of the tile sequence, and order the sequence in panels of 8 j-tiles so that a chunk keeps its
    // B panel (8 x 128 KB) L2-resident while the A tiles stream through once.
    int i0, j0;
    {
        const int nwg = gridDim.x, tj = (N + BN - 1) / BN, ti = (N + BM - 1) / BM;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = nwg >> 3, r = nwg & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int panel = wg / (8 * ti), rem = wg - panel * 8 * ti;
        const int pw = min(8, tj - panel * 8);            // width of this panel in j-tiles
        i0 = (rem / pw) * BM;
        j0 = (panel * 8 + rem % pw) * BN;
    }

    const long long aoff = t * f1_tstride + ((long long)b * Np + i0) * D;
    const long long boff = ((long long)tb * Np + j0) * D;
    const _Float16 *Ah = f1h + aoff, *Al = f1l + aoff, *Bh = f2h + boff, *Bl = f2l + boff;

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

    // Software pipeline: LDS holds k-tile kt (double buffered), register set st0/st1 hold tiles kt+1 / kt+2 in flight, so a
    // global load has two k-tiles of MFMA work (~1500 cycles) plus the co-resident block to hide behind.
    const int nk = D / BK;                     // even (D % 64 == 0 is checked on the host)
    const int l31 = lane & 31, kh = lane >> 5;
    uint4 sa0, sa1, sa2, sa3, sa4, sa5, sa6, sa7, sb0, sb1, sb2, sb3, sb4, sb5, sb6, sb7;
    BFLOW_G2R(sa, 0)
    BFLOW_R2S(sa, lds)
    BFLOW_G2R(sa, BK)
    BFLOW_G2R(sb, (2 < nk ? 2 : 0) * BK)
    __syncthreads();

#define BFLOW_COMPUTE(CUR)                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                   \
        half8 ah[2], al[2], bh[2], bl[2];                                                                \
        const int ko = ks * 32 + kh * 16;                                                                \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                  \
            const int o = (wm * 64 + m * 32 + l31) * ROWB + ko;                                          \
            ah[m] = *reinterpret_cast<const half8*>((CUR) + 0 * ARR + o);                                \
            al[m] = *reinterpret_cast<const half8*>((CUR) + 1 * ARR + o);                                \
        }                                                                                                \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                  \
            const int o = (wn * 64 + n * 32 + l31) * ROWB + ko;                                          \
            bh[n] = *reinterpret_cast<const half8*>((CUR) + 2 * ARR + o);                                \
            bl[n] = *reinterpret_cast<const half8*>((CUR) + 3 * ARR + o);                                \
        }                                                                                                \
        /* three sweeps over the 4 accumulator tiles: consecutive MFMAs never touch the same accumulator (a dependent   \
           32x32x16 MFMA would stall its 16-pass latency) */                                                           \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                    \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                hh[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], hh[m][n], 0, 0, 0);     \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                    \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                xx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], xx[m][n], 0, 0, 0);     \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                    \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                xx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], xx[m][n], 0, 0, 0);     \
    }

    for (int kt = 0; kt < nk; kt += 2) {
        // even step: compute tile kt from buffer 0; st0 (tile kt+1) -> buffer 1; refill st0 with tile kt+3
        BFLOW_COMPUTE(lds)
        __builtin_amdgcn_sched_barrier(0);
        BFLOW_R2S(sa, lds + STAGE)
        {
            const int k3 = (kt + 3 < nk) ? (kt + 3) * BK : 0;   // out-of-range prefetches re-read tile 0 (never consumed)
            BFLOW_G2R(sa, k3)
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // odd step: compute tile kt+1 from buffer 1; st1 (tile kt+2) -> buffer 0; refill st1 with tile kt+4
        BFLOW_COMPUTE(lds + STAGE)
        __builtin_amdgcn_sched_barrier(0);
        BFLOW_R2S(sb, lds)
        {
            const int k4 = (kt + 4 < nk) ? (kt + 4) * BK : 0;
            BFLOW_G2R(sb, k4)
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#undef BFLOW_COMPUTE

    float* O = out + (long long)tb * N * N;
    const bool interior = (i0 + BM <= N) && (j0 + BN <= N);   // block-uniform: no per-store predication for inner tiles
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = j0 + wn * 64 + n * 32 + l31;
                const float v = (hh[m][n][r] + xx[m][n][r] * LO_INV) / sqrt_d;
                if (interior || (row < N && col < N)) O[(long long)row * N + col] = v;
            }
        }
    }
}

}  // namespace

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
    BFLOW_REQUIRE(T > 0 && B > 0 && N > 0 && D > 0 && D % (2 * BK) == 0 && Np >= N && Np % BM == 0, BFLOW_E_ARG,
                  "corr_build_split: bad sizes T=%d B=%d D=%d N=%d Np=%d", T, B, D, N, Np);
    BFLOW_REQUIRE((long long)T * B <= 65535, BFLOW_E_LIMIT, "corr_build_split: T*B too large");
    // 80 KB of dynamic LDS (> the 64 KB default limit); idempotent, per device
    hipFuncSetAttribute(reinterpret_cast<const void*>(corr_build_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    dim3 grid(bflow::ceil_div(N, BN) * bflow::ceil_div(N, BM), T * B);
    hipLaunchKernelGGL(corr_build_split_kernel, grid, dim3(GT), 2 * STAGE, (hipStream_t)stream, (const _Float16*)f1_hi,
                       (const _Float16*)f1_lo, (const _Float16*)f2_hi, (const _Float16*)f2_lo, out, B, D, N, Np, f1_target_stride,
                       sqrtf((float)D));
    return bflow::launch_status("corr_build_split");
}
