// K5: all-pairs correlation volume  out[t,b,i,j] = <f1[.,b,:,i], f2[t,b,:,j]> / sqrt(D)
// (reference: CorrComputation._corr_dot_prod_util, models/raft_utils/corr.py:264-272).
//
// CDNA4 design
//   * the contraction is the ONE dense GEMM of the hot path -> exact-fp32 MFMA v_mfma_f32_32x32x2_f32
//     (bit-equivalent to an fmaf chain over d, so parity with the fp32 reference is round-off only);
//   * both operands are stored (D, N) with the pixel index contiguous, which IS the MFMA A/B fragment order
//     (lane l holds A[i = l&31][k = l>>5]): global rows of 128 pixels are read with coalesced 512-B float4
//     loads, staged through LDS k-major, and every fragment read is a conflict-free ds_read_b32;
//   * block tile 128x128 (4 waves, each 64x64 = 2x2 MFMA tiles, 64 accumulator VGPRs), BK = 16,
//     register-staged double buffering with one barrier per k-tile: the 64-cycle MFMA hides the staging;
//   * 1/sqrt(D) is applied in the epilogue as a true division (as the reference does).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int THREADS = 256;

template <bool VEC4>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int N, int k0, int c0, int tid, float4 (&r)[2]) {
    // tile rows k0..k0+15 (feature dim), columns c0..c0+127 (pixels); thread -> (row = tid/32 + 8*p, col4 = tid%32)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (tid >> 5) + 8 * p;
        const int col = c0 + (tid & 31) * 4;
        const float* g = src + (long long)(k0 + row) * N + col;
        if (VEC4) {
            r[p] = (col < N) ? *reinterpret_cast<const float4*>(g) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            r[p].x = (col + 0 < N) ? g[0] : 0.f;
            r[p].y = (col + 1 < N) ? g[1] : 0.f;
            r[p].z = (col + 2 < N) ? g[2] : 0.f;
            r[p].w = (col + 3 < N) ? g[3] : 0.f;
        }
    }
}

__device__ __forceinline__ void store_tile(float* lds /*[BK][128]*/, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (tid >> 5) + 8 * p;
        *reinterpret_cast<float4*>(lds + row * 128 + (tid & 31) * 4) = r[p];
    }
}

template <bool VEC4>
__global__ __launch_bounds__(THREADS) void corr_build_f32_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                 float* __restrict__ out, int B, int D, int N,
                                                                 long long f1_tstride, float sqrt_d) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BK * 128];  // [buffer][A|B][k][pixel]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, each a 64x64 sub-tile
    const int tb = blockIdx.z;                 // t*B + b
    const int t = tb / B, b = tb - t * B;
    const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;

    const float* A = f1 + t * f1_tstride + (long long)b * D * N;  // (D, N): A(i,k) = A[k*N + i]
    const float* Bm = f2 + (long long)tb * D * N;                 // (D, N): B(k,j) = Bm[k*N + j]

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float4 ra[2], rb[2];
    load_tile<VEC4>(A, N, 0, i0, tid, ra);
    load_tile<VEC4>(Bm, N, 0, j0, tid, rb);
    store_tile(lds[0][0], tid, ra);
    store_tile(lds[0][1], tid, rb);
    __syncthreads();

    const int nk = D / BK;
    const int kh = lane >> 5;    // which of the 2 k-slices of an MFMA this lane feeds
    const int l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<VEC4>(A, N, (kt + 1) * BK, i0, tid, ra);
            load_tile<VEC4>(Bm, N, (kt + 1) * BK, j0, tid, rb);
        }
        const float* As = lds[cur][0];
        const float* Bs = lds[cur][1];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = kk * 2 + kh;
            float a[2], bb[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = As[k * 128 + wm * 64 + m * 32 + l31];
#pragma unroll
            for (int n = 0; n < 2; ++n) bb[n] = Bs[k * 128 + wn * 64 + n * 32 + l31];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile(lds[cur ^ 1][0], tid, ra);
            store_tile(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* O = out + (long long)tb * N * N;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < N) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int col = j0 + wn * 64 + n * 32 + l31;
                    if (col < N) O[(long long)row * N + col] = acc[m][n][r] / sqrt_d;
                }
            }
        }
    }
}

}  // namespace

extern "C" int bflow_corr_build_f32(const float* f1, const float* f2, float* out, int T, int B, int D, int N,
                                    long long f1_target_stride, bflow_stream_t stream) {
    BFLOW_REQUIRE(f1 && f2 && out, BFLOW_E_ARG, "corr_build_f32: null pointer");
    BFLOW_REQUIRE(T > 0 && B > 0 && N > 0 && D > 0, BFLOW_E_ARG, "corr_build_f32: bad sizes T=%d B=%d D=%d N=%d", T, B, D, N);
    BFLOW_REQUIRE(D % BK == 0, BFLOW_E_ARG, "corr_build_f32: feature dim %d must be a multiple of %d", D, BK);
    BFLOW_REQUIRE((long long)T * B <= 65535, BFLOW_E_LIMIT, "corr_build_f32: T*B = %lld exceeds grid.z", (long long)T * B);
    dim3 grid(bflow::ceil_div(N, BN), bflow::ceil_div(N, BM), T * B);
    const float sqrt_d = sqrtf((float)D);
    const bool vec4 = (N % 4 == 0) && (((uintptr_t)f1 | (uintptr_t)f2) % 16 == 0) && (f1_target_stride % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (vec4)
        hipLaunchKernelGGL(corr_build_f32_kernel<true>, grid, dim3(THREADS), 0, s, f1, f2, out, B, D, N, f1_target_stride, sqrt_d);
    else
        hipLaunchKernelGGL(corr_build_f32_kernel<false>, grid, dim3(THREADS), 0, s, f1, f2, out, B, D, N, f1_target_stride, sqrt_d);
    return bflow::launch_status("corr_build_f32");
}
