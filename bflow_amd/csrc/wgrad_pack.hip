// Weight-gradient operand packing for the training path: see bflow_wgrad_pack in include/bflow_hip.h.
// The adjoint of Conv2d w.r.t. its filter contracts over PIXELS; the conv engine contracts over the 32-wide blocks of its last
// dimension.  This kernel writes both operands of that GEMM in the engine's blocked split layout with the pixel index in the block
// position: thread = (tap, k-block, row c, 8 consecutive k) -> one 16-B store per plane; reads are 8 pixels of one source row
// (stride 1: contiguous).  HBM-bound: C*K*4 B read per tap (L2 serves the taps after the first), taps*C*K*4 B written.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void wgrad_pack_kernel(const float* __restrict__ src, _Float16* __restrict__ dh, _Float16* __restrict__ dl, int B, int C,
                                                         int H, int W, int Ho, int Wo, int KW, int stride, int pad_h, int pad_w, int rows, int KB,
                                                         const float* __restrict__ scale_p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;          // (kb, c, j8) with j8 fastest: 4 threads per 64-B row
    const int tap = blockIdx.y;
    const long long per_tap = (long long)KB * rows * 4;
    if (t >= per_tap) return;
    const int j8 = (int)(t & 3);
    const long long rc = t >> 2;
    const int c = (int)(rc % rows);
    const int kb = (int)(rc / rows);
    const float scale = scale_p ? *scale_p : 1.f;
    const int r = tap / KW, q = tap - r * KW;
    const int K = B * Ho * Wo;
    half8 h8, l8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kb * 32 + j8 * 8 + i;
        float v = 0.f;
        if (k < K && c < C) {
            const int b = k / (Ho * Wo), p = k - b * (Ho * Wo);
            const int yo = p / Wo, xo = p - yo * Wo;
            const int y = yo * stride + r - pad_h, x = xo * stride + q - pad_w;
            if (y >= 0 && y < H && x >= 0 && x < W) v = src[(((long long)b * C + c) * H + y) * W + x] * scale;
        }
        _Float16 hi, lo;
        bflow::split1(v, hi, lo);
        h8[i] = hi;
        l8[i] = lo;
    }
    const long long o = (((long long)tap * KB + kb) * rows + c) * 32 + j8 * 8;
    *reinterpret_cast<half8*>(dh + o) = h8;
    *reinterpret_cast<half8*>(dl + o) = l8;
}

}  // namespace

extern "C" int bflow_wgrad_pack(const float* src, void* dst_hi, void* dst_lo, int B, int C, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                                int pad_h, int pad_w, int rows, int k_blocks, const float* scale, bflow_stream_t stream) {
    BFLOW_REQUIRE(src && dst_hi && dst_lo, BFLOW_E_ARG, "wgrad_pack: null pointer");
    BFLOW_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride >= 1 && rows >= C, BFLOW_E_ARG, "wgrad_pack: bad sizes");
    BFLOW_REQUIRE((long long)k_blocks * 32 >= (long long)B * Ho * Wo && KH * KW <= 65535, BFLOW_E_ARG, "wgrad_pack: k_blocks too small");
    BFLOW_REQUIRE((long long)B * Ho * Wo < (1LL << 31), BFLOW_E_LIMIT, "wgrad_pack: more than 2^31 pixels");
    const long long per_tap = (long long)k_blocks * rows * 4;
    dim3 grid(bflow::ceil_div(per_tap, 256), KH * KW);
    hipLaunchKernelGGL(wgrad_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst_hi, (_Float16*)dst_lo, B, C, H, W, Ho, Wo, KW,
                       stride, pad_h, pad_w, rows, k_blocks, scale);
    return bflow::launch_status("wgrad_pack");
}
