// Shared helpers for the bflow HIP kernels (gfx950 / CDNA4 only: wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bflow_hip.h"

namespace bflow {

void set_error(const char* fmt, ...);

// Checks the launch that was just enqueued; returns the C-ABI status code.
int launch_status(const char* what);

constexpr int WAVE = 64;

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Grid size for HBM-bound streaming kernels: enough blocks to fill 256 CUs x 8, grid-stride the rest
// (cdna_hip_programming.md guideline 11).
static inline int stream_grid(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

// Grid for kernels that end in a few global atomics per block (reductions): every block adds to the same 2-3 addresses, and
// 2048 blocks x fp64 atomics on one address serialise to tens of microseconds -- more than the streaming pass itself.
static inline int reduce_grid(long long work_items, int block) {
    long long g = (work_items + (long long)block * 8 - 1) / ((long long)block * 8);
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- the split-fp16 number format of the matrix-core path -------------------------------------------------------
// x ~= hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11): ~22 significant bits.  Supported range (stated in bflow_hip.h):
//   |x| <= 65504 (fp16 max): larger magnitudes SATURATE to +-65504 instead of becoming hi = inf, lo = NaN; NaN stays NaN;
//   |x| <  2^-14: hi = 0 (the matrix cores flush fp16 subnormal INPUTS) and the whole value lives in the pre-scaled lo: 11 significant
//                 bits, i.e. an absolute error <= 2^-14 * 2^-11 = 3e-8 (lo itself is normal down to |x| = 2^-25).
constexpr float SPLIT_LO_SCALE = 2048.0f, SPLIT_LO_INV = 1.0f / 2048.0f, SPLIT_MAX = 65504.0f;
__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -SPLIT_MAX), SPLIT_MAX);   // v_med3-class clamp; passes NaN through below
    const float xs = (x != x) ? x : xc;
    const float h = (fabsf(xs) >= 6.103515625e-05f) ? (float)(_Float16)xs : 0.0f;
    hi = (_Float16)h;
    lo = (_Float16)((xs - h) * SPLIT_LO_SCALE);
}

// 2-D neighbourhood gather of a few-channel fp32 NCHW tensor into a blocked split tensor (bflow_im2col_small, and the rider of the look-up
// launch): item e = ((b * CBk + kb) * P + pix) * 4 + g produces the 8 channels kk = kb * 32 + g * 8 .. + 7 of pixel pix, kk = tap * C + c,
// out[b, kb, pix, kk % 32] = x[b, c, y + r - pad_h, x + q - pad_w] (zero outside the image and for kk >= KH * KW * C).  The item index is per image
// and 32-bit (CBk * P * 4 < 2^32, checked by the callers): a 64-bit division per item cost more than the gather.
struct Im2colArgs {
    const float* x;
    _Float16 *oh, *ol;
    int C, H, W, KH, KW, pad_h, pad_w, CBk, P;
};
__device__ __forceinline__ void im2col_small_item(const Im2colArgs& m, int b, unsigned e) {   // e = (kb * P + pix) * 4 + g inside image b (32-bit)
    typedef _Float16 half8_ __attribute__((ext_vector_type(8)));
    const int K = m.KH * m.KW * m.C;
    const int g = (int)(e & 3);
    const unsigned rowi = e >> 2;                         // kb * P + pix
    const int kb = (int)(rowi / (unsigned)m.P), pix = (int)(rowi - (unsigned)kb * (unsigned)m.P);
    const long long row = ((long long)b * m.CBk + kb) * m.P + pix;
    half8_ h8, l8;
    const int y = pix / m.W, xx0 = pix - y * m.W;
    // (tap row r, tap column q, channel c) of the item's first channel by division, of the other seven by carrying: the divisions by the
    // run-time C and KW were most of this kernel's instructions (16 per item)
    const int kk0 = kb * 32 + g * 8;
    int t = kk0 / m.C, c = kk0 - t * m.C;
    int r = t / m.KW, q = t - r * m.KW;
    const bool pix_ok = pix < m.H * m.W;
    const float* xb = m.x + (long long)b * m.C * m.H * m.W;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = 0.f;
        if (kk0 + k < K && pix_ok) {
            const int yy = y + r - m.pad_h, xx = xx0 + q - m.pad_w;
            if (yy >= 0 && yy < m.H && xx >= 0 && xx < m.W) v = xb[((long long)c * m.H + yy) * m.W + xx];
        }
        _Float16 a, d;
        split1(v, a, d);
        h8[k] = a;
        l8[k] = d;
        if (++c == m.C) { c = 0; if (++q == m.KW) { q = 0; ++r; } }
    }
    *reinterpret_cast<half8_*>(m.oh + row * 32 + g * 8) = h8;
    *reinterpret_cast<half8_*>(m.ol + row * 32 + g * 8) = l8;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate non-linearities of the fused GRU epilogues (conv_engine.h) on the transcendental unit: v_exp_f32 / v_rcp_f32 (1 ulp each) instead
// of libm's expf / tanhf and a true division.  The epilogue of a batch-1 gate convolution spends ~2.7 k of its ~25 k cycles in this
// arithmetic (8 values per lane on 2 waves per SIMD; tools/gru_conv_probe.py --stamps).  Accuracy: sigmoid <= 4 ulps; tanh: absolute
// error <= 2e-7 for |x| >= 0.04 (relative <= 4e-6 there, shrinking with |x|), odd polynomial (relative 1e-7) below.
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float gate_tanh(float x) {
    const float x2 = x * x;
    const float small = x * fmaf(x2, fmaf(x2, 0.133333333f, -0.333333333f), 1.0f);   // x - x^3/3 + 2 x^5/15
    const float e = __builtin_amdgcn_exp2f(2.88539008f * x);                          // e^(2x): inf -> 1, 0 -> -1
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return fabsf(x) < 0.04f ? small : big;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace bflow

#define BFLOW_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            bflow::set_error(__VA_ARGS__);        \
            return (code);                        \
        }                                         \
    } while (0)
