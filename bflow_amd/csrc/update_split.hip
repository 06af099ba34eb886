// Bezier parameter block emission and the small im2col on blocked (B, C/32, P, 32) tensors -- what is left of the element-wise
// glue of the update block when its convolutions run on the split-fp16 engine (conv_split.hip fuses the GRU gates and the
// parameter update into its epilogue).  Thread = 8 consecutive channels of one pixel: 32-B fp32 /
// 16-B fp16 accesses, fully coalesced (a 32-channel block row is one 64-B / 128-B run).
// Reference: SepConvGRU.forward, models/raft_spline/update.py:33-48; BezierCurves.delta_update_params, bezier.py:137-139.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

using bflow::split1;   // common.h: saturating hi/lo split

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// thread = one pixel of one image: C2 <= 32 parameters
__global__ __launch_bounds__(256) void bezier_update_kernel(float* __restrict__ params, const float* __restrict__ delta, int C2,
                                                            _Float16* __restrict__ bh, _Float16* __restrict__ bl, int CBt, int cb_off,
                                                            _Float16* __restrict__ b2h, _Float16* __restrict__ b2l, int CBt2, int cb_off2, int B,
                                                            int P, int c_off) {
    const long long total = (long long)B * P;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(e / P), pix = (int)(e - (long long)b * P);
        const long long od = ((long long)b * P + pix) * 32;                         // delta: (B, 1, P, 32)
        const long long ob = (((long long)b * CBt + cb_off) * P + pix) * 32;
        half8 o1[4], o2[4];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            float v = 0.f;
            if (c < C2) {
                float* pp = params + ((long long)b * C2 + c) * P + pix;
                v = *pp;
                if (delta) {
                    v = v + delta[od + c];
                    *pp = v;
                }
            }
            _Float16 a, d;
            split1(v, a, d);
            o1[c >> 3][c & 7] = a;
            o2[c >> 3][c & 7] = d;
        }
        if (c_off > 0) {          // the parameters share their block with other channels: write the C2 values only
#pragma unroll
            for (int c = 0; c < 32; ++c)         // (compile-time register indices: a runtime-indexed loop sends o1 / o2 to scratch)
                if (c < C2) {
                    bh[ob + c_off + c] = o1[c >> 3][c & 7];
                    bl[ob + c_off + c] = o2[c >> 3][c & 7];
                }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<half8*>(bh + ob + g * 8) = o1[g];
                *reinterpret_cast<half8*>(bl + ob + g * 8) = o2[g];
            }
        }
        if (b2h) {
            const long long ob2 = (((long long)b * CBt2 + cb_off2) * P + pix) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<half8*>(b2h + ob2 + g * 8) = o1[g];
                *reinterpret_cast<half8*>(b2l + ob2 + g * 8) = o2[g];
            }
        }
    }
}

}  // namespace



extern "C" int bflow_bezier_update(float* params, const float* delta, int C2, void* blk_hi, void* blk_lo, int CB_total, int cb_off,
                                   void* blk2_hi, void* blk2_lo, int CB_total2, int cb_off2, int B, int P, int channel_in_block, bflow_stream_t stream) {
    BFLOW_REQUIRE(params && blk_hi && blk_lo && C2 > 0 && C2 <= 32 && CB_total > 0 && cb_off >= 0 && cb_off < CB_total && B > 0 && P > 0 &&
                      channel_in_block >= 0 && channel_in_block + C2 <= 32, BFLOW_E_ARG, "bezier_update: bad arguments");
    hipLaunchKernelGGL(bezier_update_kernel, dim3(bflow::stream_grid((long long)B * P, 256)), dim3(256), 0, (hipStream_t)stream, params, delta,
                       C2, (_Float16*)blk_hi, (_Float16*)blk_lo, CB_total, cb_off, (_Float16*)blk2_hi, (_Float16*)blk2_lo, CB_total2, cb_off2, B, P, channel_in_block);
    return bflow::launch_status("bezier_update");
}

namespace {
// 2-D neighbourhood gather of a few-channel fp32 NCHW tensor into a blocked split tensor: out[b, pix, tap*C + c] =
// x[b, c, y + r - pad, x + q - pad] (zero outside), K = KH*KW*C padded to a multiple of 32 with zeros.  Turns the 7x7
// convolution over the 2*deg Bezier channels (update.py:62,91) into a dense 1x1 GEMM instead of 49 mostly-empty k-tiles.
__global__ __launch_bounds__(256) void im2col_small_kernel(bflow::Im2colArgs m) {
    const unsigned per_image = (unsigned)m.CBk * (unsigned)m.P * 4u;     // 8-channel groups (bflow::im2col_small_item, common.h); image = blockIdx.y
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < per_image; e += gridDim.x * blockDim.x) bflow::im2col_small_item(m, (int)blockIdx.y, e);
}
}  // namespace

extern "C" int bflow_im2col_small(const float* x, void* out_hi, void* out_lo, int B, int C, int H, int W, int KH, int KW, int pad_h, int pad_w,
                                  int rows_per_image, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && out_hi && out_lo && B > 0 && C > 0 && H > 0 && W > 0 && KH > 0 && KW > 0, BFLOW_E_ARG, "im2col_small: bad arguments");
    const int P = rows_per_image > 0 ? rows_per_image : H * W;
    const int CBk = (KH * KW * C + 31) / 32;
    BFLOW_REQUIRE((long long)CBk * P * 4 < (1LL << 31) && B <= 65535, BFLOW_E_LIMIT, "im2col_small: image too large for 32-bit item indices");
    bflow::Im2colArgs m{x, (_Float16*)out_hi, (_Float16*)out_lo, C, H, W, KH, KW, pad_h, pad_w, CBk, P};
    hipLaunchKernelGGL(im2col_small_kernel, dim3(bflow::stream_grid((long long)CBk * P * 4, 256), B), dim3(256), 0, (hipStream_t)stream, m);
    return bflow::launch_status("im2col_small");
}
