// K13: convex x8 up-sampling of the Bezier parameters (reference: cvx_upsample, models/raft_utils/utils.py:33-48,
// called once per forward in test mode, raft.py:193-195).
//
// HBM-bound: reads the (B,576,h,w) mask once (dominant: 2304 B per low-res pixel) and writes (B,C,8h,8w).
// Thread = (low-res pixel x, sub-row i): lanes run over x so every mask read mask[b, k*64+i*8+j, y, x] is a
// coalesced row, the 9-tap softmax lives in registers, and each lane writes 8 consecutive outputs (two float4),
// i.e. a wave writes one contiguous 2-KB output row segment.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void cvx_upsample_kernel(const float* __restrict__ data, const float* __restrict__ mask,
                                                           const float* __restrict__ mask_bias, float mask_scale,
                                                           float* __restrict__ out, int B, int C, int h, int w) {
    const int N = h * w;
    const long long total = (long long)B * 8 * N;   // (b, i, y, x) with x fastest
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const int bi = (int)(idx / N);
        const int i = bi & 7, b = bi >> 3;
        const int y = n / w, x = n - y * w;
        const float* mb = mask + (long long)b * 576 * N + n;

        // softmax over the 9 taps for the 8 sub-columns j of sub-row i   (utils.py:36-37)
        float m[9][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int ch = k * 64 + i * 8 + j;
                m[k][j] = mask_scale * (mb[(long long)ch * N] + (mask_bias ? mask_bias[ch] : 0.f));
                mx = fmaxf(mx, m[k][j]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k][j] = expf(m[k][j] - mx);
                s += m[k][j];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) m[k][j] = m[k][j] / s;
        }

        for (int c = 0; c < C; ++c) {
            const float* d = data + ((long long)b * C + c) * N;
            float nb[9];  // 3x3 neighbourhood of 8*data, zero padded (F.unfold padding=1; utils.py:40)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    nb[ky * 3 + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? 8.f * d[yy * w + xx] : 0.f;
                }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += m[k][j] * nb[k];
                o[j] = acc;
            }
            float* op = out + (((long long)b * C + c) * (8 * h) + (8 * y + i)) * (8LL * w) + 8 * x;
            *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// The same operator on the mask as the conv engine leaves it: blocked fp32 (B, 18, P, 32) = bflow_conv_split's out_f32 of the 576-channel
// mask head (update.py:120-125), bias included -- no NCHW copy of the 11-MB mask in between.  Channel k*64 + i*8 + j of pixel n is element
// (i & 3) * 8 + j of the 128-B row (block 2k + (i >> 2), n).  Item = (pixel, sub-row i, half of the sub-columns): lane bits 0-2 = q =
// (i & 3) * 2 + jhalf, so eight lanes read one whole row per tap with one float4 each and a wave covers 8 pixels x 4 sub-rows; bits above:
// pixel, then i >> 2.  Twice the items of the NCHW kernel (300 instead of 150 workgroups at 60 x 80: it is latency-, not bandwidth-bound
// there); the arithmetic and its order are the NCHW kernel's, so equal masks give equal bits.
__global__ __launch_bounds__(256) void cvx_upsample_blocked_kernel(const float* __restrict__ data, const float* __restrict__ mask, float mask_scale,
                                                                   float* __restrict__ out, int B, int C, int h, int w, int P) {
    const int N = h * w;
    const long long total = (long long)B * 2 * N * 8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx & 7);
        int n, bi;
        if (total < (1LL << 31)) {                       // (uniform) 32-bit index arithmetic: a 64-bit division costs more than the softmax
            const unsigned r = (unsigned)idx >> 3;
            bi = (int)(r / (unsigned)N);
            n = (int)(r - (unsigned)bi * (unsigned)N);
        } else {
            const long long r = idx >> 3;
            n = (int)(r % N);
            bi = (int)(r / N);
        }
        const int ihi = bi & 1, b = bi >> 1;
        const int i = ihi * 4 + (q >> 1), j0 = (q & 1) * 4;
        const int y = n / w, x = n - y * w;
        float m[9][4];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(mask + (((long long)b * 18 + 2 * k + ihi) * P + n) * 32 + q * 4);
            m[k][0] = mask_scale * v.x; m[k][1] = mask_scale * v.y; m[k][2] = mask_scale * v.z; m[k][3] = mask_scale * v.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {        // softmax over the 9 taps (utils.py:36-37)
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) mx = fmaxf(mx, m[k][j]);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k][j] = expf(m[k][j] - mx);
                s += m[k][j];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) m[k][j] = m[k][j] / s;
        }
        for (int c = 0; c < C; ++c) {
            const float* d = data + ((long long)b * C + c) * N;
            float nb[9];                     // 3x3 neighbourhood of 8*data, zero padded (F.unfold padding=1; utils.py:40)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    nb[ky * 3 + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? 8.f * d[yy * w + xx] : 0.f;
                }
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += m[k][j] * nb[k];
                o[j] = acc;
            }
            *reinterpret_cast<float4*>(out + (((long long)b * C + c) * (8 * h) + (8 * y + i)) * (8LL * w) + 8 * x + j0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

}  // namespace

extern "C" int bflow_cvx_upsample_blocked(const float* data, const float* mask_blocked, float mask_scale, float* out, int B, int C, int h, int w,
                                          int mask_rows_per_image, bflow_stream_t stream) {
    BFLOW_REQUIRE(data && mask_blocked && out && B > 0 && C > 0 && h > 0 && w > 0, BFLOW_E_ARG, "cvx_upsample_blocked: bad arguments");
    const int P = mask_rows_per_image > 0 ? mask_rows_per_image : h * w;
    BFLOW_REQUIRE(P >= h * w, BFLOW_E_ARG, "cvx_upsample_blocked: mask_rows_per_image < h*w");
    BFLOW_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)mask_blocked & 15) == 0, BFLOW_E_ARG, "cvx_upsample_blocked: buffers must be 16-byte aligned");
    const long long total = (long long)B * 2 * h * w * 8;
    hipLaunchKernelGGL(cvx_upsample_blocked_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, data, mask_blocked,
                       mask_scale, out, B, C, h, w, P);
    return bflow::launch_status("cvx_upsample_blocked");
}

extern "C" int bflow_cvx_upsample(const float* data, const float* mask, const float* mask_bias, float mask_scale, float* out, int B,
                                  int C, int h, int w, bflow_stream_t stream) {
    BFLOW_REQUIRE(data && mask && out && B > 0 && C > 0 && h > 0 && w > 0, BFLOW_E_ARG, "cvx_upsample: bad arguments");
    BFLOW_REQUIRE(((uintptr_t)out & 15) == 0, BFLOW_E_ARG, "cvx_upsample: output must be 16-byte aligned");
    const long long total = (long long)B * 8 * h * w;
    hipLaunchKernelGGL(cvx_upsample_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, data, mask,
                       mask_bias, mask_scale, out, B, C, h, w);
    return bflow::launch_status("cvx_upsample");
}
