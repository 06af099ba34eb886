// Thin-output convolution on the vector ALU: KH x KW, stride 1, "same" zero padding, Cout <= 32 output channels, input = blocked split
// tensor of the conv engine.  Reference: BezierHead.conv2 (models/raft_spline/update.py:12-18: Conv2d(256, 2*degree, 3, padding=1)) with
// BezierCurves.delta_update_params (models/raft_spline/bezier.py:137-139) and the re-emission of the Bezier channel block fused behind it.
//
// Why not the MFMA engine: with 4 output channels (degree 2) a 32-channel MFMA tile is 87 % padding and the batch-1 grid is 40 workgroups
// on 256 CUs walking K = 2304 one after the other (17-18 us per GRU iteration, on the critical path).  The work itself is 88 MFLOP: the
// vector ALU does it exactly in fp32 (inputs are re-assembled as hi + lo * 2^-11, weights stay fp32 -- no split of the weights at all)
// with ALL pixels in flight at once:
//   wave = 2 consecutive pixels per group; lane = 4 input channels (lane >> 3 = channel block, lane & 7 = group of 4): 64 lanes x 4 = 256 channels per
//   pass over the input channels; per (tap, pixel) a lane loads 8 B of hi + 8 B of lo (the 64 lanes read the 8 channel blocks' 64-B rows) and
//   runs 4 x CO FMAs against weights read from LDS (CO = 4 output channels per pass: 9 taps x 4 x 256 fp32 = 36 KB per workgroup); a butterfly
//   over the wave finishes the dot products; lanes < Cout add the bias, update the fp32 NCHW accumulator (P += dP) and write the updated
//   values as ONE 32-channel block of a split tensor (channels >= Cout zero).
#include "common.h"
#include <algorithm>

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
constexpr int CO = 4;          // output channels per pass
constexpr int PPW = 2;         // pixels per wave and group (4 waves: 8 pixels per group)

struct ThinArgs {
    const _Float16 *xh, *xl;   // (B, CB, P_in, 32)
    const float* w;            // packed (taps, Cout, C) fp32
    const float* bias;         // (Cout) or null
    float* acc;                // (B, Cout, H*W) fp32, updated in place: acc += conv + bias
    _Float16 *oh, *ol;         // split output block or null: (B, CBo, P_out, 32), block cb_off receives [acc values | zeros]
    int B, H, W, CB, P_in, Cout, CBo, cb_off, P_out;
};

// Workgroup = 4 waves; the weights of CO output channels (all taps, 256 input channels, fp32: 36 KB for 3x3) sit in LDS, loaded once per
// pass; the workgroup then walks groups of 8 pixels.  At batch 1 (2880 pixels) there is less than one wave per SIMD, so nothing hides a
// load behind another wave: every load of a group (2 x 9 taps x hi/lo, the accumulator values) is issued before the first FMA.
template <int KH, int KW>
__global__ __launch_bounds__(256, 2) void conv_thin_kernel(ThinArgs a) {
    constexpr int NTAPS = KH * KW, ph = KH / 2, pw = KW / 2;
    __shared__ float4 wl[NTAPS * CO * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = a.H * a.W, C = a.CB * 32;
    const int b = blockIdx.y;
    const int n_groups = (HW + 4 * PPW - 1) / (4 * PPW);
    const int cb = lane >> 3, ch = (lane & 7) * 4;
    const bool lane_on = cb < a.CB;
    const long long plane = ((long long)b * a.CB + (lane_on ? cb : 0)) * a.P_in;
    for (int c0 = 0; c0 < a.Cout; c0 += CO) {
        if (c0) __syncthreads();
        // weights -> LDS: entry (t, k, l) = the 4 channels of lane l for output channel c0 + k, tap t
        for (int e = tid; e < NTAPS * CO * 64; e += 256) {
            const int l = e & 63, k = (e >> 6) % CO, t = e / (64 * CO);
            const bool on = (l >> 3) < a.CB && c0 + k < a.Cout;
            wl[e] = on ? *reinterpret_cast<const float4*>(a.w + ((long long)t * a.Cout + c0 + k) * C + l * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll 1
        for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
            const int n0 = (g * 4 + wave) * PPW;
            half4v xh[PPW][NTAPS], xl[PPW][NTAPS];
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int n = min(n0 + i, HW - 1);
                const int py = n / a.W, px = n - py * a.W;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int y = py + t / KW - ph, x = px + t % KW - pw;
                    const bool ok = lane_on && y >= 0 && y < a.H && x >= 0 && x < a.W;
                    const long long o = (plane + (ok ? y * a.W + x : 0)) * 32 + ch;
                    const half4v zero = {0, 0, 0, 0};
                    xh[i][t] = ok ? *reinterpret_cast<const half4v*>(a.xh + o) : zero;
                    xl[i][t] = ok ? *reinterpret_cast<const half4v*>(a.xl + o) : zero;
                }
            }
            // the accumulator value this lane will update: lane = i * CO + k
            const int ei = lane / CO, ek = lane % CO;
            const bool emit = lane < PPW * CO && c0 + ek < a.Cout && n0 + ei < HW;
            float* pa = a.acc + ((long long)b * a.Cout + c0 + ek) * HW + n0 + ei;
            const float old = emit ? *pa + (a.bias ? a.bias[c0 + ek] : 0.f) : 0.f;
            float sum[PPW][CO];
#pragma unroll
            for (int i = 0; i < PPW; ++i)
#pragma unroll
                for (int k = 0; k < CO; ++k) sum[i][k] = 0.f;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                __builtin_amdgcn_sched_barrier(0);          // one tap's LDS reads in flight at a time (the input loads stay up front)
                float v[PPW][4];
#pragma unroll
                for (int i = 0; i < PPW; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] = fmaf((float)xl[i][t][j], bflow::SPLIT_LO_INV, (float)xh[i][t][j]);
#pragma unroll
                for (int k = 0; k < CO; ++k) {
                    const float4 w = wl[(t * CO + k) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < PPW; ++i) {
                        sum[i][k] = fmaf(v[i][0], w.x, sum[i][k]);
                        sum[i][k] = fmaf(v[i][1], w.y, sum[i][k]);
                        sum[i][k] = fmaf(v[i][2], w.z, sum[i][k]);
                        sum[i][k] = fmaf(v[i][3], w.w, sum[i][k]);
                    }
                }
            }
            // finish the dot products over the 64 lanes; lane i * CO + k keeps (pixel i, output channel c0 + k)
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < PPW; ++i)
#pragma unroll
                for (int k = 0; k < CO; ++k) {
                    const float s = bflow::wave_sum(sum[i][k]);
                    if (lane == i * CO + k) mine = s;
                }
            if (emit) {
                const float v = old + mine;                 // bezier.py:137-139: params += delta (+ the conv bias)
                *pa = v;
                if (a.oh) {
                    _Float16 hi, lo;
                    bflow::split1(v, hi, lo);
                    const long long o = (((long long)b * a.CBo + a.cb_off) * a.P_out + n0 + ei) * 32 + c0 + ek;
                    a.oh[o] = hi;
                    a.ol[o] = lo;
                }
            }
            // channels [Cout, 32) of the emitted block are zero
            if (c0 == 0 && a.oh && lane >= a.Cout && lane < 32) {
#pragma unroll
                for (int i = 0; i < PPW; ++i)
                    if (n0 + i < HW) {
                        const long long o = (((long long)b * a.CBo + a.cb_off) * a.P_out + n0 + i) * 32 + lane;
                        a.oh[o] = (_Float16)0.f;
                        a.ol[o] = (_Float16)0.f;
                    }
            }
        }
    }
}

}  // namespace

extern "C" int bflow_conv_thin_acc(const void* x_hi, const void* x_lo, const float* w_packed, const float* bias, float* acc_nchw, void* out_hi,
                                   void* out_lo, int B, int H, int W, int C, int in_rows_per_image, int Cout, int KH, int KW,
                                   int out_channel_blocks, int out_block, int out_rows_per_image, bflow_stream_t stream) {
    BFLOW_REQUIRE(x_hi && x_lo && w_packed && acc_nchw, BFLOW_E_ARG, "conv_thin_acc: null pointer");
    BFLOW_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && C <= 256 && Cout >= 1 && Cout <= 32, BFLOW_E_ARG, "conv_thin_acc: needs C %% 32 == 0, C <= 256, Cout <= 32 (got C=%d Cout=%d)", C, Cout);
    BFLOW_REQUIRE((KH == 3 && KW == 3) || (KH == 1 && KW == 1), BFLOW_E_ARG, "conv_thin_acc: 3x3 and 1x1 filters are built (got %dx%d)", KH, KW);
    BFLOW_REQUIRE(in_rows_per_image >= H * W && B <= 65535, BFLOW_E_ARG, "conv_thin_acc: bad row count / batch");
    BFLOW_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), BFLOW_E_ARG, "conv_thin_acc: out_hi / out_lo go together");
    if (out_hi) BFLOW_REQUIRE(out_block >= 0 && out_block < out_channel_blocks && out_rows_per_image >= H * W, BFLOW_E_ARG, "conv_thin_acc: bad output block");
    ThinArgs a;
    a.xh = (const _Float16*)x_hi;
    a.xl = (const _Float16*)x_lo;
    a.w = w_packed;
    a.bias = bias;
    a.acc = acc_nchw;
    a.oh = (_Float16*)out_hi;
    a.ol = (_Float16*)out_lo;
    a.B = B; a.H = H; a.W = W; a.CB = C / 32; a.P_in = in_rows_per_image; a.Cout = Cout;
    a.CBo = out_channel_blocks; a.cb_off = out_block; a.P_out = out_rows_per_image;
    const int n_groups = bflow::ceil_div((long long)H * W, 4 * PPW);
    dim3 grid(std::min(n_groups, std::max(1, 1024 / B)), B);   // beyond ~4 workgroups per CU a workgroup walks several groups on one weight fill
    if (KH == 3) hipLaunchKernelGGL((conv_thin_kernel<3, 3>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_thin_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return bflow::launch_status("conv_thin_acc");
}
