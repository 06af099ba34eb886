// Thin-output convolution on the vector ALU: KH x KW, stride 1, "same" zero padding, Cout <= 32 output channels, input = blocked split
// tensor of the conv engine.  Reference: BezierHead.conv2 (models/raft_spline/update.py:12-18: Conv2d(256, 2*degree, 3, padding=1)) with
// BezierCurves.delta_update_params (models/raft_spline/bezier.py:137-139) and the re-emission of the Bezier channel block fused behind it.
//
// Why not the MFMA engine: with 4 output channels (degree 2) a 32-channel MFMA tile is 87 % padding and the batch-1 grid is 40 workgroups
// on 256 CUs walking K = 2304 one after the other (17-18 us per GRU iteration, on the critical path).  The work itself is 88 MFLOP: the
// vector ALU does it exactly in fp32 (inputs are re-assembled as hi + lo * 2^-11, weights stay fp32 -- no split of the weights at all)
// with ALL pixels in flight at once:
//   wave = 2 pixels per group; lane = (pixel, channel block, 8 input channels): 32 lanes x 8 = the 256 input channels of one pixel;
//   per tap a lane loads 16 B of hi + 16 B of lo (a wave instruction = the 8 channel-block rows of two pixels) and
//   runs 4 x CO FMAs against weights read from LDS (CO = 4 output channels per pass: 9 taps x 4 x 256 fp32 = 36 KB per workgroup); a butterfly
//   over the wave finishes the dot products; lanes < Cout add the bias, update the fp32 NCHW accumulator (P += dP) and write the updated
//   values as ONE 32-channel block of a split tensor (channels >= Cout zero).
#include "common.h"
#include <algorithm>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int CO = 4;          // output channels per pass
constexpr int GPW = 2;         // pixels per wave and group (lanes 0-31 / 32-63): 8 pixels per group and workgroup
constexpr int GPB = 1;         // groups per workgroup and pass (2: their loads are all issued together -- measured slower at batch 1:
                               // 300 workgroups leave some SIMDs with two waves of twice the work)

struct ThinArgs {
    const _Float16 *xh, *xl;   // (B, CB, P_in, 32)
    const float* w;            // packed (taps, Cout, C) fp32
    const float* bias;         // (Cout) or null
    float* acc;                // (B, Cout, H*W) fp32, updated in place: acc += conv + bias
    _Float16 *oh, *ol;         // split output block or null: (B, CBo, P_out, 32), block cb_off receives [acc values | zeros]
    int B, H, W, CB, P_in, Cout, CBo, cb_off, P_out;
    int c_off;                 // first channel of the emitted values inside block cb_off; > 0: the rest of the block is NOT touched
};

// Workgroup = 4 waves; the weights of CO output channels (all taps, 256 input channels, fp32: 36 KB for 3x3) sit in LDS, filled once per
// pass with every load in flight at once.  A wave owns 2 pixels per group: lane = (pixel, channel block, 8 channels) reads 16 B of hi
// and 16 B of lo per tap.  At batch 1 (4800 pixels, 300 workgroups) there is about one wave per SIMD, so nothing hides a load behind
// another wave: the loads of BOTH groups of a workgroup (2 x 9 taps x hi/lo, the accumulator values) are issued before the first FMA.
template <int KH, int KW>
__global__ __launch_bounds__(256, 3) void conv_thin_kernel(ThinArgs a) {
    constexpr int NTAPS = KH * KW, ph = KH / 2, pw = KW / 2;
    constexpr int WL = NTAPS * CO * 64;                       // float4 entries: (tap, k, 64 x 4 channels)
    __shared__ float4 wl[WL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = a.H * a.W, C = a.CB * 32;
    const int b = blockIdx.y;
    const int n_groups = (HW + 4 * GPW - 1) / (4 * GPW);
    const int pj = lane >> 5, cb = (lane >> 2) & 7, ch = (lane & 3) * 8;      // pixel of the pair, channel block, first of 8 channels
    const bool lane_on = cb < a.CB;
    // buffer loads: one 32-bit offset serves the hi and the lo plane, an out-of-image tap is an out-of-range offset (returns zeros)
    const int plane_bytes = a.CB * a.P_in * 64;
    const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xh + (long long)b * a.CB * a.P_in * 32), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_l = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xl + (long long)b * a.CB * a.P_in * 32), 0, plane_bytes, 0x00020000);
    const int lane_off = (cb * a.P_in * 32 + ch) * 2;
    const int NT_ = KH * KW;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, NT_ * a.Cout * a.CB * 32 * 4, 0x00020000);
    for (int c0 = 0; c0 < a.Cout; c0 += CO) {
        const int g_first = blockIdx.x * GPB;
        half8 xh[GPB][NTAPS], xl[GPB][NTAPS];
        float old[GPB];
        bool emit[GPB];
        float* pa[GPB];
        // lane e = i * CO + k (< GPW * CO) will own output (pixel i of the wave's pair, channel c0 + k)
        const int ei = lane / CO, ek = lane % CO;
        auto load_group = [&](int g0) {
#pragma unroll
            for (int u = 0; u < GPB; ++u) {
                const int n0 = ((g0 + u) * 4 + wave) * GPW;
                const int n = min(n0 + pj, HW - 1);
                const int py = n / a.W, px = n - py * a.W;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int y = py + t / KW - ph, x = px + t % KW - pw;
                    const bool ok = lane_on && n0 + pj < HW && y >= 0 && y < a.H && x >= 0 && x < a.W;
                    const unsigned o = ok ? (unsigned)(lane_off + (y * a.W + x) * 64) : 0x80000000u;
                    xh[u][t] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r_h, o, 0, 0));
                    xl[u][t] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r_l, o, 0, 0));
                }
                emit[u] = lane < GPW * CO && c0 + ek < a.Cout && n0 + ei < HW;
                pa[u] = a.acc + ((long long)b * a.Cout + c0 + ek) * HW + n0 + ei;
                old[u] = emit[u] ? *pa[u] + (a.bias ? a.bias[c0 + ek] : 0.f) : 0.f;
            }
        };
        if (c0) __syncthreads();
        if (g_first < n_groups) load_group(g_first);     // in flight under the weight fill: one memory round trip per workgroup, not two
        {   // weights -> LDS by LDS-DMA (no staging registers next to the 72 input registers): entry (t, k, half, L) = input channels
            // 8 L + 4 half .. + 3 of output channel c0 + k, tap t (lane L of a pixel reads entries L and 32 + L: consecutive 16-B slots per
            // lane, no bank conflicts); entry e lands at wl + 16 e = lane-linear per wave instruction; masked entries are out-of-range
            // offsets (zeros)
            static_assert(WL % 256 == 0, "whole wave instructions");
#pragma unroll
            for (int i = 0; i < WL / 256; ++i) {
                const int e = tid + i * 256;
                const int l = e & 63, k = (e >> 6) % CO, t = e / (64 * CO);
                const int cin = (l & 31) * 8 + (l >> 5) * 4;
                const bool on = cin < C && c0 + k < a.Cout;
                const unsigned o = on ? (unsigned)((((t * a.Cout + c0 + k) * C) + cin) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(wl + i * 256 + wave * 64), 16, o, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#pragma unroll 1
        for (int g0 = g_first; g0 < n_groups; g0 += gridDim.x * GPB) {
            if (g0 != g_first) load_group(g0);
            // x = hi + lo * 2^-11 re-assembled by ONE mixed-precision FMA per element (fp16 x fp32 + fp16), then packed fp32 FMAs: two
            // input channels per instruction against the weight pairs, even / odd partial sums per output channel.
            // One group after the other (the weights are read from LDS once per group).
            float sum[GPB][CO];
#pragma unroll
            for (int u = 0; u < GPB; ++u) {
                float2v s2[CO];
#pragma unroll
                for (int k = 0; k < CO; ++k) s2[k] = float2v{0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    __builtin_amdgcn_sched_barrier(0);      // one tap's LDS reads in flight at a time (the input loads stay up front)
                    float2v x2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        x2[j] = float2v{fmaf((float)xl[u][t][2 * j], bflow::SPLIT_LO_INV, (float)xh[u][t][2 * j]),
                                        fmaf((float)xl[u][t][2 * j + 1], bflow::SPLIT_LO_INV, (float)xh[u][t][2 * j + 1])};
#pragma unroll
                    for (int k = 0; k < CO; ++k) {
                        if (k == CO / 2) __builtin_amdgcn_sched_barrier(0);   // the weights of two output channels in registers at a time
                        const float4 w0 = wl[(t * CO + k) * 64 + (lane & 31)], w1 = wl[(t * CO + k) * 64 + 32 + (lane & 31)];
                        s2[k] = __builtin_elementwise_fma(x2[0], float2v{w0.x, w0.y}, s2[k]);
                        s2[k] = __builtin_elementwise_fma(x2[1], float2v{w0.z, w0.w}, s2[k]);
                        s2[k] = __builtin_elementwise_fma(x2[2], float2v{w1.x, w1.y}, s2[k]);
                        s2[k] = __builtin_elementwise_fma(x2[3], float2v{w1.z, w1.w}, s2[k]);
                        asm volatile("" : "+v"(s2[k]));     // pins the FMAs to their tap (LLVM otherwise sinks a group's FMAs below the loop)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < CO; ++k) sum[u][k] = s2[k][0] + s2[k][1];
            }
            // finish the dot products over the 32 lanes of a pixel; lane i * CO + k keeps (pixel i, output channel c0 + k)
#pragma unroll
            for (int u = 0; u < GPB; ++u) {
                float mine = 0.f;
#pragma unroll
                for (int k = 0; k < CO; ++k) {
                    float s = sum[u][k];
#pragma unroll
                    for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);      // within the 32-lane half
                    const float other = __shfl_xor(s, 32, 64);                          // the other pixel's total
                    const float p0 = pj == 0 ? s : other, p1 = pj == 0 ? other : s;
                    if (lane == k) mine = p0;
                    if (lane == CO + k) mine = p1;
                }
                const int n0 = ((g0 + u) * 4 + wave) * GPW;
                if (emit[u]) {                                 // (emit implies a valid pixel; a UNIFORM guard here lets the compiler sink group 1's FMAs behind it)
                    const float v = old[u] + mine;             // bezier.py:137-139: params += delta (+ the conv bias)
                    *pa[u] = v;
                    if (a.oh) {
                        _Float16 hi, lo;
                        bflow::split1(v, hi, lo);
                        const long long o = (((long long)b * a.CBo + a.cb_off) * a.P_out + n0 + ei) * 32 + a.c_off + c0 + ek;
                        a.oh[o] = hi;
                        a.ol[o] = lo;
                    }
                }
                // channels [Cout, 32) of the emitted block are zero
                if (c0 == 0 && a.oh && a.c_off == 0 && lane >= a.Cout && lane < 32) {
#pragma unroll
                    for (int i = 0; i < GPW; ++i)
                        if (n0 + i < HW) {
                            const long long o = (((long long)b * a.CBo + a.cb_off) * a.P_out + n0 + i) * 32 + lane;
                            a.oh[o] = (_Float16)0.f;
                            a.ol[o] = (_Float16)0.f;
                        }
                }
            }
        }
    }
}

}  // namespace

extern "C" int bflow_conv_thin_acc(const void* x_hi, const void* x_lo, const float* w_packed, const float* bias, float* acc_nchw, void* out_hi,
                                   void* out_lo, int B, int H, int W, int C, int in_rows_per_image, int Cout, int KH, int KW,
                                   int out_channel_blocks, int out_block, int out_rows_per_image, int out_channel_in_block, bflow_stream_t stream) {
    BFLOW_REQUIRE(x_hi && x_lo && w_packed && acc_nchw, BFLOW_E_ARG, "conv_thin_acc: null pointer");
    BFLOW_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && C <= 256 && Cout >= 1 && Cout <= 32, BFLOW_E_ARG, "conv_thin_acc: needs C %% 32 == 0, C <= 256, Cout <= 32 (got C=%d Cout=%d)", C, Cout);
    BFLOW_REQUIRE((KH == 3 && KW == 3) || (KH == 1 && KW == 1), BFLOW_E_ARG, "conv_thin_acc: 3x3 and 1x1 filters are built (got %dx%d)", KH, KW);
    BFLOW_REQUIRE(in_rows_per_image >= H * W && B <= 65535, BFLOW_E_ARG, "conv_thin_acc: bad row count / batch");
    BFLOW_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), BFLOW_E_ARG, "conv_thin_acc: out_hi / out_lo go together");
    if (out_hi) BFLOW_REQUIRE(out_block >= 0 && out_block < out_channel_blocks && out_rows_per_image >= H * W && out_channel_in_block >= 0 &&
                                  out_channel_in_block + Cout <= 32, BFLOW_E_ARG, "conv_thin_acc: bad output block");
    ThinArgs a;
    a.xh = (const _Float16*)x_hi;
    a.xl = (const _Float16*)x_lo;
    a.w = w_packed;
    a.bias = bias;
    a.acc = acc_nchw;
    a.oh = (_Float16*)out_hi;
    a.ol = (_Float16*)out_lo;
    a.B = B; a.H = H; a.W = W; a.CB = C / 32; a.P_in = in_rows_per_image; a.Cout = Cout;
    a.CBo = out_channel_blocks; a.cb_off = out_block; a.P_out = out_rows_per_image; a.c_off = out_channel_in_block;
    const int n_groups = bflow::ceil_div((long long)H * W, 4 * GPW);
    // one pass over the pixels per workgroup up to ~4 workgroups per CU, beyond that a workgroup walks several group pairs on one weight fill
    dim3 grid(std::min(bflow::ceil_div(n_groups, GPB), std::max(1, 1024 / B)), B);
    if (KH == 3) hipLaunchKernelGGL((conv_thin_kernel<3, 3>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_thin_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return bflow::launch_status("conv_thin_acc");
}
