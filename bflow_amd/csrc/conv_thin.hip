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


// ---------------------------------------------------------------------------------------------------------------------
// The same operation on the matrix cores, "taps as output channels" (round 4): the 3 x 3 filter is a dense 1 x 1 GEMM
// Y[pixel][tap * Cout + co] = sum_c x[pixel][c] w[co][c][tap]  (K = C <= 256; 9 * Cout output rows, 64 per PASS: one pass at degree 2, three at
// degree 10; Cout <= 28) followed by a shifted sum  out[p][co] = sum_tap Y[p + tap][tap * Cout + co].  The vector-ALU kernel above re-reads the nine taps of every
// pixel and the 36 KB of weights of every 8 pixels from L2 (65 MB per launch at DSEC size, 9.6 us in the captured iteration); here a
// workgroup owns a 2 x 10 pixel patch: its 4 x 12 halo (all C channels, hi + lo: 48 KB) is fetched ONCE and the weights of a pass (the units
// that hold real rows: 48 KB at degree 2) by LDS-DMA with every piece in flight at once, 8 waves = 2 pixel tiles (the 48 halo pixels) x 2 output
// tiles x 2 k-halves run 8 channel blocks x 3 split MFMAs per pass, Y goes through LDS (the next pass's weights land meanwhile) and 20 x Cout
// items do the shifted sum (partial sums of the passes in a register), the bias, P += dP and the
// re-emission of the Bezier channels.  Products are the engine's 3-pass split products (2^-22; the weights are a packed split tensor like
// any other filter: bflow_conv_pack_weights of the derived 1 x 1 filter).
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x16t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_tt;

struct ThinMArgs {
    const _Float16 *xh, *xl;   // (B, CB, P_in, 32)
    const _Float16 *wh, *wl;   // packed 1 x 1 filter (CB, cout_pad, 32): row tap * Cout + co
    int cout_pad;
    const float* bias;
    float* acc;                // (B, Cout, H*W) fp32, updated in place
    _Float16 *oh, *ol;
    int B, H, W, CB, P_in, Cout, CBo, cb_off, P_out, c_off;
};

// Patch = 2 x 10 pixels: 30 x 8 = 240 workgroups at 60 x 80 (one per CU) and the fewest bytes per CU -- a small-grid kernel's time is its
// operand fill (a CU sustains 11-25 B/clk of global -> LDS traffic when every CU fills at once: MI355X_MICROARCH.md, prologue burst), the
// matrix work is a few hundred cycles.  2 x 16 patches (150 workgroups, 144 KB each) measured 8.8 us against 9.4 us for the vector-ALU kernel.
constexpr int TM_TH = 2, TM_TW = 10, TM_HWD = TM_TW + 2, TM_HR = TM_HWD * (TM_TH + 2);   // 4 x 12 = 48 halo pixels
constexpr int TM_AU = (TM_HR + 15) / 16;                                                   // 3 one-KB units per (block, plane)
constexpr int TM_NPT = (TM_HR + 31) / 32;                                                  // pixel tiles: 2 (rows >= 48 are never used)
constexpr int TM_WU = 4;                                                                   // LDS room: 64 weight rows = 4 units per (block, plane)
constexpr int TM_PPB = 2 * TM_AU + 2 * TM_WU;                                              // DMA piece slots per channel block: 14
constexpr int TM_YS = 68;                                                                  // floats per staged Y row (64 + 4)
constexpr int TM_NW = TM_NPT * 2 * 2;                                                      // waves: pixel tiles x output tiles x k-halves = 8
constexpr int TM_YR = TM_NPT * 32;                                                         // staged Y rows per k-half

__global__ __launch_bounds__(64 * TM_NW, 1) void conv_thin_mfma_kernel(ThinMArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y;
    const int tiles_x = (a.W + TM_TW - 1) / TM_TW;
    const int ty = blockIdx.x / tiles_x, y0 = ty * TM_TH, x0 = (blockIdx.x - ty * tiles_x) * TM_TW;
    // LDS: [halo: CB x (hi 3 KB | lo 3 KB)] [weights of ONE pass: CB x (hi 4 KB | lo 4 KB)] [Y: 2 k-halves x 64 pixels x 68 floats] [1 KB scratch]
    const int O_W = a.CB * (2 * TM_AU * 1024);
    const int O_Y = O_W + a.CB * (2 * TM_WU * 1024);
    const int O_SCR = O_Y + 2 * TM_YR * TM_YS * 4;

    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const int plane_b = a.P_in * 64;
    const __amdgpu_buffer_rsrc_t r_xh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xh + (long long)b * a.CB * a.P_in * 32), 0, a.CB * plane_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xl + (long long)b * a.CB * a.P_in * 32), 0, a.CB * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;
    const __amdgpu_buffer_rsrc_t r_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, a.CB * wtile_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.wl, 0, a.CB * wtile_b, 0x00020000);
    const int nrows = 9 * a.Cout;                              // rows of the derived 1 x 1 filter: tap * Cout + co
    const int npass = (nrows + 63) >> 6;                       // 64 rows per pass (degree 2: one pass, degree 10: three)

    // ---- the halo (all C channels, hi + lo): fetched ONCE by LDS-DMA; the same instruction count in every wave (dummy pieces -> scratch) ----
    {
        const int np = a.CB * 2 * TM_AU, ni = (np + TM_NW - 1) / TM_NW;
        for (int i = 0; i < ni; ++i) {
            const int p = wave_all + TM_NW * i;
            const bool real = p < np;
            const int cb = real ? p / (2 * TM_AU) : 0, r = real ? p - cb * (2 * TM_AU) : 0;
            const int unit = r < TM_AU ? r : r - TM_AU;
            const int row = unit * 16 + urow;
            const int hy = row / TM_HWD, hx = row - hy * TM_HWD;
            const int py = y0 - 1 + hy, px = x0 - 1 + hx;
            const bool ok = real && row < TM_HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
            const unsigned off = ok ? (unsigned)(((py * a.W + px) * 32 + uchunk) * 2) : 0x80000000u;
            char* const dst = real ? lds + cb * (2 * TM_AU * 1024) + r * 1024 : lds + O_SCR;
            if (r < TM_AU) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_xh, (lptr_tt)dst, 16, off, cb * plane_b, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r_xl, (lptr_tt)dst, 16, off, cb * plane_b, 0, 0);
        }
    }
    // weights of pass `ps`: rows [64 ps, 64 ps + 64) of every channel block; 16-row units without a real row are never fetched (zeros)
    auto issue_weights = [&](int ps) {
        const int np = a.CB * 2 * TM_WU, ni = (np + TM_NW - 1) / TM_NW;
        for (int i = 0; i < ni; ++i) {
            const int p = wave_all + TM_NW * i;
            const bool real = p < np;
            const int cb = real ? p / (2 * TM_WU) : 0, u = real ? p - cb * (2 * TM_WU) : 0;
            const int unit = u & 3, row0 = ps * 64 + unit * 16;
            const unsigned off = (real && row0 < nrows) ? (unsigned)(((row0 + urow) * 32 + uchunk) * 2) : 0x80000000u;
            char* const dst = real ? lds + O_W + cb * (2 * TM_WU * 1024) + u * 1024 : lds + O_SCR;
            if (u < TM_WU) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wh, (lptr_tt)dst, 16, off, cb * wtile_b, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wl, (lptr_tt)dst, 16, off, cb * wtile_b, 0, 0);
        }
    };
    issue_weights(0);

    // ---- 8 waves: pixel tile (32 of the 64 halo rows; rows >= 48 are never used), output tile (32 of the pass's 64 rows), k-half ----
    const int ptile = wave_all % TM_NPT, ntile = (wave_all / TM_NPT) & 1, khalf = wave_all / (2 * TM_NPT);
    const int kq = khalf * 2 + kh;
    const int hp = ptile * 32 + l31, wr = ntile * 32 + l31;
    const int ao = hp * 64 + ((kq ^ ((hp >> 2) & 3)) * 16);
    const int wo = wr * 64 + ((kq ^ ((wr >> 2) & 3)) * 16);
    float* const Y = reinterpret_cast<float*>(lds + O_Y);
    // shifted sum: item = (output channel, pixel of the 2 x 10 patch); an item's partial sums over the passes stay in a register
    constexpr int NPX = TM_TH * TM_TW;                         // 20 output pixels
    constexpr int NITEM = (NPX * 28 + 64 * TM_NW - 1) / (64 * TM_NW);   // Cout <= 28: two items per thread at most
    float part[NITEM];
#pragma unroll
    for (int k = 0; k < NITEM; ++k) part[k] = 0.f;

    for (int ps = 0; ps < npass; ++ps) {
        f32x16t hh, x1, x2;
#pragma unroll
        for (int r = 0; r < 16; ++r) hh[r] = x1[r] = x2[r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (first pass: the halo too)
        __builtin_amdgcn_s_barrier();
        for (int cb = 0; cb < a.CB; ++cb) {
            const char* ab = lds + cb * (2 * TM_AU * 1024);
            const char* wb = lds + O_W + cb * (2 * TM_WU * 1024);
            const half8 xh = *reinterpret_cast<const half8*>(ab + ao);
            const half8 xl = *reinterpret_cast<const half8*>(ab + TM_AU * 1024 + ao);
            const half8 wh = *reinterpret_cast<const half8*>(wb + wo);
            const half8 wl = *reinterpret_cast<const half8*>(wb + TM_WU * 1024 + wo);
            hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hh, 0, 0, 0);     // D[output row][pixel]
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, x2, 0, 0, 0);
        }
        // Y[khalf][halo pixel][64 rows of this pass]: lane = pixel, registers 4j .. 4j+3 = rows 8j + 4 kh + 0..3 of the wave's tile.  (The previous
        // pass's readers of Y passed the barrier at the top of this pass.)
        {
            float* yr = Y + ((khalf * TM_YR + hp) * TM_YS) + ntile * 32 + 4 * kh;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(yr + 8 * j) = make_float4(hh[4 * j] + (x1[4 * j] + x2[4 * j]) * bflow::SPLIT_LO_INV,
                                                                     hh[4 * j + 1] + (x1[4 * j + 1] + x2[4 * j + 1]) * bflow::SPLIT_LO_INV,
                                                                     hh[4 * j + 2] + (x1[4 * j + 2] + x2[4 * j + 2]) * bflow::SPLIT_LO_INV,
                                                                     hh[4 * j + 3] + (x1[4 * j + 3] + x2[4 * j + 3]) * bflow::SPLIT_LO_INV);
        }
        __syncthreads();                                       // Y complete; every wave has read this pass's weights
        if (ps + 1 < npass) issue_weights(ps + 1);             // the next pass's weights land under the shifted sum
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            const int it = tid + k * 64 * TM_NW;
            if (it < NPX * a.Cout) {
                const int co = it / NPX, o = it - co * NPX;
                const int oy = o / TM_TW, ox = o - oy * TM_TW;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int row = t * a.Cout + co - ps * 64;                  // row of (tap, channel) inside this pass, if it is in it
                    if (row >= 0 && row < 64) {
                        const int hpx = (oy + t / 3) * TM_HWD + ox + t % 3;
                        part[k] += Y[hpx * TM_YS + row] + Y[(TM_YR + hpx) * TM_YS + row];
                    }
                }
            }
        }
    }
    // ---- bias + P += dP + the Bezier channels of the GRU input ----
    const int HW = a.H * a.W;
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
        const int it = tid + k * 64 * TM_NW;
        if (it < NPX * a.Cout) {
            const int co = it / NPX, o = it - co * NPX;
            const int y = y0 + o / TM_TW, x = x0 + o % TM_TW;
            if (y < a.H && x < a.W) {
                const int n = y * a.W + x;
                float* pa = a.acc + ((long long)b * a.Cout + co) * HW + n;
                const float v = *pa + (a.bias ? a.bias[co] : 0.f) + part[k];    // bezier.py:137-139: params += delta (+ the conv bias)
                *pa = v;
                if (a.oh) {
                    _Float16 hi, lo;
                    bflow::split1(v, hi, lo);
                    const long long oo = (((long long)b * a.CBo + a.cb_off) * a.P_out + n) * 32 + a.c_off + co;
                    a.oh[oo] = hi;
                    a.ol[oo] = lo;
                }
            }
        }
    }
    if (a.oh && a.c_off == 0) {                                // a block of its own: channels [Cout, 32) are zero
        for (int it = tid; it < NPX * 32; it += 64 * TM_NW) {
            const int o = it >> 5, c = it & 31;
            const int y = y0 + o / TM_TW, x = x0 + o % TM_TW;
            if (c >= a.Cout && y < a.H && x < a.W) {
                const long long oo = (((long long)b * a.CBo + a.cb_off) * a.P_out + y * a.W + x) * 32 + c;
                a.oh[oo] = (_Float16)0.f;
                a.ol[oo] = (_Float16)0.f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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

extern "C" int bflow_conv_thin_mfma_acc(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, int cout_pad, const float* bias,
                                        float* acc_nchw, void* out_hi, void* out_lo, int B, int H, int W, int C, int in_rows_per_image, int Cout,
                                        int out_channel_blocks, int out_block, int out_rows_per_image, int out_channel_in_block, bflow_stream_t stream) {
    BFLOW_REQUIRE(x_hi && x_lo && w_hi && w_lo && acc_nchw, BFLOW_E_ARG, "conv_thin_mfma_acc: null pointer");
    BFLOW_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && C <= 256 && Cout >= 1 && Cout <= 28 && cout_pad >= (9 * Cout + 63) / 64 * 64, BFLOW_E_ARG,
                  "conv_thin_mfma_acc: needs C %% 32 == 0, C <= 256, Cout <= 28, cout_pad >= 9 * Cout rounded up to 64 (got C=%d Cout=%d cout_pad=%d)", C, Cout, cout_pad);
    BFLOW_REQUIRE(in_rows_per_image >= H * W && B <= 65535, BFLOW_E_ARG, "conv_thin_mfma_acc: bad row count / batch");
    BFLOW_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), BFLOW_E_ARG, "conv_thin_mfma_acc: out_hi / out_lo go together");
    if (out_hi) BFLOW_REQUIRE(out_block >= 0 && out_block < out_channel_blocks && out_rows_per_image >= H * W && out_channel_in_block >= 0 &&
                                  out_channel_in_block + Cout <= 32, BFLOW_E_ARG, "conv_thin_mfma_acc: bad output block");
    ThinMArgs a;
    a.xh = (const _Float16*)x_hi; a.xl = (const _Float16*)x_lo; a.wh = (const _Float16*)w_hi; a.wl = (const _Float16*)w_lo;
    a.cout_pad = cout_pad; a.bias = bias; a.acc = acc_nchw; a.oh = (_Float16*)out_hi; a.ol = (_Float16*)out_lo;
    a.B = B; a.H = H; a.W = W; a.CB = C / 32; a.P_in = in_rows_per_image; a.Cout = Cout;
    a.CBo = out_channel_blocks; a.cb_off = out_block; a.P_out = out_rows_per_image; a.c_off = out_channel_in_block;
    const int lds = a.CB * TM_PPB * 1024 + 2 * TM_YR * TM_YS * 4 + 1024;      // halo + one pass of weights + Y + scratch: 132 KB at C = 256
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(bflow::ceil_div(H, TM_TH) * bflow::ceil_div(W, TM_TW), B);
    hipLaunchKernelGGL(conv_thin_mfma_kernel, grid, dim3(64 * TM_NW), lds, (hipStream_t)stream, a);
    return bflow::launch_status("conv_thin_mfma_acc");
}
