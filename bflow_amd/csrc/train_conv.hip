// Helpers of the training convolutions (bflow_amd/conv_train.py); the MFMA work itself is bflow_conv_split.
//   bflow_wgrad_pack          operands of the weight-gradient GEMM: NCHW fp32 -> blocked split with the PIXEL index in the block position
//   bflow_blocked_f32_to_nchw blocked fp32 (B, C/32, P, 32) -> NCHW fp32, optionally times a device scalar (un-scaling of gradients)
//   bflow_pow2_scale          s = 2^floor(log2(target / max|x|)) and 1/s on the device (no host synchronisation)
//   bflow_grad_stats          the same scale AND the bias gradient (per-channel sum) in one pass over an NCHW gradient
#include "common.h"
#include <algorithm>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// thread = (k-block, row n, 8 consecutive k): one 16-B store per plane.  layout 0: dst (taps, KB, rows, 32), n = c, tap = blockIdx.y;
// layout 1: dst (KB, rows, 32) with n = tap * C + c (rows >= taps * C; the rows past that are written as zeros).
__global__ __launch_bounds__(256) void wgrad_pack_kernel(const float* __restrict__ src, _Float16* __restrict__ dh, _Float16* __restrict__ dl, int B, int C,
                                                         int H, int W, int Ho, int Wo, int KW, int ntaps, int stride, int pad_h, int pad_w, int rows,
                                                         int KB, int layout, const float* __restrict__ scale_p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;          // (kb, n, j8) with j8 fastest: 4 threads per 64-B row
    if (t >= (long long)KB * rows * 4) return;
    const int j8 = (int)(t & 3);
    const long long rc = t >> 2;
    const int n = (int)(rc % rows);
    const int kb = (int)(rc / rows);
    int tap, c;
    if (layout == 0) { tap = blockIdx.y; c = n; }
    else { tap = n / C; c = n - tap * C; }
    const bool row_ok = c < C && tap < ntaps;
    const float scale = scale_p ? *scale_p : 1.f;
    const int r = tap / KW, q = tap - r * KW;
    const int K = B * Ho * Wo;
    // (b, yo, xo) of the first of the 8 pixels, then incremented
    int k = kb * 32 + j8 * 8;
    int b = k / (Ho * Wo), p = k - b * (Ho * Wo);
    int yo = p / Wo, xo = p - yo * Wo;
    half8 h8, l8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = 0.f;
        if (row_ok && k < K) {
            const int y = yo * stride + r - pad_h, x = xo * stride + q - pad_w;
            if (y >= 0 && y < H && x >= 0 && x < W) v = src[(((long long)b * C + c) * H + y) * W + x] * scale;
        }
        _Float16 hi, lo;
        bflow::split1(v, hi, lo);
        h8[i] = hi;
        l8[i] = lo;
        ++k;
        if (++xo == Wo) { xo = 0; if (++yo == Ho) { yo = 0; ++b; } }
    }
    const long long o = layout == 0 ? (((long long)tap * KB + kb) * rows + n) * 32 + j8 * 8 : ((long long)kb * rows + n) * 32 + j8 * 8;
    *reinterpret_cast<half8*>(dh + o) = h8;
    *reinterpret_cast<half8*>(dl + o) = l8;
}

// 64 pixels x 32 channels through LDS (coalesced on both sides)
__global__ __launch_bounds__(256) void blocked_f32_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int CB, int P, int C,
                                                                  const float* __restrict__ scale_p) {
    __shared__ float tile[64][33];
    const int b = blockIdx.z, cb = blockIdx.y, p0 = blockIdx.x * 64;
    const float scale = scale_p ? *scale_p : 1.f;
    {
        const int pl = threadIdx.x >> 2, c8 = (threadIdx.x & 3) * 8;
        const int pix = p0 + pl;
        if (pix < HW) {
            const float4* s4 = reinterpret_cast<const float4*>(x + (((long long)b * CB + cb) * P + pix) * 32 + c8);
            const float4 a = s4[0], c = s4[1];
            tile[pl][c8 + 0] = a.x; tile[pl][c8 + 1] = a.y; tile[pl][c8 + 2] = a.z; tile[pl][c8 + 3] = a.w;
            tile[pl][c8 + 4] = c.x; tile[pl][c8 + 5] = c.y; tile[pl][c8 + 6] = c.z; tile[pl][c8 + 7] = c.w;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = cb * 32 + ty + 4 * i, pix = p0 + tx;
        if (ch < C && pix < HW) out[((long long)b * C + ch) * HW + pix] = tile[tx][ty + 4 * i] * scale;
    }
}

// max |x| over n elements -> out[0] = 2^floor(log2(target / max)), out[1] = 1 / out[0]; work = {max bits, ticket}, zero on entry and
// left zero on exit (the last workgroup resets it), so one persistent work buffer serves every call on a stream.
__global__ __launch_bounds__(256) void pow2_scale_kernel(const float* __restrict__ x, long long n, float target, float* __restrict__ out,
                                                         unsigned* __restrict__ work) {
    float m = 0.f;
    const long long n4 = ((reinterpret_cast<size_t>(x) & 15) == 0) ? n >> 2 : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (!(m == m)) m = __builtin_inff();                        // NaN in the gradient: scale 1 (below)
        atomicMax(work, __float_as_uint(m));                        // non-negative floats order like their bit patterns
        __threadfence();
        last = atomicAdd(work + 1, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        const float amax = __uint_as_float(atomicMax(work, 0u));
        float e = floorf(log2f(target / amax));
        if (!(amax > 0.f) || !(fabsf(e) <= 60.f)) e = (amax > 0.f && e > 60.f) ? 60.f : (amax > 0.f && e < -60.f ? -60.f : 0.f);
        out[0] = exp2f(e);
        out[1] = exp2f(-e);
        work[0] = 0u;
        work[1] = 0u;
    }
}

// x (R, P, C) fp32 row-major [pixel][channel] -> blocked split (R, CB, P, 32), value * scale, channels >= C zero: the activation operand of the
// conv engine from a ROW-major matrix (bflow_norm_act_split stages the column-major / NCHW case).  thread = 8 channels of one pixel.
__global__ __launch_bounds__(256) void rows_to_split_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int R,
                                                            int P, int C, int CB, const float* __restrict__ scale_p) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const float scale = scale_p ? *scale_p : 1.f;
    const long long total = (long long)R * CB * P * 4;
    const bool vec = (C & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int chunk = (int)(i & 3);
        const long long t1 = i >> 2;
        const int p = (int)(t1 % P);
        const long long t2 = t1 / P;
        const int cb = (int)(t2 % CB), r = (int)(t2 / CB);
        const int c0 = cb * 32 + chunk * 8;
        const float* src = x + ((long long)r * P + p) * C + c0;
        float v[8];
        if (vec && c0 + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c0 + j < C ? src[j] : 0.f;
        }
        half8 h8, l8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 hh, ll;
            bflow::split1(v[j] * scale, hh, ll);
            h8[j] = hh;
            l8[j] = ll;
        }
        *reinterpret_cast<half8*>(hi + i * 8) = h8;      // (((r * CB + cb) * P + p) * 4 + chunk) * 8 == i * 8
        *reinterpret_cast<half8*>(lo + i * 8) = l8;
    }
}

// bflow_grad_stats: ONE pass over an NCHW gradient for both things the backward of a convolution needs from all of it: the power-of-two
// scale (as pow2_scale_kernel) and the bias gradient db[c] = sum over (b, y, x).  Two launches instead of tickets and fences (a release
// fence is an L2 write-back on this chip): grad_stats_kernel -- one WAVE per item = (plane, segment of GS_SEG elements), four independent
// 16-B loads per lane, per-item sums to `partial`, per-workgroup max to `partial + items` --, grad_stats_final_kernel -- one workgroup adds
// them per channel in a fixed order (deterministic, nothing to zero beforehand).
constexpr int GS_SEG = 1024;
__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ x, int planes, int HW, int segs, float* __restrict__ partial) {
    __shared__ float sm[4];
    const int items = planes * segs;
    const bool vec = (HW & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m = 0.f;
    for (int item = blockIdx.x * 4 + wave; item < items; item += gridDim.x * 4) {
        const int plane = item / segs, seg = item - plane * segs;
        const float* base = x + (long long)plane * HW;
        float s = 0.f;
        if (vec) {
            const int n4 = HW >> 2, j0 = seg * (GS_SEG / 4) + lane;
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(base)[min(j0 + 64 * k, n4 - 1)];   // clamped: no branch around a load
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = j0 + 64 * k < n4;
                const float sk = (v[k].x + v[k].y) + (v[k].z + v[k].w);
                const float mk = fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
                s += in ? sk : 0.f;
                m = fmaxf(m, in ? mk : 0.f);
            }
        } else {
            const int i0 = seg * GS_SEG + lane;
#pragma unroll 4
            for (int k = 0; k < GS_SEG / 64; ++k) {
                const float v = base[min(i0 + 64 * k, HW - 1)];
                const bool in = i0 + 64 * k < HW;
                s += in ? v : 0.f;
                m = fmaxf(m, in ? fabsf(v) : 0.f);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) partial[item] = s;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) sm[wave] = (m == m) ? m : __builtin_inff();      // NaN in the gradient: scale 1 (below)
    __syncthreads();
    if (threadIdx.x == 0) partial[items + blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__global__ __launch_bounds__(256) void grad_stats_final_kernel(const float* __restrict__ partial, int B, int C, int segs, int blocks, float target,
                                                               float* __restrict__ out, float* __restrict__ db) {
    __shared__ float sm[4];
    const int items = B * C * segs;
    float m = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 256) m = fmaxf(m, partial[items + i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    __shared__ float s_scale;
    if (threadIdx.x == 0) {
        const float amax = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        float e = floorf(log2f(target / amax));
        if (!(amax > 0.f) || !(fabsf(e) <= 60.f)) e = (amax > 0.f && e > 60.f) ? 60.f : (amax > 0.f && e < -60.f ? -60.f : 0.f);
        out[0] = s_scale = exp2f(e);
        out[1] = exp2f(-e);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) out[2 + c] = s_scale;       // the scale once per channel: bflow_norm_act_split's scale_a of the staging pass
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float* p = partial + (long long)(b * C + c) * segs;
            for (int sg = 0; sg < segs; ++sg) acc += p[sg];
        }
        db[c] = acc;
    }
}

// dw[co, ci, tap] = inv * sum_g part[...]: the engine's blocked fp32 partial results of the k-chunks -> the filter gradient in (Cout, Cin, KH*KW)
// order.  orientation 0: part (taps, G, CB, rows = Cin, 32) [co = 32 cb + j];  orientation 1: part (G, NB, rows = Cout, 32) [n = 32 nb + j = tap*Cin + ci].
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G, int Cout, int Cin, int taps,
                                                           int blocks, int rows, int orientation, const float* __restrict__ inv_p) {
    // thread = one element of a chunk's partial result in ITS order (coalesced reads over the G chunks); the small dw takes the scattered writes
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_img = (long long)blocks * rows * 32;
    const int j = (int)(i & 31);
    const int row = (int)((i >> 5) % rows);
    const int blk = (int)(((i >> 5) / rows) % blocks);
    float acc = 0.f;
    if (orientation == 0) {
        const int tap = (int)(i / per_img);
        const int co = blk * 32 + j, ci = row;
        if (tap >= taps || co >= Cout || ci >= Cin) return;
        const float* p = part + (long long)tap * G * per_img + ((long long)blk * rows + row) * 32 + j;
        for (int g = 0; g < G; ++g) acc += p[(long long)g * per_img];
        dw[((long long)co * Cin + ci) * taps + tap] = acc * (inv_p ? *inv_p : 1.f);
    } else {
        const int n = blk * 32 + j, co = row;
        if (i >= per_img || n >= taps * Cin || co >= Cout) return;
        const int tap = n / Cin, ci = n - tap * Cin;
        const float* p = part + ((long long)blk * rows + row) * 32 + j;
        for (int g = 0; g < G; ++g) acc += p[(long long)g * per_img];
        dw[((long long)co * Cin + ci) * taps + tap] = acc * (inv_p ? *inv_p : 1.f);
    }
}

}  // namespace

static int bflow_grad_stats_blocks(int B, int C, int HW) {
    const long long items = (long long)B * C * bflow::ceil_div(HW, GS_SEG);
    return (int)std::max<long long>(1, std::min<long long>((items + 3) / 4, 1024));
}

extern "C" int bflow_wgrad_pack(const float* src, void* dst_hi, void* dst_lo, int B, int C, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                                int pad_h, int pad_w, int rows, int k_blocks, int taps_in_rows, const float* scale, bflow_stream_t stream) {
    BFLOW_REQUIRE(src && dst_hi && dst_lo, BFLOW_E_ARG, "wgrad_pack: null pointer");
    BFLOW_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride >= 1, BFLOW_E_ARG, "wgrad_pack: bad sizes");
    BFLOW_REQUIRE(rows >= (taps_in_rows ? KH * KW * C : C), BFLOW_E_ARG, "wgrad_pack: %d rows do not hold the operand", rows);
    BFLOW_REQUIRE((long long)k_blocks * 32 >= (long long)B * Ho * Wo && KH * KW <= 65535, BFLOW_E_ARG, "wgrad_pack: k_blocks too small");
    BFLOW_REQUIRE((long long)B * Ho * Wo < (1LL << 31) - 64, BFLOW_E_LIMIT, "wgrad_pack: more than 2^31 pixels");
    const long long per = (long long)k_blocks * rows * 4;
    dim3 grid(bflow::ceil_div(per, 256), taps_in_rows ? 1 : KH * KW);
    hipLaunchKernelGGL(wgrad_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst_hi, (_Float16*)dst_lo, B, C, H, W, Ho, Wo, KW,
                       KH * KW, stride, pad_h, pad_w, rows, k_blocks, taps_in_rows ? 1 : 0, scale);
    return bflow::launch_status("wgrad_pack");
}

extern "C" int bflow_blocked_f32_to_nchw(const float* x, float* out, int B, int HW, int C, int channel_blocks, int rows_per_image, const float* scale,
                                         bflow_stream_t stream) {
    BFLOW_REQUIRE(x && out && B > 0 && HW > 0 && C > 0 && channel_blocks * 32 >= C && rows_per_image >= HW && B <= 65535, BFLOW_E_ARG,
                  "blocked_f32_to_nchw: bad arguments");
    dim3 grid(bflow::ceil_div(HW, 64), bflow::ceil_div(C, 32), B);
    hipLaunchKernelGGL(blocked_f32_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, HW, channel_blocks, rows_per_image, C, scale);
    return bflow::launch_status("blocked_f32_to_nchw");
}

extern "C" int bflow_pow2_scale(const float* x, long long n, float target, float* out2, void* work8, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && out2 && work8 && n > 0 && target > 0.f, BFLOW_E_ARG, "pow2_scale: bad arguments");
    const int blocks = (int)std::min<long long>(512, (n + 16383) / 16384);
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, (hipStream_t)stream, x, n, target, out2, (unsigned*)work8);
    return bflow::launch_status("pow2_scale");
}

extern "C" int bflow_rows_to_split(const float* x, void* out_hi, void* out_lo, int R, int P, int C, const float* scale, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && out_hi && out_lo && R > 0 && P > 0 && C > 0, BFLOW_E_ARG, "rows_to_split: bad arguments");
    const int CB = (C + 31) / 32;
    const long long total = (long long)R * CB * P * 4;
    hipLaunchKernelGGL(rows_to_split_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)out_hi, (_Float16*)out_lo,
                       R, P, C, CB, scale);
    return bflow::launch_status("rows_to_split");
}

extern "C" int bflow_grad_stats(const float* x, int B, int C, int HW, float target, float* out2, float* partial, float* dbias, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && out2 && partial && dbias && B > 0 && C > 0 && HW > 0 && target > 0.f, BFLOW_E_ARG, "grad_stats: bad arguments");
    const int segs = bflow::ceil_div(HW, GS_SEG);
    BFLOW_REQUIRE((long long)B * C * segs < (1LL << 30), BFLOW_E_LIMIT, "grad_stats: too many planes");
    const int blocks = bflow_grad_stats_blocks(B, C, HW);
    hipLaunchKernelGGL(grad_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B * C, HW, segs, partial);
    hipLaunchKernelGGL(grad_stats_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, B, C, segs, blocks, target, out2, dbias);
    return bflow::launch_status("grad_stats");
}

extern "C" int bflow_wgrad_reduce(const float* part, float* dw, int G, int Cout, int Cin, int taps, int blocks, int rows, int orientation,
                                  const float* inv_scale, bflow_stream_t stream) {
    BFLOW_REQUIRE(part && dw && G > 0 && Cout > 0 && Cin > 0 && taps > 0 && blocks > 0 && rows > 0, BFLOW_E_ARG, "wgrad_reduce: bad arguments");
    BFLOW_REQUIRE(orientation == 0 ? (blocks * 32 >= Cout && rows >= Cin) : (blocks * 32 >= taps * Cin && rows >= Cout), BFLOW_E_ARG,
                  "wgrad_reduce: partial results too small");
    const long long n = (long long)blocks * rows * 32 * (orientation == 0 ? taps : 1);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bflow::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, part, dw, G, Cout, Cin, taps, blocks, rows,
                       orientation, inv_scale);
    return bflow::launch_status("wgrad_reduce");
}
