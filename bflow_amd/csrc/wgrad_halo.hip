// Weight gradient of a stride-1 "same" convolution straight from the conv engine's activation layout:
//     dW[co, ci, r, q] = sum_{b, y, x} dY[b, y, x, co] * X[b, y + r - ph, x + q - pw, ci]
// Both operands are blocked split tensors (B, C/32, P, 32) -- X as the forward staged it, dY as the input-gradient pass staged it --
// so nothing is re-packed (bflow_wgrad_pack writes KH*KW shifted copies of X; this kernel reads X once per item through a halo).
// The contraction index is the PIXEL: the MFMA fragments need 8 consecutive pixels of one channel, i.e. a column of the
// row-major [pixel][32 channels] LDS tile.  gfx950's ds_read_b64_tr_b16 delivers exactly that (tools/micro/tr_b16.hip: inside a
// group of 16 lanes out[l][j] = M[4 j + (l >> 2)][l & 3], M[i][.] = the 8 bytes at lane i's address): fetch lane i reads row
// p0 + (i >> 2), 8-B chunk (i & 3) of the 32-B half row holding the group's 16 channels.
//   workgroup (4 waves, one per SIMD, 512 registers): output tile = 64 output channels x CIB input-channel blocks x all taps; items =
//     (image, 8 x 16 pixel patch) dealt round-robin over the k-split; per item the dY patch (128 rows x 2 blocks) and the X halo
//     ((8+KH-1) x (16+KW-1) rows x CIB blocks) are staged by LDS-DMA (out-of-image rows = out-of-range offsets = zeros);
//   wave w owns output block m = w & 1 and the units (tap, input block) u = (w >> 1), (w >> 1) + 2, ...: per 16-pixel step one dY fragment
//     pair + per unit one X fragment pair (fetched one step ahead), 3 MFMAs per unit: hi*hi, lo*hi, hi*lo into fp32 accumulators;
//     two staging buffers: the next item's LDS-DMA pieces are issued between the steps of the current one;
//   at the end every wave adds its tiles to dW[tap][co][ci] (fp32 atomics: k-split workgroups and taps never share an address
//     within a wave, different workgroups do).
#include "common.h"
#include <algorithm>
#include <type_traits>

#ifndef WG_ABL
#define WG_ABL 0       // tools/wgrad_ablate.sh: 1 = no MFMAs, 2 = no staging after the first item, 4 = no fragment reads, 8 = no atomics
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct WgradArgs {
    const _Float16 *xh, *xl;   // (B, CBi, P, 32)
    const _Float16 *gh, *gl;   // (B, CBo, P, 32)   dY (pre-scaled)
    float* dw;                 // (taps, Cout_pad64, Cin_pad) fp32, zero on entry
    int B, H, W, CBi, CBo, P, cout_pad, cin_pad, ksplit, pad_h, pad_w;
};

template <int IMM>
__device__ __forceinline__ half4 tr_read(unsigned addr) {     // asynchronous: the destination is valid only after SETTLE
    half4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
    return v;
}

// The reads are asynchronous: their destination registers are only valid after the wait, and the compiler must know it -- a statement
// after the wait therefore "redefines" them ("+v"; volatile asms keep their order), otherwise it may copy a destination (packing two
// half4 into a half8) before the data arrived.
#define TR_DEF(A, B, C, D) asm volatile("" : "+v"(A), "+v"(B), "+v"(C), "+v"(D));

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int KH, int KW, int CIB>
__global__ __launch_bounds__(256, 1) void wgrad_halo_kernel(WgradArgs a) {
    constexpr int TH = 8, TW = 16, NTAPS = KH * KW;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);
    constexpr int KX = (HR + 63) / 64, XU = 4 * KX;          // 1-KB pieces per (block, plane) of the halo: a multiple of 4, wave w stages pieces w + 4 k
    constexpr int X_PLANE = XU * 1024, G_PLANE = 128 * 64;   // bytes
    constexpr int O_X = 2 * 2 * G_PLANE;                     // dY: [2 blocks][2 planes][128 rows][64 B]; then X: [CIB][2 planes][XU KB]
    constexpr int BUF = O_X + CIB * 2 * X_PLANE;             // one staging buffer; two of them: the next item lands under the current one's MFMAs
    constexpr int UNITS = NTAPS * CIB, UPW = (UNITS + 1) / 2;   // wave w: output block m = w & 1, units (w >> 1) + 2 k, k < UPW
    constexpr int NDMA = 8 + 2 * CIB * KX, SLOTS = 8 * UPW;  // LDS-DMA instructions per wave and item / (step, unit) slots they are spread over
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wu = wave >> 1;
    const int cig = blockIdx.x, cot = blockIdx.y, ks = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int items = a.B * tiles_x * tiles_y;

    f32x16 hh[UPW], xx[UPW];
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hh[u][r] = 0.f; xx[u][r] = 0.f; }

    // ---- fragment addressing (see the header): group of 16 lanes g = (lane >> 4) & 1 -> channels 16 g .., k-group kg = lane >> 5 -> pixels + 8 kg
    const int i16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    const unsigned frag = (unsigned)(size_t)lds + (i16 >> 2) * 64 + (16 * g16 + 4 * (i16 & 3)) * 2 + kg * 8 * 64;
    const unsigned g_frag = frag + wm * 2 * G_PLANE;         // + buffer, + step * 1024 (immediate)
    unsigned x_frag[UPW];                                    // + buffer, + step * HWD * 64 (immediate)
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = min(wu + 2 * u, UNITS - 1);
        const int tap = unit / CIB, cibl = unit - tap * CIB, r = tap / KW, q = tap - r * KW;
        x_frag[u] = frag + O_X + (cibl * 2) * X_PLANE + (r * HWD + q) * 64;
    }
    const bool last_unit = wu + 2 * (UPW - 1) < UNITS;       // only the last unit is ragged over the waves

    // ---- staging (LDS-DMA, 1-KB pieces, plain row-major): lane -> (row of the 16-row piece, 16-B chunk); piece p of a plane belongs to wave p & 3
    const int urow = lane >> 2, uchunk2 = (lane & 3) * 16;
    int hy[KX], hx[KX];                                      // halo position of this lane's row in X piece wave + 4 k
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int row = (wave + 4 * k) * 16 + urow;
        hy[k] = row / HWD;
        hx[k] = row < HR ? row - hy[k] * HWD : (1 << 20);    // rows past the halo: never inside the image
    }
    unsigned voff_g[2], voff_x[KX];
    rsrc_t rgh, rgl, rxh, rxl;
    auto stage_setup = [&](int it, bool on) {                // per item: image base -> 4 resources, per-lane byte offsets (or out of range = zeros)
        const int b = it / (tiles_x * tiles_y), pt = it - b * (tiles_x * tiles_y);
        const int y0 = (pt / tiles_x) * TH, x0 = (pt - (pt / tiles_x) * tiles_x) * TW;
        const long long og = ((long long)b * a.CBo) * a.P * 32, ox = ((long long)b * a.CBi) * a.P * 32;
        rgh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gh + og), 0, a.CBo * a.P * 64, 0x00020000);
        rgl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gl + og), 0, a.CBo * a.P * 64, 0x00020000);
        rxh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xh + ox), 0, a.CBi * a.P * 64, 0x00020000);
        rxl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xl + ox), 0, a.CBi * a.P * 64, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = y0 + wave + 4 * j, x = x0 + urow;
            voff_g[j] = (on & (y < a.H) & (x < a.W)) ? (unsigned)((y * a.W + x) * 64 + uchunk2) : 0x80000000u;
        }
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            const int y = y0 - a.pad_h + hy[k], x = x0 - a.pad_w + hx[k];
            voff_x[k] = (on & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W)) ? (unsigned)((y * a.W + x) * 64 + uchunk2) : 0x80000000u;
        }
    };
    auto dma = [&](auto D_, int buf) {                       // DMA instruction D of this wave's share
        constexpr int D = decltype(D_)::value;
        char* const dst = lds + buf * BUF;
        if constexpr (D < 8) {                               // dY: piece wave + 4 j of (block, plane)
            constexpr int j = D & 1, plane = (D >> 1) & 1, blk = D >> 2;
            const int cb = cot * 2 + blk;
            const unsigned off = cb < a.CBo ? voff_g[j] + (unsigned)(cb * a.P * 64) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane ? rgl : rgh, (lptr_t)(dst + (blk * 2 + plane) * G_PLANE + (wave + 4 * j) * 1024), 16, off, 0, 0, 0);
        } else {                                             // X halo: piece wave + 4 k of (input block, plane)
            constexpr int e = D - 8, k = e % KX, plane = (e / KX) & 1, cibl = e / (2 * KX);
            const int cb = cig * CIB + cibl;
            const unsigned off = cb < a.CBi ? voff_x[k] + (unsigned)(cb * a.P * 64) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane ? rxl : rxh, (lptr_t)(dst + O_X + (cibl * 2 + plane) * X_PLANE + (wave + 4 * k) * 1024), 16, off, 0, 0, 0);
        }
    };

    // fragments of one 16-pixel step: g = this wave's dY block {hi px 0..7 | hi 8..15 | lo | lo}, x[u] the same for unit u's X rows
    struct Frags { half4 g[4]; half4 x[UPW][4]; };
    auto cat = [](half4 p, half4 q) { return half8{p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]}; };
    auto read_g = [&](auto S_, unsigned bo, Frags& f) {
        constexpr int O = decltype(S_)::value * 1024;
        f.g[0] = tr_read<O>(g_frag + bo); f.g[1] = tr_read<O + 256>(g_frag + bo);
        f.g[2] = tr_read<O + G_PLANE>(g_frag + bo); f.g[3] = tr_read<O + G_PLANE + 256>(g_frag + bo);
    };
    auto read_x = [&](auto S_, int u, unsigned bo, Frags& f) {
        constexpr int O = decltype(S_)::value * HWD * 64;
        f.x[u][0] = tr_read<O>(x_frag[u] + bo); f.x[u][1] = tr_read<O + 256>(x_frag[u] + bo);
        f.x[u][2] = tr_read<O + X_PLANE>(x_frag[u] + bo); f.x[u][3] = tr_read<O + X_PLANE + 256>(x_frag[u] + bo);
    };
    auto settle = [&](Frags& f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TR_DEF(f.g[0], f.g[1], f.g[2], f.g[3])
#pragma unroll
        for (int u = 0; u < UPW; ++u) TR_DEF(f.x[u][0], f.x[u][1], f.x[u][2], f.x[u][3])
    };

    int cur = 0;
    if (ks < items) {
        stage_setup(ks, true);
        static_for<NDMA>([&](auto D_) { dma(D_, 0); });
    }
    for (int it = ks; it < items; it += a.ksplit) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of item `it` landed ...
        __syncthreads();                                     // ... everybody's did, and everybody is done reading the other buffer
        const bool more = it + a.ksplit < items;
        stage_setup(more ? it + a.ksplit : it, more && !(WG_ABL & 2));   // no next item: every piece out of range (zeros into the idle buffer)
        const unsigned bo = cur * BUF;
        cur ^= 1;
        // ---- 8 steps of 16 pixels (one patch row each).  Program order per (step s, unit u) slot: the fragment reads of (s + 1, u), this slot's
        //      share of the next item's LDS-DMA, the three MFMAs of (s, u) -- pinned by sched_barriers: one wave per SIMD, so whatever is not
        //      issued between MFMAs is paid on top of them.
        Frags f[2];
        read_g(std::integral_constant<int, 0>{}, bo, f[0]);
#pragma unroll
        for (int u = 0; u < UPW; ++u)
            if (u + 1 < UPW || last_unit) read_x(std::integral_constant<int, 0>{}, u, bo, f[0]);
        settle(f[0]);
        static_for<8>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            Frags& c = f[S & 1];
            Frags& n = f[(S + 1) & 1];
            const half8 ah = cat(c.g[0], c.g[1]), al = cat(c.g[2], c.g[3]);
            if constexpr (S < 7) { if (!(WG_ABL & 4)) read_g(std::integral_constant<int, S + 1>{}, bo, n); }
            static_for<UPW>([&](auto U_) {
                constexpr int U = decltype(U_)::value;
                if (U + 1 < UPW || last_unit) {
                    if constexpr (S < 7) { if (!(WG_ABL & 4)) read_x(std::integral_constant<int, S + 1>{}, U, bo, n); }
                    const half8 bh = cat(c.x[U][0], c.x[U][1]), bl = cat(c.x[U][2], c.x[U][3]);
                    if (WG_ABL & 1) {
                        hh[U][0] += (float)bh[0] + (float)ah[0]; xx[U][0] += (float)bl[0] + (float)al[0];
                    } else {
                        hh[U] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hh[U], 0, 0, 0);      // D[co][ci]
                        xx[U] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, xx[U], 0, 0, 0);
                        xx[U] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, xx[U], 0, 0, 0);
                    }
                }
                constexpr int slot = S * UPW + U;            // DMA instructions d with d * SLOTS / NDMA == slot
                static_for<NDMA>([&](auto D_) {
                    if constexpr (decltype(D_)::value * SLOTS / NDMA == slot) dma(D_, cur);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (S < 7) { if (!(WG_ABL & 4)) settle(n); }
        });
    }
    // ---- dW[tap][co][ci] += tile: lane = ci column (lane & 31), register r = co row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wu + 2 * u;
        if (unit >= UNITS) continue;
        const int tap = unit / CIB, cibl = unit - tap * CIB;
        const int ci = (cig * CIB + cibl) * 32 + (lane & 31);
        if (ci >= a.cin_pad) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cot * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < a.cout_pad && (!(WG_ABL & 8) || hh[u][r] == 123.f)) atomicAdd(a.dw + ((long long)tap * a.cout_pad + co) * a.cin_pad + ci, hh[u][r] + xx[u][r] * bflow::SPLIT_LO_INV);
        }
    }
}

// dw[co][ci][tap] = inv * acc[tap][co][ci]; acc is left ZERO (the accumulator is a persistent buffer: the next call adds into zeros again)
__global__ __launch_bounds__(256) void wgrad_finish_kernel(float* __restrict__ acc, float* __restrict__ dw, int taps, int Cout, int Cin, int cout_pad,
                                                           int cin_pad, const float* __restrict__ inv_p) {
    const long long total = (long long)taps * cout_pad * cin_pad;
    const float inv = inv_p ? *inv_p : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % cin_pad);
        const long long t1 = i / cin_pad;
        const int co = (int)(t1 % cout_pad), tap = (int)(t1 / cout_pad);
        const float v = acc[i];
        acc[i] = 0.f;
        if (co < Cout && ci < Cin) dw[((long long)co * Cin + ci) * taps + tap] = v * inv;
    }
}

template <int KH, int KW, int CIB>
void launch(const WgradArgs& a, int items, hipStream_t s) {
    constexpr int HR = (16 + KW - 1) * (8 + KH - 1), XU = 4 * ((HR + 63) / 64);
    const int lds = 2 * (2 * 2 * 128 * 64 + CIB * 2 * XU * 1024);     // two staging buffers
    auto kern = wgrad_halo_kernel<KH, KW, CIB>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(bflow::ceil_div(a.cin_pad / 32, CIB), a.cout_pad / 64, a.ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    (void)items;
}

}  // namespace

extern "C" int bflow_conv_wgrad_finish(float* dw_acc, float* dw, int taps, int Cout, int Cin, int cin_pad, const float* inv_scale, bflow_stream_t stream) {
    BFLOW_REQUIRE(dw_acc && dw && taps > 0 && Cout > 0 && Cin > 0 && cin_pad >= Cin && cin_pad % 32 == 0, BFLOW_E_ARG, "conv_wgrad_finish: bad arguments");
    const int cout_pad = (Cout + 63) / 64 * 64;
    const long long total = (long long)taps * cout_pad * cin_pad;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dw_acc, dw, taps, Cout, Cin, cout_pad,
                       cin_pad, inv_scale);
    return bflow::launch_status("conv_wgrad_finish");
}

extern "C" int bflow_conv_wgrad_halo(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw_acc, int B, int H, int W, int Cin_pad,
                                     int Cout, int rows_per_image, int KH, int KW, bflow_stream_t stream) {
    BFLOW_REQUIRE(x_hi && x_lo && dy_hi && dy_lo && dw_acc, BFLOW_E_ARG, "conv_wgrad_halo: null pointer");
    BFLOW_REQUIRE(B > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cin_pad % 32 == 0 && Cout > 0 && rows_per_image >= H * W, BFLOW_E_ARG, "conv_wgrad_halo: bad sizes");
    const int shape = (KH == 3 && KW == 3) ? 1 : (KH == 1 && KW == 5) ? 2 : (KH == 5 && KW == 1) ? 3 : (KH == 1 && KW == 1) ? 4 : 0;
    BFLOW_REQUIRE(shape, BFLOW_E_ARG, "conv_wgrad_halo: 3x3, 1x5, 5x1 and 1x1 filters (stride 1, same padding) are built (got %dx%d)", KH, KW);
    BFLOW_REQUIRE((long long)((Cin_pad > ((Cout + 31) / 32) * 32 ? Cin_pad : ((Cout + 31) / 32) * 32) / 32) * rows_per_image * 64 < (1LL << 31), BFLOW_E_LIMIT,
                  "conv_wgrad_halo: an image exceeds the 2 GB buffer-addressing window");
    WgradArgs a;
    a.xh = (const _Float16*)x_hi; a.xl = (const _Float16*)x_lo; a.gh = (const _Float16*)dy_hi; a.gl = (const _Float16*)dy_lo;
    a.dw = dw_acc;
    a.B = B; a.H = H; a.W = W; a.CBi = Cin_pad / 32; a.CBo = (Cout + 31) / 32; a.P = rows_per_image;
    a.cout_pad = (Cout + 63) / 64 * 64; a.cin_pad = Cin_pad;
    a.pad_h = KH / 2; a.pad_w = KW / 2;
    const int items = B * bflow::ceil_div(H, 8) * bflow::ceil_div(W, 16);
    const int cib = shape == 1 ? 1 : 2;
    const int tiles = bflow::ceil_div(Cin_pad / 32, cib) * (a.cout_pad / 64);
    a.ksplit = std::max(1, std::min(items, 256 / tiles));       // one workgroup per CU (512 registers per wave): never a second round
    hipStream_t s = (hipStream_t)stream;
    if (shape == 1) launch<3, 3, 1>(a, items, s);
    else if (shape == 2) launch<1, 5, 2>(a, items, s);
    else if (shape == 3) launch<5, 1, 2>(a, items, s);
    else launch<1, 1, 2>(a, items, s);
    return bflow::launch_status("conv_wgrad_halo");
}
