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
//   wave w owns the units (tap, input block) u = w, w+4, w+8: per 16-pixel step 2 x 2 dY fragment pairs + per unit 2 X fragment pairs,
//     3 MFMAs per (output block, unit): hi*hi, lo*hi, hi*lo into fp32 accumulators (2 x UPW x 2 tiles);
//   at the end every wave adds its tiles to dW[tap][co][ci] (fp32 atomics: k-split workgroups and taps never share an address
//     within a wave, different workgroups do).
#include "common.h"
#include <algorithm>

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

__device__ __forceinline__ half4 tr_read(const char* p) {
    half4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}

// The reads are asynchronous: their destination registers are only valid after the wait, and the compiler must know it -- the wait
// statement therefore "redefines" them ("+v"), otherwise it may copy a destination (packing two half4 into a half8) before the data arrived.
#define TR_WAIT(A, B, C, D) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A), "+v"(B), "+v"(C), "+v"(D)::"memory");

template <int KH, int KW, int CIB>
__global__ __launch_bounds__(256, 1) void wgrad_halo_kernel(WgradArgs a) {
    constexpr int TH = 8, TW = 16, NTAPS = KH * KW;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);
    constexpr int XU = (HR + 15) / 16;                       // 1-KB pieces per (block, plane) of the halo
    constexpr int X_PLANE = XU * 1024, G_PLANE = 128 * 64;   // bytes
    constexpr int O_X = 2 * 2 * G_PLANE;                     // dY: [2 blocks][2 planes][128 rows][64 B]; then X: [CIB][2 planes][XU KB]
    constexpr int UNITS = NTAPS * CIB, UPW = (UNITS + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cig = blockIdx.x, cot = blockIdx.y, ks = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int items = a.B * tiles_x * tiles_y;

    f32x16 hh[2][UPW], xx[2][UPW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < UPW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hh[m][u][r] = 0.f; xx[m][u][r] = 0.f; }

    // fragment addressing (see the header): group of 16 lanes g = (lane >> 4) & 1 -> channels 16 g .., k-group kg = lane >> 5 -> pixels + 8 kg
    const int i16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    const int frag_off = (i16 >> 2) * 64 + (16 * g16 + 4 * (i16 & 3)) * 2;   // + row0 * 64

    const int urow = lane >> 2, uchunk = (lane & 3) * 8;     // LDS-DMA lane -> (row of the 16-row piece, 16-B chunk): plain row-major
    for (int it = ks; it < items; it += a.ksplit) {
        const int b = it / (tiles_x * tiles_y), pt = it - b * (tiles_x * tiles_y);
        const int y0 = (pt / tiles_x) * TH, x0 = (pt - (pt / tiles_x) * tiles_x) * TW;
        __syncthreads();                                     // the previous item's fragments are consumed
        // ---- stage dY: 2 blocks x 2 planes x 8 pieces = 32 pieces; X: CIB x 2 planes x XU pieces; dealt over the 4 waves
        {
            constexpr int NG = 32, NX = CIB * 2 * XU;
            for (int p = wave; p < NG + NX; p += 4) {
                if (p < NG) {
                    const int blk = p >> 4, plane = (p >> 3) & 1, piece = p & 7;
                    const int row = piece * 16 + urow;                       // pixel of the patch: y = row >> 4, x = row & 15
                    const int y = y0 + (row >> 4), x = x0 + (row & 15);
                    const int cb = cot * 2 + blk;
                    const bool ok = y < a.H && x < a.W && cb < a.CBo;
                    const _Float16* base = plane ? a.gl : a.gh;
                    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + ((long long)b * a.CBo) * a.P * 32), 0, a.CBo * a.P * 64, 0x00020000);
                    const unsigned off = ok ? (unsigned)((((cb * a.P) + y * a.W + x) * 32 + uchunk) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + (blk * 2 + plane) * G_PLANE + piece * 1024), 16, off, 0, 0, 0);
                } else {
                    const int q = p - NG;
                    const int cibl = q / (2 * XU), plane = (q / XU) & 1, piece = q % XU;
                    const int row = piece * 16 + urow;
                    const int hy = row / HWD, hx = row - hy * HWD;
                    const int y = y0 - a.pad_h + hy, x = x0 - a.pad_w + hx;
                    const int cb = cig * CIB + cibl;
                    const bool ok = row < HR && y >= 0 && y < a.H && x >= 0 && x < a.W && cb < a.CBi;
                    const _Float16* base = plane ? a.xl : a.xh;
                    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + ((long long)b * a.CBi) * a.P * 32), 0, a.CBi * a.P * 64, 0x00020000);
                    const unsigned off = ok ? (unsigned)((((cb * a.P) + y * a.W + x) * 32 + uchunk) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + O_X + (cibl * 2 + plane) * X_PLANE + piece * 1024), 16, off, 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- 8 steps of 16 pixels (one patch row each)
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
            const int prow = s * 16 + kg * 8;                // first of this lane group's 8 pixels: patch row s, column 8 kg
            half8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const char* t = lds + (m * 2) * G_PLANE + prow * 64 + frag_off;
                half4 h0 = tr_read(t), h1 = tr_read(t + 4 * 64), l0 = tr_read(t + G_PLANE), l1 = tr_read(t + G_PLANE + 4 * 64);
                TR_WAIT(h0, h1, l0, l1)
                ah[m] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                al[m] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
#pragma unroll
            for (int u = 0; u < UPW; ++u) {
                const int unit = wave + 4 * u;
                if (unit < UNITS) {
                    const int tap = unit / CIB, cibl = unit - tap * CIB;
                    const int r = tap / KW, q = tap - r * KW;
                    const int R0 = (s + r) * HWD + kg * 8 + q;     // halo row of the first pixel
                    const char* t = lds + O_X + (cibl * 2) * X_PLANE + R0 * 64 + frag_off;
                    half4 h0 = tr_read(t), h1 = tr_read(t + 4 * 64), l0 = tr_read(t + X_PLANE), l1 = tr_read(t + X_PLANE + 4 * 64);
                    TR_WAIT(h0, h1, l0, l1)
                    const half8 bh = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const half8 bl = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        hh[m][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, hh[m][u], 0, 0, 0);      // D[co][ci]
                        xx[m][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, xx[m][u], 0, 0, 0);
                        xx[m][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, xx[m][u], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- dW[tap][co][ci] += tile: lane = ci column (lane & 31), register r = co row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wave + 4 * u;
        if (unit >= UNITS) continue;
        const int tap = unit / CIB, cibl = unit - tap * CIB;
        const int ci = (cig * CIB + cibl) * 32 + (lane & 31);
        if (ci >= a.cin_pad) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cot * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < a.cout_pad) atomicAdd(a.dw + ((long long)tap * a.cout_pad + co) * a.cin_pad + ci, hh[m][u][r] + xx[m][u][r] * bflow::SPLIT_LO_INV);
            }
    }
}

template <int KH, int KW, int CIB>
void launch(const WgradArgs& a, int items, hipStream_t s) {
    constexpr int HR = (16 + KW - 1) * (8 + KH - 1), XU = (HR + 15) / 16;
    const int lds = 2 * 2 * 128 * 64 + CIB * 2 * XU * 1024;
    auto kern = wgrad_halo_kernel<KH, KW, CIB>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(bflow::ceil_div(a.cin_pad / 32, CIB), a.cout_pad / 64, a.ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    (void)items;
}

}  // namespace

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
    const int cib = shape == 1 ? 1 : shape == 4 ? 4 : 2;
    const int tiles = bflow::ceil_div(Cin_pad / 32, cib) * (a.cout_pad / 64);
    a.ksplit = std::max(1, std::min(items, (256 + tiles - 1) / tiles));
    hipStream_t s = (hipStream_t)stream;
    if (shape == 1) launch<3, 3, 1>(a, items, s);
    else if (shape == 2) launch<1, 5, 2>(a, items, s);
    else if (shape == 3) launch<5, 1, 2>(a, items, s);
    else launch<1, 1, 4>(a, items, s);
    return bflow::launch_status("conv_wgrad_halo");
}
