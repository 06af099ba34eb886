// Shared pieces of the split-fp16 MFMA conv engine (conv_split.hip, conv_stream.h): vector types, the argument block of a convolution
// and the LDS-transposing epilogue.  Internal to the library; every including translation unit gets its own (anonymous-namespace) copy.
#pragma once
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr float LO_INV = 1.0f / 2048.0f;
constexpr int CBM = 128, CT = 256;

using bflow::split1;   // common.h: saturating hi/lo split

struct ConvArgs {
    const _Float16 *xh, *xl;   // (B, CB1, P_in, 32): channel blocks [0, CB1)
    const _Float16 *x2h, *x2l; // (B, CB - CB1, P_in, 32): channel blocks [CB1, CB)  (virtual concatenation) or null
    const _Float16 *wh, *wl;   // (ntaps*CB, Cout_pad, 32)
    int H, W, CB, CB1, P_in, Ho, Wo, Cout, cout_pad;
    const float* addend;       // blocked fp32 (B, ceil(Cout/32), P_out, 32) added before the activation, or null
    int KH, KW, stride, pad_h, pad_w, n_tiles;
    float* out_f32;            // blocked (B, CBo, P_out, 32) fp32 or null
    _Float16 *oh, *ol;         // blocked split or null
    int CBo, cb_off, P_out;    // channel blocks / first channel block / rows per image of the output buffers
    const float *scale, *shift;   // per output channel or null
    int act;
    double* stats;             // (R, B, Cout, 2) or null: workgroup w adds into replica w % R (spreads the fp64 atomics)
    int stats_reps;            // R >= 1
    long long stats_rep_stride;
    int gate;                  // 0 | 1 (zr) | 2 (blend): fused GRU gates, see bflow_conv_desc_t
    const _Float16 *gh, *gl;   // h planes (B, CBo, P_out, 32)
    const float* gz;           // z (B, CBo, P_out, 32)
    float* acc;                // fp32 (B, Cout, Ho*Wo) accumulated in place, or null
    int w_sets;                // > 1: image b multiplies weight set b % w_sets (generic kernel only; the weight-gradient GEMMs)
    int keep_pad;              // split output: channels >= Cout of the last block are left untouched instead of zeroed (Cout % 4 == 0)
    const float* xraw;         // NIN halo kernel: pre-normalisation fp32 input (B, CB, P_in, 32) + its InstanceNorm statistics
    const double* xstats;      // (R, B, CB*32, 2)
    int xstats_reps;
    float xeps;
    unsigned long long* stamps;   // H8_STAMPS builds only (tools/conv_stamps.sh)
};

// InstanceNorm / affine coefficients of one channel: y = x * mul + add (shared by the normalisation kernel and the NIN halo kernel).
// `stats` holds per-(image, channel) (sum, sum of squares) in fp64 -- or, with eps < 0, the coefficients (mul, add) themselves.
__device__ __forceinline__ void norm_coeffs(const double* stats, const float* scale, const float* shift, int b, int c, int C, int HW,
                                            float eps, float& mul, float& add, int reps, long long rep_stride) {
    if (c >= C) { mul = 0.f; add = 0.f; return; }   // padded channels of the last block stay zero
    if (stats) {   // F.instance_norm: biased variance, eps inside the sqrt
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < reps; ++r) {
            s1 += stats[r * rep_stride + ((long long)b * C + c) * 2];
            s2 += stats[r * rep_stride + ((long long)b * C + c) * 2 + 1];
        }
        if (eps < 0.f) {   // DIRECT table (eps = -1): the entries ARE (mul, add) -- GroupNorm's per-(image, channel) affine map, any sign of gamma
            mul = (float)s1;
            add = (float)s2;
            return;
        }
        const double mean = s1 / HW;
        double var = s2 / HW - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        mul = rstd;
        add = (float)(-mean) * rstd;
    } else {
        mul = scale ? scale[c] : 1.f;
        add = shift ? shift[c] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue shared by both kernels.  The MFMAs are issued with the WEIGHT fragment as the A operand and the activation
// fragment as B, so an accumulator tile is D[channel][pixel]: lane = pixel, and registers 4j..4j+3 of a lane are 4 CONSECUTIVE
// channels (8j + 4*(lane>>5) + 0..3) of that pixel.
// A lane-per-pixel store (every lane a different 64/128-B row) is issue-bound in the memory pipeline (~600 cycles per wave
// instruction, measured: the epilogue was 5 of the 10 us of an empty-loop launch), so each wave first transposes its 32x32
// tile through a private LDS slab (float4 writes, row stride 36 floats = conflict-free) and then walks it in MEMORY order:
// lane -> (row lane/8 + 8*it, channels 4*(lane&7)..+3).  Everything after the accumulation (scale/shift, addend, activation,
// hi/lo split, statistics) happens on that side, so the addend loads and all stores are fully coalesced (1 KB fp32 / 512 B
// fp16 contiguous per wave instruction).
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define CONV_STG_STRIDE 36                       // floats per staged pixel row (32 + 4 pad)
// Output stores of the shared epilogue.  CONV_NT_STORES = 1 (tools A/B, tools/build_flag_variant.sh): non-temporal -- the consumer is the NEXT
// kernel, which starts with a cold L2 anyway, so nothing is lost by writing around it, and the lines leave during the epilogue instead of
// in the write-back at the end of the kernel.
#ifndef CONV_NT_STORES
#define CONV_NT_STORES 0
#endif
template <typename V>
__device__ __forceinline__ void epi_store(V* p, V v) {
    if (CONV_NT_STORES) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// `pixel_of(row)` maps row 0..31 of the wave's slab to the pixel row of the output image (or -1: outside, not written).
// Sum over the 8 lanes that share lane & 7 (xor 8, 16, 32) on the vector ALU: DPP row rotation inside a 16-lane row, then the
// gfx950 row / half swaps -- 6 VALU instructions instead of 3 ds_bpermute round trips through the LDS crossbar (the statistics
// reduction was 3.3 k of the epilogue's 17 k cycles per wave on the InstanceNorm convolutions).
__device__ __forceinline__ float sum_lanes_mod8(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    // v_permlane16_swap x, y: odd rows of x <-> even rows of y; with x = y = v: x = {r0, r0, r2, r2}, y = {r1, r1, r3, r3}.
    // (inline asm: the builtin hands back only the first of the two results)
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    v = x + y;
    x = v; y = v;                                   // v_permlane32_swap: upper half of x <-> lower half of y
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}

// GROUPS = 2 (8- / 10-wave split-k kernels) or 4 (12-wave kernel: 2 k-halves x 2 channel-block parities): each k-group stages ITS partial
// sums in its own slab set; after ONE workgroup barrier every group walks the GROUPS slabs of its pixel slab index `wave` -- group g takes
// the 4 / GROUPS row groups from g * 4 / GROUPS on and adds the partials -- so the k-split is combined for free on the read side and the
// sigmoid / tanh / split work is spread over all waves.
template <int NT, int NW = 4, int GROUPS = 1, typename PixelOf>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&hh)[NT], f32x16 (&xx)[NT], int b, PixelOf pixel_of, int n0,
                                              int lane, int wave, int tid, bool writer, float* red, int grp = 0) {
    constexpr int BN = 32 * NT;
    constexpr int RS = CONV_STG_STRIDE;
    static_assert(GROUPS == 1 || GROUPS == 2 || GROUPS == 4, "row groups must divide over the k-groups");
    constexpr int NIT = 4 / GROUPS;                       // row groups of 8 pixels per wave on the read side
    const int it0 = grp * NIT;
    constexpr int SLABS = NW / GROUPS;                    // pixel slabs (waves per k-group): 4, or 5 in the 10-wave kernel
    const int wave_all = wave + SLABS * grp;              // statistics scratch is per wave of the whole workgroup
    float* const stg0 = red + 2 * NW * BN + wave * NT * (32 * RS);   // the NT slabs of pixel slab `wave` (group 0's partial sums),
    constexpr int GSTRIDE = SLABS * NT * (32 * RS);                  // group 1's set lies GSTRIDE floats further; behind the statistics scratch
    const int kh = lane >> 5, l31 = lane & 31;
    const int rr = lane >> 3, ch = (lane & 7) * 4;        // read side: rows rr + 8*it, channels ch .. ch+3 of the block
    if (writer || GROUPS > 1) {
        // all channel blocks are staged first (one slab each): ONE write -> read round trip through LDS per wave
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (n0 + n * 32 >= a.Cout) break;
            float* stg = stg0 + grp * GSTRIDE + n * (32 * RS);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 4 * j;
                *reinterpret_cast<float4*>(stg + l31 * RS + 8 * j + 4 * kh) =
                    make_float4(hh[n][r0] + xx[n][r0] * LO_INV, hh[n][r0 + 1] + xx[n][r0 + 1] * LO_INV, hh[n][r0 + 2] + xx[n][r0 + 2] * LO_INV,
                                hh[n][r0 + 3] + xx[n][r0 + 3] * LO_INV);
            }
        }
    }
    if (GROUPS == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
    if (writer || GROUPS > 1) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int cbase = n0 + n * 32;
            if (cbase >= a.Cout) break;                   // channel blocks past the padded output are never written
            const float* stg = stg0 + n * (32 * RS);
            float sc[4], sh[4];
            bool cok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cc = cbase + ch + k;
                cok[k] = cc < a.Cout;
                sc[k] = (a.scale && cok[k]) ? a.scale[cc] : 1.f;
                sh[k] = (a.shift && cok[k]) ? a.shift[cc] : 0.f;
            }
            const long long ob = (((long long)b * a.CBo + a.cb_off + (n0 >> 5) + n) * a.P_out) * 32 + ch;
            const long long ab = (((long long)b * ((a.Cout + 31) >> 5) + (n0 >> 5) + n) * a.P_out) * 32 + ch;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            float4 raws[NIT];                                // all slab reads (and addend loads) in flight before the first is consumed
            float4 ads[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) raws[it] = *reinterpret_cast<const float4*>(stg + ((it0 + it) * 8 + rr) * RS + ch);
            if (GROUPS > 1) {                                // + the other k-groups' partial sums (group order: the same sum in every build)
#pragma unroll
                for (int g = 1; g < GROUPS; ++g)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const float4 o4 = *reinterpret_cast<const float4*>(stg + g * GSTRIDE + ((it0 + it) * 8 + rr) * RS + ch);
                        raws[it].x += o4.x; raws[it].y += o4.y; raws[it].z += o4.z; raws[it].w += o4.w;
                    }
            }
            if (a.addend) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m_ = pixel_of((it0 + it) * 8 + rr);
                    ads[it] = *reinterpret_cast<const float4*>(a.addend + ab + (long long)(m_ >= 0 ? m_ : 0) * 32);
                }
            }
            // gate operands (h as hi/lo halves, z) of the four row groups: same idea, one latency instead of four
            const int cbk = (n0 >> 5) + n;
            const bool r_half = a.gate == 1 && cbk >= a.CBo;
            const bool need_h = a.gate == 2 || a.gate == 3 || r_half;
            const long long gbase = (((long long)b * a.CBo + (r_half ? cbk - a.CBo : cbk)) * a.P_out) * 32 + ch;
            half4v g_hh[NIT], g_hl[NIT];
            float4 g_z[NIT];
            if (need_h) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m_ = pixel_of((it0 + it) * 8 + rr);
                    const long long go = gbase + (long long)(m_ >= 0 ? m_ : 0) * 32;
                    g_hh[it] = *reinterpret_cast<const half4v*>(a.gh + go);
                    g_hl[it] = *reinterpret_cast<const half4v*>(a.gl + go);
                    if (a.gate == 2) g_z[it] = *reinterpret_cast<const float4*>(a.gz + go);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = (it0 + it) * 8 + rr;
                const int m = pixel_of(row);
                const bool mok = m >= 0;
                const float4 raw = raws[it];
                float v[4] = {raw.x * sc[0] + sh[0], raw.y * sc[1] + sh[1], raw.z * sc[2] + sh[2], raw.w * sc[3] + sh[3]};
                if (a.addend && mok) {
                    const float4 ad = ads[it];
                    v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (a.act == 1) v[k] = fmaxf(v[k], 0.f);
                    else if (a.act == 2) v[k] = tanhf(v[k]);
                    if (!cok[k]) v[k] = 0.f;                // padded channels of the last block are written as zeros
                    if (mok) { s1[k] += v[k]; s2[k] += v[k] * v[k]; }
                }
                if (mok && a.acc) {                             // acc[b, c, m] += v; continue with the updated value
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cok[k]) {
                            float* pa = a.acc + ((long long)b * a.Cout + cbase + ch + k) * (a.Ho * a.Wo) + m;
                            v[k] += *pa;
                            *pa = v[k];
                        }
                }
                if (mok && a.gate) {
                    // SepConvGRU gates (update.py:38-47).  Block index inside the (B, CBo, P, 32) gate buffers: the z half of a
                    // "zr" convolution and a "blend" convolution map 1:1, the r half is shifted down by CBo blocks.
                    const long long gb = gbase + (long long)m * 32;
                    if (a.gate == 1 && !r_half) {
                        epi_store(reinterpret_cast<f32x4v*>(a.out_f32 + gb), f32x4v{bflow::gate_sigmoid(v[0]), bflow::gate_sigmoid(v[1]),
                                                                                     bflow::gate_sigmoid(v[2]), bflow::gate_sigmoid(v[3])});
                    } else {
                        const half4v hh4 = g_hh[it], hl4 = g_hl[it];
                        float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (a.gate == 2) z4 = g_z[it];
                        const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
                        half4v h4, l4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float h = (float)hh4[k] + (float)hl4[k] * LO_INV;
                            const float o = (a.gate == 1) ? bflow::gate_sigmoid(v[k]) * h
                                          : (a.gate == 2) ? (1.f - zz[k]) * h + zz[k] * bflow::gate_tanh(v[k])
                                                          : fmaxf(v[k] + h, 0.f);          // gate 3: the residual block's relu(x + y), extractor.py:55
                            _Float16 x1, x2;
                            split1(o, x1, x2);
                            h4[k] = x1;
                            l4[k] = x2;
                        }
                        epi_store(reinterpret_cast<half4v*>(a.oh + gb), h4);
                        epi_store(reinterpret_cast<half4v*>(a.ol + gb), l4);
                    }
                } else if (mok) {
                    if (a.out_f32) epi_store(reinterpret_cast<f32x4v*>(a.out_f32 + ob + (long long)m * 32), f32x4v{v[0], v[1], v[2], v[3]});
                    if (a.oh && !(a.keep_pad && !cok[0])) {     // keep_pad: a group of 4 pad channels belongs to somebody else
                        half4v h4, l4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            _Float16 x1, x2;
                            split1(v[k], x1, x2);
                            h4[k] = x1;
                            l4[k] = x2;
                        }
                        epi_store(reinterpret_cast<half4v*>(a.oh + ob + (long long)m * 32), h4);
                        epi_store(reinterpret_cast<half4v*>(a.ol + ob + (long long)m * 32), l4);
                    }
                }
            }
            if (a.stats) {                                  // per-channel sums over this wave's 32 pixels: lanes with equal lane&7
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[k] = sum_lanes_mod8(s1[k]);
                    s2[k] = sum_lanes_mod8(s2[k]);
                    if (lane < 8) {
                        red[(0 * NW + wave_all) * BN + n * 32 + ch + k] = s1[k];
                        red[(1 * NW + wave_all) * BN + n * 32 + ch + k] = s2[k];
                    }
                }
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid - which * BN;
            const int col = n0 + c;
            if (col < a.Cout) {
                const float* p = red + which * NW * BN + c;
                double sum = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += (double)p[w * BN];
                atomicAdd(a.stats + (long long)(blockIdx.x % a.stats_reps) * a.stats_rep_stride + ((long long)b * a.Cout + col) * 2 + which, sum);
            }
        }
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

}  // namespace
