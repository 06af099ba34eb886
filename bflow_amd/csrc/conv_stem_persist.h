// Persistent form of the row-window stem kernel (round 6; included by conv_split.hip behind conv_stem_rows_kernel).  BasicEncoder.conv1 of
// the InstanceNorm feature encoder (extractor.py:63,110-113): 7x7 / stride 2 on a few-channel fp32 input, fp32 output + statistics.
//
// What the per-patch kernel loses its time to (tools/stem_ablate.sh on the row-window kernel, 5 x 480 x 640 -> 64 channels, one box:
// 120 us = epilogue 58 + patch loads 34 + MFMAs 31 + weight stream 17, the parts ADD UP -- profiles/r06_stem_persist.txt): every workgroup
// streams the whole 88-KB weight set through a two-slot ring with a full wait + barrier per k-block, loads its patch before its first MFMA
// and drains its stores behind its last one.  Here:
//   * ONE 8-wave workgroup per CU keeps the WHOLE packed weight set resident in LDS (CH = 5: 11 k-blocks x 8 KB), loaded once (NWV = 8,
//     WREG = 0: the product; the 4-wave forms with weight fragments in registers are tools variants, measured slower);
//   * a WAVE is the unit of work: it owns slabs of 2 x 16 output pixels x 64 channels, with a PRIVATE LDS image of the slab's 9 x 37 x CH
//     input window (split once into hi / lo planes, [row][column][channel] as in the row-window kernel: the same fragments, the same
//     MFMA sequence per accumulator -> bit-identical outputs).  Nothing is shared between waves but the read-only weights: the k-loop has
//     NO barrier and no counted wait at all;
//   * the input window of slab i + 1 is requested into registers before slab i's k-loop and converted after it; slab i's 32 stores per lane
//     are issued behind its k-loop and drain under slab i + 1's MFMAs (the wave never waits for a store);
//   * a workgroup walks a CONTIGUOUS range of slabs (vertical neighbours first: their windows overlap in 5 of 9 rows and are served by the
//     CU's vector cache), a wave keeps its InstanceNorm sums in fp64 registers and issues its atomics when the image changes or it is done;
//   * the vector ALU work per slab is what the matrix work is measured against (one wave's MFMAs cover the other wave's VALU phases, not its
//     own): interior slabs take per-lane offsets computed ONCE (window loads: 6 relative offsets + a scalar offset per channel; stores: one
//     lane offset + 16 scalar offsets), and the hi / lo split runs on packed conversions, two values per instruction, with the clamp as the
//     hardware's (FP16_OVFL) and the "hi = 0 below 2^-14" select explicit: 5 instead of 12 instructions per value.
#pragma once

#ifndef STEMP_ABL
#define STEMP_ABL 0     // tools/stem_ablate.sh persist: 1 no stores, 2 no window conversion writes, 4 no window loads, 8 no MFMA
#endif

// NWV = waves per workgroup (one workgroup per CU); WREG: weight fragments of ALL k-steps that live in registers (NWV = 4: one wave per SIMD
// with the whole 512-register file): 0 none, 1 the hi plane (42 fragments = 168 registers: the operand of two of a step's three MFMA groups;
// the k-loop then reads the window fragments and the lo plane from LDS), 2 both planes (336 registers: spills, measured slower).
template <int CH, int NWV, int WREG>
__global__ __launch_bounds__(64 * NWV, 1) void conv_stem_persist_kernel(ConvArgs a, StemArgs sa, int n_slabs, int per_wave, int tiles_x, int rows2) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = 7, STRIDE = 2, NT = 2;
    constexpr int SR = STRIDE * 1 + KS, PC = STRIDE * 15 + KS;     // input rows / columns of a 2 x 16 slab: 9 x 37
    // halves per window row: even (rows start on 4-byte boundaries) and = 32 mod 64, i.e. 16 banks: the 32 lanes of a fragment read are 16
    // pixels (2 CH halves = CH dwords apart: every bank at most once for odd CH) of TWO output rows, 2 window rows = `pitch` dwords apart -- the
    // second row's 16 banks are exactly the ones the first leaves free (with the per-patch kernel's pitch of 186: two 2-way conflicts per read)
    constexpr int pitch = (PC * CH + 1 - 32 + 63) / 64 * 64 + 32;
    static_assert(pitch >= PC * CH + 1 && pitch % 64 == 32, "window row pitch");
    constexpr int plane_h = SR * pitch + 64;                       // + slack read by the last pixels' window surplus (zero)
    constexpr int SLAB_B = (2 * plane_h * 2 + 15) & ~15;           // bytes of a wave's private (hi, lo) image
    constexpr int KSTEPS = (KS * CH + 15) >> 4;                    // 16-deep steps per filter row
    constexpr int nsteps = KS * KSTEPS;
    constexpr int nkb = (nsteps + 1) >> 1;                         // 32-wide weight k-blocks
    constexpr int W_TILE = 2 * NT * 2048;                          // 64 weight rows x 64 B x (hi, lo)
    constexpr int NPAIR = (PC + 1) / 2;                            // a lane converts PAIRS of window pixels: 2 CH halves = CH dwords per plane
    constexpr int NPX = (SR * NPAIR + 63) / 64;                    // pixel pairs per lane
    // (the LAST pair of an odd window row has one pixel: (CH + 1) / 2 dwords, the last half = the row's pad half)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    char* const wbase = lds;
    _Float16* const ph_ = reinterpret_cast<_Float16*>(lds + nkb * W_TILE + wave * SLAB_B);
    _Float16* const pl_ = ph_ + plane_h;

    // split1's clamp as the hardware's, for the life of the wave: FP16_OVFL = a finite value beyond +-65504 converts to +-65504 (+-inf inputs,
    // which split1 saturates, become (inf, NaN)).  "hi = 0 below 2^-14" stays an explicit select: flushing fp16 RESULTS by mode (corr_lookup_tile.hip
    // phase D) would also zero subnormal lo values, which the matrix cores do use -- measured: outputs then differ from the per-patch kernel's.
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // hwreg(HW_REG_MODE, 23, 1): FP16_OVFL

    // ---- the weight set, once: k-block kb by the wave group kb % (NWV / 4), pieces as in conv_stem_rows_kernel (waves 0-1 hi, 2-3 lo)
    {
        const int urow = lane >> 2;
        const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
        const int w4 = wave & 3;
        const int wtile_b = a.cout_pad * 64;
        const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)((w4 >> 1) ? a.wl : a.wh), 0, nkb * wtile_b, 0x00020000);
        for (int kb = wave >> 2; kb < nkb; kb += NWV / 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int unit = (w4 * 2 + j) & 3;
                const unsigned wvo = (unsigned)(((unit * 16 + urow) * 32 + uchunk) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(wbase + kb * W_TILE + ((w4 >> 1) ? NT * 2048 : 0) + unit * 1024), 16, wvo, kb * wtile_b, 0, 0);
            }
        }
    }
    // the wave's image: zero once (the slack, the rows' pad halves and everything a conversion does not cover)
    for (int i = lane; i < SLAB_B / 4; i += 64) reinterpret_cast<unsigned*>(ph_)[i] = 0u;

    // ---- this lane's constants
    const int yo = l31 >> 4, xo = l31 & 15;
    const int win0 = (STRIDE * yo) * pitch + STRIDE * xo * CH + kh * 8;       // halves; window start of filter row 0
    const int sw = (l31 >> 2) & 3;
    const int c4 = l31 * 4;
    float sc[NT], sh[NT];
    bool cok[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = n * 32 + l31;
        cok[n] = c < a.Cout;
        sc[n] = (a.scale && cok[n]) ? a.scale[c] : 1.f;
        sh[n] = (a.shift && cok[n]) ? a.shift[c] : 0.f;
    }
    // window pixel pairs of this lane: row / first column, LDS offset (halves) and -- for interior slabs -- the byte offset of its two pixels
    // relative to the window's first pixel (lanes past the last pair repeat it: their loads stay in range, their values are not written)
    int ppr[NPX], ppc[NPX], plo[NPX];
    unsigned prel[NPX][2];
    bool pwr[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int p = lane + 64 * j;
        pwr[j] = p < SR * NPAIR;
        const int q = pwr[j] ? p : SR * NPAIR - 1;
        ppr[j] = q / NPAIR;
        ppc[j] = (q - ppr[j] * NPAIR) * 2;
        plo[j] = ppr[j] * pitch + ppc[j] * CH;
        prel[j][0] = (unsigned)((ppr[j] * sa.W + ppc[j]) * 4);
        prel[j][1] = (unsigned)((ppr[j] * sa.W + (ppc[j] + 1 < PC ? ppc[j] + 1 : ppc[j])) * 4);      // (the pad pixel: any in-range value, it meets zero weights)
    }
    const long long img = (long long)sa.H * sa.W;
    const long long plane = (long long)a.P_out * 32;
    const int plane_img_b = (int)(img * 4);                        // (8 input planes < 2 GB: checked by the launcher)

    // slabs [s_lo, s_hi) of this workgroup; wave w takes s_lo + w, + 8, ...: (image, column tile, row pair), row pair fastest
    const int s_lo = blockIdx.x * per_wave * NWV;
    const int s_hi = min(n_slabs, s_lo + per_wave * NWV);
    const int per_img = tiles_x * rows2;

    float pv[NPX][2 * CH];
    auto slab_coords = [&](int s, int& b, int& y0, int& x0) {
        b = s / per_img;
        const int r = s - b * per_img;
        const int tx = r / rows2;
        y0 = (r - tx * rows2) * 2;
        x0 = tx * 16;
    };
    auto issue_loads = [&](int s) {
        int b, y0, x0;
        slab_coords(s, b, y0, x0);
        const int gy0 = y0 * STRIDE - a.pad_h, gx0 = x0 * STRIDE - a.pad_w;
        const long long x_base = sa.B_src > 0 ? ((long long)(b % sa.B_src) * sa.C_src + sa.win[b / sa.B_src]) * img : (long long)b * sa.Cin * img;
        const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(sa.x + x_base), 0, (int)(CH * img * 4), 0x00020000);
        if (gy0 >= 0 && gy0 + SR <= sa.H && gx0 >= 0 && gx0 + PC <= sa.W) {
            // interior window: lane offsets computed once, the slab and the channel plane in the scalar offset
            const int so = (gy0 * sa.W + gx0) * 4;
#pragma unroll
            for (int j = 0; j < NPX; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        pv[j][e * CH + c] = (STEMP_ABL & 4) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, prel[j][e], so + c * plane_img_b, 0));
        } else {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int gy = gy0 + ppr[j];
                const bool rok = pwr[j] && gy >= 0 && gy < sa.H;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int gx = gx0 + ppc[j] + e;
                    const bool ok = rok && ppc[j] + e < PC && gx >= 0 && gx < sa.W;  // (out of the image: an out-of-range offset = 0)
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const unsigned off = ok ? (unsigned)(((c * sa.H + gy) * sa.W + gx) * 4) : 0x80000000u;
                        pv[j][e * CH + c] = (STEMP_ABL & 4) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, off, 0, 0));
                    }
                }
            }
        }
    };

    double d1[NT], d2[NT];                                         // InstanceNorm sums of the current image (lanes 0-31: one channel each)
#pragma unroll
    for (int n = 0; n < NT; ++n) { d1[n] = 0.0; d2[n] = 0.0; }
    int stat_b = -1;
    auto flush_stats = [&]() {
        if (a.stats && stat_b >= 0 && lane < 32) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = n * 32 + l31;
                if (col < a.Cout) {
                    double* p = a.stats + (long long)(blockIdx.x % a.stats_reps) * a.stats_rep_stride + ((long long)stat_b * a.Cout + col) * 2;
                    atomicAdd(p, d1[n]);
                    atomicAdd(p + 1, d2[n]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) { d1[n] = 0.0; d2[n] = 0.0; }
    };
    // scalar part of the store offset of accumulator register r: pixel (r >> 3, (r & 3) + 8 ((r >> 2) & 1)) of the slab
    int sto[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sto[r] = ((r >> 3) * a.Wo + (r & 3) + 8 * ((r >> 2) & 1)) * 128;

    int s = s_lo + wave;
    if (s < s_hi) issue_loads(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the weight pieces; the first window rides along)
    __syncthreads();                                               // weights resident, images zeroed: the only barrier of the kernel
    half8 rwh[WREG >= 1 ? nsteps : 1][NT], rwl[WREG >= 2 ? nsteps : 1][NT];
    if constexpr (WREG >= 1) {
#pragma unroll
        for (int st = 0; st < nsteps; ++st) {
            const char* wt_ = wbase + (st >> 1) * W_TILE;
            const int co_ = ((((st & 1) * 2 + kh) ^ sw)) * 16;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int o_ = (n * 32 + l31) * 64 + co_;
                rwh[st][n] = *reinterpret_cast<const half8*>(wt_ + o_);
                if constexpr (WREG >= 2) rwl[st][n] = *reinterpret_cast<const half8*>(wt_ + NT * 2048 + o_);
            }
        }
    }

    for (; s < s_hi; s += NWV) {
        int b, y0, x0;
        slab_coords(s, b, y0, x0);
        if (b != stat_b) {
            flush_stats();
            stat_b = b;
        }
        // ---- the window -> hi / lo image: one conversion per input element, two elements per instruction
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            if (pwr[j]) {
                const int nd = ppc[j] + 1 < PC ? CH : (CH + 1) / 2;   // dwords of this pair (the next row starts right behind the pad half)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const f32x2_ x = {pv[j][2 * c], pv[j][2 * c + 1]};
                    const f32x2_ xf = {fabsf(x[0]) >= 6.103515625e-05f ? x[0] : 0.f, fabsf(x[1]) >= 6.103515625e-05f ? x[1] : 0.f};
                    const f16x2_ hi = __builtin_convertvector(xf, f16x2_);
                    const f32x2_ rem = (x - __builtin_convertvector(hi, f32x2_)) * bflow::SPLIT_LO_SCALE;      // (exact: hi is x rounded to 11 bits)
                    const f16x2_ lo = __builtin_convertvector(rem, f16x2_);
                    if (c < nd && !(STEMP_ABL & 2)) {
                        *reinterpret_cast<f16x2_*>(ph_ + plo[j] + 2 * c) = hi;    // a pair starts on a 4-byte boundary; lanes CH dwords apart
                        *reinterpret_cast<f16x2_*>(pl_ + plo[j] + 2 * c) = lo;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + NWV < s_hi) issue_loads(s + NWV);                  // lands under this slab's k-loop
        __builtin_amdgcn_sched_barrier(0);

        f32x16 hh[NT], xx[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hh[n][r] = 0.f;
                xx[n][r] = 0.f;
            }
        // ---- k-loop: fragments of step st + 1 are read while the MFMAs of step st run
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        half8 cxh, cxl, cwh[NT], cwl[NT], nxh, nxl, nwh[NT], nwl[NT];
#define STEMP_READ(XH, XL, WH, WL, ST)                                                                                  \
        {                                                                                                                \
            constexpr int r_ = (ST) / KSTEPS, ks_ = (ST) - r_ * KSTEPS;                                                  \
            const int wo_ = win0 + r_ * pitch + ks_ * 16;                                                                \
            const unsigned* qh_ = reinterpret_cast<const unsigned*>(ph_ + wo_);                                          \
            const unsigned* ql_ = reinterpret_cast<const unsigned*>(pl_ + wo_);                                          \
            const u32x4 vh_ = {qh_[0], qh_[1], qh_[2], qh_[3]}, vl_ = {ql_[0], ql_[1], ql_[2], ql_[3]};                  \
            XH = __builtin_bit_cast(half8, vh_);                                                                         \
            XL = __builtin_bit_cast(half8, vl_);                                                                         \
            if constexpr (WREG < 2) {                                                                                    \
                const char* wt_ = wbase + ((ST) >> 1) * W_TILE;                                                          \
                const int co_ = (((((ST) & 1) * 2 + kh) ^ sw)) * 16;                                                     \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                         \
                    const int o_ = (n * 32 + l31) * 64 + co_;                                                            \
                    if constexpr (WREG < 1) WH[n] = *reinterpret_cast<const half8*>(wt_ + o_);                           \
                    WL[n] = *reinterpret_cast<const half8*>(wt_ + NT * 2048 + o_);                                       \
                }                                                                                                        \
            }                                                                                                            \
        }
        STEMP_READ(cxh, cxl, cwh, cwl, 0)
        static_for<0, nsteps>([&](auto stc) __attribute__((always_inline)) {
            constexpr int st = decltype(stc)::value;
            if constexpr (st + 1 < nsteps) STEMP_READ(nxh, nxl, nwh, nwl, st + 1)
            // (per accumulator the per-patch kernel's sequence -- hh: xh wh; xx: xh wl, then xl wh --, issued so that two MFMAs lie between
            //  the two updates of an xx accumulator: a lone wave on a SIMD has nobody to cover a dependent pair)
            if constexpr ((STEMP_ABL & 8) != 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (cxh[0] == (_Float16)123.f && (WREG >= 1 ? rwh[st][n][0] : cwh[n][0]) == (_Float16)77.f && (WREG >= 2 ? rwl[st][n][1] : cwl[n][1]) == cxl[1]) hh[n][0] += 1.f;
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cxh, WREG >= 1 ? rwh[st][n] : cwh[n], hh[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cxh, WREG >= 2 ? rwl[st][n] : cwl[n], xx[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cxl, WREG >= 1 ? rwh[st][n] : cwh[n], xx[n], 0, 0, 0);
            }
            if constexpr (st + 1 < nsteps) {
                cxh = nxh;
                cxl = nxl;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (WREG < 1) cwh[n] = nwh[n];
                    if constexpr (WREG < 2) cwl[n] = nwl[n];
                }
            }
        });
#undef STEMP_READ

        // ---- epilogue (conv_epilogue_direct's arithmetic): register r of a lane = pixel (y0 + (r >> 3), x0 + (r & 3) + 8 ((r >> 2) & 1) + 4 kh)
        //      of channel lane & 31: two complete 128-B rows of the blocked output per wave store; the stores are not waited for
        const bool inner = y0 + 1 < a.Ho && x0 + 15 < a.Wo;
        const unsigned lane_off = (unsigned)(((y0 * a.Wo + x0 + 4 * kh) * 128) + c4);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (n * 32 >= a.Cout) break;
            float* ob = a.out_f32 + ((long long)b * a.CBo + a.cb_off + n) * plane;
            const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)(plane * 4), 0x00020000);
            float s1 = 0.f, s2 = 0.f;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = (hh[n][r] + xx[n][r] * LO_INV) * sc[n] + sh[n];
                if (a.act == 1) v[r] = fmaxf(v[r], 0.f);
                else if (a.act == 2) v[r] = tanhf(v[r]);
                if (!cok[n]) v[r] = 0.f;
            }
            if (inner) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!(STEMP_ABL & 1) || v[r] == 1.2345f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ro, lane_off, sto[r], CONV_NT_STORES_ENC ? 2 : 0);
                    s1 += v[r];
                    s2 += v[r] * v[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int y = y0 + (r >> 3), x = x0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh;
                    const bool in = y < a.Ho && x < a.Wo;
                    const unsigned off = in ? (unsigned)((y * a.Wo + x) * 128 + c4) : 0x80000000u;
                    if (!(STEMP_ABL & 1) || v[r] == 1.2345f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ro, off, 0, CONV_NT_STORES_ENC ? 2 : 0);
                    if (in) { s1 += v[r]; s2 += v[r] * v[r]; }
                }
            }
            if (a.stats) {                                         // the other 16 pixels of this channel sit in the other half of the wave
                float p = s1, q = s1;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
                s1 = p + q;
                p = s2; q = s2;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
                s2 = p + q;
                d1[n] += (double)s1;
                d2[n] += (double)s2;
            }
        }
    }
    flush_stats();
#endif
}
