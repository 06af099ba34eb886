// Persistent ("streaming") form of the 3 x 3 halo convolution for outputs that are pre-normalisation fp32 (+ InstanceNorm statistics):
// every 3 x 3 of the feature encoder (extractor.py:47-55,103-125) on grids that fill the chip more than once.  Included by
// conv_split.hip inside its anonymous namespace (uses ConvArgs, slab_row / slab_col, the LDS image of conv_halo_kernel<2, 3, 3, TR>).
//
// Why: conv_halo_kernel's workgroups live for CB x 9 k-steps only; a third of a workgroup's life is its prologue (first halo + weight
// tiles straight from HBM, every workgroup of a round at once) and its epilogue (32 store instructions per lane + the statistics
// reduction), during which its SIMDs have no matrix work of their own: the layer-1 launch ran at 0.33-0.36 of the split format's matrix
// peak for three rounds ("the lever that is left is a persistent kernel that overlaps tile i's store drain with
// tile i+1's first halo under counted vmcnt over loads AND stores -- the structure K5 has").  This is that kernel:
//   * <= 2 x 256 persistent workgroups; a workgroup owns a CONTIGUOUS range of the item list (image, channel tile, 8 x 16 patch) and walks
//     it as ONE k-loop: the halo double buffer and the 4-slot weight ring run on across item boundaries, so item i+1's first halo and
//     weight tiles land under item i's last taps (the prologue exists once per workgroup);
//   * the accumulators are transposed (D[pixel][channel]: one register of a wave = two complete 128-B rows of the blocked fp32 output);
//     at an item boundary they are folded into 32 "drain" registers (hi + lo 2^-11) and the first MFMAs of the next item start from a
//     zero C operand; the drain registers are scaled, added into the lane's statistics and STORED between the MFMAs of item i+1's
//     k-steps, 1-2 per step -- no store phase, no LDS, no barrier of its own;
//   * vmcnt counts LDS-DMA pieces and stores alike and retires them in order (gfx9): every step issues a compile-time number of both
//     (out-of-range stores of edge patches / of the empty drain of the first item are dropped by the buffer bounds check, not skipped),
//     so "the weight tile of the next step has landed" stays a fixed s_waitcnt immediate.  The last two steps of an item issue no
//     stores, which makes the prologue's counts equal to the steady state's;
//   * a range never leaves its (image, channel tile) group without a flush, so the InstanceNorm sums stay in two registers per lane
//     and channel block for the whole range: ONE reduction + 128 fp64 atomics per workgroup instead of one per item.
// Template: CBT = input channel blocks (2, 3, 4: 64 / 96 / 128 channels); 64-channel output tiles (NT = 2).
template <int CBT>
__global__ __launch_bounds__(CT, 2) void conv_halo_stream_kernel(ConvArgs a, int per, int n_patches, int tiles_x, int total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 2, KH = 3, KW = 3, NW = 4;
    constexpr int TH = 2 * NW, TW = 16;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);
    constexpr int A_UNITS = ((HR + 15) / 16 + NW / 2 - 1) / (NW / 2) * (NW / 2);
    constexpr int AP = A_UNITS / (NW / 2);
    constexpr int A_PLANE = A_UNITS * 1024, A_BUF = 2 * A_PLANE;
    constexpr int NTAPS = KH * KW;
    constexpr int B_PLANE = NT * 2048, B_SLOT = 2 * B_PLANE;
    constexpr int NBP = 4 * NT / NW;
    constexpr int SB = 4, LA = 2;
    constexpr int O_B = 2 * A_BUF;
    constexpr int NSTEPS = CBT * NTAPS;
    constexpr int NDR = NT * 16;                       // drain registers = stores per item and lane
    constexpr int DSTEPS = NSTEPS - 2;                 // steps that carry stores (the last two of an item carry none)
    static_assert(NDR <= 2 * DSTEPS, "at most two stores per step");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;

    // ---- per-lane constants of the LDS-DMA pieces (independent of the item)
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const int plane_b = a.P_in * 64;
    const bool lo_a = wave & 1, lo_w = wave / (NW / 2);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(lo_w ? a.wl : a.wh), 0, NTAPS * CBT * wtile_b, 0x00020000);
    char* const a_dst = lds + (lo_a ? A_PLANE : 0) + (wave >> 1) * 1024;
    char* const w_dst = lds + O_B + (lo_w ? B_PLANE : 0);
    const int R0 = (wave * 2 + slab_row(l31)) * HWD + slab_col(l31);
    const int wsw = (l31 >> 2) & 3;
    const long long oplane = (long long)a.P_out * 32;  // floats of one channel block of one output image
    const float act_floor = a.act == 1 ? 0.f : -__builtin_inff();      // (tanh epilogues stay on conv_halo_kernel: see the dispatch)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    int q = blockIdx.x * per;
    const int q_end = q + per < total ? q + per : total;
    while (q < q_end) {
        // ---- one segment: items q .. seg_end - 1, all in one (image, channel tile) group
        const int group = q / n_patches;
        int mt = q - group * n_patches;
        const int seg_end = (group + 1) * n_patches < q_end ? (group + 1) * n_patches : q_end;
        int left = seg_end - q;                         // items still to start
        q = seg_end;
        const int b = group / a.n_tiles, n0 = (group - b * a.n_tiles) * (32 * NT);

        const rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_a ? a.xl : a.xh) + (long long)b * CBT * a.P_in * 32), 0, CBT * plane_b, 0x00020000);
        unsigned wvo[NBP];
#pragma unroll
        for (int j = 0; j < NBP; ++j) {
            const int r = n0 + ((wave * NBP + j) % (2 * NT)) * 16 + urow;
            wvo[j] = (unsigned)((r * 32 + uchunk) * 2);
        }
        rsrc_t r_o[NT];
        float sc[NT], sh[NT];
        bool cok[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int c = n0 + n * 32 + l31;
            cok[n] = c < a.Cout;
            sc[n] = (a.scale && cok[n]) ? a.scale[c] : 1.f;
            sh[n] = (a.shift && cok[n]) ? a.shift[c] : 0.f;
            // channel blocks past the padded output do not exist: a zero-sized resource drops their stores
            const bool blk = n0 + n * 32 < a.Cout;
            r_o[n] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_f32 + ((long long)b * a.CBo + a.cb_off + (n0 >> 5) + (blk ? n : 0)) * oplane), 0,
                                                       blk ? (int)(oplane * 4) : 0, 0x00020000);
        }
        float s1[NT] = {0.f, 0.f}, s2[NT] = {0.f, 0.f};

        // LDS-DMA source offsets of the halo of patch m (all out of range when there is no such item: the pieces are issued all the same,
        // the counts stay); formed where the pieces are issued -- twelve registers that would otherwise live through the whole k-loop
        auto halo_offset = [&](int i, int y0, int x0, bool exists) -> unsigned {
            const int row = ((wave >> 1) + (NW / 2) * i) * 16 + urow;
            const int hy = row / HWD, hx = row - hy * HWD;
            const int py = y0 - 1 + hy, px = x0 - 1 + hx;
            const bool ok = exists && row < HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
            return ok ? (unsigned)(((py * a.W + px) * 32 + uchunk) * 2) : 0x80000000u;
        };
        // drain side: byte offsets (inside one channel block of the output image) of the two pixel rows of this lane's slab half, for
        // the item being drained; 0x80000000 = nothing to store
        unsigned dbase[2] = {0x80000000u, 0x80000000u};
        int dxlim = 0;                                  // columns of the drained patch inside the image (<= 16)
        float dr[NT][16];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) dr[n][r] = 0.f;
        auto drain_target = [&](int m) {
            const int ty = m / tiles_x;
            const int y0 = ty * TH, x0 = (m - ty * tiles_x) * TW;
            const int yw = y0 + wave * 2;
            // register r of lane (channel, kh): column x0 + r, slab row (popcount(r >> 2) + kh) & 1: dbase[p] = row p ^ kh ... indexed by popcount parity
            const int ya = yw + kh, yb = yw + (kh ^ 1);
            dbase[0] = ya < a.H ? (unsigned)((ya * a.W + x0) * 128 + l31 * 4) : 0x80000000u;
            dbase[1] = yb < a.H ? (unsigned)((yb * a.W + x0) * 128 + l31 * 4) : 0x80000000u;
            dxlim = a.W - x0;
        };
        // the q-th store of the drained item: channel block q / 16, register q % 16
        auto drain_store = [&](auto qc) __attribute__((always_inline)) {
            constexpr int qq = decltype(qc)::value;
            constexpr int n = qq / 16, r = qq % 16;
            constexpr int par = __builtin_popcount((unsigned)(r >> 2)) & 1;
            float v = fmaxf(fmaf(dr[n][r], sc[n], sh[n]), act_floor);      // act 0 / 1 (ReLU): a clamp from below, no branch between the MFMAs
            if (!cok[n]) v = 0.f;
            const unsigned off = r < dxlim ? dbase[par] : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_o[n], off, r * 128, CONV_NT_STORES_ENC ? 2 : 0);
            if (off != 0x80000000u) { s1[n] += v; s2[n] = fmaf(v, v, s2[n]); }
        };

#define STREAM_ISSUE_A(M, EXISTS, CBI, BUF)                                                                              \
    {                                                                                                                    \
        const int ty_ = (M) / tiles_x;                                                                                   \
        const int y0_ = ty_ * TH, x0_ = ((M) - ty_ * tiles_x) * TW;                                                      \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (lptr_t)(a_dst + (BUF) * A_BUF + i * (NW / 2) * 1024), 16,     \
                                                     halo_offset(i, y0_, x0_, (EXISTS)), (CBI) * PLANE_STRIDE, 0, 0);         \
    }
#define STREAM_ISSUE_B(CBI, TAP, SLOT)                                                                                   \
    {                                                                                                                    \
        const int so_ = ((TAP) * CBT + (CBI)) * WTILE_STRIDE;                                                            \
        _Pragma("unroll") for (int j = 0; j < NBP; ++j)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(w_dst + (SLOT) * B_SLOT + ((wave * NBP + j) % (2 * NT)) * 1024), 16, \
                                                     wvo[j], so_, 0, 0);                                                 \
    }
    // fragments of ONE 16-deep half (KS) of a tap: 2 activation + 4 weight ds_read_b128.  The two halves of a tap have a register set each
    // (24 registers) and are read half a step ahead of their MFMAs: set 0 of tap s + 1 behind the first half of tap s's matrix work, set 1 of
    // tap s at the top of step s -- a whole-step look-ahead (conv_halo_kernel) needs 96 fragment registers, which the drain registers take here
#define STREAM_READ(XH, XL, WH, WL, ABUF, WSLOT, TAP, KS)                                                                \
    {                                                                                                                    \
        const int R_ = R0 + ((TAP) / KW) * HWD + ((TAP) % KW);                                                           \
        const int sw_ = (R_ >> 2) & 3;                                                                                   \
        const int ao_ = R_ * 64 + ((((KS) * 2 + kh) ^ sw_) * 16);                                                        \
        XH = *reinterpret_cast<const half8*>((ABUF) + ao_);                                                              \
        XL = *reinterpret_cast<const half8*>((ABUF) + A_PLANE + ao_);                                                    \
        const int co_ = (((KS) * 2 + kh) ^ wsw) * 16;                                                                    \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                                 \
            const int wo_ = (n * 32 + l31) * 64 + co_;                                                                   \
            WH[n] = *reinterpret_cast<const half8*>((WSLOT) + wo_);                                                      \
            WL[n] = *reinterpret_cast<const half8*>((WSLOT) + B_PLANE + wo_);                                            \
        }                                                                                                                \
    }

        // ---- prologue of the segment: first halo, first LA + 1 weight tiles
#define PLANE_STRIDE plane_b
#define WTILE_STRIDE wtile_b
        STREAM_ISSUE_A(mt, true, 0, 0)
#pragma unroll
        for (int t = 0; t <= LA; ++t) STREAM_ISSUE_B(0, t, t)
        f32x16 hh[NT], xx[NT];
        half8 xh0, xl0, wh0[NT], wl0[NT], xh1, xl1, wh1[NT], wl1[NT];   // fragment set 0 / 1 = k-half 0 / 1 of a tap
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA * NBP) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STREAM_READ(xh0, xl0, wh0, wl0, lds, lds + O_B, 0, 0)
        int islot = LA + 1, rslot = 0;                  // rslot: ring slot of the tap being computed
        int hbuf = 0;                                   // halo buffer of the phase (item, cb) being computed

#undef PLANE_STRIDE
#undef WTILE_STRIDE
#define PLANE_STRIDE plane_v
#define WTILE_STRIDE wtile_v
        while (left > 0) {
            --left;
            // (opaque copies: hipcc otherwise hoists every (tap, block) * stride product and every LDS destination out of this loop --
            //  ~100 loop-invariant SGPRs, spilled into VGPR lanes and, at 3-4 channel blocks, into scratch, whose loads would break the
            //  counted vmcnt waits)
            int wtile_v = wtile_b, plane_v = plane_b;
            asm volatile("" : "+s"(wtile_v), "+s"(plane_v));
            static_for<0, NSTEPS>([&](auto sc_) __attribute__((always_inline)) {
                constexpr int st = decltype(sc_)::value;
                constexpr int cb = st / NTAPS, t = st % NTAPS;
                // stores of this step: NDR spread over the first DSTEPS steps
                constexpr int q_lo = st < DSTEPS ? (st * NDR + DSTEPS - 1) / DSTEPS : NDR;
                constexpr int q_hi = st + 1 < DSTEPS ? ((st + 1) * NDR + DSTEPS - 1) / DSTEPS : NDR;
                // ops younger than the weight tile of step st + 1 (issued in step st - 2): the stores of steps st - 2 and st - 1, the tile of
                // step st - 1 and, when that was tap 1, the next phase's halo
                constexpr int sp2 = (st + NSTEPS - 2) % NSTEPS, sp1 = (st + NSTEPS - 1) % NSTEPS;
                constexpr int n2 = (sp2 + 1 < DSTEPS ? ((sp2 + 1) * NDR + DSTEPS - 1) / DSTEPS : NDR) - (sp2 < DSTEPS ? (sp2 * NDR + DSTEPS - 1) / DSTEPS : NDR);
                constexpr int n1 = (sp1 + 1 < DSTEPS ? ((sp1 + 1) * NDR + DSTEPS - 1) / DSTEPS : NDR) - (sp1 < DSTEPS ? (sp1 * NDR + DSTEPS - 1) / DSTEPS : NDR);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + n2 + n1 + (t == 2 ? AP : 0)) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (t == 1) {
                    if (cb + 1 < CBT) {
                        STREAM_ISSUE_A(mt, true, cb + 1, hbuf ^ 1)
                    } else {                            // the next item's first block (or nothing: out-of-range pieces, the counts stay)
                        STREAM_ISSUE_A(mt + 1, left > 0, 0, hbuf ^ 1)
                    }
                }
                {
                    constexpr int tn = (t + LA + 1) % NTAPS;
                    constexpr int cbn = (cb + (t + LA + 1) / NTAPS) % CBT;
                    STREAM_ISSUE_B(cbn, tn, islot)
                }
                if (++islot == SB) islot = 0;
                __builtin_amdgcn_sched_barrier(0);
                const char* abuf = lds + hbuf * A_BUF;
                const char* wcur = lds + O_B + rslot * B_SLOT;
                if (++rslot == SB) rslot = 0;
                const char* wnext = lds + O_B + rslot * B_SLOT;
                constexpr bool first = st == 0;
                // second half of THIS tap -> set 1, first half's matrix work (set 0, read half a step ago), the drain of the previous item,
                // first half of the NEXT tap -> set 0 (its tile landed before this step's barrier), second half's matrix work
                STREAM_READ(xh1, xl1, wh1, wl1, abuf, wcur, t, 1)
#pragma unroll
                for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh0, wh0[n], first ? zero16 : hh[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh0, wl0[n], first ? zero16 : xx[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl0, wh0[n], xx[n], 0, 0, 0);
                static_for<q_lo, q_hi>([&](auto qc) __attribute__((always_inline)) { drain_store(qc); });
                if (t + 1 < NTAPS) {
                    STREAM_READ(xh0, xl0, wh0, wl0, abuf, wnext, (t + 1) % NTAPS, 0)
                } else {
                    STREAM_READ(xh0, xl0, wh0, wl0, lds + (hbuf ^ 1) * A_BUF, wnext, 0, 0)
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh1, wh1[n], hh[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh1, wl1[n], xx[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl1, wh1[n], xx[n], 0, 0, 0);
                // (the reads of set 0 above must be ISSUED before the next step's barrier releases the slot they read to the DMA two steps on;
                //  the counted wait below is in program order behind them)
                if (t == NTAPS - 1) hbuf ^= 1;
            });
            // ---- item boundary: fold the accumulators into the drain registers (the next item's first MFMAs start from zero)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) dr[n][r] = fmaf(xx[n][r], LO_INV, hh[n][r]);
            drain_target(mt);
            ++mt;
        }
#undef STREAM_READ
#undef STREAM_ISSUE_A
#undef STREAM_ISSUE_B
#undef PLANE_STRIDE
#undef WTILE_STRIDE
        // ---- the last item of the segment has nobody to ride on: plain store sequence, then the statistics of the whole segment
        static_for<0, NDR>([&](auto qc) __attribute__((always_inline)) { drain_store(qc); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // also: the look-ahead pieces must not land in LDS that `red` re-uses
        __builtin_amdgcn_s_barrier();
        if (a.stats) {
            float* red = reinterpret_cast<float*>(lds);
            constexpr int BN = 32 * NT;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float x = s1[n], y = s1[n];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
                const float t1 = x + y;
                x = s2[n]; y = s2[n];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
                const float t2 = x + y;
                if (lane < 32) {
                    red[(0 * NW + wave) * BN + n * 32 + l31] = t1;
                    red[(1 * NW + wave) * BN + n * 32 + l31] = t2;
                }
            }
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
            __syncthreads();                                        // `red` is LDS the next segment's prologue refills
        }
    }
#endif
}
