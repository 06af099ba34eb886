// Persistent ("streaming") form of the 3 x 3 halo convolution for outputs that are pre-normalisation fp32 (+ InstanceNorm statistics):
// every 3 x 3 of the feature encoder (extractor.py:47-55,103-125) on grids that fill the chip more than twice.  Included by
// conv_split.hip inside its anonymous namespace (uses ConvArgs, slab_row / slab_col, the LDS image of conv_halo_kernel<2, 3, 3, TR>).
//
// Why: conv_halo_kernel's workgroups live for CB x 9 k-steps only; a third of a workgroup's life is its prologue (first halo + weight
// tiles straight from HBM, every workgroup of a round at once) and its epilogue (32 store instructions per lane + the statistics
// reduction), during which its SIMDs have no matrix work of their own: the layer-1 launch ran at 0.33-0.36 of the split format's matrix
// peak for three rounds ("the lever that is left is a persistent kernel that overlaps tile i's store drain with
// tile i+1's first halo under counted vmcnt over loads AND stores -- the structure K5 has").  This is that kernel:
//   * <= 2 x 256 persistent workgroups; a workgroup owns ONE channel tile of a CONTIGUOUS range of the (image, 8 x 16 patch) list and walks
//     it as ONE k-loop: the halo double buffer and the 4-slot weight ring run on across item boundaries, so item i+1's first halo and
//     weight tiles land under item i's last taps (the prologue exists once per workgroup);
//   * the accumulators are transposed (D[pixel][channel]: one register of a wave = two complete 128-B rows of the blocked fp32 output);
//     at an item boundary they are folded into 32 "drain" registers (hi + lo 2^-11) and the first MFMAs of the next item start from a
//     zero C operand; the drain registers are scaled, added into the lane's statistics and STORED between the MFMAs of item i+1's
//     first two channel blocks, 2-3 per k-step -- no store phase, no LDS, no barrier of its own;
//   * vmcnt counts LDS-DMA pieces and stores alike and retires them in order (gfx9): every step issues a compile-time number of both
//     (out-of-range stores of edge patches / of the empty drain of the first item are dropped by the buffer bounds check, not skipped),
//     so "the weight tile of the next step has landed" stays a fixed s_waitcnt immediate.  The last two taps of a channel block issue no
//     stores, which makes every block's counts -- and the prologue's -- independent of its neighbours;
//   * a range never leaves its (image, channel tile) group without a flush, so the InstanceNorm sums stay in two registers per lane
//     and channel block for the whole range: ONE reduction + 128 fp64 atomics per workgroup instead of one per item.
// The k-loop of an item is three instantiations of one 9-tap body -- channel block 0 (zero C operand, drains output block 0 of the previous
// item), block 1 (drains output block 1), blocks >= 2 (no drain) -- so the number of input channel blocks is a run-time value (>= 2) and
// the unrolled code (and its register pressure) does not grow with it.  64-channel output tiles (NT = 2).
//
// CSTREAM_ABL (tools/enc_stream_ablate.sh only; timing builds with WRONG results): 1 no stores, 2 no in-loop LDS-DMA, 4 no fragment reads,
// 5 no barrier / vmcnt wait, 6 a third of the fragment reads less (the weight fragments of the second k-half), 7 half the weight pieces (no
// counted waits) -- 6 and 7 price a 64 x 64 wave tile / a weight tile per CU before anybody writes them.
#ifndef CSTREAM_ABL
#define CSTREAM_ABL 0
#endif
// CSTREAM_PIN (tools A/B): 1 = scheduling barriers hold the fragment reads where the source puts them (set 1 of a tap at the top of its step,
// set 0 of the next tap between the two halves of the matrix work) instead of where the scheduler sinks them (right in front of their use)
#ifndef CSTREAM_PIN
#define CSTREAM_PIN 0
#endif
// CSTREAM_PRIO (tools A/B): 1 = s_setprio 1 around the matrix instructions of a half step (the wave that has its fragments issues ahead of its
// SIMD mate's loads and address arithmetic)
#ifndef CSTREAM_PRIO
#define CSTREAM_PRIO 0
#endif
// NIN (as conv_halo_kernel<..., NIN>): the input is the previous convolution's PRE-NORMALISATION fp32 output + its InstanceNorm statistics
// (ConvArgs.xraw / xstats); a thread loads 4 channels of a halo row as a float4, applies relu((x - mean) * rstd) with the coefficients the
// normalisation kernel would use, splits and writes the halo buffer by ds_write -- the LDS image the LDS-DMA path produces.  The six loads
// of a channel block go in three thirds (taps 1, 3, 5 of the block before; written at taps 3, 5, 7): eight registers, not twenty-four.
template <bool NIN>
__global__ __launch_bounds__(CT, 2) void conv_halo_stream_kernel(ConvArgs a, int per_old, int per_young, int n_patches, int tiles_x, int total, int n_images) {
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
    constexpr int DTAPS = NTAPS - 2;                   // taps of a channel block that carry stores (its last two carry none)
    constexpr int NHL = (HR + 31) / 32, NHH = NHL / 3; // NIN: halo loads per thread and channel block, per third
    static_assert(NHL == 3 * NHH && NHH == 2 && (A_UNITS * 16 - HR) * 64 >= 128 * 4, "NIN: two equal halves; room for the coefficient tables behind the halo rows");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int CB = a.CB;
#ifdef H8_STAMPS   // tools/conv_stamps.sh build: workgroup w -> [4 w]: shader clock / 100 MHz wall clock at its start and end
    if (a.stamps && tid == 0) { a.stamps[blockIdx.x * 4 + 0] = __builtin_readcyclecounter(); a.stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif

    // ---- per-lane constants of the LDS-DMA pieces (independent of the item)
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const int plane_b = a.P_in * 64;
    const bool lo_a = wave & 1, lo_w = wave / (NW / 2);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(lo_w ? a.wl : a.wh), 0, NTAPS * CB * wtile_b, 0x00020000);
    char* const a_dst = lds + (lo_a ? A_PLANE : 0) + (wave >> 1) * 1024;
    char* const w_dst = lds + O_B + (lo_w ? B_PLANE : 0);
    const int R0 = (wave * 2 + slab_row(l31)) * HWD + slab_col(l31);
    const int wsw = (l31 >> 2) & 3;
    const long long oplane = (long long)a.P_out * 32;  // floats of one channel block of one output image
    const float act_floor = a.act == 1 ? 0.f : -__builtin_inff();      // act 0 / 1 (tanh epilogues stay on conv_halo_kernel: see the dispatch)
    // Interior patches (the whole halo inside the image: 84 % of the patches at 240 x 320): the source offset of a halo piece is a per-lane
    // CONSTANT (halo row, column -> hrel) + a per-patch SCALAR that rides in the instruction's scalar offset, so a piece costs no vector
    // instruction at all; only the patches on the image border (and the out-of-range pieces behind a range's last item) form their
    // offsets lane by lane with the four bounds checks (`halo_offset` / `nin_load`'s slow path: ~25 instructions per piece).
    unsigned hrel[NIN ? NHL : AP];
    if constexpr (NIN) {
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int hy = row / HWD, hx = row - hy * HWD;
            hrel[i] = row < HR ? (unsigned)(((hy * a.W + hx) * 32 + (tid & 7) * 4) * 4) : 0x80000000u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = ((wave >> 1) + (NW / 2) * i) * 16 + urow;
            const int hy = row / HWD, hx = row - hy * HWD;
            hrel[i] = row < HR ? (unsigned)(((hy * a.W + hx) * 32 + uchunk) * 2) : 0x80000000u;
        }
    }
    auto interior = [&](int y0, int x0) -> bool { return y0 >= 1 && x0 >= 1 && y0 + TH < a.H && x0 + TW < a.W; };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // workgroup -> range.  A range = `per` consecutive patches of the (image, patch) list x ONE channel tile; the ranges are numbered with
    // the channel tile fastest, and workgroup w (which runs on XCD w % 8) takes range (w % 8) G/8 + w / 8: an XCD owns CONSECUTIVE ranges,
    // i.e. both channel tiles of the same patches -- walked at the same time by neighbouring workgroups, so that the halo they share is
    // fetched into that XCD's L2 once (as conv_halo_kernel's "all channel tiles of a patch on one XCD, back to back") -- and a band of
    // ~19 patch rows of an image at the layer-1 size, whose vertically adjacent patches share halo rows
    //
    // Uneven ranges (round 5, profiles/r05_enc_stream_clock.txt): the two workgroups of a CU do not run at the same speed -- the one that was
    // dispatched first (blockIdx < gridDim / 2: the dispatcher fills one slot of every CU before it doubles up) needs 16.2 k cycles per item, its
    // younger mate 21.2 k (the SIMD's issue arbitration prefers the older wave), at the same clock; with equal ranges the older half of the grid
    // finishes at 0.76 of the launch and the rest of it runs one workgroup per CU.  The older workgroup of a pair can therefore take per_old, the
    // younger per_young <= per_old consecutive items (the dispatch's BFLOW_CONV_STREAM_SHARE; equal by default: see there).  The grid is a multiple of 16 n_tiles: an XCD (blockIdx % 8) owns gridDim / 8 n_tiles ranges
    // = the older ones first, then the younger ones, all consecutive.
    const int nx = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int chunk = xcd * nx + j;
    const int n0 = (j % a.n_tiles) * (32 * NT);
    const int rx2 = (nx / a.n_tiles) >> 1, k = j / a.n_tiles;          // pairs of ranges per XCD; this workgroup's range inside the XCD's band
    const bool older = k < rx2;
    int q = xcd * rx2 * (per_old + per_young) + (older ? k * per_old : rx2 * per_old + (k - rx2) * per_young);
    const int q_end = min(q + (older ? per_old : per_young), total);
    while (q < q_end) {
        // ---- one segment: patches q .. seg_end - 1 of ONE image
        const int b = q / n_patches;
        int mt = q - b * n_patches;
        const int seg_end = (b + 1) * n_patches < q_end ? (b + 1) * n_patches : q_end;
        int left = seg_end - q;                         // items still to start
        q = seg_end;

        const rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_a ? a.xl : a.xh) + (long long)b * CB * a.P_in * 32), 0, CB * plane_b, 0x00020000);
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
        // NIN: mul / add per input channel of image b in the rows HR .. 16 A_UNITS of halo buffer 0 that no tap ever reads (hi plane: mul, lo: add)
        float* const nmul = reinterpret_cast<float*>(lds + HR * 64);
        float* const nadd = reinterpret_cast<float*>(lds + A_PLANE + HR * 64);
        const int ng = tid & 7;
        rsrc_t r_raw = r_w;
        if constexpr (NIN) {
            r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xraw + (long long)b * CB * a.P_in * 32), 0, CB * a.P_in * 128, 0x00020000);
            for (int c = tid; c < CB * 32; c += CT) {
                float m_, a_;
                norm_coeffs(a.xstats, nullptr, nullptr, b, c, CB * 32, a.H * a.W, a.xeps, m_, a_, a.xstats_reps, (long long)n_images * CB * 32 * 2);
                nmul[c] = m_;
                nadd[c] = a_;
            }
        }
        float4 nv[NHH];                                 // the third of a channel block's raw halo that is in flight
        unsigned nval = 0;                              // bit i: row i of that third lies inside the image
        // loads of third `hf` of block `cbi` of patch m (zeros when there is no such item); written to halo buffer `buf` by nin_write
        auto nin_load = [&](int m, bool exists, int cbi, auto hfc) __attribute__((always_inline)) {
            constexpr int hf = decltype(hfc)::value;
            const int ty = m / tiles_x;
            const int y0 = ty * TH, x0 = (m - ty * tiles_x) * TW;
            if (exists && interior(y0, x0)) {
                const int sbase = cbi * a.P_in * 128 + ((y0 - 1) * a.W + (x0 - 1)) * 128;
                nval = (1u << NHH) - 1;                 // (rows >= HR are never written: nin_write checks the row)
#pragma unroll
                for (int i = 0; i < NHH; ++i)
                    nv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_raw, hrel[hf * NHH + i], sbase, 0));
                return;
            }
            nval = 0;
#pragma unroll
            for (int i = 0; i < NHH; ++i) {
                int row = (tid >> 3) + 32 * (hf * NHH + i);
                asm volatile("" : "+v"(row));
                const int hy = row / HWD, hx = row - hy * HWD;
                const int py = y0 - 1 + hy, px = x0 - 1 + hx;
                const bool ok = exists & (row < HR) & ((unsigned)py < (unsigned)a.H) & ((unsigned)px < (unsigned)a.W);
                nval |= ok ? (1u << i) : 0u;
                nv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_raw, ok ? (unsigned)(((py * a.W + px) * 32 + ng * 4) * 4) : 0x80000000u,
                                                                                      cbi * a.P_in * 128, 0));
            }
        };
        auto nin_write = [&](int cbi, int buf, auto hfc, auto ic) __attribute__((always_inline)) {
            constexpr int hf = decltype(hfc)::value, i = decltype(ic)::value;
            const float4 m4 = *reinterpret_cast<const float4*>(nmul + cbi * 32 + ng * 4);
            const float4 a4 = *reinterpret_cast<const float4*>(nadd + cbi * 32 + ng * 4);
            const int row = (tid >> 3) + 32 * (hf * NHH + i);
            if (row < HR) {
                const bool in_ = (nval >> i) & 1;      // zero padding applies to the NORMALISED activation
                const float v_[4] = {in_ ? fmaxf(fmaf(nv[i].x, m4.x, a4.x), 0.f) : 0.f, in_ ? fmaxf(fmaf(nv[i].y, m4.y, a4.y), 0.f) : 0.f,
                                     in_ ? fmaxf(fmaf(nv[i].z, m4.z, a4.z), 0.f) : 0.f, in_ ? fmaxf(fmaf(nv[i].w, m4.w, a4.w), 0.f) : 0.f};
                half4v h4_, l4_;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    _Float16 x1_, x2_;
                    split1(v_[k], x1_, x2_);
                    h4_[k] = x1_;
                    l4_[k] = x2_;
                }
                char* d_ = lds + buf * A_BUF + row * 64 + ((((ng >> 1) ^ ((row >> 2) & 3))) << 4) + ((ng & 1) << 3);
                *reinterpret_cast<half4v*>(d_) = h4_;
                *reinterpret_cast<half4v*>(d_ + A_PLANE) = l4_;
            }
        };

        // LDS-DMA source offsets of the halo of patch m (all out of range when there is no such item: the pieces are issued all the same,
        // the counts stay); formed where the pieces are issued -- twelve registers that would otherwise live through the whole k-loop
        auto halo_offset = [&](int i, int y0, int x0, bool exists) -> unsigned {
            int row = ((wave >> 1) + (NW / 2) * i) * 16 + urow;
            asm volatile("" : "+v"(row));               // border patches only: re-formed here, not carried through the k-loop in 12 registers
            const int hy = row / HWD, hx = row - hy * HWD;
            const int py = y0 - 1 + hy, px = x0 - 1 + hx;
            const bool ok = exists & (row < HR) & ((unsigned)py < (unsigned)a.H) & ((unsigned)px < (unsigned)a.W);
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
            // register r of lane (channel, kh): column x0 + r, slab row (popcount(r >> 2) + kh) & 1 -> dbase[parity of popcount(r >> 2)]
            const int ya = yw + kh, yb = yw + (kh ^ 1);
            dbase[0] = ya < a.H ? (unsigned)((ya * a.W + x0) * 128 + l31 * 4) : 0x80000000u;
            dbase[1] = yb < a.H ? (unsigned)((yb * a.W + x0) * 128 + l31 * 4) : 0x80000000u;
            dxlim = a.W - x0;
        };
        // store of register r of output block n of the drained item
        auto drain_store = [&](auto nc, auto rc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value, r = decltype(rc)::value;
            constexpr int par = __builtin_popcount((unsigned)(r >> 2)) & 1;
            float v = fmaxf(fmaf(dr[n][r], sc[n], sh[n]), act_floor);      // ReLU = a clamp from below: no branch between the MFMAs
            if (!cok[n]) v = 0.f;
            const unsigned off = r < dxlim ? dbase[par] : 0x80000000u;
            if (CSTREAM_ABL != 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_o[n], off, r * 128, CONV_NT_STORES_ENC ? 2 : 0);
            if (off != 0x80000000u) { s1[n] += v; s2[n] = fmaf(v, v, s2[n]); }
        };

#define STREAM_ISSUE_A(M, EXISTS, CBI, BUF)                                                                              \
    {                                                                                                                    \
        const int ty_ = (M) / tiles_x;                                                                                   \
        const int y0_ = ty_ * TH, x0_ = ((M) - ty_ * tiles_x) * TW;                                                      \
        if constexpr (!NIN) {                                                                                            \
            if ((EXISTS) && interior(y0_, x0_)) {                                                                        \
                const int sb_ = (CBI) * plane_b + ((y0_ - 1) * a.W + (x0_ - 1)) * 64;                                    \
                _Pragma("unroll") for (int i = 0; i < AP; ++i)                                                           \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (lptr_t)(a_dst + (BUF) * A_BUF + i * (NW / 2) * 1024), 16, hrel[i], sb_, 0, 0); \
            } else {                                                                                                     \
                _Pragma("unroll") for (int i = 0; i < AP; ++i)                                                           \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (lptr_t)(a_dst + (BUF) * A_BUF + i * (NW / 2) * 1024), 16, \
                                                             halo_offset(i, y0_, x0_, (EXISTS)), (CBI) * plane_b, 0, 0); \
            }                                                                                                            \
        }                                                                                                                \
    }
#define STREAM_ISSUE_B(CBI, TAP, SLOT)                                                                                   \
    {                                                                                                                    \
        const int so_ = ((TAP) * CB + (CBI)) * wtile_b;                                                                  \
        _Pragma("unroll") for (int j = 0; j < (CSTREAM_ABL == 7 ? NBP / 2 : NBP); ++j)   /* 7: half the weight pieces (a tile per CU) */ \
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
        const int co_ = (((KS) * 2 + kh) ^ wsw) * 16;                                                                    \
        if (CSTREAM_ABL == 4) {                                                                                          \
            asm volatile("" : "+v"(XH), "+v"(XL), "+v"(WH[0]), "+v"(WL[0]), "+v"(WH[1]), "+v"(WL[1]));                    \
        } else {                                                                                                         \
            XH = *reinterpret_cast<const half8*>((ABUF) + ao_);                                                          \
            XL = *reinterpret_cast<const half8*>((ABUF) + A_PLANE + ao_);                                                \
            if (CSTREAM_ABL == 6 && (KS) == 1) {  /* a third of the reads less: what a 64 x 64 wave tile would read per MFMA */    \
                asm volatile("" : "+v"(WH[0]), "+v"(WL[0]), "+v"(WH[1]), "+v"(WL[1]));                                    \
            } else {                                                                                                     \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                         \
                    const int wo_ = (n * 32 + l31) * 64 + co_;                                                           \
                    WH[n] = *reinterpret_cast<const half8*>((WSLOT) + wo_);                                              \
                    WL[n] = *reinterpret_cast<const half8*>((WSLOT) + B_PLANE + wo_);                                    \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
    }

        // ---- prologue of the segment: first halo, first LA + 1 weight tiles
        if constexpr (NIN) {
            __syncthreads();                                        // coefficient tables complete
            static_for<0, 3>([&](auto hfc) __attribute__((always_inline)) {
                nin_load(mt, true, 0, hfc);
                static_for<0, NHH>([&](auto ic) __attribute__((always_inline)) { nin_write(0, 0, hfc, ic); });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to the other waves at the barrier below
        } else {
            STREAM_ISSUE_A(mt, true, 0, 0)
        }
#pragma unroll
        for (int t = 0; t <= LA; ++t) STREAM_ISSUE_B(0, t, t)
        f32x16 hh[NT], xx[NT];
        half8 xh0, xl0, wh0[NT], wl0[NT], xh1, xl1, wh1[NT], wl1[NT];   // fragment set 0 / 1 = k-half 0 / 1 of a tap
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA * NBP) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STREAM_READ(xh0, xl0, wh0, wl0, lds, lds + O_B, 0, 0)
        int islot = LA + 1, rslot = 0;                  // ring slot to fill next / of the tap being computed
        int hbuf = 0;                                   // halo buffer of the phase (item, channel block) being computed

        // The 9 taps of channel block `cb` of the current item.  KIND 0: cb == 0 (zero C operand at tap 0; drains output block 0 of the
        // previous item), 1: cb == 1 (drains output block 1), 2: cb >= 2 (no drain).  Stores per tap (KIND < 2): 16 over the first DTAPS taps.
        auto block9 = [&](auto kind_c, const int cb) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
            const bool last_cb = cb + 1 == CB;
            static_for<0, NTAPS>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                // drain registers of this tap: r_lo .. r_hi - 1 of output block KIND
                constexpr int r_lo = (KIND < 2 && t < DTAPS) ? (t * 16 + DTAPS - 1) / DTAPS : 16;
                constexpr int r_hi = (KIND < 2 && t + 1 < DTAPS) ? ((t + 1) * 16 + DTAPS - 1) / DTAPS : 16;
                // ops younger than the weight tile of tap t + 1 (issued two steps ago): the stores of the last two steps, the tile issued one step
                // ago and, when that step was tap 1, the next phase's halo.  Taps DTAPS .. NTAPS - 1 carry no stores, so steps -2 / -1 of a
                // block contribute none whatever block preceded it.
                constexpr int tm2 = (t + NTAPS - 2) % NTAPS, tm1 = (t + NTAPS - 1) % NTAPS;
                constexpr int n2 = (KIND < 2 && t >= 2 && tm2 < DTAPS) ? ((tm2 + 1 < DTAPS ? ((tm2 + 1) * 16 + DTAPS - 1) / DTAPS : 16) - (tm2 * 16 + DTAPS - 1) / DTAPS) : 0;
                constexpr int n1 = (KIND < 2 && t >= 1 && tm1 < DTAPS) ? ((tm1 + 1 < DTAPS ? ((tm1 + 1) * 16 + DTAPS - 1) / DTAPS : 16) - (tm1 * 16 + DTAPS - 1) / DTAPS) : 0;
                if (CSTREAM_ABL != 5 && CSTREAM_ABL != 1 && CSTREAM_ABL != 2 && CSTREAM_ABL != 7) {
                    // (NIN: the raw loads of taps 1, 3 and 5 stand where the halo pieces of tap 1 stand)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + n2 + n1 + (NIN ? ((t == 2 || t == 4 || t == 6) ? NHH : 0) : (t == 2 ? AP : 0))) : "memory");
                }
                if (CSTREAM_ABL != 5) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (CSTREAM_ABL != 2) {
                    if constexpr (NIN) {
                        // next phase = block cb + 1 of this item, or block 0 of the next one (or nothing: out-of-range loads, the counts stay)
                        const int m_n = last_cb ? mt + 1 : mt, cb_n = last_cb ? 0 : cb + 1;
                        const bool ex_n = !last_cb || left > 0;
                        // third k: written at tap 2k + 3 (its loads, issued two taps before, have landed behind that tap's counted wait), then the
                        // next third is requested into the same registers; the buffer is read from tap NTAPS - 1 on
                        if (t == 3 || t == 5 || t == 7) {
                            static_for<0, NHH>([&](auto ic) __attribute__((always_inline)) { nin_write(cb_n, hbuf ^ 1, std::integral_constant<int, (t - 3) / 2>{}, ic); });
                            if (t == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible at the barrier of tap NTAPS - 1
                        }
                        if (t == 1 || t == 3 || t == 5) nin_load(m_n, ex_n, cb_n, std::integral_constant<int, (t - 1) / 2>{});
                    } else if (t == 1) {
                        if (!last_cb) {
                            STREAM_ISSUE_A(mt, true, cb + 1, hbuf ^ 1)
                        } else {                        // the next item's first block (or nothing: out-of-range pieces, the counts stay)
                            STREAM_ISSUE_A(mt + 1, left > 0, 0, hbuf ^ 1)
                        }
                    }
                    constexpr int tn = (t + LA + 1) % NTAPS;
                    const int cbn = (t + LA + 1 < NTAPS) ? cb : (last_cb ? 0 : cb + 1);
                    STREAM_ISSUE_B(cbn, tn, islot)
                }
                if (++islot == SB) islot = 0;
                __builtin_amdgcn_sched_barrier(0);
                const char* abuf = lds + hbuf * A_BUF;
                const char* wcur = lds + O_B + rslot * B_SLOT;
                if (++rslot == SB) rslot = 0;
                const char* wnext = lds + O_B + rslot * B_SLOT;
                constexpr bool first = KIND == 0 && t == 0;
                // second half of THIS tap -> set 1, first half's matrix work (set 0, read half a step ago), the drain of the previous item,
                // first half of the NEXT tap -> set 0 (its tile landed before this step's barrier), second half's matrix work
                STREAM_READ(xh1, xl1, wh1, wl1, abuf, wcur, t, 1)
                if (CSTREAM_PIN) __builtin_amdgcn_sched_barrier(0);
                if (CSTREAM_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh0, wh0[n], first ? zero16 : hh[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh0, wl0[n], first ? zero16 : xx[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl0, wh0[n], xx[n], 0, 0, 0);
                if (CSTREAM_PRIO) __builtin_amdgcn_s_setprio(0);
                if (CSTREAM_PIN) __builtin_amdgcn_sched_barrier(0);
                static_for<r_lo, r_hi>([&](auto rc) __attribute__((always_inline)) { drain_store(std::integral_constant<int, (KIND < 2 ? KIND : 0)>{}, rc); });
                if (t + 1 < NTAPS) {
                    STREAM_READ(xh0, xl0, wh0, wl0, abuf, wnext, (t + 1) % NTAPS, 0)
                } else {
                    STREAM_READ(xh0, xl0, wh0, wl0, lds + (hbuf ^ 1) * A_BUF, wnext, 0, 0)
                }
                if (CSTREAM_PIN) __builtin_amdgcn_sched_barrier(0);
                if (CSTREAM_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh1, wh1[n], hh[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh1, wl1[n], xx[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl1, wh1[n], xx[n], 0, 0, 0);
                if (CSTREAM_PRIO) { __builtin_amdgcn_s_setprio(0); if (CSTREAM_PIN) __builtin_amdgcn_sched_barrier(0); }
            });
            hbuf ^= 1;
        };

        while (left > 0) {
            --left;
            block9(std::integral_constant<int, 0>{}, 0);
            block9(std::integral_constant<int, 1>{}, 1);
#pragma nounroll
            for (int cb = 2; cb < CB; ++cb) block9(std::integral_constant<int, 2>{}, cb);
            // ---- item boundary: fold the accumulators into the drain registers (the next item's first MFMAs start from a zero C operand)
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
        // ---- the last item of the segment has nobody to ride on: plain store sequence, then the statistics of the whole segment
        static_for<0, NT>([&](auto nc) __attribute__((always_inline)) {
            static_for<0, 16>([&](auto rc) __attribute__((always_inline)) { drain_store(nc, rc); });
        });
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
                int rep = chunk % a.stats_reps;
                asm volatile("" : "+s"(rep));               // (formed here: hoisted to the kernel's top the address costs four registers for its whole life)
                if (col < a.Cout) {
                    const float* p = red + which * NW * BN + c;
                    double sum = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += (double)p[w * BN];
                    atomicAdd(a.stats + (long long)rep * a.stats_rep_stride + ((long long)b * a.Cout + col) * 2 + which, sum);
                }
            }
            __syncthreads();                                        // `red` is LDS the next segment's prologue refills
        }
    }
#ifdef H8_STAMPS
    if (a.stamps && tid == 0) { a.stamps[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); a.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
#endif
}
