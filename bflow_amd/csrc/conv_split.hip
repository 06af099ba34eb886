// Implicit-GEMM convolution on the split-fp16 MFMA engine (see split_gemm.hip for the number format).
//
// Activations are "blocked channels-last split" tensors: two fp16 planes (hi, lo), each laid out (B, C/32, P, 32) --
// 32-channel blocks, P >= H*W pixel rows per image, value = hi + lo * 2^-11.  One (filter tap, channel block) pair is one
// 32-deep k-tile whose rows (pixels) are 64-B contiguous runs, i.e. exactly the k-contiguous operand order the fp16 MFMA
// wants AND full-line, channel-spread global reads -- no im2col buffer, no transposes:
//       out[pix, co] = sum_{tap, c} x[pix + tap, c] * w[co, tap, c].
// Weights are pre-split once into (KH*KW*C/32, Cout_pad, 32) planes (same k-tile-major order).
//
//   block  : 128 output pixels (of ONE image) x BN = 32*NT output channels, 4 waves, wave tile 32 x BN
//   loads  : global_load_lds_dwordx4 (LDS-DMA, no staging registers) into a 3-stage ring of k-tiles; rows outside the
//            image / beyond the last pixel read a zero page (= zero padding); XOR-swizzled 64-B LDS rows (swizzle on the
//            per-lane source address and on the fragment read); counted vmcnt + ONE raw s_barrier per k-tile
//   MFMA   : per 16-deep step NT x { hi*hi, hi*lo, lo*hi } v_mfma_f32_32x32x16_f16, fp32 accumulation; the fragments of
//            both 16-deep steps are fetched before the first MFMA so LDS latency hides behind the matrix pipe
//   epilogue (fused): per-channel affine (conv bias / folded BatchNorm), ReLU, optional per-(image, channel) sum / sum-of-
//            squares for InstanceNorm (fp64 atomics, one per channel per block), blocked fp32 and/or blocked split stores
//            at a channel-block offset (writes straight into concatenated buffers; a 32x32 MFMA tile is one contiguous
//            2-KB / 4-KB run).
#include "conv_engine.h"

#ifndef CONV_NT_STORES_ENC
#define CONV_NT_STORES_ENC 0  // tools A/B: 1 = non-temporal stores in the direct (transposed-accumulator) epilogue of the encoder's 3x3s and stem
#endif
#ifndef CONV_SPREAD_DMA
#define CONV_SPREAD_DMA 0     // tools A/B (tools/build_flag_variant.sh): 1 = the small-grid kernels issue a step's LDS-DMA pieces between its taps
#endif

#ifdef H8_STAMPS   // tools/conv_stamps.sh build: s_memtime stamps of every wave of conv_halo8_kernel at its phase boundaries (16 x u64 per wave)
static unsigned long long* g_h8_stamp_buf = nullptr;
extern "C" __attribute__((visibility("default"))) void bflow_conv_set_stamp_buffer(void* p) { g_h8_stamp_buf = (unsigned long long*)p; }
#define H8STAMP(i) \
    if (a.stamps && lane == 0) a.stamps[((blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave_all) * 16 + (i)] = __builtin_readcyclecounter();
#define H8STAMP_RT(i) \
    if (a.stamps && lane == 0) a.stamps[((blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave_all) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define H8STAMP(i)
#define H8STAMP_RT(i)
#endif

namespace {


// ---------------------------------------------------------------------------------------------------------------------
// Direct epilogue of the TRANSPOSED halo kernel (TR = true: activation fragment as the A operand, weight fragment as B, so an
// accumulator tile is D[pixel][channel]): lane = output channel (lane & 31), and register r of a lane is pixel column x0 + r of the
// wave's 2 x 16 slab, patch row (popcount(r >> 2) + (lane >> 5)) & 1 (slab_row / slab_col of pixel (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
// One accumulator register of a wave is therefore TWO COMPLETE 128-B pixel rows of the blocked fp32 output (32 channels x 4 B each):
// it is stored as it is -- no LDS transpose, no slab round trip, no wave barrier (the store stream shape K5 uses: 4.7-5.0 TB/s alone,
// tools/micro/store_patterns.hip) -- the per-channel scale / shift is ONE value per lane and the InstanceNorm sums are plain register
// adds over the 16 registers plus one half swap.  Used for the convolutions that write pre-normalisation fp32 (+ statistics): every
// 3 x 3 of the InstanceNorm feature encoder (extractor.py:27-31,47-55).
// ---------------------------------------------------------------------------------------------------------------------
// `offset_of(r)`: byte offset of (the pixel of accumulator register r of this lane, channel lane & 31) inside one channel block of the
// output image, or 0x80000000 (out of range: the store is dropped by the buffer bounds check) for pixels outside the image.
template <int NT, int NW = 4, typename OffsetOf>
__device__ __forceinline__ void conv_epilogue_direct(const ConvArgs& a, f32x16 (&hh)[NT], f32x16 (&xx)[NT], int b, OffsetOf offset_of, int n0, int lane,
                                                     int wave, int tid, float* red) {
    constexpr int BN = 32 * NT;
    const int l31 = lane & 31;
    const long long plane = (long long)a.P_out * 32;                 // floats of one channel block of one image
    unsigned offs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) offs[r] = offset_of(r);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int cbase = n0 + n * 32;
        if (cbase >= a.Cout) break;
        const int c = cbase + l31;
        const bool cok = c < a.Cout;
        const float sc = (a.scale && cok) ? a.scale[c] : 1.f, sh = (a.shift && cok) ? a.shift[c] : 0.f;
        float* ob = a.out_f32 + ((long long)b * a.CBo + a.cb_off + (n0 >> 5) + n) * plane;
        const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)(plane * 4), 0x00020000);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = (hh[n][r] + xx[n][r] * LO_INV) * sc + sh;
            if (a.act == 1) v = fmaxf(v, 0.f);
            else if (a.act == 2) v = tanhf(v);
            if (!cok) v = 0.f;                                   // padded channels of the last block are written as zeros
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, offs[r], 0, CONV_NT_STORES_ENC ? 2 : 0);
            if (offs[r] != 0x80000000u) { s1 += v; s2 += v * v; }
        }
        if (a.stats) {                                           // the other 16 pixels of this channel sit in the other half of the wave
            float x = s1, y = s1;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
            s1 = x + y;
            x = s2; y = s2;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
            s2 = x + y;
            if (lane < 32) {
                red[(0 * NW + wave) * BN + n * 32 + l31] = s1;
                red[(1 * NW + wave) * BN + n * 32 + l31] = s2;
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

// NT = output-channel tile / 32;  S = depth of the LDS ring (k-tiles in flight = S - 1): 3 for grids that fill the chip
// (2 workgroups per CU hide each other's latency), deeper for small grids (batch-1 update block: <= 1 workgroup per CU, so
// the whole 160 KB of LDS can go into prefetch depth).
//   KG = k-groups per workgroup: with KG = 2 the workgroup has 8 waves; waves 0-3 and 4-7 own the same output tile but alternate
//        k-tiles (each group with its own LDS ring) and are summed through LDS at the end.  A small grid then runs 2 waves per
//        SIMD, so one group's LDS/issue latency hides behind the other's MFMAs (intra-workgroup split-K).
//   `bx` = the workgroup's index in ITS convolution's 1-D grid (blockIdx.x of a plain launch; a pair launch, below, subtracts the
//        first convolution's grid).
template <int NT, int S, int KG>
__device__ __forceinline__ void conv_split_body(const ConvArgs& a, const int bx) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses device-only builtins (buffer resources); the host pass only needs the stub
    constexpr int CSTAGES = S;
    constexpr int BN = 32 * NT;
    constexpr int BU = (NT <= 2) ? 1 : 2;                 // B units (16 rows x 64 B) per wave per plane
    constexpr int B_ARR = BU * 4 * 1024;                  // LDS bytes of one B plane (64 or 128 rows)
    constexpr int A_ARR = CBM * 64;
    constexpr int O_AH = 0, O_AL = A_ARR, O_BH = 2 * A_ARR, O_BL = 2 * A_ARR + B_ARR;
    constexpr int STAGE = 2 * A_ARR + 2 * B_ARR;          // 24 KB (NT <= 2) or 32 KB
    constexpr int NLOADS = 4 + 2 * BU;                    // LDS-DMA instructions per wave per k-tile
    extern __shared__ __attribute__((aligned(16))) char lds[];   // the only shared object: ring of CSTAGES stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & 3, grp = wave_all >> 2;      // wave inside its k-group, k-group
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int HoWo = a.Ho * a.Wo;
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2 each); the 1-D grid is decoded so
    // that ALL channel tiles of one pixel tile run on the SAME XCD, back to back: the pixel tile's activation rows are then
    // fetched from the Infinity Cache once per XCD instead of once per channel tile, and re-read from L2 (4x lower latency).
    int m0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = bx & 7, slot = bx >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        m0 = mt * CBM;
        n0 = (slot - (slot / ntn) * ntn) * BN;
        if (m0 >= HoWo) return;      // padding workgroups of the last group of 8 pixel tiles (uniform per workgroup)
    }
    const int ntaps = a.KH * a.KW;
    const int nk = ntaps * a.CB;
    char* const ring = lds + grp * (CSTAGES * STAGE);        // this k-group's LDS ring

    // ---- LDS-DMA sources.  Everything per-iteration is kept to a handful of instructions (the loop is otherwise bound by
    // address arithmetic, not by the matrix cores): the planes are addressed through BUFFER descriptors, so a padded / invalid
    // row is simply an out-of-range offset (the hardware returns zeros); per lane there is ONE constant byte offset per staged
    // row plus a bit mask of the filter taps that fall inside the image; the (tap, channel-block) part of the address is a
    // running scalar.  unit = 16 rows x 64 B; lane -> row lane/4, logical 16-B chunk (lane&3) ^ ((lane>>4)&3).
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;   // in halves
    int aoffb[2];                 // byte offset of (row's centre pixel - padding origin), may be negative
    unsigned long long vmask[2];  // bit t: filter tap t of this row lies inside the image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + (wave * 2 + j) * 16 + urow;
        const bool rok = m < HoWo;
        const int ho = m / a.Wo, wo = m - ho * a.Wo;
        const int hb = ho * a.stride - a.pad_h, wb = wo * a.stride - a.pad_w;
        aoffb[j] = ((hb * a.W + wb) * 32 + uchunk) * 2;
        unsigned long long mk = 0;
        for (int t = 0; t < ntaps; ++t) {
            const int r = t / a.KW, q = t - r * a.KW;
            const int hi_ = hb + r, wi_ = wb + q;
            if (rok && hi_ >= 0 && hi_ < a.H && wi_ >= 0 && wi_ < a.W) mk |= 1ull << t;
        }
        vmask[j] = mk;
    }
    unsigned wvo[BU];             // byte offset of this lane's weight row inside a k-tile
#pragma unroll
    for (int j = 0; j < BU; ++j) {
        int r = n0 + (wave * BU + j) * 16 + urow;
        r = r < a.cout_pad ? r : a.cout_pad - 1;          // NT = 3 stages 128 rows of a 96-wide tile: clamp (never consumed)
        wvo[j] = (unsigned)((r * 32 + uchunk) * 2);
    }
    const int CB2 = a.CB - a.CB1;
    const int plane_b = a.P_in * 64;                       // bytes of one channel block of one image
    const rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xh + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_l1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xl + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_h2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2h + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const rsrc_t r_l2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2l + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;                   // bytes of one weight k-tile
    const long long wset = a.w_sets > 1 ? (long long)(b % a.w_sets) * nk * (wtile_b / 2) : 0;   // halfs
    const rsrc_t r_wh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wh + wset), 0, nk * wtile_b, 0x00020000);
    const rsrc_t r_wl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wl + wset), 0, nk * wtile_b, 0x00020000);

    // running (uniform) position of the k-tile to be issued next: tap it, column iq inside the filter row, channel block icb,
    // byte offset of the tap itoff = (r*W + q)*64, flat k-tile index ik.  k-group g takes k-tiles g, g+KG, ...
    int it = 0, iq = 0, icb = grp, itoff = 0, ik = grp;
    bool ivalid = true;
    auto settle = [&]() {
        while (icb >= a.CB) {
            icb -= a.CB; ++it; ++iq; itoff += 64;
            if (iq == a.KW) { iq = 0; itoff += (a.W - a.KW) * 64; }
        }
        if (it >= ntaps) { ivalid = false; it = 0; iq = 0; icb = 0; itoff = 0; ik = 0; }   // tail: zero rows, any weights
    };
    settle();
    auto advance = [&]() { icb += KG; ik += KG; if (ivalid) settle(); else { icb = 0; ik = 0; } };

#define CONV_ISSUE(SB)                                                                                                   \
    {                                                                                                                    \
        char* sb = (SB);                                                                                                 \
        const bool first_ = icb < a.CB1;                         /* which of the two concatenated sources (uniform) */   \
        const rsrc_t rh_ = first_ ? r_h1 : r_h2;                                                                         \
        const rsrc_t rl_ = first_ ? r_l1 : r_l2;                                                                         \
        const int ub_ = (first_ ? icb : icb - a.CB1) * plane_b + itoff;                                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
            const bool ok = ivalid && ((vmask[j] >> it) & 1ull);                                                         \
            const unsigned vo = ok ? (unsigned)(aoffb[j] + ub_) : 0x80000000u;      /* out of range -> zeros */          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rh_, (lptr_t)(sb + O_AH + (wave * 2 + j) * 1024), 16, vo, 0, 0, 0);  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl_, (lptr_t)(sb + O_AL + (wave * 2 + j) * 1024), 16, vo, 0, 0, 0);  \
        }                                                                                                                \
        const int wso_ = ik * wtile_b;                                                                                   \
        _Pragma("unroll") for (int j = 0; j < BU; ++j) {                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wh, (lptr_t)(sb + O_BH + (wave * BU + j) * 1024), 16, wvo[j], wso_, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wl, (lptr_t)(sb + O_BL + (wave * BU + j) * 1024), 16, wvo[j], wso_, 0, 0); \
        }                                                                                                                \
    }

    f32x16 hh[NT], xx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hh[n][r] = 0.f;
            xx[n][r] = 0.f;
        }

    // (tap, channel block) of the k-tile to be issued next; the ring always runs two k-tiles ahead of the compute
#pragma unroll
    for (int st = 0; st < CSTAGES - 1; ++st) {
        CONV_ISSUE(ring + st * STAGE)
        advance();
    }

    const int sw = (l31 >> 2) & 3;
    const int aro = (wave * 32 + l31) * 64;
    const int nsteps = (nk + KG - 1) / KG;
    int islot = CSTAGES - 1, cslot = 0;        // ring slot to fill / to consume
    for (int kt = 0; kt < nsteps; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((CSTAGES - 2) * NLOADS) : "memory");   // k-tile kt landed; S-2 newer tiles may fly
        __builtin_amdgcn_s_barrier();          // ... for every wave; the slot of k-tile kt-1 is free again
        __builtin_amdgcn_sched_barrier(0);
        CONV_ISSUE(ring + islot * STAGE)
        advance();
        if (++islot == CSTAGES) islot = 0;
        __builtin_amdgcn_sched_barrier(0);
        const char* cur = ring + cslot * STAGE;
        if (++cslot == CSTAGES) cslot = 0;
        half8 ah[2], al[2], bh[2][NT], bl[2][NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {       // fragments of BOTH 16-deep steps first ...
            const int co = ((ks * 2 + kh) ^ sw) * 16;
            ah[ks] = *reinterpret_cast<const half8*>(cur + O_AH + aro + co);
            al[ks] = *reinterpret_cast<const half8*>(cur + O_AL + aro + co);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int bo = (n * 32 + l31) * 64 + co;
                bh[ks][n] = *reinterpret_cast<const half8*>(cur + O_BH + bo);
                bl[ks][n] = *reinterpret_cast<const half8*>(cur + O_BL + bo);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {       // ... then the MFMAs (three sweeps: no back-to-back dependent accumulators)
#pragma unroll
            for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks][n], ah[ks], hh[n], 0, 0, 0);   // D[channel][pixel]
#pragma unroll
            for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks][n], ah[ks], xx[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks][n], al[ks], xx[n], 0, 0, 0);
        }
    }
#undef CONV_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // all LDS-DMA traffic has landed: the ring can be reused for the reduction

    // ---- with KG = 2 the k-groups are combined on the read side of the epilogue (each group stages its partial sums)
    const bool writer = (grp == 0);

    // ---- epilogue (lane = pixel wave*32 + l31 of this tile) --------------------------------------------------------------
    const int mw = m0 + wave * 32;
    conv_epilogue<NT, 4 * KG, KG>(a, hh, xx, b, [=](int row) { return mw + row < HoWo ? mw + row : -1; }, n0, lane, wave, tid, writer,
                                  reinterpret_cast<float*>(lds), grp);
#endif
}

template <int NT, int S, int KG>
__global__ __launch_bounds__(CT * KG, 2) void conv_split_kernel(ConvArgs a) {
    conv_split_body<NT, S, KG>(a, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// PAIR launches (round 4).  Two INDEPENDENT convolutions that resolve to the same small-grid kernel run as ONE grid: workgroups
// [0, n0) are the first convolution's, [n0, n0 + n1) the second's (both counts are multiples of 8, so the XCD decode of either body is
// unchanged).  The batch-1 motion encoder (update.py:88-97) has two such pairs -- convc1 | convf1 and convc2 | convf2 -- that used
// to sit on two hardware queues: every cross-queue edge of the captured graph cost the main chain 5-7 us (profiles/r04_iteration_launches.txt:
// the 6.7 us hole before `conv`, the 4.6 us one after the Bezier head), more than the overlap saved at this size.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int S, int KG>
__global__ __launch_bounds__(CT * KG, 2) void conv_split_pair_kernel(ConvArgs a0, ConvArgs a1, int n0) {
    if ((int)blockIdx.x < n0) conv_split_body<NT, S, KG>(a0, (int)blockIdx.x);
    else conv_split_body<NT, S, KG>(a1, (int)blockIdx.x - n0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct-activation variant of the generic kernel (round 3) for the shapes the halo kernels do not take -- 1x1 and stride-2 filters:
// the generic kernel stages the 128-pixel activation tile of EVERY (tap, channel block) through LDS next to the weight tile (16 + 4 NT KB
// per k-tile) and is bound by the ~25 B/clk a CU moves global -> LDS (the stride-2 3x3 of encoder layer2 ran 98 us for 13 us of
// matrix work; the 1x1 convc1 of the update block 15 us for 3.4).  But the activation fragment of a lane is PRIVATE to its wave (a wave
// owns 32 pixels): here it never touches LDS.  Lane (pixel, k half) loads its four 16-B fragments (hi / lo x two 16-deep steps) of a
// k-tile straight from global memory into registers -- consecutive pixels are consecutive 64-B rows, so a wave instruction reads
// 32 rows x 32 B; padding / out-of-image taps are out-of-range buffer offsets (zeros) -- two k-tiles ahead; only the weight tile
// (shared by the waves) goes through a 3-stage LDS-DMA ring.  One counted vmcnt + one barrier per k-tile cover both streams.
//   KG = 2: 8 waves, the two groups alternate k-tiles and are combined in the shared epilogue (small grids: two waves per SIMD).
//   TR: transposed accumulators D[pixel][channel] + the direct fp32 epilogue (fp32 + statistics outputs; KG = 1 only).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int KG, bool TR, int NSTG>   // NSTG: k-tiles in flight + 1 (3 .. 6)
__global__ __launch_bounds__(CT * KG, 2) void conv_direct_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!(TR && KG > 1), "the direct epilogue has no k-group combine");
    constexpr int BN = 32 * NT;
    constexpr int STAGE = 2 * NT * 2048;                  // weight tile: BN rows x 64 B x (hi, lo)
    constexpr int NLW = NT;                               // weight DMA pieces (1 KB) per wave per k-tile: 4 NT pieces over 4 waves
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & 3, grp = wave_all >> 2;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int HoWo = a.Ho * a.Wo;
    int m0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        m0 = mt * CBM;
        n0 = (slot - (slot / ntn) * ntn) * BN;
        if (m0 >= HoWo) return;
    }
    const int ntaps = a.KH * a.KW;
    const int nk = ntaps * a.CB;
    char* const ring = lds + grp * (NSTG * STAGE);

    // ---- activation side: this lane's pixel, its byte offset for tap (0, 0) of channel block 0 and the mask of valid taps
    int aoffb;
    unsigned long long vmask = 0;
    {
        const int m = m0 + wave * 32 + l31;
        const bool rok = m < HoWo;
        const int ho = m / a.Wo, wo = m - ho * a.Wo;
        const int hb = ho * a.stride - a.pad_h, wb = wo * a.stride - a.pad_w;
        aoffb = ((hb * a.W + wb) * 32 + kh * 8) * 2;
        for (int t = 0; t < ntaps; ++t) {
            const int r = t / a.KW, q = t - r * a.KW;
            const int hi_ = hb + r, wi_ = wb + q;
            if (rok && hi_ >= 0 && hi_ < a.H && wi_ >= 0 && wi_ < a.W) vmask |= 1ull << t;
        }
    }
    // ---- weight side: piece j of this wave: plane (wave * NLW + j) / (2 NT), unit (16 rows) (wave * NLW + j) % (2 NT)
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    unsigned wvo[NLW];
#pragma unroll
    for (int j = 0; j < NLW; ++j) {
        int r = n0 + ((wave * NLW + j) % (2 * NT)) * 16 + urow;
        r = r < a.cout_pad ? r : a.cout_pad - 1;
        wvo[j] = (unsigned)((r * 32 + uchunk) * 2);
    }
    const int CB2 = a.CB - a.CB1;
    const int plane_b = a.P_in * 64;
    const rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xh + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_l1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xl + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_h2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2h + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const rsrc_t r_l2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2l + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, nk * wtile_b, 0x00020000);
    const rsrc_t r_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.wl, 0, nk * wtile_b, 0x00020000);

    // running (uniform) position of the k-tile to be issued next (as in conv_split_kernel)
    int it = 0, iq = 0, icb = grp, itoff = 0, ik = grp;
    bool ivalid = true;
    auto settle = [&]() {
        while (icb >= a.CB) {
            icb -= a.CB; ++it; ++iq; itoff += 64;
            if (iq == a.KW) { iq = 0; itoff += (a.W - a.KW) * 64; }
        }
        if (it >= ntaps) { ivalid = false; it = 0; iq = 0; icb = 0; itoff = 0; ik = 0; }
    };
    settle();
    auto advance = [&]() { icb += KG; ik += KG; if (ivalid) settle(); else { icb = 0; ik = 0; } };

    half8 xh[NSTG][2], xl[NSTG][2];       // activation fragments of three k-tiles (consumed / landing / just requested)
#define DIRECT_ISSUE(SLOT)                                                                                               \
    {                                                                                                                    \
        char* sb = ring + (SLOT) * STAGE;                                                                                \
        const int wso_ = ik * wtile_b;                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NLW; ++j) {                                                                \
            const int pc_ = wave * NLW + j;                                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pc_ / (2 * NT) ? r_wl : r_wh, (lptr_t)(sb + pc_ * 1024), 16, wvo[j], wso_, 0, 0); \
        }                                                                                                                \
        const bool first_ = icb < a.CB1;                                                                                 \
        const rsrc_t rh_ = first_ ? r_h1 : r_h2;                                                                         \
        const rsrc_t rl_ = first_ ? r_l1 : r_l2;                                                                         \
        const bool ok_ = ivalid && ((vmask >> it) & 1ull);                                                               \
        const unsigned vo_ = ok_ ? (unsigned)(aoffb + (first_ ? icb : icb - a.CB1) * plane_b + itoff) : 0x80000000u;     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
            xh[SLOT][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rh_, ok_ ? vo_ + ks * 32 : vo_, 0, 0)); \
            xl[SLOT][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rl_, ok_ ? vo_ + ks * 32 : vo_, 0, 0)); \
        }                                                                                                                \
        advance();                                                                                                       \
    }

    f32x16 hh[NT], xx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hh[n][r] = 0.f;
            xx[n][r] = 0.f;
        }

    static_for<0, NSTG - 1>([&](auto u) __attribute__((always_inline)) { DIRECT_ISSUE(decltype(u)::value) });
    const int sw = (l31 >> 2) & 3;
    const int nsteps = (nk + KG - 1) / KG;
    // one k-tile: everything of the issue group NSTG - 1 steps back (its weight pieces AND its activation fragments) has landed when at
    // most NSTG - 2 groups (NLW + 4 operations each) are still in flight; the barrier publishes the weight tile and frees the stage read
    // one step ago
#define DIRECT_STEP(U)                                                                                                   \
    if (kt0 + (U) < nsteps) {                                                                                            \
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((NSTG - 2) * (NLW + 4)) : "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        DIRECT_ISSUE(((U) + NSTG - 1) % NSTG)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        const char* wt = ring + (U) * STAGE;                                                                             \
        half8 wh[2][NT], wl[2][NT];                                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
            const int co = ((ks * 2 + kh) ^ sw) * 16;                                                                    \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                             \
                const int wo = (n * 32 + l31) * 64 + co;                                                                 \
                wh[ks][n] = *reinterpret_cast<const half8*>(wt + wo);                                                    \
                wl[ks][n] = *reinterpret_cast<const half8*>(wt + NT * 2048 + wo);                                        \
            }                                                                                                            \
        }                                                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
            if constexpr (TR) {                                                                                          \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[U][ks], wh[ks][n], hh[n], 0, 0, 0); \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[U][ks], wl[ks][n], xx[n], 0, 0, 0); \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[U][ks], wh[ks][n], xx[n], 0, 0, 0); \
            } else {                                                                                                     \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks][n], xh[U][ks], hh[n], 0, 0, 0); \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks][n], xh[U][ks], xx[n], 0, 0, 0); \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks][n], xl[U][ks], xx[n], 0, 0, 0); \
            }                                                                                                            \
        }                                                                                                                \
    }
    for (int kt0 = 0; kt0 < nsteps; kt0 += NSTG) {
        static_for<0, NSTG>([&](auto u) __attribute__((always_inline)) { DIRECT_STEP(decltype(u)::value) });
    }
#undef DIRECT_STEP
#undef DIRECT_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // all LDS-DMA traffic has landed: the ring can be reused by the epilogue

    const int mw = m0 + wave * 32;
    if constexpr (TR) {
        const int kh_ = lane >> 5, c4 = (lane & 31) * 4;
        conv_epilogue_direct<NT>(a, hh, xx, b, [=](int r) {
            const int m = mw + (r & 3) + 8 * (r >> 2) + 4 * kh_;
            return m < HoWo ? (unsigned)(m * 128 + c4) : 0x80000000u; }, n0, lane, wave, tid, reinterpret_cast<float*>(lds));
    } else {
        conv_epilogue<NT, 4 * KG, KG>(a, hh, xx, b, [=](int row) { return mw + row < HoWo ? mw + row : -1; }, n0, lane, wave, tid, grp == 0,
                                      reinterpret_cast<float*>(lds), grp);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Halo variant for stride-1 "same" convolutions (3x3, 1x5, 5x1): the kernels above are bound by the global->LDS fill rate of
// a CU (~25 B/clk measured, L2-resident data), not by the matrix cores, and the generic kernel re-stages the activation tile
// for EVERY tap.  Here a workgroup owns an 8 x 16 pixel patch; per 32-channel block its (8+KH-1) x (16+KW-1) halo patch is
// staged ONCE (out-of-image pixels are out-of-range buffer offsets = zeros, so there are no tap masks at all) and every tap
// reads it at a constant row offset r*(16+KW-1) + q.  Only the weight tile is streamed per tap (4-slot ring), so the bytes
// staged per MFMA drop ~2.3x (3x3, 64-channel tile) and 32-channel tiles (2x the workgroups for the small batch-1 grids of
// the update block) become affordable.
//   k-step = (channel block, tap), taps unrolled: the s_waitcnt immediates are compile-time functions of the tap.
//   LDS: 2 halo buffers (12 units x 1 KB x 2 planes each) + 4 weight slots (NT x 4 KB) = 64 / 80 KB -> 2 workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
// Pixel of tile column n (= MFMA column = lane & 31) inside a wave's 2 x 16 pixel slab.  ds_read_b128 is serviced in four groups of 16
// lanes, {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32 for the upper half; MI355X_MICROARCH.md, LDS): with the obvious
// row = n >> 4, col = n & 15 a group mixes columns {0-3, 12-15} of one patch row with {4-11} of the next, and the 18/20-row
// halo pitch then puts two pairs of lanes on the same 16-B bank slot (measured: 25 % of the LDS cycles were conflicts).  Here every
// group gets the 16 CONSECUTIVE halo rows of one patch row, which the XOR swizzle spreads over all 16 slots for any pitch / tap.
__device__ __forceinline__ int slab_row(int n) { return __builtin_popcount((unsigned)(n >> 2) & 7u) & 1; }
__device__ __forceinline__ int slab_col(int n) { return ((n >> 3) & 3) * 4 + (n & 3); }


// (A 16 x 16 patch / 8-wave / up-to-128-channel variant -- half the weight bytes per MFMA -- was measured 10-25 % SLOWER on
// every encoder and batch-8 shape: eight waves that meet at one barrier per tap serialise more than two independent 4-wave
// workgroups per CU do.  NW stays a constant so that the index arithmetic below reads generally.)
// NIN: the input is the previous convolution's PRE-NORMALISATION fp32 output + its InstanceNorm statistics (ConvArgs.xraw / xstats): the
// halo patch of a channel block is loaded into registers (float4 = 4 channels of a halo row per thread), turned into
// relu((x - mean) * rstd) with the coefficients the normalisation kernel would use, split into hi / lo and written to the halo buffer
// by ds_write -- the same LDS image the LDS-DMA path produces from a normalised split tensor, so everything downstream is unchanged and
// the result is bit-identical to "normalise, then convolve", without the normalisation pass over the activation (38 us per 64-channel
// half-resolution map of the 5 event windows).
// (Round 4's HT instantiation -- the workgroups of a half-empty last channel tile, Cout % 64 in (0, 32], skip the empty half's fragment reads and
// MFMAs: a quarter of the 96-channel layers' matrix work -- was bit-identical and neutral in three alternating pairs
// (profiles/r04_half_tile_ab.txt); removed in round 5.)
template <int NT, int KH, int KW, bool TR = false, bool NIN = false>    // TR: transposed accumulators D[pixel][channel] + the direct epilogue
__global__ __launch_bounds__(CT, 2) void conv_halo_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 4;
    constexpr int TH = 2 * NW, TW = 16;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);     // halo patch: rows of 64 B per plane
    constexpr int A_UNITS = ((HR + 15) / 16 + NW / 2 - 1) / (NW / 2) * (NW / 2);   // 16-row LDS-DMA units per plane, a multiple of NW/2
    constexpr int AP = A_UNITS / (NW / 2);                          // pieces per wave per channel block (plane = wave parity)
    constexpr int A_PLANE = A_UNITS * 1024, A_BUF = 2 * A_PLANE;
    constexpr int NTAPS = KH * KW;
    constexpr int B_PLANE = NT * 2048, B_SLOT = 2 * B_PLANE;        // NT*32 weight rows x 64 B per plane
    constexpr int NBP = 4 * NT / NW;                                // weight pieces per wave per tap
    static_assert((4 * NT) % NW == 0, "weight tile must split evenly over the waves");
    constexpr int SB = 4, LA = 2;                                   // weight ring slots, look-ahead in taps
    static_assert(NTAPS >= 4, "the halo of the next block must land within a block");
    constexpr int O_B = 2 * A_BUF;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    // (de-phasing the co-resident workgroups of the first round by a pseudo-random start delay of up to 8 / 15 / 30 k cycles was measured:
    // 98.7 -> 98.9 / 101.7 / 109.3 us on the layer-1 launch -- phase alignment is not what the two workgroups of a CU lose time to)
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    int y0, x0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int mt = (slot / ntn) * 8 + xcd;                      // all channel tiles of a patch on one XCD, back to back
        if (mt >= tiles_x * tiles_y) return;
        n0 = (slot - (slot / ntn) * ntn) * 32 * NT;
        const int ty = mt / tiles_x;
        y0 = ty * TH;
        x0 = (mt - ty * tiles_x) * TW;
    }

    // ---- LDS-DMA sources ------------------------------------------------------------------------------------------
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;        // halves; LDS chunk = logical chunk ^ ((row >> 2) & 3)
    // halo: wave w stages plane (w & 1), units (w >> 1) + (NW/2) i
    unsigned aoff[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = ((wave >> 1) + (NW / 2) * i) * 16 + urow;
        const int hy = row / HWD, hx = row - hy * HWD;
        const int py = y0 - a.pad_h + hy, px = x0 - a.pad_w + hx;
        const bool ok = row < HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
        aoff[i] = ok ? (unsigned)(((py * a.W + px) * 32 + uchunk) * 2) : 0x80000000u;
    }
    // weights: piece p = w*NBP + j of a tap's 4 NT pieces: plane p / (2 NT) = w / (NW/2), unit p % (2 NT)
    unsigned wvo[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const int r = n0 + ((wave * NBP + j) % (2 * NT)) * 16 + urow;    // < cout_pad (a multiple of 128)
        wvo[j] = (unsigned)((r * 32 + uchunk) * 2);
    }
    const int CB2 = a.CB - a.CB1;
    const int plane_b = a.P_in * 64;
    const bool lo_a = wave & 1, lo_w = wave / (NW / 2);
    const rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_a ? a.xl : a.xh) + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_a ? a.x2l : a.x2h) + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(lo_w ? a.wl : a.wh), 0, NTAPS * a.CB * wtile_b, 0x00020000);
    char* const a_dst = lds + (lo_a ? A_PLANE : 0) + (wave >> 1) * 1024;
    char* const w_dst = lds + O_B + (lo_w ? B_PLANE : 0);

#define HALO_ISSUE_A(CBI, BUF)                                                                                           \
    {                                                                                                                    \
        const int cbi_ = (CBI) < a.CB ? (CBI) : a.CB - 1;             /* past the last block: a harmless re-load */      \
        const bool first_ = cbi_ < a.CB1;                                                                                \
        const rsrc_t ra_ = first_ ? r_a1 : r_a2;                                                                         \
        const int so_ = (first_ ? cbi_ : cbi_ - a.CB1) * plane_b;                                                        \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lptr_t)(a_dst + (BUF) * A_BUF + i * (NW / 2) * 1024), 16, aoff[i], so_, 0, 0); \
    }
#define HALO_ISSUE_B(CBI, TAP, SLOT)                                                                                     \
    {                                                                                                                    \
        const int so_ = ((TAP) * a.CB + (CBI)) * wtile_b;             /* past the last block: in range, unused */        \
        _Pragma("unroll") for (int j = 0; j < NBP; ++j)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(w_dst + (SLOT) * B_SLOT + ((wave * NBP + j) % (2 * NT)) * 1024), 16, \
                                                     wvo[j], so_, 0, 0);                                                 \
    }

    f32x16 hh[NT], xx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hh[n][r] = 0.f;
            xx[n][r] = 0.f;
        }

    // ---- NIN: thread -> 4 channels (tid & 7) of halo rows (tid >> 3) + 32 i; float4 loads, normalise + ReLU + split, ds_write_b64
    constexpr int NHL = NIN ? (HR + 31) / 32 : 0;                  // halo loads per thread per channel block
    // coefficient tables (mul / add per input channel, <= 128 channels): in the rows HR .. 16 A_UNITS of halo buffer 0 that no tap ever
    // reads (hi plane: mul, lo plane: add) -- one more KB of LDS would cost the second workgroup of the CU
    static_assert(!NIN || (A_UNITS * 16 - HR) * 64 >= 128 * 4, "no room for the coefficient tables");
    float* const nmul = reinterpret_cast<float*>(lds + HR * 64);
    float* const nadd = reinterpret_cast<float*>(lds + A_PLANE + HR * 64);
    unsigned noff[NHL > 0 ? NHL : 1];
    float4 nv[NHL > 0 ? NHL : 1];
    const int ng = tid & 7;
    rsrc_t r_raw = r_a1;
    if constexpr (NIN) {
#pragma unroll
        for (int i = 0; i < NHL; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int hy = row / HWD, hx = row - hy * HWD;
            const int py = y0 - a.pad_h + hy, px = x0 - a.pad_w + hx;
            const bool ok = row < HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
            noff[i] = ok ? (unsigned)(((py * a.W + px) * 32 + ng * 4) * 4) : 0x80000000u;
        }
        r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xraw + (long long)b * a.CB * a.P_in * 32), 0, a.CB * a.P_in * 128, 0x00020000);
        for (int c = tid; c < a.CB * 32; c += CT) {                // coefficients of every input channel of image b, once per workgroup
            float m_, a_;
            norm_coeffs(a.xstats, nullptr, nullptr, b, c, a.CB * 32, a.H * a.W, a.xeps, m_, a_, a.xstats_reps, (long long)gridDim.z * a.CB * 32 * 2);
            nmul[c] = m_;
            nadd[c] = a_;
        }
    }
#define NIN_LOAD(CBI)                                                                                                    \
    {                                                                                                                    \
        const int cbi_ = (CBI) < a.CB ? (CBI) : a.CB - 1;                                                                \
        _Pragma("unroll") for (int i = 0; i < NHL; ++i)                                                                  \
            nv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_raw, noff[i], cbi_ * a.P_in * 128, 0)); \
    }
#define NIN_WRITE(CBI, BUF, I0, I1)                                                                                      \
    {                                                                                                                    \
        const int cbi_ = (CBI) < a.CB ? (CBI) : a.CB - 1;                                                                \
        const float4 m4 = *reinterpret_cast<const float4*>(nmul + cbi_ * 32 + ng * 4);                                   \
        const float4 a4 = *reinterpret_cast<const float4*>(nadd + cbi_ * 32 + ng * 4);                                   \
        _Pragma("unroll") for (int i = (I0); i < (I1) && i < NHL; ++i) {                                                 \
            const int row = (tid >> 3) + 32 * i;                                                                         \
            if (row < HR) {                                                                                              \
                const bool in_ = noff[i] != 0x80000000u;          /* zero padding applies to the NORMALISED activation */  \
                const float v_[4] = {in_ ? fmaxf(fmaf(nv[i].x, m4.x, a4.x), 0.f) : 0.f, in_ ? fmaxf(fmaf(nv[i].y, m4.y, a4.y), 0.f) : 0.f, \
                                     in_ ? fmaxf(fmaf(nv[i].z, m4.z, a4.z), 0.f) : 0.f, in_ ? fmaxf(fmaf(nv[i].w, m4.w, a4.w), 0.f) : 0.f}; \
                half4v h4_, l4_;                                                                                         \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                          \
                    _Float16 x1_, x2_;                                                                                   \
                    split1(v_[k], x1_, x2_);                                                                             \
                    h4_[k] = x1_;                                                                                        \
                    l4_[k] = x2_;                                                                                        \
                }                                                                                                        \
                char* d_ = lds + (BUF) * A_BUF + row * 64 + ((((ng >> 1) ^ ((row >> 2) & 3))) << 4) + ((ng & 1) << 3);    \
                *reinterpret_cast<half4v*>(d_) = h4_;                                                                    \
                *reinterpret_cast<half4v*>(d_ + A_PLANE) = l4_;                                                          \
            }                                                                                                            \
        }                                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        /* visible to the other waves at the next barrier */   \
    }

    if constexpr (NIN) {
        NIN_LOAD(0)
    } else {
        HALO_ISSUE_A(0, 0)
    }
#pragma unroll
    for (int t = 0; t <= LA; ++t) HALO_ISSUE_B(0, t, t)
    if constexpr (NIN) {
        __syncthreads();                                           // coefficient table complete
        NIN_WRITE(0, 0, 0, NHL)
    }

    // this lane's pixel inside the patch (MFMA B-operand row = pixel) and its halo row for tap (0, 0)
    const int R0 = (wave * 2 + slab_row(l31)) * HWD + slab_col(l31);
    const int wsw = (l31 >> 2) & 3;
    // Register pipelining: the fragments of tap s+1 are read from LDS while the MFMAs of tap s run (one wave per SIMD and
    // workgroup: nothing else hides the ds_read latency).  `cur` is consumed, `nxt` is in flight.
    half8 cxh[2], cxl[2], cwh[2][NT], cwl[2][NT], nxh[2], nxl[2], nwh[2][NT], nwl[2][NT];
#define HALO_READ(XH, XL, WH, WL, ABUF, WSLOT, TAP)                                                                      \
    {                                                                                                                    \
        const int R_ = R0 + ((TAP) / KW) * HWD + ((TAP) % KW);                                                           \
        const int sw_ = (R_ >> 2) & 3;                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
            const int ao_ = R_ * 64 + (((ks * 2 + kh) ^ sw_) * 16);                                                      \
            XH[ks] = *reinterpret_cast<const half8*>((ABUF) + ao_);                                                      \
            XL[ks] = *reinterpret_cast<const half8*>((ABUF) + A_PLANE + ao_);                                            \
            const int co_ = ((ks * 2 + kh) ^ wsw) * 16;                                                                  \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                             \
                const int wo_ = (n * 32 + l31) * 64 + co_;                                                               \
                WH[ks][n] = *reinterpret_cast<const half8*>((WSLOT) + wo_);                                              \
                WL[ks][n] = *reinterpret_cast<const half8*>((WSLOT) + B_PLANE + wo_);                                    \
            }                                                                                                            \
        }                                                                                                                \
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA * NBP) : "memory");      // halo 0 and weight tile 0 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    HALO_READ(cxh, cxl, cwh, cwl, lds, lds + O_B, 0)
    int islot = LA + 1, rslot = 1;              // ring slot to fill / to read next
    for (int cb = 0; cb < a.CB; ++cb) {
        const char* abuf = lds + (cb & 1) * A_BUF;
        const char* abuf_next = lds + ((cb + 1) & 1) * A_BUF;
        const bool last_cb = cb + 1 == a.CB;
        static_for<0, NTAPS>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            // Step s = (cb, t) consumes the fragments read one step ago and READS tap s+1, whose weight tile was issued at step
            // s-2 (after that step's halo issue, if any).  Issued since: the tile of step s-1 and, when s-1 was tap 1, the next
            // block's halo.  A slot / halo buffer is refilled two barriers after its last reader (hipcc lets fragment reads
            // complete after the next barrier): ring of 4 = {being drained, being read, 2 in flight}.
            if (t == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + (NIN ? NHL : AP)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NIN) {
                if (t == 1) NIN_LOAD(cb + 1)
                // the loads have landed behind the counted wait of tap 3; buffer (cb + 1) & 1 was last read two barriers before tap 1.  The
                // conversion is spread over taps 3 .. NTAPS - 2 (its ~40 VALU instructions per item sit between the taps' MFMAs instead
                // of stalling all eight waves of the CU at one tap); the buffer is read from tap NTAPS - 1 on
                if constexpr (NTAPS >= 6) {
                    constexpr int slots = NTAPS - 4;                             // taps 3 .. NTAPS - 2
                    constexpr int per = (NHL + slots - 1) / slots;
                    if (t >= 3 && t <= NTAPS - 2) NIN_WRITE(cb + 1, (cb + 1) & 1, (t - 3) * per, (t - 2) * per)
                } else {
                    if (t == 3) NIN_WRITE(cb + 1, (cb + 1) & 1, 0, NHL)
                }
            } else {
                if (t == 1) HALO_ISSUE_A(cb + 1, (cb + 1) & 1)
            }
            {
                constexpr int tn = (t + LA + 1) % NTAPS;
                const int cbn = cb + (t + LA + 1) / NTAPS;
                HALO_ISSUE_B(cbn, tn, islot)
            }
            if (++islot == SB) islot = 0;
            __builtin_amdgcn_sched_barrier(0);
            const char* wnext = lds + O_B + rslot * B_SLOT;
            if (++rslot == SB) rslot = 0;
            if (t + 1 < NTAPS) HALO_READ(nxh, nxl, nwh, nwl, abuf, wnext, (t + 1) % NTAPS)
            else if (!last_cb) HALO_READ(nxh, nxl, nwh, nwl, abuf_next, wnext, 0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (TR) {        // D[pixel][channel]
#pragma unroll
                    for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cxh[ks], cwh[ks][n], hh[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cxh[ks], cwl[ks][n], xx[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cxl[ks], cwh[ks][n], xx[n], 0, 0, 0);
                    // (a timing-only build with 8 instead of 12 MFMA-times per tap + the 16 fp16 -> fp8 conversions the fp8 cross terms of K5 would
                    // need here measured 98 -> 94.5 us on the layer-1 launch and -1.7 % on the frame: not worth the parity margin of every layer)
                } else {
#pragma unroll
                    for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cwh[ks][n], cxh[ks], hh[n], 0, 0, 0);   // D[channel][pixel]
#pragma unroll
                    for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cwl[ks][n], cxh[ks], xx[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NT; ++n) xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cwh[ks][n], cxl[ks], xx[n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                cxh[ks] = nxh[ks];
                cxl[ks] = nxl[ks];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    cwh[ks][n] = nwh[ks][n];
                    cwl[ks][n] = nwl[ks][n];
                }
            }
        });
    }
#undef HALO_READ
#undef HALO_ISSUE_A
#undef HALO_ISSUE_B
#undef NIN_LOAD
#undef NIN_WRITE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int W = a.W, H = a.H, yw = y0 + wave * 2;
    if constexpr (TR) {
        // register r of lane (channel, kh): pixel column x0 + r, patch row (popcount(r >> 2) + kh) & 1 of the wave's 2 x 16 slab
        const int kh_ = lane >> 5, c4 = (lane & 31) * 4;
        conv_epilogue_direct<NT>(a, hh, xx, b, [=](int r) {
            const int y = yw + ((__builtin_popcount((unsigned)(r >> 2)) + kh_) & 1), x = x0 + r;
            return (y < H && x < W) ? (unsigned)((y * W + x) * 128 + c4) : 0x80000000u; }, n0, lane, wave, tid, reinterpret_cast<float*>(lds));
    } else {
        conv_epilogue<NT>(a, hh, xx, b, [=](int row) {
            const int y = yw + slab_row(row), x = x0 + slab_col(row);
            return (y < H && x < W) ? y * W + x : -1; }, n0, lane, wave, tid, true, reinterpret_cast<float*>(lds));
    }
#endif
}

#include "conv_stream.h"

// ---------------------------------------------------------------------------------------------------------------------
// Small-grid halo variant (batch-1 update block: 40 patches x Cout/32 workgroups <= one per CU).  With one 4-wave workgroup
// per CU every SIMD holds a single wave and each (wait, barrier, fragment reads, 6 MFMAs) step is a serial latency chain
// (~400 cycles for 192 cycles of matrix work, measured).  Here
//   * 8 waves: wave (s, g) owns pixel slab s like above but only k-half g of every 32-channel block (the two 16-deep MFMA
//     sub-steps are split across the groups, which share the staged halo AND the weight tiles); two waves per SIMD
//     interleave, the partial sums are combined through LDS at the end;
//   * a step is 3 taps (a filter row of a 3x3; 3 + 2 taps of a 1x5 / 5x1): ~3x fewer barriers, LDS 64-72 KB so that two
//     workgroups fit a CU (the 320 workgroups of the 256-channel z|r convolution then run in one round);
//   * 32-channel tile only (NT = 1), 2 weight slots; the fragment reads of a step are pinned before the next barrier
//     (sched_barrier), so a slot / halo buffer may be refilled one barrier after its last reader;
//   * the 8 waves split every stage into whole 1-KB pieces plus, when the count is not a multiple of 8, one half piece
//     (lanes 0-31 or 32-63 active).
// ---------------------------------------------------------------------------------------------------------------------
template <int KH, int KW>
__device__ __forceinline__ void conv_halo8_body(const ConvArgs& a, const int bx) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 8, TW = 16;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);
    constexpr int A_UNITS = (HR + 15) / 16;
    constexpr int A_PLANE = A_UNITS * 1024, A_BUF = 2 * A_PLANE;
    constexpr int NTAPS = KH * KW;
    constexpr int TPS = 3;                                          // taps per step (the last step of a 5-tap filter has 2)
    constexpr int NST = (NTAPS + TPS - 1) / TPS;                    // steps per channel block (3 or 2)
    constexpr int B_TAP = 4096, B_SLOT = TPS * B_TAP;               // per tap: 32 weight rows x 64 B x 2 planes
    constexpr int O_B = 2 * A_BUF;
    // wave (q, plane): plane = wave_all & 1 (hi / lo), q = wave_all >> 1.  Per plane a halo buffer is A_UNITS 1-KB pieces and a
    // weight slot 2*TPS (tap-major, 2 units per tap); wave q takes pieces q + 4 i, and when the count is 4 k + 2 the last two
    // pieces are split into half pieces (q >> 1), half (q & 1): lanes 0-31 or 32-63 active.
    constexpr int NFA = A_UNITS / 4;                                // whole pieces per wave
    constexpr bool HALF_A = (A_UNITS % 4) != 0;
    static_assert(A_UNITS % 2 == 0, "stages must split into whole + half pieces over 4 waves per plane");
    constexpr int NIA = NFA + (HALF_A ? 1 : 0);                     // halo load instructions per wave per channel block
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & 3, grp = wave_all >> 2;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    int y0, x0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = bx & 7, slot = bx >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        if (mt >= tiles_x * tiles_y) return;
        n0 = (slot - (slot / ntn) * ntn) * 32;
        const int ty = mt / tiles_x;
        y0 = ty * TH;
        x0 = (mt - ty * tiles_x) * TW;
    }

    // ---- LDS-DMA sources ------------------------------------------------------------------------------------------
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const int q = wave_all >> 1;
    const bool lo_p = wave_all & 1;
    const bool half_on = (lane >> 5) == (q & 1);                    // lanes of this wave's half piece
    unsigned aoff[NIA];
    int a_unit[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int unit = (i < NFA) ? q + 4 * i : 4 * NFA + (q >> 1);
        a_unit[i] = unit;
        const int row = unit * 16 + urow;
        const int hy = row / HWD, hx = row - hy * HWD;
        const int py = y0 - a.pad_h + hy, px = x0 - a.pad_w + hx;
        const bool ok = row < HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
        aoff[i] = ok ? (unsigned)(((py * a.W + px) * 32 + uchunk) * 2) : 0x80000000u;
    }
    const unsigned wvo0 = (unsigned)(((n0 + urow) * 32 + uchunk) * 2), wvo1 = wvo0 + 16 * 64;   // weight rows of unit 0 / 1
    const int CB2 = a.CB - a.CB1;
    const int plane_b = a.P_in * 64;
    const rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_p ? a.xl : a.xh) + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_p ? a.x2l : a.x2h) + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(lo_p ? a.wl : a.wh), 0, NTAPS * a.CB * wtile_b, 0x00020000);
    char* const a_dst = lds + (lo_p ? A_PLANE : 0);
    char* const w_dst = lds + O_B + (lo_p ? 2048 : 0);

#define H8_ISSUE_A(CBI, BUF, SEL, NSEL)                                                                                  \
    {                                                                                                                    \
        const int cbi_ = (CBI) < a.CB ? (CBI) : a.CB - 1;                                                                \
        const bool first_ = cbi_ < a.CB1;                                                                                \
        const rsrc_t ra_ = first_ ? r_a1 : r_a2;                                                                         \
        const int so_ = (first_ ? cbi_ : cbi_ - a.CB1) * plane_b;                                                        \
        _Pragma("unroll") for (int i = 0; i < NIA; ++i)                                                                  \
            if (i % (NSEL) == (SEL) && (i < NFA || half_on))                                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lptr_t)(a_dst + (BUF) * A_BUF + a_unit[i] * 1024), 16, aoff[i], so_, 0, 0); \
    }
    // weight slot for step (CBI, ST): taps ST*TPS .. ; piece idx -> tap idx / 2, unit idx & 1.  ST is a compile-time constant.
#define H8_ISSUE_B(CBI, ST, SLOT, SEL, NSEL)                                                                             \
    {                                                                                                                    \
        constexpr int ntp_ = (NTAPS - (ST) * TPS) < TPS ? (NTAPS - (ST) * TPS) : TPS;      /* taps of this step */       \
        constexpr int nfb_ = (2 * ntp_) / 4;                                                                             \
        constexpr bool halfb_ = ((2 * ntp_) % 4) != 0;                                                                   \
        _Pragma("unroll") for (int i = 0; i < nfb_ + (halfb_ ? 1 : 0); ++i) {                                            \
            const int idx_ = (i < nfb_) ? q + 4 * i : 4 * nfb_ + (q >> 1);                                               \
            const int so_ = (((ST) * TPS + (idx_ >> 1)) * a.CB + (CBI)) * wtile_b;   /* past the end: out of range, never consumed */ \
            if (i % (NSEL) == (SEL) && (i < nfb_ || half_on))                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(w_dst + (SLOT) * B_SLOT + (idx_ >> 1) * B_TAP + (idx_ & 1) * 1024), 16, \
                                                         (idx_ & 1) ? wvo1 : wvo0, so_, 0, 0);                           \
        }                                                                                                                \
    }

    f32x16 hh[1], x1, x2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        hh[0][r] = 0.f;
        x1[r] = 0.f;
        x2[r] = 0.f;
    }

    H8STAMP_RT(14) H8STAMP(0)
    H8_ISSUE_A(0, 0, 0, 1)
    H8_ISSUE_B(0, 0, 0, 0, 1)
    H8STAMP(1)

    const int R0 = (wave * 2 + slab_row(l31)) * HWD + slab_col(l31);
    const int kq = grp * 2 + kh;                                    // this lane's 16-B k-chunk of the 64-B row
    const int wro = l31 * 64 + ((kq ^ ((l31 >> 2) & 3)) * 16);      // weight fragment offset inside a (tap, plane) tile
    int cur = 0;
    for (int cb = 0; cb < a.CB; ++cb) {
        const char* abuf = lds + (cb & 1) * A_BUF;
        static_for<0, NTAPS>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            constexpr int st = t / TPS, j = t % TPS;
            if (j == 0) {                          // ---- step boundary
                // in flight, oldest first: [halo of block cb+1 (issued at st 0, AFTER that step's weights)], weights of this step
                if (!CONV_SPREAD_DMA && NST > 1 && st == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (cb == 0 && st == 0) { H8STAMP(2) }
                if (cb == 1 && st == 0) { H8STAMP(3) }
                if (!CONV_SPREAD_DMA) {
                    constexpr int stn = (st + 1) % NST;
                    const int cbn = cb + (st + 1) / NST;
                    H8_ISSUE_B(cbn, stn, cur ^ 1, 0, 1)
                    if (st == 0) H8_ISSUE_A(cb + 1, (cb + 1) & 1, 0, 1)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (CONV_SPREAD_DMA) {   // A/B build: the step's LDS-DMA pieces dealt over its taps (issued in front of each tap's fragment reads) instead of one burst
                constexpr int ntc = (NTAPS - st * TPS) < TPS ? (NTAPS - st * TPS) : TPS;
                constexpr int stn = (st + 1) % NST;
                const int cbn = cb + (st + 1) / NST;
                if (st == 0) H8_ISSUE_A(cb + 1, (cb + 1) & 1, j, ntc)
                H8_ISSUE_B(cbn, stn, cur ^ 1, j, ntc)
            }
            const char* wcur = lds + O_B + cur * B_SLOT;
            const int R = R0 + (t / KW) * HWD + (t % KW);
            const int ao = R * 64 + ((kq ^ ((R >> 2) & 3)) * 16);
            const half8 xh = *reinterpret_cast<const half8*>(abuf + ao);
            const half8 xl = *reinterpret_cast<const half8*>(abuf + A_PLANE + ao);
            const half8 wh = *reinterpret_cast<const half8*>(wcur + j * B_TAP + wro);
            const half8 wl = *reinterpret_cast<const half8*>(wcur + j * B_TAP + 2048 + wro);
            hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hh[0], 0, 0, 0);   // D[channel][pixel]
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, x2, 0, 0, 0);
            if (j == TPS - 1 || t == NTAPS - 1) {
                __builtin_amdgcn_sched_barrier(0);   // fragment reads complete before the next barrier releases the refill
                cur ^= 1;
            }
        });
    }
#undef H8_ISSUE_A
#undef H8_ISSUE_B
    H8STAMP(4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    H8STAMP(5)

    // ---- the two k-halves are combined on the read side of the shared epilogue (each group stages its partial sums)
    f32x16 xx[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) xx[0][r] = x1[r] + x2[r];
    const int W = a.W, H = a.H, yw = y0 + wave * 2;
    conv_epilogue<1, 8, 2>(a, hh, xx, b, [=](int row) {
        const int y = yw + slab_row(row), x = x0 + slab_col(row);
        return (y < H && x < W) ? y * W + x : -1; }, n0, lane, wave, tid, grp == 0, reinterpret_cast<float*>(lds), grp);
#ifdef H8_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    H8STAMP(6) H8STAMP_RT(15)
#endif
}

template <int KH, int KW>
__global__ __launch_bounds__(2 * CT, 2) void conv_halo8_kernel(ConvArgs a) {
    conv_halo8_body<KH, KW>(a, (int)blockIdx.x);
}

template <int KH, int KW>     // two independent convolutions as one grid (see conv_split_pair_kernel)
__global__ __launch_bounds__(2 * CT, 2) void conv_halo8_pair_kernel(ConvArgs a0, ConvArgs a1, int n0) {
    if ((int)blockIdx.x < n0) conv_halo8_body<KH, KW>(a0, (int)blockIdx.x);
    else conv_halo8_body<KH, KW>(a1, (int)blockIdx.x - n0);
}

// ---------------------------------------------------------------------------------------------------------------------
// 10-wave variant of the small-grid kernel: a 10 x 16 pixel patch = 5 pixel slabs x 2 k-groups.  On the DSEC-size grid (60 x 80) the 8 x 16
// patches of conv_halo8_kernel leave the last patch row half empty (60 = 7.5 x 8) and a 256-channel convolution (z|r of the GRU, the first
// head convolution) becomes 40 x 8 = 320 workgroups on 256 CUs: 64 CUs carry two and set the launch time (22 us against 15.6 us for
// the 160-workgroup q convolution of the same depth).  10 x 16 patches tile 60 x 80 exactly: 30 x 8 = 240 workgroups, one per CU, 25 %
// more pixels each at the same weight traffic.  Same stages, rings and epilogue as above; the LDS-DMA pieces are dealt over 5 waves
// per plane with dummy pieces so that every wave issues the same count.
// ---------------------------------------------------------------------------------------------------------------------
template <int KH, int KW, int NSL>                                  // NSL = pixel slabs of 2 x 16 pixels = waves per k-group: 5 (or 3, see the dispatch)
__global__ __launch_bounds__(128 * NSL, 1) void conv_halo10_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 2 * NSL, TW = 16;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);
    constexpr int A_UNITS = (HR + 15) / 16;
    constexpr int A_PLANE = A_UNITS * 1024, A_BUF = 2 * A_PLANE;
    constexpr int NTAPS = KH * KW;
    constexpr int TPS = 3;                                          // taps per step (the last step of a 5-tap filter has 2)
    constexpr int NST = (NTAPS + TPS - 1) / TPS;                    // steps per channel block (3 or 2)
    constexpr int B_TAP = 4096, B_SLOT = TPS * B_TAP;               // per tap: 32 weight rows x 64 B x 2 planes
    constexpr int O_B = 2 * A_BUF;
    // wave (q, plane): plane = wave_all & 1 (hi / lo), q = wave_all >> 1 (0 .. 4).  Per plane a halo buffer is A_UNITS 1-KB pieces and a
    // weight slot 2*TPS; wave q takes pieces q + 5 i.  Every wave issues the SAME number of LDS-DMA instructions (the counted vmcnt
    // waits are compile-time): indices past the end are dummy pieces -- out-of-range source offsets (nothing fetched) into a scratch KB.
    constexpr int NIA = (A_UNITS + NSL - 1) / NSL;                  // halo load instructions per wave per channel block
    constexpr int O_SCR = O_B + 2 * B_SLOT;                         // 1 KB scratch behind the weight slots
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all >= NSL ? 1 : 0, wave = wave_all - grp * NSL;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    int y0, x0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        if (mt >= tiles_x * tiles_y) return;
        n0 = (slot - (slot / ntn) * ntn) * 32;
        const int ty = mt / tiles_x;
        y0 = ty * TH;
        x0 = (mt - ty * tiles_x) * TW;
    }

    // ---- LDS-DMA sources ------------------------------------------------------------------------------------------
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const int q = wave_all >> 1;
    const bool lo_p = wave_all & 1;
    unsigned aoff[NIA];
    int a_unit[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int unit = q + NSL * i;
        a_unit[i] = unit;
        const int row = unit * 16 + urow;
        const int hy = row / HWD, hx = row - hy * HWD;
        const int py = y0 - a.pad_h + hy, px = x0 - a.pad_w + hx;
        const bool ok = unit < A_UNITS && row < HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
        aoff[i] = ok ? (unsigned)(((py * a.W + px) * 32 + uchunk) * 2) : 0x80000000u;
    }
    const unsigned wvo0 = (unsigned)(((n0 + urow) * 32 + uchunk) * 2), wvo1 = wvo0 + 16 * 64;   // weight rows of unit 0 / 1
    const int CB2 = a.CB - a.CB1;
    const int plane_b = a.P_in * 64;
    const rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_p ? a.xl : a.xh) + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_p ? a.x2l : a.x2h) + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(lo_p ? a.wl : a.wh), 0, NTAPS * a.CB * wtile_b, 0x00020000);
    char* const a_dst = lds + (lo_p ? A_PLANE : 0);
    char* const w_dst = lds + O_B + (lo_p ? 2048 : 0);

#define H8_ISSUE_A(CBI, BUF, SEL, NSEL)                                                                                  \
    {                                                                                                                    \
        const int cbi_ = (CBI) < a.CB ? (CBI) : a.CB - 1;                                                                \
        const bool first_ = cbi_ < a.CB1;                                                                                \
        const rsrc_t ra_ = first_ ? r_a1 : r_a2;                                                                         \
        const int so_ = (first_ ? cbi_ : cbi_ - a.CB1) * plane_b;                                                        \
        _Pragma("unroll") for (int i = 0; i < NIA; ++i)                                                                  \
            if (i % (NSEL) == (SEL))                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lptr_t)(a_unit[i] < A_UNITS ? a_dst + (BUF) * A_BUF + a_unit[i] * 1024 : lds + O_SCR), 16, \
                                                     aoff[i], so_, 0, 0);                                                \
    }
    // weight slot for step (CBI, ST): taps ST*TPS .. ; piece idx -> tap idx / 2, unit idx & 1.  ST is a compile-time constant.
#define H8_ISSUE_B(CBI, ST, SLOT, SEL, NSEL)                                                                             \
    {                                                                                                                    \
        constexpr int ntp_ = (NTAPS - (ST) * TPS) < TPS ? (NTAPS - (ST) * TPS) : TPS;      /* taps of this step */       \
        constexpr int nib_ = (2 * ntp_ + NSL - 1) / NSL;                                                                 \
        _Pragma("unroll") for (int i = 0; i < nib_; ++i) {                                                               \
            const int idx_ = q + NSL * i;                                                                                \
            const bool real_ = idx_ < 2 * ntp_;                                                                          \
            const int so_ = (((ST) * TPS + (idx_ >> 1)) * a.CB + (CBI)) * wtile_b;   /* past the end: out of range, never consumed */ \
            if (i % (NSEL) == (SEL))                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(real_ ? w_dst + (SLOT) * B_SLOT + (idx_ >> 1) * B_TAP + (idx_ & 1) * 1024 : lds + O_SCR), 16, \
                                                     real_ ? ((idx_ & 1) ? wvo1 : wvo0) : 0x80000000u, so_, 0, 0);       \
        }                                                                                                                \
    }

    f32x16 hh[1], x1, x2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        hh[0][r] = 0.f;
        x1[r] = 0.f;
        x2[r] = 0.f;
    }

    H8_ISSUE_A(0, 0, 0, 1)
    H8_ISSUE_B(0, 0, 0, 0, 1)

    const int R0 = (wave * 2 + slab_row(l31)) * HWD + slab_col(l31);
    const int kq = grp * 2 + kh;                                    // this lane's 16-B k-chunk of the 64-B row
    const int wro = l31 * 64 + ((kq ^ ((l31 >> 2) & 3)) * 16);      // weight fragment offset inside a (tap, plane) tile
    int cur = 0;
    for (int cb = 0; cb < a.CB; ++cb) {
        const char* abuf = lds + (cb & 1) * A_BUF;
        static_for<0, NTAPS>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            constexpr int st = t / TPS, j = t % TPS;
            if (j == 0) {                          // ---- step boundary
                // in flight, oldest first: [halo of block cb+1 (issued at st 0, AFTER that step's weights)], weights of this step
                if (!CONV_SPREAD_DMA && NST > 1 && st == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (!CONV_SPREAD_DMA) {
                    constexpr int stn = (st + 1) % NST;
                    const int cbn = cb + (st + 1) / NST;
                    H8_ISSUE_B(cbn, stn, cur ^ 1, 0, 1)
                    if (st == 0) H8_ISSUE_A(cb + 1, (cb + 1) & 1, 0, 1)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (CONV_SPREAD_DMA) {   // A/B build: the step's LDS-DMA pieces dealt over its taps (issued in front of each tap's fragment reads) instead of one burst
                constexpr int ntc = (NTAPS - st * TPS) < TPS ? (NTAPS - st * TPS) : TPS;
                constexpr int stn = (st + 1) % NST;
                const int cbn = cb + (st + 1) / NST;
                if (st == 0) H8_ISSUE_A(cb + 1, (cb + 1) & 1, j, ntc)
                H8_ISSUE_B(cbn, stn, cur ^ 1, j, ntc)
            }
            const char* wcur = lds + O_B + cur * B_SLOT;
            const int R = R0 + (t / KW) * HWD + (t % KW);
            const int ao = R * 64 + ((kq ^ ((R >> 2) & 3)) * 16);
            const half8 xh = *reinterpret_cast<const half8*>(abuf + ao);
            const half8 xl = *reinterpret_cast<const half8*>(abuf + A_PLANE + ao);
            const half8 wh = *reinterpret_cast<const half8*>(wcur + j * B_TAP + wro);
            const half8 wl = *reinterpret_cast<const half8*>(wcur + j * B_TAP + 2048 + wro);
            hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hh[0], 0, 0, 0);   // D[channel][pixel]
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, x2, 0, 0, 0);
            if (j == TPS - 1 || t == NTAPS - 1) {
                __builtin_amdgcn_sched_barrier(0);   // fragment reads complete before the next barrier releases the refill
                cur ^= 1;
            }
        });
    }
#undef H8_ISSUE_A
#undef H8_ISSUE_B
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- the two k-halves are combined on the read side of the shared epilogue (each group stages its partial sums)
    f32x16 xx[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) xx[0][r] = x1[r] + x2[r];
    const int W = a.W, H = a.H, yw = y0 + wave * 2;
    conv_epilogue<1, 2 * NSL, 2>(a, hh, xx, b, [=](int row) {
        const int y = yw + slab_row(row), x = x0 + slab_col(row);
        return (y < H && x < W) ? y * W + x : -1; }, n0, lane, wave, tid, grp == 0, reinterpret_cast<float*>(lds), grp);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// 12-wave variant of the small-grid kernel for the <= 128-output-channel convolutions of the batch-1 update block (q of both GRU halves,
// the motion encoder's last 3x3).  On the 60 x 80 grid those are 40 patches x 4 channel tiles = 160 workgroups of the 8-wave kernel: 96
// CUs idle and every busy CU carries four 32 x 32 wave tiles of matrix work.  The chip holds 600 such tiles, i.e. three per CU is the
// floor.  A 6 x 16 patch = NSL = 3 pixel slabs gives 50 patches x 4 = 200 workgroups with THREE tiles each -- but three slabs x two
// k-halves = 6 waves load the four SIMDs 2 / 2 / 1 / 1 and nothing is gained (measured in round 3: 12.2 vs 11.9 us).  So the k range is
// split FOUR ways: wave (slab, khalf, parity) -- besides its 16-deep half of every 32-channel block a wave only takes the channel blocks
// of its PARITY.  A stage holds BP = 2 consecutive channel blocks (halo + weights of both), the two parities work on them concurrently:
// 12 waves = 3 per SIMD, each SIMD carries 3/4 of the matrix work of the 8-wave kernel's, in half as many (twice as long) steps, so the
// per-step skeleton (wait, barrier, DMA issue) is paid half as often and hidden by three waves instead of two.
// Everything else is conv_halo10_kernel: stages, rings, one barrier per 3-tap step, the shared epilogue -- with four partial-sum slab sets
// that the four groups add on the read side (group g takes row group g).  A block past the end (odd CB) is staged as ZERO weights.
// ---------------------------------------------------------------------------------------------------------------------
template <int KH, int KW, int NSL, int BP>
__global__ __launch_bounds__(128 * NSL * BP, 1) void conv_halo_bp_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 2 * NSL, TW = 16;
    constexpr int HWD = TW + KW - 1, HR = HWD * (TH + KH - 1);
    constexpr int A_UNITS = (HR + 15) / 16;
    constexpr int A_PLANE = A_UNITS * 1024, A_BLK = 2 * A_PLANE, A_BUF = BP * A_BLK;     // one stage = BP channel blocks x (hi, lo)
    constexpr int NTAPS = KH * KW;
    constexpr int TPS = 3;                                          // taps per step (the last step of a 5-tap filter has 2)
    constexpr int NST = (NTAPS + TPS - 1) / TPS;                    // steps per stage (3 or 2)
    constexpr int B_TAP = 4096, B_BLK = TPS * B_TAP, B_SLOT = BP * B_BLK;   // per tap and block: 32 weight rows x 64 B x 2 planes
    constexpr int O_B = 2 * A_BUF;
    constexpr int NW = 2 * BP * NSL;                                // waves
    constexpr int NQ = NW / 2;                                      // DMA roles per plane: wave (q, plane), plane = wave_all & 1, q = wave_all >> 1
    // per plane a stage is BP * A_UNITS halo pieces of 1 KB and a weight slot BP * 2 * ntp; wave q takes pieces q + NQ i.  Every wave issues the
    // SAME number of LDS-DMA instructions (the counted vmcnt waits are compile-time): indices past the end are dummy pieces.
    constexpr int NIA = (BP * A_UNITS + NQ - 1) / NQ;               // halo load instructions per wave per stage
    constexpr int O_SCR = O_B + 2 * B_SLOT;                         // 1 KB scratch behind the weight slots
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all / NSL, wave = wave_all - grp * NSL;    // compute role: pixel slab `wave`, k-group grp = parity * 2 + khalf
    const int khalf = grp & 1, bpar = grp >> 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    int y0, x0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        if (mt >= tiles_x * tiles_y) return;
        n0 = (slot - (slot / ntn) * ntn) * 32;
        const int ty = mt / tiles_x;
        y0 = ty * TH;
        x0 = (mt - ty * tiles_x) * TW;
    }

    // ---- LDS-DMA sources ------------------------------------------------------------------------------------------
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const int q = wave_all >> 1;
    const bool lo_p = wave_all & 1;
    unsigned aoff[NIA];
    int a_dst_off[NIA], a_blk[NIA];                                 // destination inside a stage (or the scratch KB) and block of the stage
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int pi = q + NQ * i;
        const int blk = pi / A_UNITS, unit = pi - blk * A_UNITS;
        const bool real = pi < BP * A_UNITS;
        const int row = unit * 16 + urow;
        const int hy = row / HWD, hx = row - hy * HWD;
        const int py = y0 - a.pad_h + hy, px = x0 - a.pad_w + hx;
        const bool ok = real && row < HR && py >= 0 && py < a.H && px >= 0 && px < a.W;
        aoff[i] = ok ? (unsigned)(((py * a.W + px) * 32 + uchunk) * 2) : 0x80000000u;
        a_blk[i] = real ? blk : 0;
        a_dst_off[i] = real ? blk * A_BLK + unit * 1024 + (lo_p ? A_PLANE : 0) : -1;
    }
    const unsigned wvo0 = (unsigned)(((n0 + urow) * 32 + uchunk) * 2), wvo1 = wvo0 + 16 * 64;   // weight rows of unit 0 / 1
    const int CB2 = a.CB - a.CB1;
    const int plane_b = a.P_in * 64;
    const rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_p ? a.xl : a.xh) + (long long)b * a.CB1 * a.P_in * 32), 0, a.CB1 * plane_b, 0x00020000);
    const rsrc_t r_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)((lo_p ? a.x2l : a.x2h) + (long long)b * CB2 * a.P_in * 32), 0, CB2 * plane_b, 0x00020000);
    const int wtile_b = a.cout_pad * 64;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(lo_p ? a.wl : a.wh), 0, NTAPS * a.CB * wtile_b, 0x00020000);
    char* const w_dst = lds + O_B + (lo_p ? 2048 : 0);

    // halo of stage SG (channel blocks SG * BP ...) into stage buffer BUF; a block past the end re-reads the last one (its weights are zero)
#define HB_ISSUE_A(SG, BUF, SEL, NSEL)                                                                                   \
    {                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NIA; ++i) if (i % (NSEL) == (SEL)) {                                       \
            const int cb0_ = (SG) * BP + a_blk[i];                                                                       \
            const int cbi_ = cb0_ < a.CB ? cb0_ : a.CB - 1;                                                              \
            const bool first_ = cbi_ < a.CB1;                                                                            \
            const int so_ = (first_ ? cbi_ : cbi_ - a.CB1) * plane_b;                                                    \
            char* const dst_ = a_dst_off[i] >= 0 ? lds + (BUF) * A_BUF + a_dst_off[i] : lds + O_SCR;                     \
            if (first_) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a1, (lptr_t)dst_, 16, aoff[i], so_, 0, 0);            \
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a2, (lptr_t)dst_, 16, aoff[i], so_, 0, 0);                   \
        }                                                                                                                \
    }
    // weight slot for step (SG, ST): taps ST*TPS .. of the BP blocks; piece idx -> block idx / (2 ntp), tap (idx % (2 ntp)) / 2, unit idx & 1.
    // ST is a compile-time constant.  A block >= CB: out-of-range source = ZEROS in LDS (its MFMAs run on them and add nothing).
#define HB_ISSUE_B(SG, ST, SLOT, SEL, NSEL)                                                                              \
    {                                                                                                                    \
        constexpr int ntp_ = (NTAPS - (ST) * TPS) < TPS ? (NTAPS - (ST) * TPS) : TPS;      /* taps of this step */       \
        constexpr int nib_ = (BP * 2 * ntp_ + NQ - 1) / NQ;                                                              \
        _Pragma("unroll") for (int i = 0; i < nib_; ++i) if (i % (NSEL) == (SEL)) {                                      \
            const int pj_ = q + NQ * i;                                                                                  \
            const int blk_ = pj_ / (2 * ntp_), idx_ = pj_ - blk_ * (2 * ntp_);                                           \
            const int cbi_ = (SG) * BP + blk_;                                                                           \
            const bool real_ = pj_ < BP * 2 * ntp_;                                                                      \
            const bool data_ = real_ && cbi_ < a.CB;                                                                     \
            const int so_ = data_ ? (((ST) * TPS + (idx_ >> 1)) * a.CB + cbi_) * wtile_b : 0;                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(real_ ? w_dst + (SLOT) * B_SLOT + blk_ * B_BLK + (idx_ >> 1) * B_TAP + (idx_ & 1) * 1024 : lds + O_SCR), 16, \
                                                     data_ ? ((idx_ & 1) ? wvo1 : wvo0) : 0x80000000u, so_, 0, 0);       \
        }                                                                                                                \
    }

    f32x16 hh[1], x1, x2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        hh[0][r] = 0.f;
        x1[r] = 0.f;
        x2[r] = 0.f;
    }

    HB_ISSUE_A(0, 0, 0, 1)
    HB_ISSUE_B(0, 0, 0, 0, 1)

    const int R0 = (wave * 2 + slab_row(l31)) * HWD + slab_col(l31);
    const int kq = khalf * 2 + kh;                                  // this lane's 16-B k-chunk of the 64-B row
    const int wro = l31 * 64 + ((kq ^ ((l31 >> 2) & 3)) * 16);      // weight fragment offset inside a (tap, plane) tile
    const int nstages = (a.CB + BP - 1) / BP;
    int cur = 0;
    for (int sg = 0; sg < nstages; ++sg) {
        const char* abuf = lds + (sg & 1) * A_BUF + bpar * A_BLK;
        static_for<0, NTAPS>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            constexpr int st = t / TPS, j = t % TPS;
            if (j == 0) {                          // ---- step boundary
                // in flight, oldest first: [halo of stage sg+1 (issued at st 0, AFTER that step's weights)], weights of this step
                if (!CONV_SPREAD_DMA && NST > 1 && st == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (!CONV_SPREAD_DMA) {
                    constexpr int stn = (st + 1) % NST;
                    const int sgn = sg + (st + 1) / NST;
                    HB_ISSUE_B(sgn, stn, cur ^ 1, 0, 1)
                    if (st == 0) HB_ISSUE_A(sg + 1, (sg + 1) & 1, 0, 1)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (CONV_SPREAD_DMA) {
                constexpr int ntc = (NTAPS - st * TPS) < TPS ? (NTAPS - st * TPS) : TPS;
                constexpr int stn = (st + 1) % NST;
                const int sgn = sg + (st + 1) / NST;
                if (st == 0) HB_ISSUE_A(sg + 1, (sg + 1) & 1, j, ntc)
                HB_ISSUE_B(sgn, stn, cur ^ 1, j, ntc)
            }
            const char* wcur = lds + O_B + cur * B_SLOT + bpar * B_BLK;
            const int R = R0 + (t / KW) * HWD + (t % KW);
            const int ao = R * 64 + ((kq ^ ((R >> 2) & 3)) * 16);
            const half8 xh = *reinterpret_cast<const half8*>(abuf + ao);
            const half8 xl = *reinterpret_cast<const half8*>(abuf + A_PLANE + ao);
            const half8 wh = *reinterpret_cast<const half8*>(wcur + j * B_TAP + wro);
            const half8 wl = *reinterpret_cast<const half8*>(wcur + j * B_TAP + 2048 + wro);
            hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hh[0], 0, 0, 0);   // D[channel][pixel]
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, x2, 0, 0, 0);
            if (j == TPS - 1 || t == NTAPS - 1) {
                __builtin_amdgcn_sched_barrier(0);   // fragment reads complete before the next barrier releases the refill
                cur ^= 1;
            }
        });
    }
#undef HB_ISSUE_A
#undef HB_ISSUE_B
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- the 2 * BP partial sums are combined on the read side of the shared epilogue (each group stages its partial sums)
    f32x16 xx[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) xx[0][r] = x1[r] + x2[r];
    const int W = a.W, H = a.H, yw = y0 + wave * 2;
    conv_epilogue<1, NW, 2 * BP>(a, hh, xx, b, [=](int row) {
        const int y = yw + slab_row(row), x = x0 + slab_col(row);
        return (y < H && x < W) ? y * W + x : -1; }, n0, lane, wave, tid, grp == 0, reinterpret_cast<float*>(lds), grp);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem kernel: KS x KS stride-2 convolution of a FEW-channel fp32 NCHW tensor (the 7x7/2 entry convolution of BasicEncoder,
// extractor.py:63,110: 5 / 8 / 25 / 41 / 3 input channels -> 64).  With so few channels a 32-channel k-block per tap would be
// 85-97 % padding, and an im2col tensor in HBM would be 8x the input.  Instead the k index runs over (channel, tap) TIGHTLY
// (K = C*KS*KS rounded up to 32 per chunk of <= 8 channels) and the im2col happens in LDS:
//   * a workgroup owns an 8 x 16 output patch x 64 channels; per channel chunk its (2*8+KS-2) x (2*16+KS-2) fp32 input patch is
//     loaded once (zero padded);
//   * per k-block the 128 x 32 activation tile is BUILT in LDS by the vector ALU (gather through a k -> patch-offset table,
//     fp32 -> hi/lo split, 16-B swizzled stores) into one of two tile buffers while the matrix cores consume the other;
//   * weights are an ordinary packed (k-block, Cout_pad, 32) split tensor (the host presents them as a 1x1 convolution over
//     K_total channels, chunk-major, (c, r, q) inside a chunk), streamed by LDS-DMA, double buffered;
//   * same accumulator layout and epilogue as the other kernels (bias / folded BatchNorm / ReLU / statistics / f32 / split).
// ---------------------------------------------------------------------------------------------------------------------
struct StemArgs {
    const float* x;            // (B, Cin, H, W) fp32
    int Cin, H, W;
    int chunk;                 // channels per chunk (<= 8)
    int kblocks_per_chunk;     // ceil(chunk * KS * KS / 32) (the LAST chunk may hold fewer channels; its table marks the rest as zero)
    // channel windows of a wider source tensor (B_src, C_src, H, W): image n reads channels [win[n / B_src], + Cin) of source image
    // n % B_src -- RAFTSpline.gen_voxel_grids (raft.py:88-99) followed by torch.cat, without the copy.  B_src = 0: x is plain.
    int B_src, C_src;
    int win[8];
    int layout;                // 0: k = (channel, tap) im2col tiles built in LDS (conv_stem_kernel); 1: row windows (conv_stem_rows_kernel)
    // General input (row-window kernel only; `general` = 0 keeps the plain path above untouched): the Cin channels of an image are
    //   segment 0: channels [0, c1) -- window group g = n / B_src reads channels [win[g], + c1) of image n % B_src of ITS OWN source tensor
    //              xw[g] (C_src channels per image): the two images of raft.py:136 without torch.cat, or the voxel windows as before;
    //   segment 1: channels [c1, Cin) -- image n % B_src of x2 (Cin - c1 channels per image): `img0` behind the context bins (raft.py:137-140);
    // element type per segment (0 fp32, 1 uint8) and `norm`: 2 * (v / 255) - 1 applied to in-image elements (raft.py:134; padding stays 0).
    int general, c1;
    const void* xw[8];
    const void* x2;
    int dt0, dt1, norm0, norm1;
};

#ifndef STEM_ABL
#define STEM_ABL 0      // tools/stem_ablate.sh: 1 no epilogue, 2 no tile build / patch split, 4 no patch load, 8 no MFMA, 16 no weight DMA (rows kernel)
#endif
template <int KS, int STRIDE, bool TR = false>   // TR: transposed accumulators + direct fp32 epilogue, as in conv_halo_kernel
__global__ __launch_bounds__(CT, 2) void conv_stem_kernel(ConvArgs a, StemArgs sa) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 2, TH = 8, TW = 16;
    constexpr int PR = STRIDE * (TH - 1) + KS, PC = STRIDE * (TW - 1) + KS;     // input patch rows / cols (21 x 37)
    constexpr int MAXC = 8;
    constexpr int A_TILE = 2 * CBM * 64;                  // 128 pixel rows x 64 B x (hi, lo)
    constexpr int W_TILE = 2 * NT * 2048;                 // 64 weight rows x 64 B x (hi, lo)
    constexpr int O_A = 0, O_W = 2 * A_TILE, O_P = O_W + 2 * W_TILE, O_T = O_P + MAXC * PR * PC * 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* patch = reinterpret_cast<float*>(lds + O_P);
    int* koff = reinterpret_cast<int*>(lds + O_T);        // [kblocks_per_chunk * 32] patch offsets of the chunk's k values, -1 = zero

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    int y0, x0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        if (mt >= tiles_x * tiles_y) return;
        n0 = (slot - (slot / ntn) * ntn) * 32 * NT;
        const int ty = mt / tiles_x;
        y0 = ty * TH;
        x0 = (mt - ty * tiles_x) * TW;
    }
    const int gy0 = y0 * STRIDE - a.pad_h, gx0 = x0 * STRIDE - a.pad_w;   // image position of patch element (0, 0)

    // weights: 8 pieces (2 planes x 4 units of 16 rows) per k-block, 2 per wave: plane = wave >> 1, units (2 wave + j) & 3
    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    unsigned wvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wvo[j] = (unsigned)(((n0 + ((wave * 2 + j) & 3) * 16 + urow) * 32 + uchunk) * 2);
    const int wtile_b = a.cout_pad * 64;
    const int nchunks = (sa.Cin + sa.chunk - 1) / sa.chunk;
    const int nkb = nchunks * sa.kblocks_per_chunk;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)((wave >> 1) ? a.wl : a.wh), 0, nkb * wtile_b, 0x00020000);
    char* const w_dst = lds + O_W + ((wave >> 1) ? NT * 2048 : 0);
#define STEM_ISSUE_W(KB, BUF)                                                                                            \
    {                                                                                                                    \
        const int so_ = ((KB) < nkb ? (KB) : nkb - 1) * wtile_b;                                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(w_dst + (BUF) * W_TILE + ((wave * 2 + j) & 3) * 1024), 16, wvo[j], so_, 0, 0); \
    }

    // builder: thread -> pixel p = tid & 127 of the patch, k half h = tid >> 7 (16 of the 32 k values of a k-block)
    const int bp = tid & 127, bh = tid >> 7;
    const int pbase = ((bp >> 4) * STRIDE) * PC + (bp & 15) * STRIDE;         // patch offset of the pixel's tap (0, 0)
    const int bsw = (bp >> 2) & 3;
    auto build_tile = [&](int kbl, int buf) {                                 // k-block kbl of the current chunk -> A tile `buf`
        char* dst = lds + O_A + buf * A_TILE + bp * 64;
        const int4* tk = reinterpret_cast<const int4*>(koff + kbl * 32 + bh * 16);   // 16 offsets: the same for every pixel of this half
        int o[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int4 t4 = tk[g4];
            o[4 * g4] = t4.x; o[4 * g4 + 1] = t4.y; o[4 * g4 + 2] = t4.z; o[4 * g4 + 3] = t4.w;
        }
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = patch[pbase + max(o[i], 0)];        // branch-free: clamped address, select below
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
            half8 h8, l8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 hi, lo;
                split1(o[g8 * 8 + i] >= 0 ? v[g8 * 8 + i] : 0.f, hi, lo);
                h8[i] = hi;
                l8[i] = lo;
            }
            const int pos = ((bh * 2 + g8) ^ bsw) * 16;                      // swizzled 16-B chunk of the 64-B row
            *reinterpret_cast<half8*>(dst + pos) = h8;
            *reinterpret_cast<half8*>(dst + CBM * 64 + pos) = l8;
        }
    };

    f32x16 hh[NT], xx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hh[n][r] = 0.f;
            xx[n][r] = 0.f;
        }

    const int sw = (l31 >> 2) & 3;
    const int aro = (wave * 32 + l31) * 64;
    STEM_ISSUE_W(0, 0)
    int kb = 0;                                                               // global k-block index (weights)
    for (int ck = 0; ck < nchunks; ++ck) {
        const int c_first = ck * sa.chunk;
        const int nch = min(sa.chunk, sa.Cin - c_first);
        __syncthreads();                                                      // previous chunk's patch / table / tiles are drained
        // ---- input patch of this chunk (zero outside the image) and its k -> offset table
        {   // Buffer loads: an out-of-image element is an out-of-range offset (returns 0) -- no branch, so the NLD loads of a thread are
            // all in flight together (with plain loads hipcc guards each with a branch + s_waitcnt: one HBM latency per element,
            // 21 k cycles per workgroup measured).
            constexpr int NLD = (MAXC * PR * PC + CT - 1) / CT;
            float pv[NLD];
            const int total = nch * PR * PC;
            const long long img = (long long)sa.H * sa.W;
            const long long x_base = sa.B_src > 0 ? ((long long)(b % sa.B_src) * sa.C_src + sa.win[b / sa.B_src] + c_first) * img
                                                  : ((long long)b * sa.Cin + c_first) * img;
            const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(sa.x + x_base), 0, (int)(nch * img * 4), 0x00020000);
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int i = tid + j * CT;
                const int c = i / (PR * PC), rem = i - c * (PR * PC);
                const int pr = rem / PC, pc = rem - pr * PC;
                const int gy = gy0 + pr, gx = gx0 + pc;
                const bool ok = i < total && gy >= 0 && gy < sa.H && gx >= 0 && gx < sa.W;
                const unsigned off = ok ? (unsigned)(((c * sa.H + gy) * sa.W + gx) * 4) : 0x80000000u;
                pv[j] = (STEM_ABL & 4) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, off, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int i = tid + j * CT;
                if (i < total) patch[i] = pv[j];
            }
        }
        for (int k = tid; k < sa.kblocks_per_chunk * 32; k += CT) {
            const int c = k / (KS * KS), rq = k - c * (KS * KS);
            const int r = rq / KS, q = rq - r * KS;
            koff[k] = c < nch ? (c * PR + r) * PC + q : -1;
        }
        __syncthreads();
        if (!(STEM_ABL & 2)) build_tile(0, kb & 1);
        for (int kbl = 0; kbl < sa.kblocks_per_chunk; ++kbl, ++kb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // weight tile kb landed
            __syncthreads();                                                  // ... everywhere; activation tile kb is complete
            STEM_ISSUE_W(kb + 1, (kb + 1) & 1)
            if (kbl + 1 < sa.kblocks_per_chunk && !(STEM_ABL & 2)) build_tile(kbl + 1, (kb + 1) & 1);   // VALU work under this block's MFMAs
            const char* at = lds + O_A + (kb & 1) * A_TILE;
            const char* wt = lds + O_W + (kb & 1) * W_TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int co = ((ks * 2 + kh) ^ sw) * 16;
                const half8 xh = *reinterpret_cast<const half8*>(at + aro + co);
                const half8 xl = *reinterpret_cast<const half8*>(at + CBM * 64 + aro + co);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int wo = (n * 32 + l31) * 64 + co;
                    const half8 wh = *reinterpret_cast<const half8*>(wt + wo);
                    const half8 wl = *reinterpret_cast<const half8*>(wt + NT * 2048 + wo);
                    if (!(STEM_ABL & 8)) {
                    if constexpr (TR) {
                    hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, hh[n], 0, 0, 0);
                    xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, xx[n], 0, 0, 0);
                    xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, xx[n], 0, 0, 0);
                    } else {
                    hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hh[n], 0, 0, 0);
                    xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, xx[n], 0, 0, 0);
                    xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, xx[n], 0, 0, 0);
                    }
                    } else { hh[n][0] += (float)wh[0] * (float)xh[0] + (float)wl[1] * (float)xl[1]; }
                }
            }
        }
    }
#undef STEM_ISSUE_W
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int Wo = a.Wo, Ho = a.Ho, yw = y0 + wave * 2;
    if (STEM_ABL & 1) {
        float acc_ = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_ += hh[n][r] + xx[n][r];
        if (acc_ == 12345.678f) a.out_f32[tid] = acc_;
        return;
    }
    if constexpr (TR) {
        // register r of lane (channel, kh): pixel (r & 3) + 8 (r >> 2) + 4 kh of the wave's 2 x 16 row-major slab
        const int kh_ = lane >> 5, c4 = (lane & 31) * 4;
        conv_epilogue_direct<NT>(a, hh, xx, b, [=](int r) {
            const int y = yw + (r >> 3), x = x0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh_;
            return (y < Ho && x < Wo) ? (unsigned)((y * Wo + x) * 128 + c4) : 0x80000000u; }, n0, lane, wave, tid, reinterpret_cast<float*>(lds));
    } else {
        conv_epilogue<NT>(a, hh, xx, b, [=](int row) {
            const int y = yw + (row >> 4), x = x0 + (row & 15);
            return (y < Ho && x < Wo) ? y * Wo + x : -1; }, n0, lane, wave, tid, true, reinterpret_cast<float*>(lds));
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem kernel, row-window form (round 3; StemArgs.layout = 1).  The im2col form above spends its time building 128 x 32 activation
// tiles on the vector ALU (a table gather + split per k element: 128 KB of LDS writes per workgroup for 32 KB of output; 50 of the
// launch's 157 us).  Here the input patch of a channel chunk is split ONCE (one conversion per input element: 8x less) and laid out
// [row][column][channel] in LDS, so that the 7 x CH values a filter ROW needs for one output pixel are CONTIGUOUS: k = (q, c) is a
// window of 7 CH halves starting at column 2 x_out.  A lane's MFMA fragment is 8 consecutive halves of that window -- four
// ds_read_b32 (the window starts on a 4-byte, not a 16-byte boundary; adjacent pixels are 4 CH bytes apart: 5 dwords for CH = 5, so the
// 32 lanes of a read spread over all 32 banks) -- and nothing is ever gathered.  K per filter row is 7 CH rounded up to 16 (CH = 5: 48
// for 35; the surplus columns are the next pixels' finite values against ZERO weights), 7 rows per chunk; the weights are an ordinary
// packed tensor in (chunk, row, window) order, two 16-deep steps per 32-wide k-block, streamed by LDS-DMA as before.
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, bool TR, bool GEN = false>   // GEN: the general input (StemArgs.general) -- its own instantiation, so that the plain
                                                          // kernel keeps its 158 registers (3 workgroups per CU; one merged kernel needed 202: 84 -> 110 us)
__global__ __launch_bounds__(CT, 2) void conv_stem_rows_kernel(ConvArgs a, StemArgs sa) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 2, TH = 8, TW = 16;
    constexpr int PR = STRIDE * (TH - 1) + KS, PC = STRIDE * (TW - 1) + KS;     // input patch rows / cols (21 x 37)
    constexpr int W_TILE = 2 * NT * 2048;                 // 64 weight rows x 64 B x (hi, lo)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int CH = sa.chunk;
    const int pitch = (PC * CH + 1) & ~1;                 // halves per patch row (even: rows start on 4-byte boundaries)
    const int plane_h = PR * pitch + 64;                  // + slack read by the last pixels' window surplus (zeroed)
    const int O_W = ((2 * plane_h * 2 + 15) & ~15);       // weight tiles behind the two planes
    _Float16* const ph_ = reinterpret_cast<_Float16*>(lds);
    _Float16* const pl_ = ph_ + plane_h;
    const int KSTEPS = (KS * CH + 15) >> 4;               // 16-deep steps per filter row
    const int nsteps = KS * KSTEPS;                       // per chunk
    const int kbpc = (nsteps + 1) >> 1;                   // 32-wide weight k-blocks per chunk

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    int y0, x0, n0;
    {
        const int ntn = a.n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int mt = (slot / ntn) * 8 + xcd;
        if (mt >= tiles_x * tiles_y) return;
        n0 = (slot - (slot / ntn) * ntn) * 32 * NT;
        const int ty = mt / tiles_x;
        y0 = ty * TH;
        x0 = (mt - ty * tiles_x) * TW;
    }
    const int gy0 = y0 * STRIDE - a.pad_h, gx0 = x0 * STRIDE - a.pad_w;

    const int urow = lane >> 2;
    const int uchunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    unsigned wvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wvo[j] = (unsigned)(((n0 + ((wave * 2 + j) & 3) * 16 + urow) * 32 + uchunk) * 2);
    const int wtile_b = a.cout_pad * 64;
    const int nchunks = (sa.Cin + CH - 1) / CH;
    const int nkb = nchunks * kbpc;
    const rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)((wave >> 1) ? a.wl : a.wh), 0, nkb * wtile_b, 0x00020000);
    char* const w_dst = lds + O_W + ((wave >> 1) ? NT * 2048 : 0);
#define STEMR_ISSUE_W(KB, BUF)                                                                                           \
    if (!(STEM_ABL & 16)) {                                                                                              \
        const int so_ = ((KB) < nkb ? (KB) : nkb - 1) * wtile_b;                                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(w_dst + (BUF) * W_TILE + ((wave * 2 + j) & 3) * 1024), 16, wvo[j], so_, 0, 0); \
    }

    f32x16 hh[NT], xx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hh[n][r] = 0.f;
            xx[n][r] = 0.f;
        }
    // this lane's pixel of the patch and the half index of its window start (row 0)
    const int yo = 2 * wave + (l31 >> 4), xo = l31 & 15;
    const int win0 = (STRIDE * yo) * pitch + STRIDE * xo * CH + kh * 8;
    const int sw = (l31 >> 2) & 3;

    for (int i = tid; i < 64; i += CT) { ph_[PR * pitch + i] = (_Float16)0.f; pl_[PR * pitch + i] = (_Float16)0.f; }   // the slack
    STEMR_ISSUE_W(0, 0)
    int kb = 0;
    for (int ck = 0; ck < nchunks; ++ck) {
        const int c_first = ck * CH;
        const int nch = min(CH, sa.Cin - c_first);
        __syncthreads();                                                      // the previous chunk's patch is drained
        {   // ---- the chunk's input patch: thread = patch pixel, all channels; out-of-image = out-of-range buffer offset = 0
            const long long img = (long long)sa.H * sa.W;
            constexpr int NPX = (PR * PC + CT - 1) / CT;
            float pv[NPX][8];
            if constexpr (!GEN) {
            const long long x_base = sa.B_src > 0 ? ((long long)(b % sa.B_src) * sa.C_src + sa.win[b / sa.B_src] + c_first) * img
                                                  : ((long long)b * sa.Cin + c_first) * img;
            const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(sa.x + x_base), 0, (int)(nch * img * 4), 0x00020000);
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int p = tid + j * CT;
                const int pr = p / PC, pc = p - pr * PC;
                const int gy = gy0 + pr, gx = gx0 + pc;
                const bool ok = p < PR * PC && gy >= 0 && gy < sa.H && gx >= 0 && gx < sa.W;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned off = (ok && c < nch) ? (unsigned)(((c * sa.H + gy) * sa.W + gx) * 4) : 0x80000000u;
                    pv[j][c] = (STEM_ABL & 4) ? 1.f : c < CH ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, off, 0, 0)) : 0.f;
                }
            }
            } else {
            // two segments, each fp32 or uint8, each with the image normalisation or without: BOTH segments' loads are issued for every channel,
            // the one the channel does not belong to with an out-of-range offset (returns 0, costs no traffic) -- no per-channel branch, every
            // load of the thread in flight together.  The element is the sum of the two (one of them is 0 by construction).
            const int g = sa.B_src > 0 ? b / sa.B_src : 0, bi = sa.B_src > 0 ? b - g * sa.B_src : b;
            const int es0 = sa.dt0 ? 1 : 4, es1 = sa.dt1 ? 1 : 4;
            const int n0c = max(0, min(nch, sa.c1 - c_first));                 // channels of this chunk that come from segment 0
            const int cs1 = max(sa.c1 - c_first, 0);                           // chunk-local channel where segment 1 starts
            const int n1c = max(0, nch - cs1);
            const char* base0 = reinterpret_cast<const char*>(sa.xw[g]) + ((long long)bi * sa.C_src + sa.win[g] + c_first) * img * es0;
            const char* base1 = reinterpret_cast<const char*>(sa.x2) + ((long long)bi * (sa.Cin - sa.c1) + max(c_first - sa.c1, 0)) * img * es1;
            const rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)base0, 0, (int)(n0c * img * es0), 0x00020000);
            const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(sa.x2 ? base1 : base0), 0, sa.x2 ? (int)(n1c * img * es1) : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int p = tid + j * CT;
                const int pr = p / PC, pc = p - pr * PC;
                const int gy = gy0 + pr, gx = gx0 + pc;
                const bool ok = p < PR * PC && gy >= 0 && gy < sa.H && gx >= 0 && gx < sa.W;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c >= CH) { pv[j][c] = 0.f; continue; }
                    const bool in0 = ok && c < n0c, in1 = ok && c >= cs1 && c < nch;
                    const unsigned o0 = in0 ? (unsigned)(((c * sa.H + gy) * sa.W + gx) * es0) : 0x80000000u;
                    const unsigned o1 = in1 ? (unsigned)((((c - cs1) * sa.H + gy) * sa.W + gx) * es1) : 0x80000000u;
                    const float v0 = sa.dt0 ? (float)(unsigned)__builtin_amdgcn_raw_buffer_load_b8(r0, o0, 0, 0)
                                            : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, o0, 0, 0));
                    const float v1 = sa.dt1 ? (float)(unsigned)__builtin_amdgcn_raw_buffer_load_b8(r1, o1, 0, 0)
                                            : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, o1, 0, 0));
                    // raft.py:134: 2 * (x / 255) - 1, the reference's operation order (a true division; 2 q is exact, so the contraction of
                    // the last two steps into one FMA rounds identically); zero padding is applied to the NORMALISED image: padding stays 0
                    const float q0 = (sa.norm0 && in0) ? 2.0f * (v0 / 255.0f) - 1.0f : v0;
                    const float q1 = (sa.norm1 && in1) ? 2.0f * (v1 / 255.0f) - 1.0f : v1;
                    pv[j][c] = q0 + q1;
                }
            }
            }
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int p = tid + j * CT;
                if (p < PR * PC && !((STEM_ABL & 2) && pv[j][0] != 12345.f)) {
                    const int pr = p / PC, pc = p - pr * PC;
                    const int o = pr * pitch + pc * CH;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < CH) {
                            _Float16 hi, lo;
                            split1(pv[j][c], hi, lo);
                            ph_[o + c] = hi;
                            pl_[o + c] = lo;
                        }
                    if (pc == PC - 1 && (PC * CH & 1)) { ph_[o + CH] = (_Float16)0.f; pl_[o + CH] = (_Float16)0.f; }   // the pitch's pad half
                }
            }
        }
        for (int kbl = 0; kbl < kbpc; ++kbl, ++kb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // weight tile kb landed
            __syncthreads();                                                  // ... everywhere; (kbl = 0: the patch is complete)
            STEMR_ISSUE_W(kb + 1, (kb + 1) & 1)
            const char* wt = lds + O_W + (kb & 1) * W_TILE;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int st = 2 * kbl + h;
                if (st < nsteps) {                                            // (an odd step count leaves the last half k-block empty)
                    const int r = st / KSTEPS, ks = st - r * KSTEPS;
                    const int wo_ = win0 + r * pitch + ks * 16;              // halves; 4-byte aligned
                    const unsigned* qh = reinterpret_cast<const unsigned*>(ph_ + wo_);
                    const unsigned* ql = reinterpret_cast<const unsigned*>(pl_ + wo_);
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 vh = {qh[0], qh[1], qh[2], qh[3]}, vl = {ql[0], ql[1], ql[2], ql[3]};
                    const half8 xh = __builtin_bit_cast(half8, vh), xl = __builtin_bit_cast(half8, vl);
                    const int co = ((h * 2 + kh) ^ sw) * 16;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int wo = (n * 32 + l31) * 64 + co;
                        const half8 wh = *reinterpret_cast<const half8*>(wt + wo);
                        const half8 wl = *reinterpret_cast<const half8*>(wt + NT * 2048 + wo);
                        if constexpr ((STEM_ABL & 8) != 0) {
                            if (xh[0] == (_Float16)123.f && wh[0] == (_Float16)77.f && wl[1] == xl[1]) hh[n][0] += 1.f;
                        } else if constexpr (TR) {
                            hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, hh[n], 0, 0, 0);
                            xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, xx[n], 0, 0, 0);
                            xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, xx[n], 0, 0, 0);
                        } else {
                            hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, hh[n], 0, 0, 0);
                            xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, xx[n], 0, 0, 0);
                            xx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, xx[n], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
#undef STEMR_ISSUE_W
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int Wo = a.Wo, Ho = a.Ho, yw = y0 + wave * 2;
    if constexpr ((STEM_ABL & 1) != 0) {
        float acc_ = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_ += hh[n][r] + xx[n][r];
        if (acc_ == 1.2345f) a.out_f32[tid] = acc_;
    } else if constexpr (TR) {
        const int kh_ = lane >> 5, c4 = (lane & 31) * 4;
        conv_epilogue_direct<NT>(a, hh, xx, b, [=](int r) {
            const int y = yw + (r >> 3), x = x0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh_;
            return (y < Ho && x < Wo) ? (unsigned)((y * Wo + x) * 128 + c4) : 0x80000000u; }, n0, lane, wave, tid, reinterpret_cast<float*>(lds));
    } else {
        conv_epilogue<NT>(a, hh, xx, b, [=](int row) {
            const int y = yw + (row >> 4), x = x0 + (row & 15);
            return (y < Ho && x < Wo) ? y * Wo + x : -1; }, n0, lane, wave, tid, true, reinterpret_cast<float*>(lds));
    }
#endif
}

#include "conv_stem_persist.h"

// ---------------------------------------------------------------------------------------------------------------------
// weights (Cout, Cin, KH, KW) fp32 -> split (KH*KW*CB, Cout_pad, 32), zero padded (k-tile-major)
// ---------------------------------------------------------------------------------------------------------------------
// adjoint != 0: the filter of the INPUT-GRADIENT convolution, read from the forward weight w (Cin, Cout, KH, KW) [the roles swap: this
// pack's "Cout" rows are the forward's input channels]: w'[co][c][tap] = w[c][co][ntaps - 1 - tap]  (flipped in space, transposed in channels)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ wh, _Float16* __restrict__ wl,
                                                           int Cout, int Cin, int ntaps, int Cout_pad, int CB, int adjoint) {
    const long long total = (long long)ntaps * CB * Cout_pad * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c32 = (int)(i & 31);
        const long long t1 = i >> 5;
        const int co = (int)(t1 % Cout_pad);
        const int kt = (int)(t1 / Cout_pad);
        const int tap = kt / CB, c = (kt - tap * CB) * 32 + c32;
        float v = 0.f;
        if (co < Cout && c < Cin) v = adjoint ? w[((long long)c * Cout + co) * ntaps + (ntaps - 1 - tap)] : w[((long long)co * Cin + c) * ntaps + tap];
        _Float16 h, l;
        split1(v, h, l);
        wh[i] = h;
        wl[i] = l;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// per-plane sum / sum of squares of an NCHW fp32 tensor (InstanceNorm statistics of a MIOpen conv output)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int HW) {
    __shared__ double sh[2][4];
    const float* p = x + (long long)blockIdx.x * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x * 4; i + 3 < HW; i += 256 * 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int i = (HW & ~3) + threadIdx.x; i < HW; i += 256) {
        s1 += p[i];
        s2 += (double)p[i] * p[i];
    }
    s1 = bflow::wave_sum(s1);
    s2 = bflow::wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s1;
        sh[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 2) stats[(long long)blockIdx.x * 2 + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// ---------------------------------------------------------------------------------------------------------------------
// normalise / affine + activation + residual -> blocked split.  One kernel for every "between two convolutions" step of
// the encoder:   out = act_out( res + act_a( norm_a(a) ) ),   res in { nothing, split tensor, norm_b(b) }
//   a, b   : fp32 blocked (B, CB, P, 32), or (a only) plain NCHW (B, C, HW) which is transposed on the fly
//   norm_* : InstanceNorm from (sum, sumsq) statistics [stats != null]  or per-channel affine [scale/shift]  or identity
// Thread = 8 consecutive channels of one pixel (32-B fp32 reads, 16-B fp16 writes).
// ---------------------------------------------------------------------------------------------------------------------

struct NormArgs {
    const float* a; const double* stats_a; const float *scale_a, *shift_a; int a_nchw; int act_a;
    const float* b; const double* stats_b;
    const _Float16 *rh, *rl;
    int act_out;
    _Float16 *oh, *ol; float* out_f32;
    int B, HW, C, CB, P; float eps;
    int tiles_per_block;   // consecutive 64-pixel tiles walked by one block (amortises the coefficient set-up on big grids)
    int stats_reps;        // replicas of the statistics tables (summed here)
    int act_b;             // 1: the second branch is relu(norm_b(b)) (a residual that was never materialised), 0: norm_b(b)
};


__global__ __launch_bounds__(256) void norm_act_split_kernel(NormArgs p) {
    __shared__ float tile[32][65];                    // NCHW input only: 32 channels x 64 pixels
    __shared__ float coef[4][32];                     // (mul_a, add_a, mul_b, add_b) of this block's 32 channels: computed ONCE
    const int b = blockIdx.z, cb = blockIdx.y;        // (the fp64 mean / variance / rsqrt per element dominated this kernel)
    if (threadIdx.x < 32) {
        const int c = cb * 32 + threadIdx.x;
        float m, a2;
        norm_coeffs(p.stats_a, p.scale_a, p.shift_a, b, c, p.C, p.HW, p.eps, m, a2, p.stats_reps, (long long)p.B * p.C * 2);
        coef[0][threadIdx.x] = m;
        coef[1][threadIdx.x] = a2;
        if (p.b) {
            norm_coeffs(p.stats_b, nullptr, nullptr, b, c, p.C, p.HW, p.eps, m, a2, p.stats_reps, (long long)p.B * p.C * 2);
            coef[2][threadIdx.x] = m;
            coef[3][threadIdx.x] = a2;
        }
    }
    __syncthreads();
    const int pl = threadIdx.x >> 2, c8 = (threadIdx.x & 3) * 8;
    for (int tix = 0; tix < p.tiles_per_block; ++tix) {
        const int p0 = (blockIdx.x * p.tiles_per_block + tix) * 64;
        if (p0 >= p.HW) break;
        const int pix = p0 + pl;
        if (p.a_nchw) {
            __syncthreads();
            const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cb * 32 + ty + 4 * i, px = p0 + tx;
                tile[ty + 4 * i][tx] = (c < p.C && px < p.HW) ? p.a[((long long)b * p.C + c) * p.HW + px] : 0.f;
            }
            __syncthreads();
        }
        if (pix >= p.HW) continue;
        const long long o = (((long long)b * p.CB + cb) * p.P + pix) * 32 + c8;
        float v[8];
        if (p.a_nchw) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tile[c8 + k][pl];
        } else {
            const float4 v0 = *reinterpret_cast<const float4*>(p.a + o), v1 = *reinterpret_cast<const float4*>(p.a + o + 4);
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        }
        float bv[8];
        if (p.b) {
            const float4 v0 = *reinterpret_cast<const float4*>(p.b + o), v1 = *reinterpret_cast<const float4*>(p.b + o + 4);
            bv[0] = v0.x; bv[1] = v0.y; bv[2] = v0.z; bv[3] = v0.w; bv[4] = v1.x; bv[5] = v1.y; bv[6] = v1.z; bv[7] = v1.w;
        }
        half8 rh8, rl8;
        if (p.rh) {
            rh8 = *reinterpret_cast<const half8*>(p.rh + o);
            rl8 = *reinterpret_cast<const half8*>(p.rl + o);
        }
        half8 oh8, ol8;
        float of[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = cb * 32 + c8 + k;
            float x = fmaf(v[k], coef[0][c8 + k], coef[1][c8 + k]);   // (explicit fma: the NIN halo kernel applies the same expression, bit for bit)
            if (p.act_a == 1) x = fmaxf(x, 0.f);
            if (p.b) {
                const float yb = fmaf(bv[k], coef[2][c8 + k], coef[3][c8 + k]);
                x += p.act_b == 1 ? fmaxf(yb, 0.f) : yb;
            }
            if (p.rh) x += (float)rh8[k] + (float)rl8[k] * LO_INV;
            if (p.act_out == 1) x = fmaxf(x, 0.f);
            else if (p.act_out == 2) x = tanhf(x);
            if (c >= p.C) x = 0.f;
            _Float16 h, l;
            split1(x, h, l);
            oh8[k] = h;
            ol8[k] = l;
            of[k] = x;
        }
        if (p.oh) {
            *reinterpret_cast<half8*>(p.oh + o) = oh8;
            *reinterpret_cast<half8*>(p.ol + o) = ol8;
        }
        if (p.out_f32) {
            *reinterpret_cast<float4*>(p.out_f32 + o) = make_float4(of[0], of[1], of[2], of[3]);
            *reinterpret_cast<float4*>(p.out_f32 + o + 4) = make_float4(of[4], of[5], of[6], of[7]);
        }
    }
}

// blocked split -> fp32 NCHW (leaving the engine, e.g. towards the MIOpen-based update block)
__global__ __launch_bounds__(256) void split_to_nchw_kernel(const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                            float* __restrict__ out, int HW, int CB, int P, int c_first, int c_count,
                                                            long long out_bs) {
    __shared__ float tile[64][33];                    // 64 pixels x 32 channels
    const int b = blockIdx.z, cb = blockIdx.y, p0 = blockIdx.x * 64;   // cb counts blocks from c_first/32
    {
        const int pl = threadIdx.x >> 2, c8 = (threadIdx.x & 3) * 8;
        const int pix = p0 + pl;
        if (pix < HW) {
            const long long o = (((long long)b * CB + (c_first >> 5) + cb) * P + pix) * 32 + c8;
            const half8 h = *reinterpret_cast<const half8*>(xh + o), l = *reinterpret_cast<const half8*>(xl + o);
#pragma unroll
            for (int k = 0; k < 8; ++k) tile[pl][c8 + k] = (float)h[k] + (float)l[k] * LO_INV;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = cb * 32 + ty + 4 * i, pix = p0 + tx;
        if (c < c_count && pix < HW) out[b * out_bs + (long long)c * HW + pix] = tile[tx][ty + 4 * i];
    }
}

}  // namespace

// A convolution that a PAIR launch (bflow_conv_split_pair) can take: its resolved kernel arguments and grid instead of a launch.
struct PairPlan {
    int kind = 0;        // 0 = launched on its own (no pair variant of its kernel); 1 = conv_split_kernel<2, 3, 2>; 2 = conv_halo8_kernel<3, 3>
    ConvArgs a;
    unsigned gx = 0;     // workgroups (a multiple of 8)
    int lds = 0;
};

static int conv_split_impl(const bflow_conv_desc_t* d, bflow_stream_t stream, PairPlan* plan) {
    BFLOW_REQUIRE(d && ((d->x_hi && d->x_lo) || (d->x_raw && d->x_stats)) && d->w_hi && d->w_lo, BFLOW_E_ARG, "conv_split: null operand");
    if (d->x_raw)
        BFLOW_REQUIRE(d->x_stats && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && !d->x2_hi && d->out_f32 && !d->out_hi &&
                          !d->addend && !d->gate && !d->acc_nchw && d->weight_sets <= 1 && d->C <= 128 && d->tile_n == 64,
                      BFLOW_E_ARG, "conv_split: x_raw (normalised-on-load input) needs a stride-1 3x3 with fp32 (+ stats) output, C <= 128, 64-channel tiles");
    BFLOW_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 32 == 0, BFLOW_E_ARG, "conv_split: C=%d must be a multiple of 32", d->C);
    BFLOW_REQUIRE(d->KH > 0 && d->KW > 0 && (d->stride == 1 || d->stride == 2) && d->Cout > 0, BFLOW_E_ARG, "conv_split: bad filter");
    BFLOW_REQUIRE(d->out_f32 || (d->out_hi && d->out_lo), BFLOW_E_ARG, "conv_split: no output");
    const int Ho = (d->H + 2 * d->pad_h - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad_w - d->KW) / d->stride + 1;
    BFLOW_REQUIRE(Ho > 0 && Wo > 0, BFLOW_E_ARG, "conv_split: empty output");
    BFLOW_REQUIRE(d->B <= 65535, BFLOW_E_LIMIT, "conv_split: batch too large");
    BFLOW_REQUIRE(d->KH * d->KW <= 64, BFLOW_E_LIMIT, "conv_split: more than 64 filter taps");
    {
        const long long P_in = d->in_rows_per_image > 0 ? d->in_rows_per_image : (long long)d->H * d->W;
        BFLOW_REQUIRE((long long)(d->C / 32) * P_in * 64 < (1LL << 31) && (long long)d->KH * d->KW * (d->C / 32) * d->cout_pad * 64 < (1LL << 31),
                      BFLOW_E_LIMIT, "conv_split: an image / the weights exceed the 2 GB buffer-addressing window");
    }
    const int NT = d->tile_n / 32;
    BFLOW_REQUIRE(NT >= 2 && NT <= 4 && d->tile_n % 32 == 0, BFLOW_E_ARG, "conv_split: tile_n must be 64, 96 or 128");
    BFLOW_REQUIRE(d->cout_pad >= d->Cout, BFLOW_E_ARG, "conv_split: cout_pad < Cout");
    BFLOW_REQUIRE(d->out_channel_offset % 32 == 0, BFLOW_E_ARG, "conv_split: channel offset must be a multiple of 32");
    ConvArgs a;
    a.xh = (const _Float16*)d->x_hi; a.xl = (const _Float16*)d->x_lo; a.wh = (const _Float16*)d->w_hi; a.wl = (const _Float16*)d->w_lo;
    a.x2h = (const _Float16*)d->x2_hi; a.x2l = (const _Float16*)d->x2_lo; a.addend = d->addend;
    a.CB1 = (d->x2_hi && d->x2_lo) ? d->x_split_channels / 32 : d->C / 32;
    BFLOW_REQUIRE(a.CB1 > 0 && a.CB1 <= d->C / 32 && d->x_split_channels % 32 == 0, BFLOW_E_ARG, "conv_split: bad two-source split");
    a.H = d->H; a.W = d->W; a.CB = d->C / 32; a.P_in = d->in_rows_per_image > 0 ? d->in_rows_per_image : d->H * d->W;
    a.Ho = Ho; a.Wo = Wo; a.Cout = d->Cout; a.cout_pad = d->cout_pad;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
    a.out_f32 = d->out_f32; a.oh = (_Float16*)d->out_hi; a.ol = (_Float16*)d->out_lo;
    const int out_c = d->out_channel_stride > 0 ? d->out_channel_stride : (d->Cout + 31) / 32 * 32;
    BFLOW_REQUIRE(out_c % 32 == 0 && (d->gate == 1 ? d->Cout == 2 * out_c : d->out_channel_offset + d->Cout <= out_c), BFLOW_E_ARG,
                  "conv_split: bad output channel layout");
    a.CBo = out_c / 32; a.cb_off = d->out_channel_offset / 32; a.P_out = d->out_rows_per_image > 0 ? d->out_rows_per_image : Ho * Wo;
    a.scale = d->scale; a.shift = d->shift; a.act = d->act; a.stats = d->stats;
    a.stats_reps = d->stats_replicas > 0 ? d->stats_replicas : 1; a.stats_rep_stride = (long long)d->B * d->Cout * 2;
    a.acc = d->acc_nchw;
    a.w_sets = d->weight_sets > 1 ? d->weight_sets : 1;
    a.keep_pad = d->keep_pad_channels;
#ifdef H8_STAMPS
    a.stamps = g_h8_stamp_buf;
#else
    a.stamps = nullptr;
#endif
    BFLOW_REQUIRE(!d->keep_pad_channels || (d->Cout % 4 == 0 && !d->gate && !d->out_f32), BFLOW_E_ARG, "conv_split: keep_pad_channels needs Cout %% 4 == 0, split output only");
    a.xraw = d->x_raw; a.xstats = d->x_stats; a.xstats_reps = d->x_stats_replicas > 0 ? d->x_stats_replicas : 1; a.xeps = d->x_eps;
    a.gate = d->gate; a.gh = (const _Float16*)d->gate_h_hi; a.gl = (const _Float16*)d->gate_h_lo; a.gz = d->gate_z;
    if (d->gate) {
        BFLOW_REQUIRE(d->gate >= 1 && d->gate <= 3 && d->gate_h_hi && d->gate_h_lo && d->out_hi && d->out_lo && !d->stats && (d->act == 0 || d->gate == 3) &&
                          d->out_channel_offset == 0 && d->Cout % 32 == 0, BFLOW_E_ARG, "conv_split: bad gate arguments");
        BFLOW_REQUIRE(d->gate == 1 ? (d->out_f32 && d->Cout == 2 * out_c) : ((d->gate_z || d->gate == 3) && d->Cout == out_c), BFLOW_E_ARG,
                      "conv_split: gate buffers must hold Cout/2 (zr) or Cout (blend, residual) channels");
        BFLOW_REQUIRE(d->gate != 3 || (!d->out_f32 && !d->acc_nchw), BFLOW_E_ARG, "conv_split: the residual epilogue writes the split output only");
    }
    a.n_tiles = bflow::ceil_div(d->Cout, d->tile_n);
    const int m_tiles8 = (bflow::ceil_div((long long)Ho * Wo, CBM) + 7) / 8 * 8;   // pixel tiles, padded to the 8 XCDs
    dim3 grid(m_tiles8 * a.n_tiles, 1, d->B);
    hipStream_t s = (hipStream_t)stream;
    const long long nblocks = (long long)bflow::ceil_div((long long)Ho * Wo, CBM) * a.n_tiles * d->B;
    // halo kernel: stride 1, "same" padding, 3x3 / 1x5 / 5x1
    const char* force = getenv("BFLOW_CONV_KERNEL");           // tests: "generic" forces the generic kernel for every shape
    const bool same = d->stride == 1 && d->pad_w == (d->KW - 1) / 2 && d->pad_h == (d->KH - 1) / 2;
    const int shape = (d->KH == 3 && d->KW == 3) ? 1 : (d->KH == 1 && d->KW == 5) ? 2 : (d->KH == 5 && d->KW == 1) ? 3 : 0;
    if (same && shape && a.w_sets == 1 && !(force && strncmp(force, "halo", 4) != 0)) {
        const int patches = bflow::ceil_div(d->H, 8) * bflow::ceil_div(d->W, 16);
        // 64-channel tiles unless that leaves most CUs without a workgroup (batch-1 update block: 40 patches)
        const int nt = (a.xraw || (long long)patches * d->B * bflow::ceil_div(d->Cout, 64) >= 200) ? 2 : 1;
        a.n_tiles = bflow::ceil_div(d->Cout, 32 * nt);
        dim3 hgrid((patches + 7) / 8 * 8 * a.n_tiles, 1, d->B);
        static const int halo_lds_pad = [] { const char* e = getenv("BFLOW_HALO_LDS_PAD"); return e ? atoi(e) : 0; }();   // tools: forces 1 workgroup per CU
#define LAUNCH_HALO(N, KHH, KWW)                                                                                       \
    {                                                                                                                  \
        constexpr int units_ = (((16 + (KWW) - 1) * (8 + (KHH) - 1) + 15) / 16 + 1) / 2 * 2;                           \
        const int lds = 2 * 2 * units_ * 1024 + 4 * (N) * 4096 + halo_lds_pad;   /* (pad: occupancy probe, tools) */       \
        if ((N) == 2 && (KHH) == 3 && a.xraw) {   /* ... and the input normalised on its way into LDS */         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<2, 3, 3, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((conv_halo_kernel<2, 3, 3, true, true>), hgrid, dim3(CT), lds, s, a);                   \
        } else if ((N) == 2 && (KHH) == 3 && direct) {   /* fp32 (+ statistics) output: transposed accumulators, stores without an LDS transpose */ \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<2, 3, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((conv_halo_kernel<2, 3, 3, true>), hgrid, dim3(CT), lds, s, a);                         \
        } else {                                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<N, KHH, KWW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((conv_halo_kernel<N, KHH, KWW>), hgrid, dim3(CT), lds, s, a);                           \
        }                                                                                                              \
    }
#define LAUNCH_HALO_NSL(KHH, KWW, NSLL, PATCHES)                                                                       \
    {                                                                                                                  \
        const int lds = 2 * 2 * (((16 + (KWW) - 1) * (2 * (NSLL) + (KHH) - 1) + 15) / 16) * 1024 + 2 * 3 * 4096 + 1024;  \
        dim3 gridn(((PATCHES) + 7) / 8 * 8 * a.n_tiles, 1, d->B);                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo10_kernel<KHH, KWW, NSLL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((conv_halo10_kernel<KHH, KWW, NSLL>), gridn, dim3(128 * (NSLL)), lds, s, a);                \
    }
#define LAUNCH_HALO10(KHH, KWW) LAUNCH_HALO_NSL(KHH, KWW, 5, patches10)
#define LAUNCH_HALO12(KHH, KWW)                                                                                        \
    {                                                                                                                  \
        constexpr int au_ = ((16 + (KWW) - 1) * (6 + (KHH) - 1) + 15) / 16;                                            \
        const int main_ = 2 * 2 * 2 * au_ * 1024 + 2 * 2 * 3 * 4096 + 1024;                                            \
        const int epi_ = (2 * 12 * 32 + 4 * 3 * 32 * CONV_STG_STRIDE) * 4;                                             \
        const int lds = main_ > epi_ ? main_ : epi_;                                                                   \
        dim3 gridn((patches6 + 7) / 8 * 8 * a.n_tiles, 1, d->B);                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_bp_kernel<KHH, KWW, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((conv_halo_bp_kernel<KHH, KWW, 3, 2>), gridn, dim3(768), lds, s, a);                        \
    }
#define LAUNCH_HALO8(KHH, KWW)                                                                                         \
    {                                                                                                                  \
        const int lds = 2 * 2 * (((16 + (KWW) - 1) * (8 + (KHH) - 1) + 15) / 16) * 1024 + 2 * 3 * 4096;                   \
        if (plan && (KHH) == 3 && (KWW) == 3 && !a.stats) { plan->kind = 2; plan->a = a; plan->gx = hgrid.x; plan->lds = lds; return 0; } \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo8_kernel<KHH, KWW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((conv_halo8_kernel<KHH, KWW>), hgrid, dim3(2 * CT), lds, s, a);                             \
    }
        static const bool no_direct = getenv("BFLOW_CONV_NO_DIRECT") != nullptr;      // A/B timing (tools/)
        const bool direct = !no_direct && a.out_f32 && !a.oh && !a.addend && !a.gate && !a.acc;
        const bool small8 = nt == 1 && !(force && strcmp(force, "halo4") == 0);   // small grids: the 8-wave split-k variant
        // 10 x 16 patches when the 8 x 16 grid needs a second workgroup on some CUs and the 10 x 16 grid does not
        const int patches10 = bflow::ceil_div(d->H, 10) * bflow::ceil_div(d->W, 16);
        const long long wg8 = (long long)patches * d->B * a.n_tiles, wg10 = (long long)patches10 * d->B * a.n_tiles;
        // (6 x 16 patches = 3 slabs on SIX waves, two k-groups, were measured in round 3: q 12.2 vs 11.9 us -- 2 / 2 / 1 / 1 waves per SIMD)
        const bool forced_variant = force && strncmp(force, "halo", 4) == 0 && force[4];
        // (a pair of 3x3s on 10 x 16 patches -- convc2 | convf2 at 60 x 80 as 180 + 60 workgroups, one per CU, instead of 240 + 80 of 8 x 16 -- was
        //  built and measured in round 4: 22.3 vs 19.2 us for the launch, 3.587-3.614 vs 3.570-3.597 ms per frame; removed)
        const bool ten = small8 && (forced_variant ? strcmp(force, "halo10") == 0 : (wg8 > 256 && wg10 <= 256));
        // 6 x 16 patches on 12 waves (conv_halo_bp_kernel: four k-groups) when the 8 x 16 grid leaves more than a quarter of the chip idle and
        // the 6 x 16 grid still fits one round: the <= 128-channel convolutions at 60 x 80 (160 -> 200 workgroups carrying 3 instead of 4
        // wave tiles each).  BFLOW_CONV_KERNEL=halo12 / halo8x16 force either side (tests, A/B).
        const int patches6 = bflow::ceil_div(d->H, 6) * bflow::ceil_div(d->W, 16);
        const long long wg6 = (long long)patches6 * d->B * a.n_tiles;
        static const bool no_h12 = getenv("BFLOW_CONV_NO_HALO12") != nullptr;
        const bool twelve = small8 && !ten && !plan && (forced_variant ? strcmp(force, "halo12") == 0 : (!no_h12 && wg8 <= 192 && wg6 <= 256 && wg6 > wg8));
        // the persistent form (conv_stream.h) for the feature encoder's 3x3s on grids of more than two rounds: fp32 (+ statistics) output.
        // Measured per shape (tools/enc_stream_probe.py, profiles/r05_enc_stream_ab.txt): 64 -> 64 (two input blocks, one channel tile): -5...-9 %;
        // 96 -> 96 / 128 -> 128 (27 / 36 k-steps per item, two channel tiles, the 96-channel one half empty): +-0 at batch 40, +3 % at batch 5
        // (ranges of 3 patches); with the input normalised on load (x_raw: conv2 of every residual block, 3 of the 4 layer-1 launches) 64 -> 64
        // 114 -> 96 us / 900 -> 749 us (-16 %), 96 -> 96 72 -> 67 / 533 -> 468 us -- the kernel runs any block count >= 2 (tested), the dispatch
        // takes it where it wins: x_raw always, else two input blocks.  BFLOW_CONV_STREAM=all (tests, A/B) takes it wherever it can run, =0 never.
        static const int stream_mode = [] { const char* e = getenv("BFLOW_CONV_STREAM"); return !e ? 1 : !strcmp(e, "0") ? 0 : !strcmp(e, "all") ? 2 : 1; }();
        static const long long stream_min_items = [] { const char* e = getenv("BFLOW_CONV_STREAM_MIN_ITEMS"); return e ? atoll(e) : 1024LL; }();   // tools A/B
        const long long items = (long long)patches * d->B * a.n_tiles;
        if (shape == 1 && nt == 2 && direct && (!a.xraw || a.CB <= 4) && !a.x2h && a.act != 2 && stream_mode && !force && a.CB >= 2 && (a.CB == 2 || a.xraw || stream_mode == 2) &&
            items >= stream_min_items && items < (1LL << 30) && a.n_tiles <= 32) {
            // ranges of `per` patches x one channel tile; <= 512 of them (two workgroups per CU), a multiple of 8 x n_tiles (whole XCDs of
            // whole patch ranges; a few trailing workgroups may own no patch)
            const long long bp = (long long)patches * d->B;
            // <= 512 workgroups (two per CU) = 256 / n_tiles pairs of ranges per channel tile; a pair = an older and a younger workgroup (see the
            // kernel: blockIdx < gridDim / 2 is dispatched first), which share pair_sum consecutive items share_old : 1 - share_old.
            // BFLOW_CONV_STREAM_SHARE (percent for the older one; tools A/B).  Default 50 = equal ranges: 57 : 43 shortens the launch ALONE by 2-3 %
            // (profiles/r05_enc_stream_clock.txt) but not the frame (303.8 vs 301.2 frames/s over three alternating pairs, c4_strong equal:
            // profiles/r05_stream_share_frame_ab.txt) -- next to the context encoder's launches the early-finishing older workgroups are the CUs
            // (and the LDS) those launches get
            static const int share_pct = [] { const char* e = getenv("BFLOW_CONV_STREAM_SHARE"); const int v = e ? atoi(e) : 0; return v >= 50 && v <= 80 ? v : 50; }();
            const int g = 512 / (16 * a.n_tiles) * (16 * a.n_tiles);   // whole pairs of ranges per XCD for every channel tile (n_tiles = 3: 480)
            const int pairs = g / a.n_tiles / 2;
            const int pair_sum = (int)bflow::ceil_div(bp, pairs);
            int per_old = (pair_sum * share_pct + 50) / 100;
            if (per_old >= pair_sum && pair_sum > 1) per_old = pair_sum - 1;
            const int per_young = pair_sum - per_old;
            const int lds = 2 * 2 * 12 * 1024 + 4 * 2 * 4096;
            const int tiles_x = bflow::ceil_div(d->W, 16);
            if (a.xraw) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_stream_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                hipLaunchKernelGGL(conv_halo_stream_kernel<true>, dim3(g), dim3(CT), lds, s, a, per_old, per_young, patches, tiles_x, (int)bp, d->B);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_stream_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                hipLaunchKernelGGL(conv_halo_stream_kernel<false>, dim3(g), dim3(CT), lds, s, a, per_old, per_young, patches, tiles_x, (int)bp, d->B);
            }
            return bflow::launch_status("conv_split(stream)");
        }
        if (shape == 1) { if (nt == 2) LAUNCH_HALO(2, 3, 3) else if (ten) LAUNCH_HALO10(3, 3) else if (twelve) LAUNCH_HALO12(3, 3) else if (small8) LAUNCH_HALO8(3, 3) else LAUNCH_HALO(1, 3, 3) }
        else if (shape == 2) { if (nt == 2) LAUNCH_HALO(2, 1, 5) else if (ten) LAUNCH_HALO10(1, 5) else if (twelve) LAUNCH_HALO12(1, 5) else if (small8) LAUNCH_HALO8(1, 5) else LAUNCH_HALO(1, 1, 5) }
        else { if (nt == 2) LAUNCH_HALO(2, 5, 1) else if (ten) LAUNCH_HALO10(5, 1) else if (twelve) LAUNCH_HALO12(5, 1) else if (small8) LAUNCH_HALO8(5, 1) else LAUNCH_HALO(1, 5, 1) }
#undef LAUNCH_HALO
#undef LAUNCH_HALO8
#undef LAUNCH_HALO10
#undef LAUNCH_HALO12
#undef LAUNCH_HALO_NSL
        return bflow::launch_status("conv_split(halo)");
    }
    // every other inference shape (1x1, stride 2): the direct-activation kernel; the generic kernel keeps the weight-set GEMMs of the
    // training path (and BFLOW_CONV_KERNEL=generic for tests / A-B timing)
    static const bool no_directk = getenv("BFLOW_CONV_NO_DIRECT_KERNEL") != nullptr;   // A/B timing (tools/)
    // (grids that fill the chip several times only: on the <= 320-workgroup grids of the batch-1 update block a 32-channel, two-k-group variant of it
    // measured 17-19 us for convc1 against 16.4 us for the generic split-k kernel -- there the chain of dependent L2 round trips, not
    // the LDS fill rate, sets the time -- so those stay on the generic kernel)
    const long long mt_ = bflow::ceil_div((long long)Ho * Wo, CBM) * d->B;
    if (a.w_sets == 1 && !force && !no_directk && mt_ * bflow::ceil_div(d->Cout, 64) >= 600) {   // (376-workgroup grids of layer3: 39.7 vs 34.0 us, generic)
        static const bool no_direct = getenv("BFLOW_CONV_NO_DIRECT") != nullptr;
        const int nt = (d->Cout > 64 && d->Cout <= 96) ? 3 : 2;     // 64-channel tiles; 96 when that is the whole layer
        a.n_tiles = bflow::ceil_div(d->Cout, 32 * nt);
        dim3 dgrid(m_tiles8 * a.n_tiles, 1, d->B);
        const bool tr = !no_direct && nt > 1 && a.out_f32 && !a.oh && !a.addend && !a.gate && !a.acc;
#define LAUNCH_DIRECT(N, KGG, TRR, DD)                                                                                 \
    {                                                                                                                  \
        const int ring_ = (KGG) * (DD) * 2 * (N) * 2048;                                                               \
        const int epi_ = (TRR) ? 2 * 4 * 32 * (N) * 4 : (2 * 4 * (KGG) * 32 * (N) + 4 * (KGG) * (N) * 32 * CONV_STG_STRIDE) * 4; \
        const int lds = ring_ > epi_ ? ring_ : epi_;                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct_kernel<N, KGG, TRR, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((conv_direct_kernel<N, KGG, TRR, DD>), dgrid, dim3(CT * (KGG)), lds, s, a);                 \
    }
        // prefetch depth (k-tiles in flight + 1): 3; BFLOW_CONV_DIRECT_DEPTH = 4 for A/B (measured equal on the encoder shapes; a depth of 6 on
        // the batch-1 grids was slower than 3)
        static const int depth_env = [] { const char* e = getenv("BFLOW_CONV_DIRECT_DEPTH"); return e ? atoi(e) : 0; }();
        if (nt == 2) {
            if (depth_env == 4) { if (tr) LAUNCH_DIRECT(2, 1, true, 4) else LAUNCH_DIRECT(2, 1, false, 4) }
            else { if (tr) LAUNCH_DIRECT(2, 1, true, 3) else LAUNCH_DIRECT(2, 1, false, 3) }
        } else {
            if (depth_env == 4) { if (tr) LAUNCH_DIRECT(3, 1, true, 4) else LAUNCH_DIRECT(3, 1, false, 4) }
            else { if (tr) LAUNCH_DIRECT(3, 1, true, 3) else LAUNCH_DIRECT(3, 1, false, 3) }
        }
#undef LAUNCH_DIRECT
        return bflow::launch_status("conv_split(direct)");
    }
    const bool deep = nblocks <= 320;   // at most ~1 workgroup per CU: spend the LDS on prefetch depth instead of co-residency
#define LAUNCH(N, SS, KGG)                                                                                             \
    {                                                                                                                  \
        const int lds = (KGG) * (SS) * (2 * CBM * 64 + 2 * ((N) <= 2 ? 1 : 2) * 4 * 1024);                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<N, SS, KGG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((conv_split_kernel<N, SS, KGG>), grid, dim3(CT * (KGG)), lds, s, a);                        \
    }
    if (plan && NT == 2 && deep && !a.stats && a.w_sets == 1) {
        plan->kind = 1; plan->a = a; plan->gx = grid.x; plan->lds = 2 * 3 * (2 * CBM * 64 + 2 * 4 * 1024);
        return 0;
    }
    if (NT == 2) { if (deep) LAUNCH(2, 3, 2) else LAUNCH(2, 3, 1) }
    else if (NT == 3) { if (deep) LAUNCH(3, 5, 1) else LAUNCH(3, 3, 1) }
    else { if (deep) LAUNCH(4, 5, 1) else LAUNCH(4, 3, 1) }
#undef LAUNCH
    return bflow::launch_status("conv_split");
}

extern "C" int bflow_conv_split(const bflow_conv_desc_t* d, bflow_stream_t stream) { return conv_split_impl(d, stream, nullptr); }

// Two independent convolutions (neither reads what the other writes; disjoint outputs; same batch) on one stream.  When both resolve to the
// same small-grid kernel that has a pair variant -- the generic split-k kernel (1x1 / im2col GEMMs) or the 8-wave 3x3 halo kernel -- they
// are ONE launch (conv_split_pair_kernel / conv_halo8_pair_kernel); otherwise two consecutive launches.  Results are those of two
// bflow_conv_split calls bit for bit (same kernel bodies, same tiles).  *fused (optional) receives 1 when one launch was made.
extern "C" int bflow_conv_split_pair(const bflow_conv_desc_t* d0, const bflow_conv_desc_t* d1, int* fused, bflow_stream_t stream) {
    if (fused) *fused = 0;
    BFLOW_REQUIRE(d0 && d1 && d0->B == d1->B, BFLOW_E_ARG, "conv_split_pair: two descriptors of the same batch size expected");
    static const bool no_pair = getenv("BFLOW_CONV_NO_PAIR") != nullptr;      // tools A/B: always two launches
    // the pair runs as ONE grid in no particular order: a convolution that reads what the other writes (or writes the same channel blocks of
    // the same buffer) is not a pair -- two ordered launches instead
    auto reads = [](const bflow_conv_desc_t* r, const void* p) { return p && (p == r->x_hi || p == r->x_lo || p == r->x2_hi || p == r->x2_lo || p == (const void*)r->x_raw ||
                                                                             p == (const void*)r->addend || p == r->gate_h_hi || p == (const void*)r->gate_z); };
    auto writes_into = [&](const bflow_conv_desc_t* w, const bflow_conv_desc_t* r) { return reads(r, w->out_hi) || reads(r, w->out_lo) || reads(r, (const void*)w->out_f32) ||
                                                                                           reads(r, (const void*)w->acc_nchw); };
    const bool same_out = (d0->out_hi && d0->out_hi == d1->out_hi) || (d0->out_f32 && d0->out_f32 == d1->out_f32);
    const bool overlap = same_out && d0->out_channel_offset < d1->out_channel_offset + (d1->Cout + 31) / 32 * 32 &&
                         d1->out_channel_offset < d0->out_channel_offset + (d0->Cout + 31) / 32 * 32;
    if (no_pair || writes_into(d0, d1) || writes_into(d1, d0) || overlap) {
        const int rc = conv_split_impl(d0, stream, nullptr);
        return rc != 0 ? rc : conv_split_impl(d1, stream, nullptr);
    }
    PairPlan p0, p1;
    int rc = conv_split_impl(d0, stream, &p0);
    if (rc != 0) return rc;
    rc = conv_split_impl(d1, stream, &p1);
    if (rc != 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (p0.kind && p0.kind == p1.kind) {
        const int lds = p0.lds > p1.lds ? p0.lds : p1.lds;
        dim3 grid(p0.gx + p1.gx, 1, d0->B);
        if (p0.kind == 1) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_pair_kernel<2, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((conv_split_pair_kernel<2, 3, 2>), grid, dim3(CT * 2), lds, s, p0.a, p1.a, (int)p0.gx);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo8_pair_kernel<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((conv_halo8_pair_kernel<3, 3>), grid, dim3(2 * CT), lds, s, p0.a, p1.a, (int)p0.gx);
        }
        if (fused) *fused = 1;
        return bflow::launch_status("conv_split_pair");
    }
    // no common pair kernel: whatever was only planned is launched on its own, through the ordinary route
    if (p0.kind) { rc = conv_split_impl(d0, stream, nullptr); if (rc != 0) return rc; }
    if (p1.kind) { rc = conv_split_impl(d1, stream, nullptr); if (rc != 0) return rc; }
    return 0;
}

extern "C" int bflow_conv_pack_weights(const float* w, void* w_hi, void* w_lo, int Cout, int Cin, int KH, int KW, int cout_pad, int cin_pad,
                                       bflow_stream_t stream) {
    BFLOW_REQUIRE(w && w_hi && w_lo && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && cout_pad >= Cout && cin_pad >= Cin && cin_pad % 32 == 0,
                  BFLOW_E_ARG, "conv_pack_weights: bad arguments");
    const long long total = (long long)cout_pad * KH * KW * cin_pad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)w_hi,
                       (_Float16*)w_lo, Cout, Cin, KH * KW, cout_pad, cin_pad / 32, 0);
    return bflow::launch_status("conv_pack_weights");
}

extern "C" int bflow_conv_pack_weights_adjoint(const float* w, void* w_hi, void* w_lo, int Cout, int Cin, int KH, int KW, int cout_pad, int cin_pad,
                                               bflow_stream_t stream) {
    BFLOW_REQUIRE(w && w_hi && w_lo && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && cout_pad >= Cout && cin_pad >= Cin && cin_pad % 32 == 0,
                  BFLOW_E_ARG, "conv_pack_weights_adjoint: bad arguments");
    const long long total = (long long)cout_pad * KH * KW * cin_pad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)w_hi,
                       (_Float16*)w_lo, Cout, Cin, KH * KW, cout_pad, cin_pad / 32, 1);
    return bflow::launch_status("conv_pack_weights_adjoint");
}

extern "C" int bflow_plane_stats(const float* x, double* stats, long long planes, int HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && stats && planes > 0 && HW > 0 && planes < (1LL << 31), BFLOW_E_ARG, "plane_stats: bad arguments");
    BFLOW_REQUIRE(((uintptr_t)x & 15) == 0 && HW % 4 == 0, BFLOW_E_ARG, "plane_stats: needs 16-B aligned planes (HW %% 4 == 0)");
    hipLaunchKernelGGL(plane_stats_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, stats, HW);
    return bflow::launch_status("plane_stats");
}

extern "C" int bflow_norm_act_split(const bflow_norm_desc_t* d, bflow_stream_t stream) {
    BFLOW_REQUIRE(d && d->a && d->B > 0 && d->HW > 0 && d->C > 0, BFLOW_E_ARG, "norm_act_split: bad arguments");
    BFLOW_REQUIRE((d->out_hi && d->out_lo) || d->out_f32, BFLOW_E_ARG, "norm_act_split: no output");
    BFLOW_REQUIRE(!(d->b && d->a_is_nchw), BFLOW_E_ARG, "norm_act_split: second branch requires blocked inputs");
    NormArgs p;
    p.a = d->a; p.stats_a = d->stats_a; p.scale_a = d->scale_a; p.shift_a = d->shift_a; p.a_nchw = d->a_is_nchw; p.act_a = d->act_a;
    p.act_b = d->act_b;
    p.b = d->b; p.stats_b = d->stats_b; p.rh = (const _Float16*)d->res_hi; p.rl = (const _Float16*)d->res_lo; p.act_out = d->act_out;
    p.oh = (_Float16*)d->out_hi; p.ol = (_Float16*)d->out_lo; p.out_f32 = d->out_f32; p.B = d->B; p.HW = d->HW; p.C = d->C;
    p.CB = (d->C + 31) / 32; p.P = d->rows_per_image > 0 ? d->rows_per_image : d->HW; p.eps = d->eps;
    p.stats_reps = d->stats_replicas > 0 ? d->stats_replicas : 1;
    const long long tiles = (long long)bflow::ceil_div(d->HW, 64) * p.CB * d->B;
    p.tiles_per_block = tiles >= 8192 ? 4 : tiles >= 4096 ? 2 : 1;       // keep >= ~2000 blocks in flight
    dim3 grid(bflow::ceil_div(d->HW, 64 * p.tiles_per_block), p.CB, d->B);
    hipLaunchKernelGGL(norm_act_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return bflow::launch_status("norm_act_split");
}

extern "C" int bflow_split_to_nchw(const void* x_hi, const void* x_lo, float* out, int B, int HW, int C, int c_first, int c_count,
                                   long long out_batch_stride, bflow_stream_t stream) {
    BFLOW_REQUIRE(x_hi && x_lo && out && B > 0 && HW > 0 && C > 0 && C % 32 == 0 && c_first >= 0 && c_first % 32 == 0 && c_count > 0 &&
                      c_first + c_count <= C, BFLOW_E_ARG, "split_to_nchw: bad arguments");
    dim3 grid(bflow::ceil_div(HW, 64), bflow::ceil_div(c_count, 32), B);
    hipLaunchKernelGGL(split_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_hi, (const _Float16*)x_lo, out, HW,
                       C / 32, HW, c_first, c_count, out_batch_stride);
    return bflow::launch_status("split_to_nchw");
}

extern "C" int bflow_conv_stem(const bflow_stem_desc_t* d, bflow_stream_t stream) {
    BFLOW_REQUIRE(d && d->x && d->w_hi && d->w_lo && d->B > 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, BFLOW_E_ARG,
                  "conv_stem: bad arguments");
    BFLOW_REQUIRE(d->ksize == 7 && d->stride == 2 && d->pad == 3, BFLOW_E_LIMIT, "conv_stem: only the 7x7 stride-2 pad-3 stem is built");
    BFLOW_REQUIRE((d->out_hi && d->out_lo) || d->out_f32, BFLOW_E_ARG, "conv_stem: no output");
    BFLOW_REQUIRE(d->cout_pad >= d->Cout && d->cout_pad % 64 == 0, BFLOW_E_ARG, "conv_stem: cout_pad must be a multiple of 64 >= Cout");
    const int Ho = (d->H + 2 * d->pad - d->ksize) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->ksize) / d->stride + 1;
    BFLOW_REQUIRE(Ho > 0 && Wo > 0, BFLOW_E_ARG, "conv_stem: empty output");
    BFLOW_REQUIRE((long long)8 * d->H * d->W * 4 < (1LL << 31), BFLOW_E_LIMIT, "conv_stem: 8 input planes exceed the 2 GB buffer-addressing window");
    StemArgs sa;
    sa.x = d->x; sa.Cin = d->Cin; sa.H = d->H; sa.W = d->W;
    sa.chunk = d->Cin < 8 ? d->Cin : 8;
    sa.kblocks_per_chunk = (sa.chunk * d->ksize * d->ksize + 31) / 32;
    sa.B_src = 0; sa.C_src = d->Cin;
    for (int i = 0; i < 8; ++i) sa.win[i] = 0;
    if (d->n_windows > 0) {
        const int wwidth = d->x2 ? d->Cin - d->x2_channels : d->Cin;          // channels a window supplies (the rest comes from x2)
        BFLOW_REQUIRE(d->n_windows <= 8 && d->window_starts && d->B % d->n_windows == 0 && d->src_channels >= wwidth && wwidth > 0, BFLOW_E_ARG,
                      "conv_stem: bad channel windows");
        sa.B_src = d->B / d->n_windows; sa.C_src = d->src_channels;
        for (int i = 0; i < d->n_windows; ++i) {
            BFLOW_REQUIRE(d->window_starts[i] >= 0 && d->window_starts[i] + wwidth <= d->src_channels, BFLOW_E_ARG, "conv_stem: window %d out of range", i);
            sa.win[i] = d->window_starts[i];
        }
    }
    sa.layout = d->layout;
    BFLOW_REQUIRE(d->layout >= 0 && d->layout <= 3, BFLOW_E_ARG, "conv_stem: layout must be 0 (im2col tiles), 1 (row windows), 2 / 3 (row windows, persistent / per-patch form forced)");
    const bool rows_layout = d->layout >= 1;
    sa.general = 0; sa.c1 = d->Cin; sa.x2 = nullptr; sa.dt0 = sa.dt1 = sa.norm0 = sa.norm1 = 0;
    for (int i = 0; i < 8; ++i) sa.xw[i] = d->x;
    if (d->window_bases || d->x2 || d->x_dtype || d->x_image_norm) {
        BFLOW_REQUIRE(rows_layout, BFLOW_E_ARG, "conv_stem: multi-source / uint8 / normalised input needs the row-window kernel (layout 1)");
        BFLOW_REQUIRE((d->x_dtype == 0 || d->x_dtype == 1) && (d->x2_dtype == 0 || d->x2_dtype == 1), BFLOW_E_ARG, "conv_stem: element types are 0 (fp32) or 1 (uint8)");
        BFLOW_REQUIRE(!d->x2 || (d->x2_channels > 0 && d->x2_channels < d->Cin), BFLOW_E_ARG, "conv_stem: x2_channels must be in (0, Cin)");
        BFLOW_REQUIRE(!d->window_bases || d->n_windows > 0, BFLOW_E_ARG, "conv_stem: window_bases needs n_windows");
        sa.general = 1;
        sa.c1 = d->x2 ? d->Cin - d->x2_channels : d->Cin;
        if (d->n_windows > 0) {
            BFLOW_REQUIRE(d->src_channels >= sa.c1, BFLOW_E_ARG, "conv_stem: source narrower than the window");
            for (int i = 0; i < d->n_windows; ++i) {
                BFLOW_REQUIRE(d->window_starts[i] + sa.c1 <= d->src_channels, BFLOW_E_ARG, "conv_stem: window %d out of range", i);
                sa.xw[i] = d->window_bases ? (const void*)d->window_bases[i] : (const void*)d->x;
                BFLOW_REQUIRE(sa.xw[i], BFLOW_E_ARG, "conv_stem: null window base %d", i);
            }
        } else {
            sa.C_src = sa.c1;
        }
        sa.x2 = d->x2; sa.dt0 = d->x_dtype; sa.dt1 = d->x2_dtype; sa.norm0 = d->x_image_norm; sa.norm1 = d->x2_image_norm;
    }
    const int ksteps_row = (d->ksize * sa.chunk + 15) / 16;                       // layout 1: 16-deep steps per filter row
    const int kb_expected = ((d->Cin + sa.chunk - 1) / sa.chunk) * (rows_layout ? (d->ksize * ksteps_row + 1) / 2 : sa.kblocks_per_chunk);
    BFLOW_REQUIRE(d->k_blocks == kb_expected, BFLOW_E_ARG,
                  "conv_stem: the packed weights hold %d k-blocks, expected %d (chunks of %d channels, layout %d)", d->k_blocks, kb_expected, sa.chunk, d->layout);
    ConvArgs a = {};
    a.wh = (const _Float16*)d->w_hi; a.wl = (const _Float16*)d->w_lo;
    a.H = d->H; a.W = d->W; a.Ho = Ho; a.Wo = Wo; a.Cout = d->Cout; a.cout_pad = d->cout_pad;
    a.KH = a.KW = d->ksize; a.stride = d->stride; a.pad_h = a.pad_w = d->pad;
    a.out_f32 = d->out_f32; a.oh = (_Float16*)d->out_hi; a.ol = (_Float16*)d->out_lo;
    a.CBo = (d->Cout + 31) / 32; a.cb_off = 0; a.P_out = d->out_rows_per_image > 0 ? d->out_rows_per_image : Ho * Wo;
    BFLOW_REQUIRE(a.P_out >= Ho * Wo, BFLOW_E_ARG, "conv_stem: out_rows_per_image < Ho*Wo");
    a.scale = d->scale; a.shift = d->shift; a.act = d->act; a.stats = d->stats;
    a.stats_reps = d->stats_replicas > 0 ? d->stats_replicas : 1; a.stats_rep_stride = (long long)d->B * d->Cout * 2;
    a.n_tiles = bflow::ceil_div(d->Cout, 64);
    const int patches = bflow::ceil_div(Ho, 8) * bflow::ceil_div(Wo, 16);
    dim3 grid((patches + 7) / 8 * 8 * a.n_tiles, 1, d->B);
    const int lds = 2 * (2 * CBM * 64) + 2 * (2 * 2 * 2048) + 8 * 21 * 37 * 4 + sa.kblocks_per_chunk * 32 * 4;
    static const bool no_direct = getenv("BFLOW_CONV_NO_DIRECT") != nullptr;      // A/B timing (tools/)
    if (rows_layout) {
        const int pitch = (37 * sa.chunk + 1) & ~1, plane_h = 21 * pitch + 64;
        const int body = ((2 * plane_h * 2 + 15) & ~15) + 2 * (2 * 2 * 2048);
        const bool tr = !no_direct && a.out_f32 && !a.oh;
        // the persistent form (conv_stem_persist.h): plain 5-channel fp32 input (the event encoders of every DSEC / MultiFlow configuration with
        // five correlation bins), fp32 + statistics output, one channel tile; from 4 slabs per wave on (below that the 2048 waves of the
        // launch are not filled evenly).  Measured: profiles/r06_stem_persist.txt.  BFLOW_STEM_PERSIST=0: never (A/B).
        {
            static const bool persist_on = [] { const char* e = getenv("BFLOW_STEM_PERSIST"); return !(e && !strcmp(e, "0")); }();
            const int rows2 = bflow::ceil_div(Ho, 2), tiles_x = bflow::ceil_div(Wo, 16);
            const long long n_slabs = (long long)d->B * rows2 * tiles_x;
            const bool can = tr && !sa.general && d->Cin == 5 && a.n_tiles == 1 && n_slabs < (1LL << 30);
            BFLOW_REQUIRE(d->layout != 2 || can, BFLOW_E_LIMIT, "conv_stem: layout 2 (persistent form) needs a plain 5-channel fp32 input, fp32 output and <= 64 output channels");
            if (can && (d->layout == 2 || (d->layout == 1 && persist_on && n_slabs >= 8192))) {
                constexpr int CH = 5, CUS = 256;
                constexpr int p_pitch = (37 * CH + 1 - 32 + 63) / 64 * 64 + 32, p_plane = 9 * p_pitch + 64, p_slab = (2 * p_plane * 2 + 15) & ~15;   // (as in the kernel)
                constexpr int p_nkb = (7 * ((7 * CH + 15) >> 4) + 1) >> 1;
                // 8 waves per CU, every fragment read from LDS in the k-loop.  BFLOW_STEM_PERSIST_FORM=4 (tools A/B): 4 waves (one per SIMD), the hi-plane
                // weight fragments of all 21 k-steps in registers (168 of 414) -- built, bit-identical, SLOWER (99-102 vs 80-83 us at 5 images, 635 vs
                // 520 us at 40: a lone wave per SIMD has nobody to cover its conversion / epilogue phases; profiles/r06_stem_persist.txt)
                static const bool form8 = [] { const char* e = getenv("BFLOW_STEM_PERSIST_FORM"); return !(e && !strcmp(e, "4")); }();
                const int nwv = form8 ? 8 : 4;
                const int p_lds = p_nkb * (2 * 2 * 2048) + nwv * p_slab;
                const int per_wave = (int)bflow::ceil_div(n_slabs, (long long)CUS * nwv);
                const int g = (int)bflow::ceil_div(n_slabs, (long long)per_wave * nwv);
                if (form8) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_persist_kernel<CH, 8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, p_lds);
                    hipLaunchKernelGGL((conv_stem_persist_kernel<CH, 8, 0>), dim3(g), dim3(512), p_lds, (hipStream_t)stream, a, sa, (int)n_slabs, per_wave, tiles_x, rows2);
                } else {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_persist_kernel<CH, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, p_lds);
                    hipLaunchKernelGGL((conv_stem_persist_kernel<CH, 4, 1>), dim3(g), dim3(256), p_lds, (hipStream_t)stream, a, sa, (int)n_slabs, per_wave, tiles_x, rows2);
                }
                return bflow::launch_status("conv_stem(persistent)");
            }
        }
        const int epi = tr ? 2 * 4 * 64 * 4 : (2 * 4 * 64 + 4 * 2 * 32 * CONV_STG_STRIDE) * 4;
        const int lds1 = body > epi ? body : epi;
#define LAUNCH_STEM_ROWS(TRR, GENN)                                                                                    \
    {                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_rows_kernel<7, 2, TRR, GENN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1); \
        hipLaunchKernelGGL((conv_stem_rows_kernel<7, 2, TRR, GENN>), grid, dim3(CT), lds1, (hipStream_t)stream, a, sa); \
    }
        if (sa.general) { if (tr) LAUNCH_STEM_ROWS(true, true) else LAUNCH_STEM_ROWS(false, true) }
        else { if (tr) LAUNCH_STEM_ROWS(true, false) else LAUNCH_STEM_ROWS(false, false) }
#undef LAUNCH_STEM_ROWS
        return bflow::launch_status("conv_stem(rows)");
    }
    if (!no_direct && a.out_f32 && !a.oh) {     // fp32 (+ statistics) output: transposed accumulators, direct stores (see conv_epilogue_direct)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel<7, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((conv_stem_kernel<7, 2, true>), grid, dim3(CT), lds, (hipStream_t)stream, a, sa);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel<7, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((conv_stem_kernel<7, 2>), grid, dim3(CT), lds, (hipStream_t)stream, a, sa);
    }
    return bflow::launch_status("conv_stem");
}
