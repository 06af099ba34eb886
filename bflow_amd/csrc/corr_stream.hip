// K5, streaming form: the all-pairs correlation volume as a register-stationary persistent kernel (gfx950 only).
// Replaces CorrComputation._corr_dot_prod_util, models/raft_utils/corr.py:264-272 (+ the 1-to-N / M-to-N reshapes :237-262).
//
// Why another structure than the 256x128 tile kernel of split_gemm.hip: at DSEC size the volume (368.6 MB) costs ~50 us of HBM
// writes and the 3-pass split-fp16 contraction ~60-85 us of matrix-core time (2.4 GHz ... the ~1.65 GHz this kernel sustains), so
// the kernel only approaches either roof if the matrix cores run WHILE the stores drain.  A tile kernel alternates a k-loop (stores
// idle) with an epilogue (matrix cores idle) in lock step on every CU.  Here nothing alternates:
//   * one persistent 8-wave workgroup per CU.  A wave keeps 32 TARGET columns (all D <= 256 channels, hi and lo planes) in 128
//     VGPRs for as long as it stays on its 256-column panel: the stationary operand costs neither LDS space nor LDS reads;
//   * the REFERENCE rows stream through a 4-slot LDS ring in chunks of 32 rows (32 KB: all channels, hi + lo), fetched by LDS-DMA
//     (buffer_load ... lds, 16 B / lane, source address pre-swizzled -> XOR-swizzled 64-B rows, conflict-free ds_read_b128), three
//     chunks ahead, one 1-KB piece per k-step;
//   * per chunk a wave runs 16 k-steps x 3 MFMAs (hi*hi, hi*lo, lo*hi) into one 32x32 accumulator pair while the PREVIOUS chunk's
//     accumulators are scaled and stored, one register (two full 128-B lines) per k-step, between the MFMAs; the fragments of the
//     next k-step (and, across the chunk boundary, of the next chunk) are read ahead.  The two waves of a SIMD keep its matrix
//     pipe busy for each other; one s_barrier per chunk is the only synchronisation.  A workgroup writes 32 rows x 1 KB per chunk.
// Work = segments (panel, chunk range): an XCD owns a contiguous range of the panel list and its workgroups walk the reference rows in
// LOCKSTEP, each on its own panel (round 3; see the comment at the segment arithmetic), so that a 32-KB row chunk is fetched from the
// fabric once per XCD and served to the other 31 workgroups by the L2 whatever the write stream evicts; the volume stores are
// non-temporal for the same reason.  A workgroup changes its panel (128 registers, 256 KB) only between rounds.
//
// vmcnt bookkeeping (gfx9 counts loads AND stores in vmcnt and retires them in order): per chunk a wave issues PW = KB/2 DMA pieces
// and exactly 16 buffer stores (out-of-range ones are dropped by the buffer bounds check, not skipped), so "my pieces of chunk i+1
// have landed" is a fixed s_waitcnt vmcnt(N) at the top of chunk i (see VMCNT_TOP).
//
// Measured on MI355X (round 2, tools/k5_probe.py): C2 (T=4, N=4800) 129-141 us vs 199-211 us for the tile kernel; steady state
// 7150 cycles per 64 reference rows per CU against a matrix-core floor of 6144 (stores cost ~900 of the difference: with them
// removed the same loop runs at 6250), 18 k cycles of prologue (first panel), 1.65 GHz sustained.
#include <cstdlib>
#include <type_traits>
#include "common.h"

// STREAM_ABL (tools/k5_ablate.sh only; timing builds with WRONG results): 1 no stores, 2 no in-loop DMA, 3 no barrier / DMA wait,
// 4 no MFMA, 5 no fragment reads
#ifndef STREAM_ABL
#define STREAM_ABL 0
#endif
// STREAM_STORE_AUX (tools A/B): cache-policy bits of the volume stores (gfx950 buffer aux: 1 sc0, 2 nt, 16 sc1)
#ifndef STREAM_STORE_AUX
#define STREAM_STORE_AUX 2   // nt: the write stream must not evict the reference rows the XCD re-reads (C5, fp8 cross terms: 2242 -> 1653 us)
#endif
#ifndef STREAM_POOL_AUX
#define STREAM_POOL_AUX 0
#endif
#ifdef STREAM_STAMPS   // tools/k5_ablate.sh "stamps" build: s_memtime stamps of wave 0 of every workgroup (64 x u64 per workgroup)
static unsigned long long* g_stamp_buf = nullptr;
extern "C" __attribute__((visibility("default"))) void bflow_k5_set_stamp_buffer(void* p) { g_stamp_buf = (unsigned long long*)p; }
#define STAMP(i)                                                                                  \
    if (a.stamps && tid == 0 && (i) < 62) a.stamps[blockIdx.x * 64 + (i)] = __builtin_readcyclecounter();
#define STAMP_RT(i) \
    if (a.stamps && tid == 0) a.stamps[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define STAMP_RT(i)
#define STAMP(i)
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr float LO_INV = 1.0f / 2048.0f;
constexpr int COLS_WG = 256;      // target columns held in registers per workgroup: 8 waves x 32
constexpr int CHUNK = 32;         // reference rows per ring slot
constexpr int SLOTS = 4;
constexpr unsigned OOB = 0x80000000u;   // byte offset beyond any volume slab (N*N*4 < 2^31 is checked on the host)

struct StreamArgs {
    const _Float16 *f1h, *f1l, *f2h, *f2l;
    float* out;
    int T, B, N, Np;
    long long f1_tstride;   // elements between the per-target f1 blocks (0 = one reference shared by all targets)
    float scale;            // POW2: 1/sqrt(D) (exact) ; else sqrt(D)
    int JP;                 // column panels per target matrix = ceil(N / 256)
    int CH2;                // pairs of 32-row chunks = ceil(N / 64)
    int n_pan;              // panels = T * B * JP
    int ph, pw;             // > 0: tiled planes (bflow_hip.h: 4x8 tiles); 0: row-major (N, N) slabs
    int PS;                 // elements of one plane of the volume: N (row-major) or tiles * 32
    // fused K6, level 0 -> 1 (corr.py:108-125,297-305): targets with more than one pyramid level get the 2 x 2 mean of their level-0 planes
    // written next to the volume (tiled planes only): pool_out (T1, B, N, PS1) in the volume's element type, pool_index[t] = row of target t
    // in it or -1
    void* pool_out;
    int pool_index[8];
    int PS1;                // elements of one level-1 plane: tiles1 * 32
    unsigned long long* stamps;   // STREAM_STAMPS builds only
};

struct Item {   // wave-uniform description of one (panel, chunk): 32 reference rows i0.. against the 256 columns of panel (t, b, jp)
    int pan, t, b, jp, c, cend, i0;   // cend: end of the chunk range of the segment the item belongs to
};

struct Cursor {   // position in the workgroup's item list (divisions only at a segment change)
    int seg, pan, t, b, jp, c, cend;
};

// MODE (arithmetic of the contraction):
//   M_SPLIT  three fp16 MFMA passes per k-step on (hi, lo) pairs: hi*hi + (hi*lo + lo*hi) 2^-11 -- fp32-class products (the default);
//   M_F16    (BASELINE configs[4] "fp16 MFMA correlation") plain fp16 operands (the hi planes alone), ONE MFMA per k-step: a third of the
//            matrix-core work; the second planes are neither loaded nor staged;
//   M_X8     hi*hi on the fp16 rate + BOTH cross terms of a 32-channel block in ONE v_mfma_f32_32x32x64_f8f6f4 (e4m3, unit scales, twice the
//            fp16 rate): the second plane of an operand is its "x8" plane -- per (row, 32-channel block) 64 B = [hi8 x 32 | lo8 x 32],
//            bflow_split_to_x8 -- so that the K = 64 of the instruction is [A_hi8 | A_lo8] . [B_lo8 | B_hi8]: lanes 0-31 (first k half) hold
//            hi8 of the streamed row / lo8 of the stationary column, lanes 32-63 the opposite.  Two matrix-pipe units per block instead
//            of three (1024 instead of 1536 cycles per chunk and wave) and less power per product (tools/micro/fp8_cross.hip: the 3-pass
//            stream runs at 1.41-1.65 GHz on random data, this one at 1.79 GHz: 1.64x in wall time).  The cross terms only carry 2^-11 of
//            the product, so their 2^-4 operand rounding leaves ~2^-16 per product (measured in tests/test_hip_parity.py).
// ST16: the volume is stored as fp16 (half the bytes) instead of fp32.
enum { M_SPLIT = 0, M_F16 = 1, M_X8 = 2 };
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// POOL: the fused level-1 pooling epilogue (one more buffer store per accumulator register, dropped by the bounds check for the targets
// that have no level 1 -- the counted vmcnt waits need the same number of operations in every chunk).
template <int KB, bool POW2, int MODE, bool ST16, bool POOL>
__global__ __launch_bounds__(512, 2) void corr_stream_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // SLOTS x (4*KB KB) ring; the only shared object
    constexpr bool F16 = MODE == M_F16, X8 = MODE == M_X8;
    constexpr int NS = 2 * KB;                 // k16 steps per chunk
    constexpr int NPL = F16 ? 1 : 2;           // operand planes
    constexpr int PW = NPL * KB / 4;           // DMA pieces (1 KB) per wave per chunk
    constexpr int SLOT_BYTES = NPL * 2 * KB * 1024;  // 32 rows x D x planes fp16
    constexpr int PLANE_BYTES = 2 * KB * 1024;
    constexpr int OB = ST16 ? 2 : 4;           // bytes per volume element
    static_assert(PW >= 1, "F16 needs D >= 128");
    constexpr int ST_PER_STEP = (16 / NS > 0 ? 16 / NS : 1);
    constexpr int OPS_PER_REG = POOL ? 2 : 1;      // buffer stores per accumulator register
    static_assert(16 % NS == 0 || NS % 16 == 0, "k-steps and accumulator registers must divide");
    static_assert(NS <= 16, "D <= 256");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int D = KB * 32;

    // ---- this workgroup's work: SEGMENTS (panel, chunk range), chunks handed out in PAIRS (64 rows: the two accumulator sets alternate
    // roles).  XCD x owns a contiguous range of the panel list (panel-major = target-major: XCD mates stream the same reference matrix).
    // Its workgroups run in LOCKSTEP over the reference rows: in a full round workgroup i takes panel (round * per_x + i) and walks ALL
    // chunks; the P mod per_x panels of the last round are shared out 2-D -- panel j gets floor or ceil(per_x / R) workgroups which split its
    // chunks evenly.  Workgroups that stream the same rows at the same time share them through the XCD's L2 whatever the write stream does
    // to its contents (with equal contiguous ranges of a panel-major list, as in round 2, the mates of an XCD are spread over the whole row
    // slice, which the kernel's own stores keep evicting: 276 MB of fabric reads for 24.6 MB of operands at DSEC size), and a workgroup
    // changes its panel only between rounds.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int p_lo = (int)((long long)xcd * a.n_pan / 8), p_hi = (int)((long long)(xcd + 1) * a.n_pan / 8);
    const int P = p_hi - p_lo, full = P / per_x, R = P - full * per_x;
    int t_pan = 0, t_c0 = 0, t_c1 = 0;      // this workgroup's share of the last (partial) round
    if (R > 0) {
        const int base = per_x / R, extra = per_x - base * R;      // the first `extra` panels get base + 1 workgroups
        int j, k, S;
        if (idx < extra * (base + 1)) {
            j = idx / (base + 1);
            k = idx - j * (base + 1);
            S = base + 1;
        } else {
            const int i2 = idx - extra * (base + 1);
            j = extra + i2 / base;
            k = i2 - (i2 / base) * base;
            S = base;
        }
        t_pan = p_lo + full * per_x + j;
        t_c0 = 2 * (int)((long long)k * a.CH2 / S);
        t_c1 = 2 * (int)((long long)(k + 1) * a.CH2 / S);
    }
    const int n = full * 2 * a.CH2 + (t_c1 - t_c0);     // chunks of this workgroup: even
    if (n <= 0) return;

    Cursor cur;
    auto set_segment = [&](Cursor& k, int seg) {
        k.seg = seg;
        const int pan = seg < full ? p_lo + seg * per_x + idx : t_pan;
        k.pan = pan;
        k.t = pan / (a.B * a.JP);
        k.b = (pan / a.JP) % a.B;
        k.jp = pan % a.JP;
        k.c = seg < full ? 0 : t_c0;
        k.cend = seg < full ? 2 * a.CH2 : t_c1;
    };
    set_segment(cur, 0);
    auto item_of = [&](const Cursor& k) -> Item {
        Item r;
        r.pan = k.pan;
        r.t = k.t;
        r.b = k.b;
        r.jp = k.jp;
        r.c = k.c;
        r.cend = k.cend;
        r.i0 = k.c * CHUNK;
        return r;
    };
    auto advance = [&](Cursor& k) {
        if (++k.c == k.cend) set_segment(k, k.seg + 1);   // next round (never called past the last item)
    };
    int fetched = 0;   // items handed out by next_item(); past the end the LAST item is repeated (its slot is never read)
    auto next_item = [&]() -> Item {
        if (fetched > 0 && fetched < n) advance(cur);
        ++fetched;
        return item_of(cur);
    };

    // ---- addressing: every global access goes through a buffer descriptor with a wave-uniform (SGPR) matrix / k-block offset and
    // ONE per-lane VGPR offset, so that the address arithmetic costs scalar instructions and almost no vector registers
    const unsigned f1_bytes = (unsigned)(((a.f1_tstride ? (long long)a.T : 1LL) * a.B * a.Np * D) * 2);
    const unsigned f2_bytes = (unsigned)(((long long)a.T * a.B * a.Np * D) * 2);
    const __amdgpu_buffer_rsrc_t r1h = __builtin_amdgcn_make_buffer_rsrc((void*)a.f1h, 0, f1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1l = __builtin_amdgcn_make_buffer_rsrc((void*)a.f1l, 0, f1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2h = __builtin_amdgcn_make_buffer_rsrc((void*)a.f2h, 0, f2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2l = __builtin_amdgcn_make_buffer_rsrc((void*)a.f2l, 0, f2_bytes, 0x00020000);
    const unsigned kb_bytes = (unsigned)a.Np * 64u;   // one 32-deep k-block of one matrix

    // ---- LDS-DMA of one chunk: piece q = wave*PW + j -> (plane, k-block, 16-row half); lane -> (row, 16-B slot) with the source
    // chunk swizzled so that the lane-linear LDS image holds logical chunk c of row r at slot c ^ ((r >> 2) & 3)
    const unsigned dma_lane = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    auto issue_piece = [&](const Item& im, int slot, int j) {
        const unsigned mat = (unsigned)((im.t * a.f1_tstride + (long long)im.b * a.Np * D) * 2) + (unsigned)im.i0 * 64u;
        const int q = wave * PW + j;
        const int plane = q / (2 * KB), kb = (q % (2 * KB)) >> 1, half = q & 1;
        const unsigned so = mat + (unsigned)kb * kb_bytes + (unsigned)half * 1024u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(plane ? r1l : r1h, (lptr_t)(lds + slot * SLOT_BYTES + q * 1024), 16, dma_lane, so, 0, 0);
    };
    auto issue = [&](const Item& im, int slot) {
#pragma unroll
        for (int j = 0; j < PW; ++j) issue_piece(im, slot, j);
    };

    // ---- target panel -> registers: lane holds column (wave*32 + l31) of the panel, 8 consecutive k per k16 step
    half8 Bh[NS], Bl[NS];                      // M_X8: Bl[2 kb], Bl[2 kb + 1] = the 32 x8 bytes of block kb (lo8 for lanes 0-31, hi8 for 32-63)
    // fused pooling: this lane's role in the level-1 store of the wave's tile (set per panel).  The 2 x 2 mean of tile (ty, tx) is 2 x 4
    // values of level-1 tile (ty / 2, tx / 2), held by the lanes with even (y, x).  The other lanes are idle -- they write the ZEROS of the
    // level-1 positions no level-0 tile reaches (below the last tile row / right of the last tile column when their count is odd), so that
    // every element of the level-1 planes is written by this kernel (the look-up weighs pad positions with 0 and needs them finite).
    unsigned p1lane = OOB;                     // byte offset inside a level-1 plane, or OOB
    bool p1pool = false;                       // this lane stores a pooled value (else 0)
    auto pool_roles = [&](int jp) {
        const int tw0 = (a.pw + 7) >> 3, th0 = (a.ph + 3) >> 2, h1 = a.ph >> 1, w1 = a.pw >> 1;
        const int tw1 = (w1 + 7) >> 3, th1 = (h1 + 3) >> 2;
        const int tile = jp * 8 + wave, ty = tile / tw0, tx = tile - ty * tw0;
        const int y = l31 >> 3, x = l31 & 7;
        const bool below_missing = !(ty & 1) && ty + 1 >= th0, right_missing = !(tx & 1) && tx + 1 >= tw0;
        const bool oy = y & 1, ox = x & 1;
        const bool active = ty < th0 && (ty >> 1) < th1 && (tx >> 1) < tw1 && (!oy || below_missing) && (!ox || right_missing);
        const int y1l = (oy ? 2 : (ty & 1) * 2) + (y >> 1), x1l = (ox ? 4 : (tx & 1) * 4) + (x >> 1);
        p1lane = active ? (unsigned)((((ty >> 1) * tw1 + (tx >> 1)) * 32 + y1l * 8 + x1l) * OB) : OOB;
        p1pool = !oy && !ox && ty * 2 + (y >> 1) < h1 && tx * 4 + (x >> 1) < w1;
    };
    auto load_panel = [&](int tb, int jp) {
        if (POOL) pool_roles(jp);
        int col = jp * COLS_WG + wave * 32 + l31;   // row-major planes: 32 consecutive target pixels per wave
        if (a.pw > 0) {
            // tiled planes: the wave's 32 columns are the 4 x 8 pixels of ONE tile, so that an accumulator register is one 128-B line of
            // the tiled plane.  Pad positions of edge tiles take a clamped (valid) pixel: they hold finite values nobody weights.
            const int tw = (a.pw + 7) >> 3, tile = jp * 8 + wave;
            const int ty = tile / tw, tx = tile - ty * tw;
            const int y = min(ty * 4 + (l31 >> 3), a.ph - 1), x = min(tx * 8 + (l31 & 7), a.pw - 1);
            col = y * a.pw + x;
        }
        col = col < a.Np ? col : a.Np - 1;   // columns >= N (row-major) / tiles past the plane are never stored
        const unsigned vo = (unsigned)col * 64u + (unsigned)kh * 16u;
        const unsigned vo8 = (unsigned)col * 64u + (unsigned)(1 - kh) * 32u;   // stationary side: first k half = lo8 (bytes 32..63), second = hi8
        const unsigned mat = (unsigned)(((long long)tb * a.Np * D) * 2);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned so = mat + (unsigned)(s >> 1) * kb_bytes;
            Bh[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r2h, vo + (s & 1) * 32, so, 0));
            if (X8) Bl[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r2l, vo8 + (s & 1) * 16, so, 0));
            else if (!F16) Bl[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r2l, vo + (s & 1) * 32, so, 0));
        }
    };

    // fragment read offsets inside a slot (bytes): row l31, logical chunk (s&1)*2 + kh
    const int sw = (l31 >> 2) & 3;
    const int fo_even = l31 * 64 + ((kh ^ sw) << 4), fo_odd = l31 * 64 + (((2 + kh) ^ sw) << 4);
    // second plane: lo fragments of the same k-step, or (M_X8) the streamed row's x8 bytes: chunks 2 kh, 2 kh + 1 (hi8 for lanes 0-31, lo8 for 32-63)
    const int fo2_even = X8 ? l31 * 64 + (((2 * kh) ^ sw) << 4) : fo_even, fo2_odd = X8 ? l31 * 64 + (((2 * kh + 1) ^ sw) << 4) : fo_odd;

    // ---- store side: one accumulator register = rows (r&3) + 8*(r>>2) + 4*kh of the wave's 32, 32 consecutive columns
    auto store_base = [&](const Item& im, __amdgpu_buffer_rsrc_t& rs) -> unsigned {
        const int row = im.i0 + 4 * kh, col = im.jp * COLS_WG + wave * 32 + l31;   // tiled: col = tile * 32 + position in the tile
        char* slab = reinterpret_cast<char*>(a.out) + (long long)(im.t * a.B + im.b) * a.N * a.PS * OB;
        rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, a.N * a.PS * OB, 0x00020000);
        return col < a.PS ? (unsigned)((row * a.PS + col) * OB) : OOB;   // rows >= N fall off the end of the slab by themselves
    };
    const unsigned rowstep = (unsigned)a.PS * (unsigned)OB;
    const unsigned rowstep1 = (unsigned)a.PS1 * (unsigned)OB;
    bool p1on = false, p1val = false;          // p1val: p1pool of the panel the pending stores belong to (latched with p1base)
    auto store_base1 = [&](const Item& im, __amdgpu_buffer_rsrc_t& rs, bool& on) -> unsigned {
        const int k1 = a.pool_index[im.t];
        on = k1 >= 0;
        p1val = p1pool;
        char* slab = reinterpret_cast<char*>(a.pool_out) + (long long)((on ? k1 : 0) * a.B + im.b) * a.N * a.PS1 * OB;
        rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, on ? a.N * a.PS1 * OB : 0, 0x00020000);   // no level 1: every store is dropped
        return p1lane != OOB ? (unsigned)((im.i0 + 4 * kh) * a.PS1 * OB) + p1lane : OOB;
    };

    __amdgpu_buffer_rsrc_t p1rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);
    unsigned p1base = OOB;
    auto put = [&](float v, const __amdgpu_buffer_rsrc_t& rs, unsigned off, auto aux) {
        constexpr int AUX = decltype(aux)::value;
        if (ST16) {
            const _Float16 h = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rs, off, 0, AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, AUX);
        }
    };
    auto store_reg = [&](float hh_, float xx_, const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned off1) {
        float v = F16 ? hh_ : fmaf(xx_, LO_INV, hh_);
        v = POW2 ? v * a.scale : v / a.scale;
        if (STREAM_ABL == 1) {
            asm volatile("" ::"v"(v), "v"(off));
            return;
        }
        if (ST16) v = (float)(_Float16)fminf(fmaxf(v, -65504.f), 65504.f);     // the pooled value is the mean of the STORED values
        put(v, rs, off, std::integral_constant<int, STREAM_STORE_AUX>{});
        if (POOL && p1on) {   // wave-uniform: only the chunks of targets with a level 1 carry the second store (see VMCNT_TOP / pooled history)
            float m = 0.f;
            {   //  F.avg_pool2d's order (corr.py:119): ((q[2y][2x] + q[2y][2x+1]) + q[2y+1][2x]) + q[2y+1][2x+1], then / 4
                const int vi = __builtin_bit_cast(int, v);
                const int sw = __builtin_amdgcn_mov_dpp(vi, 0xB1 /* quad_perm [1,0,3,2]: lane ^ 1 */, 0xf, 0xf, false);
                const float c = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0x128 /* row_ror:8: lane ^ 8 */, 0xf, 0xf, false));
                const float d = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(sw, 0x128, 0xf, 0xf, false));
                m = p1val ? (((v + __builtin_bit_cast(float, sw)) + c) + d) * 0.25f : 0.f;
            }
            // default cache policy: a level-1 line is assembled from 16-B pieces of four waves (two of them in another workgroup); the L2
            // merges them, a non-temporal partial line goes out as it is
            put(m, p1rs, off1, std::integral_constant<int, STREAM_POOL_AUX>{});
        }
    };

    // ---- pipeline state --------------------------------------------------------------------------------------------------
    STAMP(0)
    STAMP_RT(62)
    Item q0 = next_item(), q1 = next_item(), q2 = next_item();
    issue(q0, 0);
    issue(q1, 1);
    issue(q2, 2);

    f32x16 X_hh, X_xx, Y_hh, Y_xx;
#pragma unroll
    for (int r = 0; r < 16; ++r) Y_hh[r] = Y_xx[r] = 0.f;
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);   // nothing to store yet: every store is dropped
    unsigned pbase = OOB;

    // One chunk.  On entry fh[0] / fl[0] hold (or are about to receive) the k-step-0 fragments of chunk `it`, read at the end of the
    // previous chunk: the barrier at the top of chunk i orders "chunk i+1 has landed" (every wave waited for its own pieces of it
    // first), so chunk i+1 may be read before the NEXT barrier and the matrix cores restart right behind it.  The DMA pieces of chunk
    // it+3 go out one per k-step from step DMA_S0 on (their slot was last read in chunk it-1, which every wave left before this
    // chunk's barrier); per k-step: prefetch the next fragments, 3 MFMAs, one accumulator register of the previous chunk scaled and
    // stored.  vmcnt at the top = ops issued after this wave's last piece of chunk it+1 (in chunk it-2): the stores behind it there +
    // the 16 stores and PW pieces of chunk it-1.
    constexpr int DMA_S0 = NS >= 8 ? 2 : 1;
    // fused pooling: a chunk whose stores belong to a target with a level 1 issues 32 stores instead of 16.  The counted wait uses the larger
    // number only when BOTH chunks whose operations it skips over carried them (pm1 && pm2); otherwise the smaller one, which then waits for a
    // few stores more than necessary (never fewer: retirement is in order) -- that happens for two chunks at a change of target only.
    constexpr int VMCNT_TOP = (NS - (DMA_S0 + PW)) * ST_PER_STEP + 16 + PW;
    constexpr int VMCNT_TOP2 = (NS - (DMA_S0 + PW)) * ST_PER_STEP * OPS_PER_REG + 16 * OPS_PER_REG + PW;
    bool pm1 = false, pm2 = false;                 // pooled stores issued in the previous / the one before the previous chunk iteration
    static_assert(DMA_S0 + PW <= NS && VMCNT_TOP2 < 64, "DMA schedule");
    half8 fh[2], fl[2], ale;
#define STREAM_STEP(CH_, CX_, PH_, PX_)                                                                                     \
    {                                                                                                                       \
        if (STREAM_ABL != 3) {                                                                                              \
            if (POOL && pm1 && pm2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMCNT_TOP2) : "memory");                       \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMCNT_TOP) : "memory");                                           \
            __builtin_amdgcn_s_barrier();                                                                                   \
        }                                                                                                                   \
        if (POOL) { pm2 = pm1; pm1 = p1on; }   /* this iteration's stores carry the pooled ones iff p1on (latched at the end of the last one) */ \
        const Item q3 = next_item();                                                                                        \
        const char* sb = lds + (it & 3) * SLOT_BYTES;                                                                       \
        const char* sn = lds + ((it + 1) & 3) * SLOT_BYTES;                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) CH_[r] = CX_[r] = 0.f;                                               \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                                    \
            {   /* fragments of the next k-step (the next chunk's step 0 behind the last one) */                           \
                const int sn_ = (s + 1) % NS;                                                                               \
                const char* fb = (s + 1 < NS ? sb : sn) + (sn_ >> 1) * 2048;                                                \
                if (STREAM_ABL != 5) {                                                                                      \
                    fh[(s + 1) & 1] = *reinterpret_cast<const half8*>(fb + ((sn_ & 1) ? fo_odd : fo_even));                \
                    if (!F16) fl[(s + 1) & 1] = *reinterpret_cast<const half8*>(fb + PLANE_BYTES + ((sn_ & 1) ? fo2_odd : fo2_even)); \
                }                                                                                                           \
            }                                                                                                               \
            const half8 ah = fh[s & 1], al = fl[s & 1];                                                                     \
            if (X8 && !(s & 1)) ale = al;                                                                                   \
            if (STREAM_ABL != 4) {                                                                                          \
                CH_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[s], CH_, 0, 0, 0);                                      \
                if (X8) {                                                                                                   \
                    if (s & 1) {                                                                                            \
                        const i32x4 a0 = __builtin_bit_cast(i32x4, ale), a1 = __builtin_bit_cast(i32x4, al);                \
                        const i32x4 b0 = __builtin_bit_cast(i32x4, Bl[s - 1]), b1 = __builtin_bit_cast(i32x4, Bl[s]);       \
                        const i32x8 a8 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};                          \
                        const i32x8 b8 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};                          \
                        CX_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, CX_, 0, 0, 0, 0, 0, 0);               \
                    }                                                                                                       \
                } else if (!F16) {                                                                                          \
                    CX_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[s], CX_, 0, 0, 0);                                  \
                    CX_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[s], CX_, 0, 0, 0);                                  \
                }                                                                                                           \
            } else {                                                                                                        \
                CH_[s] += (float)ah[0] + (float)Bh[s][1];                                                                   \
                CX_[s] += (float)al[0] + (float)Bl[s][1];                                                                   \
            }                                                                                                               \
            _Pragma("unroll") for (int u = 0; u < ST_PER_STEP; ++u) {                                                       \
                const int r = s * ST_PER_STEP + u;                                                                          \
                store_reg(PH_[r], PX_[r], prs, pbase + (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep,                        \
                          p1base + (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep1);                                          \
            }                                                                                                               \
            if (STREAM_ABL != 2 && s >= DMA_S0 && s < DMA_S0 + PW) issue_piece(q3, (it + 3) & 3, s - DMA_S0);              \
        }                                                                                                                   \
        pbase = store_base(q0, prs);                                                                                        \
        if (POOL) p1base = store_base1(q0, p1rs, p1on);                                                                     \
        q0 = q1;                                                                                                            \
        q1 = q2;                                                                                                            \
        q2 = q3;                                                                                                            \
        ++it;                                                                                                               \
    }

    int it = 0;
    while (it < n) {
        // (next) panel: its 256 target columns -> registers.  The compiler-visible vmcnt(0) covers the panel loads AND every DMA piece /
        // store issued so far, so no wait for the panel registers appears inside the chunk loop (a static vmcnt there would also wait
        // for the in-flight stores of every chunk) and the first chunks of the panel need no counted wait of their own.
        load_panel(q0.t * a.B + q0.b, q0.jp);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (it == 0) {   // first chunk: nobody prefetched its k-step-0 fragments
            __builtin_amdgcn_s_barrier();
            fh[0] = *reinterpret_cast<const half8*>(lds + fo_even);
            if (!F16) fl[0] = *reinterpret_cast<const half8*>(lds + PLANE_BYTES + fo2_even);
        }
        STAMP(1 + it)
        const int stop = it + (q0.cend - q0.c);        // chunks left in this segment (one panel): even
        do {
            STREAM_STEP(X_hh, X_xx, Y_hh, Y_xx)
            STREAM_STEP(Y_hh, Y_xx, X_hh, X_xx)
            STAMP(1 + it)
        } while (it < stop);
    }
#undef STREAM_STEP

    // ---- drain: the last chunk's accumulators
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        store_reg(Y_hh[r], Y_xx[r], prs, pbase + (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep, p1base + (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMA pieces must not outlive the workgroup's LDS
    STAMP(61)
    STAMP_RT(63)
}

template <int KB, int MODE, bool ST16, bool POOL>
int launch_kb(const StreamArgs& a, bool pow2, hipStream_t s) {
    constexpr bool F16 = MODE == M_F16;
    const int lds = SLOTS * (F16 ? 2 : 4) * KB * 1024;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        // one persistent workgroup per CU (128 KB of LDS, 225 VGPRs); the fp16 variant needs half of both, so two per CU hide each other's
        // barriers and store bursts (BFLOW_CORR_F16_WGS: 256 | 512, tools A/B)
        static const int f16_wgs = [] { const char* e = getenv("BFLOW_CORR_F16_WGS"); return e && atoi(e) == 256 ? 256 : 512; }();
        hipLaunchKernelGGL(kern, dim3(F16 ? f16_wgs : 256), dim3(512), lds, s, a);
    };
    if (pow2) go(corr_stream_kernel<KB, true, MODE, ST16, POOL>);
    else go(corr_stream_kernel<KB, false, MODE, ST16, POOL>);
    return bflow::launch_status(MODE == M_F16 ? "corr_build(stream, fp16 operands)" : MODE == M_X8 ? "corr_build(stream, fp8 cross terms)" : "corr_build_split(stream)");
}

template <int MODE, bool ST16, bool POOL = false>
int launch_mode(const StreamArgs& a, int D, bool pow2, hipStream_t s) {
    if (D == 256) return launch_kb<8, MODE, ST16, POOL>(a, pow2, s);
    if (D == 128) return launch_kb<4, MODE, ST16, POOL>(a, pow2, s);
    if constexpr (MODE == M_SPLIT && !ST16 && !POOL) return launch_kb<2, MODE, ST16, false>(a, pow2, s);   // D = 64: the fp32 split volume only
    return BFLOW_E_ARG;
}

}  // namespace

namespace bflow {

// true when the streaming kernel supports the shape (D in {64, 128, 256}); otherwise the caller uses the tile kernel
bool corr_stream_supported(int T, int B, int D, int N, int Np) {
    // 32-bit buffer offsets: one volume slab (tiled planes are < 1.2x larger: 2^30.7 B) and each operand plane must stay below 2 GiB
    return (D == 64 || D == 128 || D == 256) && (long long)N * N * 4 < (3LL << 29) && (long long)T * B * Np * D * 2 < (1LL << 31);
}

// plane_h x plane_w > 0 (= N): the volume is written as TILED planes (see bflow_corr_build_split_tiled); 0, 0: row-major (T, B, N, N)
// arithmetic: 0 split (f*_lo = lo planes), 1 fp16 operands (f*_lo ignored, may be null), 2 fp8 cross terms (f*_lo = x8 planes, bflow_split_to_x8);
// out_fp16: `out` is an fp16 volume.  Everything but (split, fp32) needs D in {128, 256}.
// pool_out / pool_index (or null): the fused level-1 pooling epilogue (tiled planes; (split | split8, fp32) and (fp16, fp16) only; D in {128, 256})
int corr_stream_launch(const void* f1_hi, const void* f1_lo, const void* f2_hi, const void* f2_lo, void* out, int T, int B, int D, int N, int Np,
                       long long f1_target_stride, int plane_h, int plane_w, int arithmetic, bool out_fp16, void* pool_out, const int* pool_index,
                       hipStream_t stream) {
    StreamArgs a;
    a.f1h = (const _Float16*)f1_hi;
    a.f1l = (const _Float16*)f1_lo;
    a.f2h = (const _Float16*)f2_hi;
    a.f2l = (const _Float16*)f2_lo;
    a.out = (float*)out;
    a.T = T;
    a.B = B;
    a.N = N;
    a.Np = Np;
    a.f1_tstride = f1_target_stride;
    const float sq = sqrtf((float)D);
    int e;
    const bool pow2 = frexpf(sq, &e) == 0.5f;   // sqrt(D) a power of two: x / sqrt(D) == x * (1 / sqrt(D)) bit for bit
    a.scale = pow2 ? 1.0f / sq : sq;
    a.ph = plane_h;
    a.pw = plane_w;
    a.PS = plane_w > 0 ? ceil_div(plane_h, 4) * ceil_div(plane_w, 8) * 32 : N;
    a.JP = ceil_div(a.PS, COLS_WG);
    a.CH2 = ceil_div(N, 2 * CHUNK);
    a.n_pan = T * B * a.JP;
#ifdef STREAM_STAMPS
    a.stamps = g_stamp_buf;
#else
    a.stamps = nullptr;
#endif
    a.pool_out = pool_out;
    a.PS1 = plane_w > 0 ? ceil_div(plane_h / 2, 4) * ceil_div(plane_w / 2, 8) * 32 : 0;
    for (int t = 0; t < 8; ++t) a.pool_index[t] = (pool_out && pool_index && t < T) ? pool_index[t] : -1;
    if (pool_out) {
        if (plane_w <= 0 || plane_h < 2 || plane_w < 2 || T > 8 || D == 64 || (long long)N * a.PS1 * 4 >= (1LL << 31)) return BFLOW_E_ARG;
        switch (arithmetic * 2 + (out_fp16 ? 1 : 0)) {
            case 0: return launch_mode<M_SPLIT, false, true>(a, D, pow2, stream);
            case 3: return launch_mode<M_F16, true, true>(a, D, pow2, stream);
            case 4: return launch_mode<M_X8, false, true>(a, D, pow2, stream);
            default: return BFLOW_E_ARG;
        }
    }
    switch (arithmetic * 2 + (out_fp16 ? 1 : 0)) {
        case 0: return launch_mode<M_SPLIT, false>(a, D, pow2, stream);
        case 1: return launch_mode<M_SPLIT, true>(a, D, pow2, stream);
        case 2: return launch_mode<M_F16, false>(a, D, pow2, stream);
        case 3: return launch_mode<M_F16, true>(a, D, pow2, stream);
        case 4: return launch_mode<M_X8, false>(a, D, pow2, stream);
        case 5: return launch_mode<M_X8, true>(a, D, pow2, stream);
        default: return BFLOW_E_ARG;
    }
}

}  // namespace bflow
