// K1 / K2: event voxel grid (reference: VoxelGrid.convert + norm_voxel_grid, data/utils/representations.py:9-18,64-111).
//
// K1 (round 5) bins the events by grid tile, accumulates every tile in LDS in 64-bit fixed point and writes it once: no global
// atomics, no zero fill, run-to-run bit-identical (see the block comment above the kernels).
// K2 is three streaming passes (sum+count of non-zeros, sum of squared deviations, apply) with wavefront-shuffle +
// one fp64 atomic per block reductions; the scalar results stay on the device (graph-capture safe).
#include "common.h"

namespace {

// ---- K1: tile-binned, LDS-accumulated, deterministic ----------------------------------------------------------------------------------
// The grid is cut into BINS of CG channels x TH rows x 32 columns (one bin = one workgroup's LDS accumulator and one set of whole 128-B
// output rows).  An event belongs to every bin one of its 8 (float x/y) or 2 (integer x/y) neighbour cells lies in: 1 bin for most,
// 2 / 4 / 8 for events on a bin's upper borders (+24 % records at 15 x 480 x 640).  Four launches, no global atomics, no zero fill:
//   count    workgroup b owns the contiguous event chunk b: LDS histogram over the bins -> counts[b][bin]
//   scan     per bin the exclusive prefix over the chunks (in place) and the bin totals
//   place    workgroup b scans the totals (bin bases), adds its own prefix row = its private cursor per bin, and writes a 16-B record
//            (x, y, t_norm, value) per (event, bin) -- the rectification gather (f-1) happens here and in `count`
//   gather   workgroup = bin: its records add their contributions into 64-bit FIXED-POINT LDS cells (ds_add_u64, 2^-40 units), the
//            bin is converted (one rounding per cell) and stored once, as whole rows, zeros included.
// Integer addition is associative, so the result does not depend on the order in which records meet (run-to-run bit-identical,
// whatever `place` did), and it is the correctly rounded EXACT sum of the reference's own fp32 contributions (each product is formed
// with the reference's operations in the reference's order; a contribution below 2^-16 is truncated at 2^-40 absolute) -- closer to
// the real sum than the reference's sequential fp32 accumulation, from which it differs by that accumulation's round-off.
// Range: |cell sum| < 2^23 (8 M same-sign events on ONE cell).
enum { SRC_F32 = 0, SRC_I16 = 1, SRC_I32 = 2, SRC_RECT = 3 };

struct VoxGeo {
    int C, H, W;
    int cg_shift;      // channels per bin = 1 << cg_shift when ng > 1; ng == 1: the bin holds all C channels
    int CG;            // channels per bin
    int th_shift, TH;  // rows per bin (8 / 16 / 32); 32 columns per bin
    int ntx, nty;      // spatial bins
    int g_lo, g_cnt;   // channel groups of this slab
    int nbins;         // ntx * nty * g_cnt
    float denom, cm1;
    long long t0c;
};

struct VoxSrc {
    const void *x, *y, *pol;
    const long long* t;
    const float* rect;
    long long n;
    // Device-resident window (bflow_voxel_grid_rectified_window; null otherwise): {first event, event count, t0_center, t1_center} read by
    // the kernels at run time, so that ONE captured launch sequence serves every window of a recording.  x / y / pol / t then address the
    // whole recording of n_total events and `n` is the capacity the launch and its workspace were planned for.
    const long long* win;
    long long n_total;
};

struct VoxRec {        // 16 bytes
    union { float fx; int qm; };
    union { float fy; int qd; };
    float tn, value;
};

// the (up to) 2 x 2 x 2 bins of an event: per axis the first bin and, if the upper neighbour lies in another one, the second (-1 = none)
struct VoxBins { int tx[2], ty[2], tg[2]; };

__device__ __forceinline__ int vox_bin_index(const VoxGeo& g, int tg, int ty, int tx) { return ((tg - g.g_lo) * g.nty + ty) * g.ntx + tx; }

__device__ __forceinline__ void vox_axis(int lo, int limit, int shift, int out[2]) {
    const int a = (lo >= 0 && lo < limit) ? (lo >> shift) : -1;
    int b = (lo + 1 >= 0 && lo + 1 < limit) ? ((lo + 1) >> shift) : -1;
    if (b == a) b = -1;
    out[0] = a;
    out[1] = b;
}

__device__ __forceinline__ void vox_groups(const VoxGeo& g, int p0, int p1, bool v0, bool v1, int out[2]) {   // channel planes -> groups of this slab
    int a = v0 ? (p0 >> g.cg_shift) : -1, b = v1 ? (p1 >> g.cg_shift) : -1;
    if (a >= 0 && (a < g.g_lo || a >= g.g_lo + g.g_cnt)) a = -1;
    if (b >= 0 && (b < g.g_lo || b >= g.g_lo + g.g_cnt)) b = -1;
    if (b == a) b = -1;
    out[0] = a;
    out[1] = b;
}

// representations.py:58: int64 tensor - python int -> int64; / python int -> float32 true division; * (C - 1)
__device__ __forceinline__ float vox_tnorm(const VoxGeo& g, long long t) { return (float)(t - g.t0c) / g.denom * g.cm1; }

// An event as loaded: coordinates as 32-bit payloads (integer value or float bits), the polarity as the reference's `value`.
// Loading (vox_load, + vox_rectify for the DSEC map gather) and classifying (vox_classify) are separate so that a thread can have the
// loads of several events in flight before it touches the first one.
struct VoxRaw {
    long long t;
    int a, b;          // x, y
    float value;       // 2 * pol - 1 (representations.py:83)
    bool ok;
};

template <int SRC>
__device__ __forceinline__ VoxRaw vox_load(const VoxSrc& s, long long e, bool in_range) {
    VoxRaw r;
    r.ok = in_range;
    if (!in_range) { r.t = 0; r.a = r.b = 0; r.value = 0.f; return r; }
    r.t = s.t[e];
    if (SRC == SRC_F32) {
        r.a = reinterpret_cast<const int*>(s.x)[e];
        r.b = reinterpret_cast<const int*>(s.y)[e];
    } else if (SRC == SRC_I16) {
        r.a = reinterpret_cast<const short*>(s.x)[e];
        r.b = reinterpret_cast<const short*>(s.y)[e];
    } else if (SRC == SRC_I32) {
        r.a = reinterpret_cast<const int*>(s.x)[e];
        r.b = reinterpret_cast<const int*>(s.y)[e];
    } else {
        r.a = reinterpret_cast<const unsigned short*>(s.x)[e];
        r.b = reinterpret_cast<const unsigned short*>(s.y)[e];
    }
    r.value = 2.f * (SRC == SRC_RECT ? (float)reinterpret_cast<const unsigned char*>(s.pol)[e] : (float)reinterpret_cast<const signed char*>(s.pol)[e]) - 1.f;
    return r;
}

// BaseSubSequence._rectify_events (data/dsec/subsequence/base.py:137-143): rectify_map[y, x] -> (x', y'); raw coordinates outside the map
// (the reference asserts on them) drop the event and are counted
template <int SRC>
__device__ __forceinline__ void vox_rectify(const VoxGeo& g, const VoxSrc& s, VoxRaw& r, int* bad) {
    if (SRC != SRC_RECT || !r.ok) return;
    if (r.a >= g.W || r.b >= g.H) {
        if (bad) atomicAdd(bad, 1);
        r.ok = false;
        return;
    }
    const float2 xy = *reinterpret_cast<const float2*>(s.rect + ((long long)r.b * g.W + r.a) * 2);
    r.a = __float_as_int(xy.x);
    r.b = __float_as_int(xy.y);
}

// The record of an event and the bins it belongs to.  Returns false if it contributes nothing (in this slab).
template <int SRC>
__device__ __forceinline__ bool vox_classify(const VoxGeo& g, const VoxRaw& raw, VoxRec& r, VoxBins& b) {
    if (!raw.ok) return false;
    r.tn = vox_tnorm(g, raw.t);
    r.value = raw.value;
    const float tf = floorf(r.tn);
    const float tcl = fminf(fmaxf(tf, -4.f), (float)g.C + 4.f);
    const int t0 = (int)tcl;
    const bool t_sane = tf == tcl;
    if (SRC == SRC_I16 || SRC == SRC_I32) {
        // representations.py:85-94: only the time bin is masked; x / y enter through the FLAT index ht*wd*t + wd*y + x handed to Tensor.put_,
        // which accepts [-numel, numel) (negative = from the end) and raises outside: an index put_ would accept lands where put_ puts it,
        // one it would raise on is dropped.  With q = wd*y + x = qd * HW + qm (0 <= qm < HW) the index of time bin tl is (tl + qd) * HW + qm:
        // pixel qm of plane p = tl + qd (+ C if negative), accepted iff -C <= tl + qd < C.
        const long long HW = (long long)g.H * g.W, q = (long long)raw.b * g.W + raw.a;
        long long qd, qm;
        if (q >= 0 && q < HW) { qd = 0; qm = q; }      // the sensor's own coordinates: no 64-bit division
        else {
            qd = q / HW;
            qm = q - qd * HW;
            if (qm < 0) { qm += HW; --qd; }
        }
        qd = qd < -(1 << 20) ? -(1 << 20) : (qd > (1 << 20) ? (1 << 20) : qd);
        r.qm = (int)qm;
        r.qd = (int)qd;
        int p0 = t0 + r.qd, p1 = t0 + 1 + r.qd;
        const bool v0 = t_sane && t0 >= 0 && t0 < g.C && p0 >= -g.C && p0 < g.C;
        const bool v1 = t_sane && t0 + 1 >= 0 && t0 + 1 < g.C && p1 >= -g.C && p1 < g.C;
        p0 += p0 < 0 ? g.C : 0;
        p1 += p1 < 0 ? g.C : 0;
        vox_groups(g, p0, p1, v0, v1, b.tg);
        const int py = r.qm / g.W, px = r.qm - py * g.W;
        b.tx[0] = px >> 5; b.tx[1] = -1;
        b.ty[0] = py >> g.th_shift; b.ty[1] = -1;
        return (b.tg[0] & b.tg[1]) >= 0;
    } else {
        r.fx = __int_as_float(raw.a);
        r.fy = __int_as_float(raw.b);
        const float xf = floorf(r.fx), yf = floorf(r.fy);
        const float xcl = fminf(fmaxf(xf, -4.f), (float)g.W + 4.f), ycl = fminf(fmaxf(yf, -4.f), (float)g.H + 4.f);
        if (!(t_sane && xf == xcl && yf == ycl)) return false;      // NaN / far outside: no neighbour cell is in the grid
        vox_axis((int)xcl, g.W, 5, b.tx);
        vox_axis((int)ycl, g.H, g.th_shift, b.ty);
        vox_groups(g, t0, t0 + 1, t0 >= 0 && t0 < g.C, t0 + 1 >= 0 && t0 + 1 < g.C, b.tg);
        return (b.tx[0] & b.tx[1]) >= 0 && (b.ty[0] & b.ty[1]) >= 0 && (b.tg[0] & b.tg[1]) >= 0;   // -1 = none: any bin on every axis
    }
}

// A workgroup's events, four per thread and round with every load issued before the first use: f(record, bins) per contributing event.
constexpr int VOX_BIN_THREADS = 1024;
template <int SRC, class F>
__device__ __forceinline__ void vox_for_events(const VoxGeo& g, const VoxSrc& s, long long lo, long long hi, int* bad, F f) {
    for (long long base = lo; base < hi; base += 4 * VOX_BIN_THREADS) {
        VoxRaw raw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long e = base + k * VOX_BIN_THREADS + threadIdx.x;
            raw[k] = vox_load<SRC>(s, e, e < hi);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) vox_rectify<SRC>(g, s, raw[k], bad);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            VoxRec r;
            VoxBins b;
            if (vox_classify<SRC>(g, raw[k], r, b)) f(r, b);
        }
    }
}

template <class F>
__device__ __forceinline__ void vox_for_bins(const VoxGeo& g, const VoxBins& b, F f) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (b.tg[k] >= 0 && b.ty[j] >= 0 && b.tx[i] >= 0) f(vox_bin_index(g, b.tg[k], b.ty[j], b.tx[i]));
}

// Workgroup -> chunk: workgroup w runs on XCD w % 8; XCD k gets the CONSECUTIVE chunks [k * nb / 8, (k + 1) * nb / 8), so that the
// records of one bin, which are ordered by chunk, are written in eight contiguous pieces by one XCD (one L2) each.
__device__ __forceinline__ int vox_chunk_of_block(int nb, int w) { return (nb & 7) ? w : (w & 7) * (nb >> 3) + (w >> 3); }

// chunk b of the events: [lo, hi)
__device__ __forceinline__ void vox_chunk(long long n, int nb, int b, long long& lo, long long& hi) {
    const long long per = ((n + nb - 1) / nb + 1023) & ~1023LL;
    lo = per * b;
    hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = n;
}

// The window descriptor -> the by-value copies of the launch's geometry / source (uniform: every lane reads the same four values).
// Clamped to the recording and to the planned capacity: no descriptor can make a kernel read outside the arrays or write outside the workspace.
template <int SRC>
__device__ __forceinline__ void vox_apply_window(VoxGeo& g, VoxSrc& s) {
    if (!s.win) return;
    long long first = s.win[0], cnt = s.win[1];
    const long long t0c = s.win[2], t1c = s.win[3];
    first = first < 0 ? 0 : (first > s.n_total ? s.n_total : first);
    cnt = cnt < 0 ? 0 : cnt;
    cnt = cnt > s.n_total - first ? s.n_total - first : cnt;
    cnt = cnt > s.n ? s.n : cnt;
    constexpr int xy_bytes = (SRC == SRC_F32 || SRC == SRC_I32) ? 4 : 2;
    s.x = reinterpret_cast<const char*>(s.x) + first * xy_bytes;
    s.y = reinterpret_cast<const char*>(s.y) + first * xy_bytes;
    s.pol = reinterpret_cast<const char*>(s.pol) + first;
    s.t += first;
    s.n = cnt;
    g.t0c = t0c;
    g.denom = (float)(t1c - t0c);            // what the host forms for a plain call (vox_run)
}

template <int SRC>
__global__ __launch_bounds__(VOX_BIN_THREADS) void voxel_count_kernel(VoxGeo g, VoxSrc s, int* __restrict__ counts, int* __restrict__ bad) {
    extern __shared__ int hist[];
    vox_apply_window<SRC>(g, s);
    for (int i = threadIdx.x; i < g.nbins; i += VOX_BIN_THREADS) hist[i] = 0;
    __syncthreads();
    const int chunk = vox_chunk_of_block(gridDim.x, blockIdx.x);
    long long lo, hi;
    vox_chunk(s.n, gridDim.x, chunk, lo, hi);
    vox_for_events<SRC>(g, s, lo, hi, bad, [&](const VoxRec&, const VoxBins& b) { vox_for_bins(g, b, [&](int bin) { atomicAdd(&hist[bin], 1); }); });
    __syncthreads();
    int* row = counts + (long long)chunk * g.nbins;
    for (int i = threadIdx.x; i < g.nbins; i += VOX_BIN_THREADS) row[i] = hist[i];
}

// counts[b][bin] -> exclusive prefix over b (in place); totals[bin].  Workgroup = 32 bins x 32 segments of the chunk axis (half a wave =
// one 128-B row piece), every thread holds its <= 16 values in registers.
__global__ __launch_bounds__(1024) void voxel_scan_kernel(int* __restrict__ counts, int* __restrict__ totals, int nb, int nbins) {
    __shared__ int seg_tot[32][33];
    const int col = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int bin = blockIdx.x * 32 + col;
    const int per = (nb + 31) >> 5;                    // <= 16
    const int b0 = seg * per;
    int v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (k < per && b0 + k < nb && bin < nbins) ? counts[(long long)(b0 + k) * nbins + bin] : 0;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = v[k];
        v[k] = sum;
        sum += c;
    }
    seg_tot[seg][col] = sum;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = seg_tot[k][col];
        off += k < seg ? c : 0;
        tot += c;
    }
    if (bin < nbins) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < per && b0 + k < nb) counts[(long long)(b0 + k) * nbins + bin] = off + v[k];
        if (seg == 0) totals[bin] = tot;
    }
}

template <int SRC>
__global__ __launch_bounds__(VOX_BIN_THREADS) void voxel_place_kernel(VoxGeo g, VoxSrc s, const int* __restrict__ prefix, const int* __restrict__ totals,
                                                                      int* __restrict__ bin_base, VoxRec* __restrict__ recs) {
    extern __shared__ int cursor[];
    __shared__ int wave_tot[VOX_BIN_THREADS / 64];
    vox_apply_window<SRC>(g, s);
    const int chunk = vox_chunk_of_block(gridDim.x, blockIdx.x);
    // exclusive scan of the bin totals: thread i owns the run [i * per, (i + 1) * per)
    const int per = (g.nbins + VOX_BIN_THREADS - 1) / VOX_BIN_THREADS;     // <= 8
    const int i0 = threadIdx.x * per;
    int sum = 0;
    for (int k = 0; k < per; ++k) sum += (i0 + k < g.nbins) ? totals[i0 + k] : 0;
    int incl = sum;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wv; ++w) base += wave_tot[w];
    const int* row = prefix + (long long)chunk * g.nbins;
    for (int k = 0; k < per; ++k) {
        if (i0 + k < g.nbins) {
            cursor[i0 + k] = base + row[i0 + k];
            if (blockIdx.x == 0) bin_base[i0 + k] = base;
            base += totals[i0 + k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == VOX_BIN_THREADS - 1) bin_base[g.nbins] = base;   // the last thread's run ends at (or beyond) nbins
    __syncthreads();
    long long lo, hi;
    vox_chunk(s.n, gridDim.x, chunk, lo, hi);
    vox_for_events<SRC>(g, s, lo, hi, nullptr, [&](const VoxRec& r, const VoxBins& b) {
        vox_for_bins(g, b, [&](int bin) {
            const int slot = atomicAdd(&cursor[bin], 1);
            *reinterpret_cast<uint4*>(recs + slot) = *reinterpret_cast<const uint4*>(&r);
        });
    });
}

// fp32 -> signed fixed point, 2^-40 units, truncated toward zero below the unit (deterministic; exact for |w| >= 2^-16)
__device__ __forceinline__ long long vox_fixed(float w) {
    const int bits = __float_as_int(w);
    const int ex = (bits >> 23) & 255;
    const long long m = (long long)((bits & 0x7fffff) | (ex ? 0x800000 : 0));
    int sh = (ex ? ex : 1) - 110;                       // value = m * 2^(ex - 150); * 2^40
    sh = sh > 38 ? 38 : sh;                             // inf / NaN / absurd magnitudes: garbage in, bounded shift
    const long long mag = sh >= 0 ? (m << sh) : (sh > -24 ? (m >> -sh) : 0);
    return bits < 0 ? -mag : mag;
}

template <bool INT_XY, int THREADS>
__global__ __launch_bounds__(THREADS) void voxel_gather_kernel(VoxGeo g, const int* __restrict__ bin_base, const VoxRec* __restrict__ recs,
                                                           float* __restrict__ grid) {
    extern __shared__ unsigned long long acc[];         // [CG][TH][32]
    const int cells = g.CG << (g.th_shift + 5);
    for (int i = threadIdx.x; i < cells; i += THREADS) acc[i] = 0ull;
    const int bin = blockIdx.x;
    const int tx = bin % g.ntx, ty = (bin / g.ntx) % g.nty, tg = bin / (g.ntx * g.nty) + g.g_lo;
    const int x_lo = tx << 5, y_lo = ty << g.th_shift, c_lo = tg * g.CG;   // ng == 1: tg = 0
    const int r_lo = bin_base[bin], r_hi = bin_base[bin + 1];
    __syncthreads();
    auto add = [&](int tl, int yl, int xl, float w) {
        const unsigned c = (unsigned)(tl - c_lo), yy = (unsigned)(yl - y_lo), xx = (unsigned)(xl - x_lo);
        if (c < (unsigned)g.CG && yy < (unsigned)g.TH && xx < 32u)
            atomicAdd(&acc[((c << g.th_shift) + yy) * 32 + xx], (unsigned long long)vox_fixed(w));
    };
    for (int i = r_lo + (int)threadIdx.x; i < r_hi; i += THREADS) {
        VoxRec r;
        *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(recs + i);
        const float tf = floorf(r.tn);
        const int t0 = (int)tf;                         // classified sane by `place`
        if (INT_XY) {
            const int py = r.qm / g.W, px = r.qm - py * g.W;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int tl = t0 + dt;
                int p = tl + r.qd;
                if (tl >= 0 && tl < g.C && p >= -g.C && p < g.C) {
                    p += p < 0 ? g.C : 0;
                    add(p, py, px, r.value * (1.f - fabsf((float)tl - r.tn)));           // representations.py:87
                }
            }
        } else {
            const float x = r.fx, y = r.fy;
            const int x0 = (int)floorf(x), y0 = (int)floorf(y);
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int xl = x0 + dx, yl = y0 + dy, tl = t0 + dt;
                        if (xl < g.W && xl >= 0 && yl < g.H && yl >= 0 && tl >= 0 && tl < g.C)
                            // representations.py:103: value * (1-|xlim-x|) * (1-|ylim-y|) * (1-|tlim-t_norm|), left to right
                            add(tl, yl, xl, r.value * (1.f - fabsf((float)xl - x)) * (1.f - fabsf((float)yl - y)) * (1.f - fabsf((float)tl - r.tn)));
                    }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += THREADS) {
        const int xx = i & 31, yy = (i >> 5) & (g.TH - 1), c = i >> (5 + g.th_shift);
        const int xl = x_lo + xx, yl = y_lo + yy, tl = c_lo + c;
        if (xl < g.W && yl < g.H && tl < g.C) grid[((long long)tl * g.H + yl) * g.W + xl] = __ll2float_rn((long long)acc[i]) * 0x1p-40f;
    }
}

// max |a - b| over n floats -> *out (float bits are ordered like unsigned ints for non-negative values); out zeroed by the caller.
// TwoStepSubSequence.__getitem__ asserts that the temporal slice shared by the two grids agrees (twostep.py:83).
__global__ __launch_bounds__(256) void maxabs_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                          unsigned int* __restrict__ out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(a[i] - b[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

__device__ __forceinline__ void block_accumulate(double v0, double v1, double* dst0, double* dst1) {
    __shared__ double sh[2][4];
    v0 = bflow::wave_sum(v0);
    v1 = bflow::wave_sum(v1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = v0;
        sh[1][wv] = v1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dst0, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        if (dst1) atomicAdd(dst1, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

// ---- K2 (round 6): TWO launches, three passes over memory (was: a memset + three kernels, four passes, two rounds of fp64 atomics).
//   norm_stats_kernel   one read: per-block partial (sum, sum of squares, count) of the non-zero entries in fp64 -> ws[block][3]
//                       (plain stores: the workspace needs no zero fill and no atomics)
//   norm_apply_kernel   every block folds the <= NORM_BLOCKS partials (L2-resident, 24 KB), derives mean / std as torch does -- fp32 mean of
//                       the masked values, unbiased std of the deviations from THAT mean: sum (v - m)^2 = S2 - 2 m S1 + n m^2, exact algebra
//                       in fp64 -- and writes (v - mean) / std over the non-zero entries.
// The input is the concatenation of up to two segments [a | b] (the merge of TwoStepSubSequence.__getitem__, twostep.py:77-85: previous
// grid | current grid without its first bin), the output one contiguous array (which may be `a` itself: in place).
constexpr int NORM_BLOCKS = 1024;

__device__ __forceinline__ void norm_acc(float v, double& s1, double& s2, double& c) {
    if (v != 0.f) {
        const double d = (double)v;
        s1 += d;
        s2 = fma(d, d, s2);
        c += 1.0;
    }
}

__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b, long long nb,
                                                         double* __restrict__ ws) {
    double s1 = 0.0, s2 = 0.0, c = 0.0;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long long)gridDim.x * blockDim.x;
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const float* g = seg ? b : a;
        const long long n = seg ? nb : na;
        if (n <= 0) continue;
        const long long head = (long long)((16 - ((uintptr_t)g & 15)) & 15) / 4;       // scalars in front of the first 16-B boundary
        const long long h = head < n ? head : n;
        const long long nv = (n - h) / 4;
        const float4* g4 = reinterpret_cast<const float4*>(g + h);
        for (long long i = tid; i < nv; i += nthr) {
            const float4 v = g4[i];
            norm_acc(v.x, s1, s2, c);
            norm_acc(v.y, s1, s2, c);
            norm_acc(v.z, s1, s2, c);
            norm_acc(v.w, s1, s2, c);
        }
        for (long long i = tid; i < h; i += nthr) norm_acc(g[i], s1, s2, c);
        for (long long i = h + nv * 4 + tid; i < n; i += nthr) norm_acc(g[i], s1, s2, c);
    }
    __shared__ double sh[3][4];
    s1 = bflow::wave_sum(s1);
    s2 = bflow::wave_sum(s2);
    c = bflow::wave_sum(c);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = s1;
        sh[1][wv] = s2;
        sh[2][wv] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) ws[(long long)blockIdx.x * 3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b, long long nb,
                                                         float* __restrict__ out, const double* __restrict__ ws, int nparts) {
    // every block folds the partials in the SAME order: all blocks apply the same (mean, std), bit for bit
    __shared__ double sh[3][4];
    double p[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nparts; i += 256) {
        p[0] += ws[i * 3];
        p[1] += ws[i * 3 + 1];
        p[2] += ws[i * 3 + 2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = bflow::wave_sum(p[k]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = p[0];
        sh[1][wv] = p[1];
        sh[2][wv] = p[2];
    }
    __syncthreads();
    const double S1 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]), S2 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]),
                 cnt = (sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3]);
    const bool any = cnt > 0.0;                                    // representations.py:11: nothing to do (the merge copy still happens)
    const float mean = any ? (float)(S1 / cnt) : 0.f;              // torch: fp32 mean of the masked values
    // unbiased std (torch.Tensor.std default, representations.py:13) of the deviations from the fp32 mean; a single element gives NaN in
    // torch and `std > 0` is then False -> mean-only branch
    const double m = (double)mean;
    const double ssd = fmax(S2 - 2.0 * m * S1 + cnt * m * m, 0.0);
    const float stdv = cnt > 1.0 ? (float)sqrt(ssd / (cnt - 1.0)) : 0.f;
    const long long n = na + nb;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long long)gridDim.x * blockDim.x;
    auto f = [&](float v) -> float { return (v != 0.f && any) ? (stdv > 0.f ? (v - mean) / stdv : v - mean) : v; };
    // 16-B vectors where source segment, segment boundary and destination allow it (torch tensors and whole bins do); scalars otherwise
    const bool vec = (((uintptr_t)a | (uintptr_t)out | (uintptr_t)(b ? b : a)) & 15) == 0 && (na & 3) == 0 && (nb & 3) == 0;
    if (vec) {
        const long long nva = na / 4, nv = n / 4;
        for (long long i = tid; i < nv; i += nthr) {
            float4 v = i < nva ? reinterpret_cast<const float4*>(a)[i] : reinterpret_cast<const float4*>(b)[i - nva];
            v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
            reinterpret_cast<float4*>(out)[i] = v;
        }
    } else {
        for (long long i = tid; i < n; i += nthr) out[i] = f(i < na ? a[i] : b[i - na]);
    }
}

struct VoxPlan {
    VoxGeo geo;          // g_lo / g_cnt / nbins are per slab
    int ng, groups_per_slab, nb;
    size_t off_totals, off_base, off_recs, bytes;
};

constexpr int VOX_MAX_BINS = 8192;

// Bin shape and workspace layout.  max_dup = the most bins one event can belong to.
static int vox_plan(long long n, int C, int H, int W, bool float_xy, VoxPlan& p, const char* what) {
    BFLOW_REQUIRE(C > 1 && H > 1 && W > 1, BFLOW_E_ARG, "%s: bad grid", what);
    BFLOW_REQUIRE(n >= 0 && n <= (1LL << 27), BFLOW_E_ARG, "%s: at most 2^27 events per call", what);
    BFLOW_REQUIRE((long long)C * H * W < (1LL << 31) && (long long)H * W <= (1LL << 24), BFLOW_E_ARG, "%s: grid too large", what);
    VoxGeo& g = p.geo;
    g.C = C; g.H = H; g.W = W;
    g.cg_shift = 3;
    g.CG = C <= 8 ? C : 8;
    p.ng = C <= 8 ? 1 : (C + 7) / 8;
    g.ntx = (W + 31) / 32;
    // rows per bin: the tallest of 32 / 16 / 8 that still leaves >= 512 bins (longer runs of records per (chunk, bin) in `place`, fewer
    // border duplicates; measured 2 M events: 15 x 480 x 640 87 -> 73 us with 32 rows, 5 x 480 x 640 75 -> 67 us with 16), 64 KB of LDS at most
    auto bins = [&](int sh) { return (long long)g.ntx * ((H + (1 << sh) - 1) >> sh) * p.ng; };
    g.th_shift = 5;
    while (g.th_shift > 3 && bins(g.th_shift) < 512) --g.th_shift;
    while (g.th_shift < 5 && bins(g.th_shift) > VOX_MAX_BINS) ++g.th_shift;
    g.TH = 1 << g.th_shift;
    g.nty = (H + g.TH - 1) >> g.th_shift;
    const int spatial = g.ntx * g.nty;
    BFLOW_REQUIRE(spatial <= VOX_MAX_BINS, BFLOW_E_ARG, "%s: grid too large", what);
    p.groups_per_slab = VOX_MAX_BINS / spatial;
    if (p.groups_per_slab > p.ng) p.groups_per_slab = p.ng;
    const int slab_bins = spatial * p.groups_per_slab;
    // chunks: 4096 events each at 2 M events; fewer, larger ones when the count matrix (chunks x bins) would outgrow the event arrays
    long long nb = (n + 4095) / 4096;
    const long long cap = slab_bins > 4096 ? 256 : 512;
    nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
    if (nb > 8) nb = (nb + 7) & ~7LL;     // whole chunks per XCD (vox_chunk_of_block)
    p.nb = (int)nb;
    const int max_dup = (float_xy ? 4 : 1) * (p.ng > 1 ? 2 : 1);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    p.off_totals = up((size_t)p.nb * slab_bins * 4);
    p.off_base = p.off_totals + up((size_t)slab_bins * 4);
    p.off_recs = p.off_base + up((size_t)(slab_bins + 1) * 4);
    p.bytes = p.off_recs + up((size_t)(n > 0 ? n : 1) * max_dup * sizeof(VoxRec));
    return 0;
}

template <int SRC>
int vox_run(const VoxSrc& s, long long t0c, long long t1c, float* grid, int C, int H, int W, void* ws, long long ws_bytes, int* bad,
            bflow_stream_t stream, const char* what) {
    constexpr bool FLOAT_XY = SRC == SRC_F32 || SRC == SRC_RECT;
    VoxPlan p;
    if (int rc = vox_plan(s.n, C, H, W, FLOAT_XY, p, what)) return rc;
    BFLOW_REQUIRE(grid, BFLOW_E_ARG, "%s: bad grid", what);
    BFLOW_REQUIRE(s.win || t1c > t0c, BFLOW_E_ARG, "%s: t1_center must be > t0_center", what);
    BFLOW_REQUIRE(s.n == 0 || (s.x && s.y && s.pol && s.t), BFLOW_E_ARG, "%s: bad event arrays", what);
    BFLOW_REQUIRE(ws && ((uintptr_t)ws & 15) == 0 && ws_bytes >= (long long)p.bytes, BFLOW_E_ARG,
                  "%s: workspace of %lld bytes needed (bflow_voxel_workspace_bytes), 16-byte aligned", what, (long long)p.bytes);
    hipStream_t st = (hipStream_t)stream;
    VoxGeo g = p.geo;
    g.t0c = t0c;
    g.denom = (float)(t1c - t0c);
    g.cm1 = (float)(C - 1);
    char* w = (char*)ws;
    int* counts = (int*)w;
    int* totals = (int*)(w + p.off_totals);
    int* base = (int*)(w + p.off_base);
    VoxRec* recs = (VoxRec*)(w + p.off_recs);
    for (int g_lo = 0; g_lo < p.ng; g_lo += p.groups_per_slab) {
        g.g_lo = g_lo;
        g.g_cnt = p.ng - g_lo < p.groups_per_slab ? p.ng - g_lo : p.groups_per_slab;
        g.nbins = g.ntx * g.nty * g.g_cnt;
        hipLaunchKernelGGL(voxel_count_kernel<SRC>, dim3(p.nb), dim3(VOX_BIN_THREADS), (size_t)g.nbins * 4, st, g, s, counts, g_lo == 0 ? bad : nullptr);
        hipLaunchKernelGGL(voxel_scan_kernel, dim3((g.nbins + 31) / 32), dim3(1024), 0, st, counts, totals, p.nb, g.nbins);
        hipLaunchKernelGGL(voxel_place_kernel<SRC>, dim3(p.nb), dim3(VOX_BIN_THREADS), (size_t)g.nbins * 4, st, g, s, counts, totals, base, recs);
        hipLaunchKernelGGL((voxel_gather_kernel<!FLOAT_XY, 1024>), dim3(g.nbins), dim3(1024), (size_t)g.CG * g.TH * 32 * 8, st, g, base, recs, grid);
    }
    return bflow::launch_status(what);
}

}  // namespace

extern "C" long long bflow_voxel_workspace_bytes(long long n_events, int C, int H, int W, int float_xy) {
    VoxPlan p;
    if (vox_plan(n_events, C, H, W, float_xy != 0, p, "voxel_workspace_bytes")) return -1;
    return (long long)p.bytes;
}

extern "C" int bflow_voxel_grid_f32xy(const float* x, const float* y, const signed char* pol, const long long* t, long long n,
                                      long long t0c, long long t1c, float* grid, int C, int H, int W, void* ws, long long ws_bytes,
                                      bflow_stream_t stream) {
    return vox_run<SRC_F32>(VoxSrc{x, y, pol, t, nullptr, n, nullptr, 0}, t0c, t1c, grid, C, H, W, ws, ws_bytes, nullptr, stream, "voxel_grid_f32xy");
}

extern "C" int bflow_voxel_grid_i16xy(const short* x, const short* y, const signed char* pol, const long long* t, long long n,
                                      long long t0c, long long t1c, float* grid, int C, int H, int W, void* ws, long long ws_bytes,
                                      bflow_stream_t stream) {
    return vox_run<SRC_I16>(VoxSrc{x, y, pol, t, nullptr, n, nullptr, 0}, t0c, t1c, grid, C, H, W, ws, ws_bytes, nullptr, stream, "voxel_grid_i16xy");
}

extern "C" int bflow_voxel_grid_i32xy(const int* x, const int* y, const signed char* pol, const long long* t, long long n,
                                      long long t0c, long long t1c, float* grid, int C, int H, int W, void* ws, long long ws_bytes,
                                      bflow_stream_t stream) {
    return vox_run<SRC_I32>(VoxSrc{x, y, pol, t, nullptr, n, nullptr, 0}, t0c, t1c, grid, C, H, W, ws, ws_bytes, nullptr, stream, "voxel_grid_i32xy");
}

static int norm_run(const float* a, long long na, const float* b, long long nb, float* out, double* ws, bflow_stream_t stream, const char* what) {
    BFLOW_REQUIRE(a && out && ws && na > 0 && nb >= 0 && (nb == 0 || b), BFLOW_E_ARG, "%s: bad arguments", what);
    hipStream_t s = (hipStream_t)stream;
    const long long n = na + nb;
    int nblk = (int)((n / 4 + 255) / 256);
    nblk = nblk < 1 ? 1 : nblk > NORM_BLOCKS ? NORM_BLOCKS : nblk;
    hipLaunchKernelGGL(norm_stats_kernel, dim3(nblk), dim3(256), 0, s, a, na, b, nb, ws);
    hipLaunchKernelGGL(norm_apply_kernel, dim3(nblk), dim3(256), 0, s, a, na, b, nb, out, ws, nblk);
    return bflow::launch_status(what);
}

extern "C" int bflow_voxel_norm(float* grid, long long n, double* ws, bflow_stream_t stream) {
    return norm_run(grid, n, nullptr, 0, grid, ws, stream, "voxel_norm");
}

extern "C" int bflow_voxel_merge_norm(const float* a, long long na, const float* b, long long nb, float* out, double* ws, bflow_stream_t stream) {
    BFLOW_REQUIRE(out != b || nb == 0, BFLOW_E_ARG, "voxel_merge_norm: the output may alias the FIRST segment only");
    return norm_run(a, na, b, nb, out, ws, stream, "voxel_merge_norm");
}

extern "C" int bflow_voxel_grid_rectified(const unsigned short* x, const unsigned short* y, const unsigned char* pol, const long long* t,
                                          long long n, const float* rectify_map, long long t0c, long long t1c, float* grid, int C, int H,
                                          int W, int* bad_count, void* ws, long long ws_bytes, bflow_stream_t stream) {
    BFLOW_REQUIRE(rectify_map && ((uintptr_t)rectify_map & 7) == 0, BFLOW_E_ARG, "voxel_grid_rectified: the map must be 8-byte aligned");
    return vox_run<SRC_RECT>(VoxSrc{x, y, pol, t, rectify_map, n, nullptr, 0}, t0c, t1c, grid, C, H, W, ws, ws_bytes, bad_count, stream,
                             "voxel_grid_rectified");
}

extern "C" int bflow_voxel_grid_rectified_window(const unsigned short* x, const unsigned short* y, const unsigned char* pol, const long long* t,
                                                 long long n_total, long long max_events, const long long* window, const float* rectify_map,
                                                 float* grid, int C, int H, int W, int* bad_count, void* ws, long long ws_bytes,
                                                 bflow_stream_t stream) {
    BFLOW_REQUIRE(rectify_map && ((uintptr_t)rectify_map & 7) == 0, BFLOW_E_ARG, "voxel_grid_rectified_window: the map must be 8-byte aligned");
    BFLOW_REQUIRE(window && ((uintptr_t)window & 7) == 0, BFLOW_E_ARG, "voxel_grid_rectified_window: the window descriptor is 4 device int64");
    BFLOW_REQUIRE(n_total > 0 && max_events > 0, BFLOW_E_ARG, "voxel_grid_rectified_window: empty recording / capacity");
    return vox_run<SRC_RECT>(VoxSrc{x, y, pol, t, rectify_map, max_events, window, n_total}, 0, 1, grid, C, H, W, ws, ws_bytes, bad_count, stream,
                             "voxel_grid_rectified_window");
}

extern "C" int bflow_maxabs_diff(const float* a, const float* b, long long n, float* out, bflow_stream_t stream) {
    BFLOW_REQUIRE(a && b && out && n > 0, BFLOW_E_ARG, "maxabs_diff: bad arguments");
    hipLaunchKernelGGL(maxabs_diff_kernel, dim3(bflow::reduce_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, (unsigned int*)out);
    return bflow::launch_status("maxabs_diff");
}
