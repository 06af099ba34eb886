// K1 / K2: event voxel grid (reference: VoxelGrid.convert + norm_voxel_grid, data/utils/representations.py:9-18,64-111).
//
// K1 is an atomic-scatter kernel: one event per lane, 8 (float x/y) or 2 (integer x/y) fp32 hardware atomic adds
// (global_atomic_add_f32) into the (C,H,W) grid; the event arrays are read with fully coalesced loads.  The order of
// accumulation differs from the reference's sequential put_(accumulate=True), so K1 is checked to a stated fp32
// tolerance, not bit-exactly.
// K2 is three streaming passes (sum+count of non-zeros, sum of squared deviations, apply) with wavefront-shuffle +
// one fp64 atomic per block reductions; the scalar results stay on the device (graph-capture safe).
#include "common.h"

namespace {

template <class XY>
__global__ __launch_bounds__(256) void voxel_scatter_kernel(const XY* __restrict__ xs, const XY* __restrict__ ys,
                                                            const signed char* __restrict__ pol, const long long* __restrict__ ts,
                                                            long long n, long long t0c, long long t1c, float* __restrict__ grid, int C,
                                                            int H, int W) {
    constexpr bool INT_XY = !__is_floating_point(XY);
    const float denom = (float)(t1c - t0c);
    const float cm1 = (float)(C - 1);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        // representations.py:58: int64 tensor / python int -> float32 true division, then * (C-1)
        const float t_norm = (float)(ts[e] - t0c) / denom * cm1;
        const float tf = floorf(t_norm);
        const float tcl = fminf(fmaxf(tf, -4.f), (float)C + 4.f);
        const int t0 = (int)tcl;
        const float value = 2.f * (float)pol[e] - 1.f;
        if (INT_XY) {
            // representations.py:85-94: only the time bin is masked; x / y enter through the FLAT index ht*wd*t + wd*y + x handed to
            // Tensor.put_, which accepts [-numel, numel) (negative = from the end) and raises outside.  The same rule here: an index
            // put_ would accept lands where put_ puts it, one it would raise on is dropped -- never an out-of-bounds write.
            const long long x = (long long)xs[e], y = (long long)ys[e];
            const long long numel = (long long)C * H * W;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int tl = t0 + dt;
                if (tl >= 0 && tl < C && tf == tcl) {
                    long long idx = ((long long)tl * H + y) * W + x;
                    if (idx < 0) idx += numel;
                    if (idx >= 0 && idx < numel) {
                        const float wgt = value * (1.f - fabsf((float)tl - t_norm));
                        atomicAdd(grid + idx, wgt);
                    }
                }
            }
        } else {
            const float x = (float)xs[e], y = (float)ys[e];
            const float xf = floorf(x), yf = floorf(y);
            const float xcl = fminf(fmaxf(xf, -4.f), (float)W + 4.f), ycl = fminf(fmaxf(yf, -4.f), (float)H + 4.f);
            const bool sane = (xf == xcl) && (yf == ycl) && (tf == tcl);
            const int x0 = (int)xcl, y0 = (int)ycl;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int xl = x0 + dx, yl = y0 + dy, tl = t0 + dt;
                        if (sane && xl < W && xl >= 0 && yl < H && yl >= 0 && tl >= 0 && tl < C) {
                            // representations.py:103: value * (1-|xlim-x|) * (1-|ylim-y|) * (1-|tlim-t_norm|), left to right
                            const float wgt = value * (1.f - fabsf((float)xl - x)) * (1.f - fabsf((float)yl - y)) *
                                              (1.f - fabsf((float)tl - t_norm));
                            atomicAdd(grid + ((long long)tl * H + yl) * W + xl, wgt);
                        }
                    }
        }
    }
}

// f-1: DSEC sample assembly.  Raw sensor events (uint16 x / y, 0/1 polarity) are rectified through the per-sequence
// map (BaseSubSequence._rectify_events, data/dsec/subsequence/base.py:137-143: rectify_map[y, x] -> (x', y') float32) and
// scattered tri-linearly in the same kernel: the rectified coordinate arrays are never materialised.  Events whose raw
// coordinates fall outside the map (the reference asserts on them) are skipped and counted in *bad.
__global__ __launch_bounds__(256) void voxel_scatter_rect_kernel(const unsigned short* __restrict__ xs, const unsigned short* __restrict__ ys,
                                                                 const unsigned char* __restrict__ pol, const long long* __restrict__ ts,
                                                                 long long n, const float* __restrict__ rect, long long t0c, long long t1c,
                                                                 float* __restrict__ grid, int C, int H, int W, int* __restrict__ bad) {
    const float denom = (float)(t1c - t0c);
    const float cm1 = (float)(C - 1);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int xr = xs[e], yr = ys[e];
        if (xr >= W || yr >= H) {
            if (bad) atomicAdd(bad, 1);
            continue;
        }
        const float2 xy = *reinterpret_cast<const float2*>(rect + ((long long)yr * W + xr) * 2);
        const float t_norm = (float)(ts[e] - t0c) / denom * cm1;
        const float tf = floorf(t_norm);
        const float tcl = fminf(fmaxf(tf, -4.f), (float)C + 4.f);
        const int t0 = (int)tcl;
        const float value = 2.f * (float)pol[e] - 1.f;
        const float x = xy.x, y = xy.y;
        const float xf = floorf(x), yf = floorf(y);
        const float xcl = fminf(fmaxf(xf, -4.f), (float)W + 4.f), ycl = fminf(fmaxf(yf, -4.f), (float)H + 4.f);
        const bool sane = (xf == xcl) && (yf == ycl) && (tf == tcl);
        const int x0 = (int)xcl, y0 = (int)ycl;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int xl = x0 + dx, yl = y0 + dy, tl = t0 + dt;
                    if (sane && xl < W && xl >= 0 && yl < H && yl >= 0 && tl >= 0 && tl < C) {
                        const float wgt = value * (1.f - fabsf((float)xl - x)) * (1.f - fabsf((float)yl - y)) * (1.f - fabsf((float)tl - t_norm));
                        atomicAdd(grid + ((long long)tl * H + yl) * W + xl, wgt);
                    }
                }
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

// ws[0] = sum, ws[1] = count, ws[2] = sum of squared deviations
__global__ __launch_bounds__(256) void norm_pass1(const float* __restrict__ g, long long n, double* ws) {
    double s = 0.0, c = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        if (v != 0.f) {
            s += (double)v;
            c += 1.0;
        }
    }
    block_accumulate(s, c, ws + 0, ws + 1);
}

__global__ __launch_bounds__(256) void norm_pass2(const float* __restrict__ g, long long n, double* ws) {
    const double cnt = ws[1];
    const float mean = cnt > 0.0 ? (float)(ws[0] / cnt) : 0.f;   // torch: fp32 mean of the masked values
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        if (v != 0.f) {
            const double d = (double)v - (double)mean;
            s += d * d;
        }
    }
    block_accumulate(s, 0.0, ws + 2, nullptr);
}

__global__ __launch_bounds__(256) void norm_pass3(float* __restrict__ g, long long n, const double* ws) {
    const double cnt = ws[1];
    if (cnt <= 0.0) return;                                        // representations.py:11: nothing to do
    const float mean = (float)(ws[0] / cnt);
    // unbiased std (torch.Tensor.std default, representations.py:13); a single element gives NaN in torch and
    // `std > 0` is then False -> mean-only branch
    const float stdv = cnt > 1.0 ? (float)sqrt(ws[2] / (cnt - 1.0)) : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        if (v != 0.f) g[i] = stdv > 0.f ? (v - mean) / stdv : v - mean;
    }
}

template <class XY>
int scatter(const XY* x, const XY* y, const signed char* pol, const long long* t, long long n, long long t0c, long long t1c, float* grid,
            int C, int H, int W, bflow_stream_t stream, const char* what) {
    BFLOW_REQUIRE(grid && C > 1 && H > 1 && W > 1, BFLOW_E_ARG, "%s: bad grid", what);
    BFLOW_REQUIRE(t1c > t0c, BFLOW_E_ARG, "%s: t1_center must be > t0_center", what);
    if (n == 0) return 0;
    BFLOW_REQUIRE(x && y && pol && t && n > 0, BFLOW_E_ARG, "%s: bad event arrays", what);
    hipLaunchKernelGGL(voxel_scatter_kernel<XY>, dim3(bflow::stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, pol, t, n,
                       t0c, t1c, grid, C, H, W);
    return bflow::launch_status(what);
}

}  // namespace

extern "C" int bflow_voxel_scatter_f32xy(const float* x, const float* y, const signed char* pol, const long long* t, long long n,
                                         long long t0c, long long t1c, float* grid, int C, int H, int W, bflow_stream_t stream) {
    return scatter<float>(x, y, pol, t, n, t0c, t1c, grid, C, H, W, stream, "voxel_scatter_f32xy");
}

extern "C" int bflow_voxel_scatter_i16xy(const short* x, const short* y, const signed char* pol, const long long* t, long long n,
                                         long long t0c, long long t1c, float* grid, int C, int H, int W, bflow_stream_t stream) {
    return scatter<short>(x, y, pol, t, n, t0c, t1c, grid, C, H, W, stream, "voxel_scatter_i16xy");
}

extern "C" int bflow_voxel_scatter_i32xy(const int* x, const int* y, const signed char* pol, const long long* t, long long n,
                                         long long t0c, long long t1c, float* grid, int C, int H, int W, bflow_stream_t stream) {
    return scatter<int>(x, y, pol, t, n, t0c, t1c, grid, C, H, W, stream, "voxel_scatter_i32xy");
}

extern "C" int bflow_voxel_norm(float* grid, long long n, double* ws, bflow_stream_t stream) {
    BFLOW_REQUIRE(grid && ws && n > 0, BFLOW_E_ARG, "voxel_norm: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws, 0, 4 * sizeof(double), s);
    if (e != hipSuccess) {
        bflow::set_error("voxel_norm: memset: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int g = bflow::stream_grid(n, 256), gr = bflow::reduce_grid(n, 256);
    hipLaunchKernelGGL(norm_pass1, dim3(gr), dim3(256), 0, s, grid, n, ws);
    hipLaunchKernelGGL(norm_pass2, dim3(gr), dim3(256), 0, s, grid, n, ws);
    hipLaunchKernelGGL(norm_pass3, dim3(g), dim3(256), 0, s, grid, n, ws);
    return bflow::launch_status("voxel_norm");
}

extern "C" int bflow_voxel_scatter_rectified(const unsigned short* x, const unsigned short* y, const unsigned char* pol, const long long* t,
                                             long long n, const float* rectify_map, long long t0c, long long t1c, float* grid, int C, int H,
                                             int W, int* bad_count, bflow_stream_t stream) {
    BFLOW_REQUIRE(grid && rectify_map && C > 1 && H > 1 && W > 1, BFLOW_E_ARG, "voxel_scatter_rectified: bad grid / map");
    BFLOW_REQUIRE(t1c > t0c, BFLOW_E_ARG, "voxel_scatter_rectified: t1_center must be > t0_center");
    BFLOW_REQUIRE(((uintptr_t)rectify_map & 7) == 0, BFLOW_E_ARG, "voxel_scatter_rectified: the map must be 8-byte aligned");
    if (n == 0) return 0;
    BFLOW_REQUIRE(x && y && pol && t && n > 0, BFLOW_E_ARG, "voxel_scatter_rectified: bad event arrays");
    hipLaunchKernelGGL(voxel_scatter_rect_kernel, dim3(bflow::stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, pol, t, n,
                       rectify_map, t0c, t1c, grid, C, H, W, bad_count);
    return bflow::launch_status("voxel_scatter_rectified");
}

extern "C" int bflow_maxabs_diff(const float* a, const float* b, long long n, float* out, bflow_stream_t stream) {
    BFLOW_REQUIRE(a && b && out && n > 0, BFLOW_E_ARG, "maxabs_diff: bad arguments");
    hipLaunchKernelGGL(maxabs_diff_kernel, dim3(bflow::reduce_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, (unsigned int*)out);
    return bflow::launch_status("maxabs_diff");
}
