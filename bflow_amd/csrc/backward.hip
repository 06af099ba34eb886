// backward passes of the look-up (K7), the pyramid pooling (K6) and the convex up-sampling (K13), and the masked
// L1 loss of the training step (utils/losses.py:6-22).  The reference gets these from autograd over grid_sample / avg_pool2d /
// unfold+softmax (models/raft_utils/corr.py:108-125,307-351, utils.py:5-21,33-48); here they are written as the adjoint of the
// forward kernels in corr_lookup.hip / upsample.hip, with the same sampling arithmetic (same round trip through normalised
// coordinates, same zero padding), so that value and gradient describe the same function.
//
// All kernels are deterministic (no floating-point atomics on shared addresses): a query pixel owns a private plane of the 4-D
// volume, so its gradient patch is accumulated in LDS by ONE thread and added to the volume gradient by the workgroup that owns
// the pixel; per-plane coordinate gradients are written separately and summed per target by the caller.
#include "common.h"

namespace {

constexpr int R = BFLOW_LOOKUP_RADIUS;
constexpr int WIN = 2 * R + 1;
constexpr int PATCH = 12;
constexpr int PSTRIDE = PATCH * PATCH + 1;
constexpr int BT = 32;          // query pixels per workgroup
constexpr int BTHREADS = 192;   // 12 row groups x 16 lanes for the patch gather / write-back

struct PlaneBwd {
    const float* base;   // forward pyramid plane (values: needed for the coordinate gradient)
    float* grad;         // gradient of that plane, accumulated in place
    int h, w;
    float inv_scale;
    int target;
};

struct LookupBwdArgs {
    PlaneBwd planes[BFLOW_MAX_PLANES];
    float coef[BFLOW_MAX_TARGETS * BFLOW_MAX_DEGREE];
    int P, T, deg;
};

__device__ __forceinline__ float roundtrip(float x, int size) {   // identical to corr_lookup.hip
    const float sm1 = (float)(size - 1);
    const float g = 2.0f * x / sm1 - 1.0f;
    return (g + 1.0f) * (sm1 / 2.0f);
}

template <bool FUSED>
__global__ __launch_bounds__(BTHREADS) void corr_lookup_bwd_kernel(LookupBwdArgs args, const float* __restrict__ src,
                                                                  const float* __restrict__ gout, float* __restrict__ gcoords,
                                                                  int B, int h1, int w1) {
    __shared__ float patch[BT * PSTRIDE];
    __shared__ float gpatch[BT * PSTRIDE];
    __shared__ float s_cx[BT], s_cy[BT];
    __shared__ int s_ox[BT], s_oy[BT];

    const int tid = threadIdx.x;
    const int N = h1 * w1;
    const int p = blockIdx.y, b = blockIdx.z;
    const int n0 = blockIdx.x * BT;
    const PlaneBwd pl = args.planes[p];
    const int npix = min(BT, N - n0);

    if (tid < BT) {
        const int n = n0 + tid;
        float cx = 0.f, cy = 0.f;
        if (n < N) {
            if (FUSED) {
                const int deg = args.deg;
                const float* pp = src + (long long)b * 2 * deg * N + n;
                const float* cf = args.coef + pl.target * deg;
                float fx = 0.f, fy = 0.f;
                for (int i = 0; i < deg; ++i) {
                    fx = fmaf(pp[(long long)i * N], cf[i], fx);
                    fy = fmaf(pp[(long long)(deg + i) * N], cf[i], fy);
                }
                const int y = n / w1, x = n - y * w1;
                cx = (float)x + fx;
                cy = (float)y + fy;
            } else {
                const float* cc = src + ((long long)(pl.target * B + b) * 2) * N + n;
                cx = cc[0];
                cy = cc[N];
            }
            cx *= pl.inv_scale;
            cy *= pl.inv_scale;
        }
        s_cx[tid] = cx;
        s_cy[tid] = cy;
        const float ccx = fminf(fmaxf(cx, -64.f), (float)pl.w + 64.f);
        const float ccy = fminf(fmaxf(cy, -64.f), (float)pl.h + 64.f);
        s_ox[tid] = (int)floorf(ccx) - (R + 1);
        s_oy[tid] = (int)floorf(ccy) - (R + 1);
    }
    for (int i = tid; i < BT * PSTRIDE; i += BTHREADS) gpatch[i] = 0.f;
    __syncthreads();

    const int r = tid >> 4, c = tid & 15;
    const int plane_sz = pl.h * pl.w;
    const long long slab0 = ((long long)b * N + n0) * plane_sz;
    // ---- value patches (zero padded), as in the forward
    for (int pix = 0; pix < npix; ++pix) {
        const int gy = s_oy[pix] + r, gx = s_ox[pix] + c;
        const bool ok = (c < PATCH) && gy >= 0 && gy < pl.h && gx >= 0 && gx < pl.w;
        const float v = ok ? pl.base[slab0 + (long long)pix * plane_sz + gy * pl.w + gx] : 0.f;
        if (c < PATCH) patch[pix * PSTRIDE + r * PATCH + c] = v;
    }
    __syncthreads();

    // ---- one thread per pixel: adjoint of the 81 bilinear samples
    if (tid < npix) {
        const int pix = tid, n = n0 + pix;
        const float cx = s_cx[pix], cy = s_cy[pix];
        const int ox = s_ox[pix], oy = s_oy[pix];
        const float* pp = patch + pix * PSTRIDE;
        float* gp = gpatch + pix * PSTRIDE;
        const float* go = gout + ((long long)b * args.P * (WIN * WIN) + (long long)p * (WIN * WIN)) * N + n;
        float gcx = 0.f, gcy = 0.f;
        for (int ky = 0; ky < WIN; ++ky) {
            float iy = roundtrip(cy + (float)(ky - R), pl.h);
            iy = fminf(fmaxf(iy, -1.0e4f), 1.0e4f);
            const float fy0 = floorf(iy);
            const float ws = iy - fy0, wn = 1.f - ws;
            const int ay = (int)fy0 - oy;
            const bool yok = (ay >= 0 && ay + 1 < PATCH);
            for (int kx = 0; kx < WIN; ++kx) {
                float ix = roundtrip(cx + (float)(kx - R), pl.w);
                ix = fminf(fmaxf(ix, -1.0e4f), 1.0e4f);
                const float fx0 = floorf(ix);
                const float we = ix - fx0, ww = 1.f - we;
                const int ax = (int)fx0 - ox;
                if (!(yok && ax >= 0 && ax + 1 < PATCH)) continue;
                const float g = go[(long long)(ky * WIN + kx) * N];
                const int q = ay * PATCH + ax;
                const float q00 = pp[q], q01 = pp[q + 1], q10 = pp[q + PATCH], q11 = pp[q + PATCH + 1];
                gp[q] += g * (ww * wn);
                gp[q + 1] += g * (we * wn);
                gp[q + PATCH] += g * (ww * ws);
                gp[q + PATCH + 1] += g * (we * ws);
                gcx += g * ((q01 - q00) * wn + (q11 - q10) * ws);
                gcy += g * ((q10 - q00) * ww + (q11 - q01) * we);
            }
        }
        // d(sample coordinate)/d(coords) = 1/2^level  (corr.py:333; the normalise/un-normalise round trip has slope 1)
        float* gc = gcoords + (((long long)p * B + b) * 2) * N + n;
        gc[0] = gcx * pl.inv_scale;
        gc[N] = gcy * pl.inv_scale;
    }
    __syncthreads();

    // ---- add the patches to the volume gradient (cells outside the plane carry the zero padding: no gradient)
    for (int pix = 0; pix < npix; ++pix) {
        const int gy = s_oy[pix] + r, gx = s_ox[pix] + c;
        if ((c < PATCH) && gy >= 0 && gy < pl.h && gx >= 0 && gx < pl.w) {
            const float g = gpatch[pix * PSTRIDE + r * PATCH + c];
            if (g != 0.f) pl.grad[slab0 + (long long)pix * plane_sz + gy * pl.w + gx] += g;
        }
    }
}

int fill_bwd_args(LookupBwdArgs& a, const bflow_plane_t* planes, float* const* grads, int P, int T) {
    BFLOW_REQUIRE(planes && grads && P > 0 && T > 0, BFLOW_E_ARG, "corr_lookup_bwd: bad plane table");
    BFLOW_REQUIRE(P <= BFLOW_MAX_PLANES && T <= BFLOW_MAX_TARGETS, BFLOW_E_LIMIT, "corr_lookup_bwd: %d planes / %d targets", P, T);
    a.P = P;
    a.T = T;
    a.deg = 0;
    for (int p = 0; p < P; ++p) {
        BFLOW_REQUIRE(planes[p].base && grads[p] && planes[p].h > 0 && planes[p].w > 0 && planes[p].level >= 0 && planes[p].level < 16 &&
                          planes[p].target >= 0 && planes[p].target < T,
                      BFLOW_E_ARG, "corr_lookup_bwd: bad descriptor for plane %d", p);
        a.planes[p].base = planes[p].base;
        a.planes[p].grad = grads[p];
        a.planes[p].h = planes[p].h;
        a.planes[p].w = planes[p].w;
        a.planes[p].inv_scale = 1.0f / (float)(1 << planes[p].level);
        a.planes[p].target = planes[p].target;
    }
    return 0;
}

// grad_prev (M, h, w) += 0.25 * grad_cur (M, h/2, w/2) broadcast over the 2x2 cells; an odd last row / column was dropped by
// avg_pool2d (corr.py:119) and gets nothing.
__global__ __launch_bounds__(256) void corr_pool2x2_bwd_kernel(const float* __restrict__ gcur, float* __restrict__ gprev, long long M,
                                                               int h, int w) {
    const int h2 = h / 2, w2 = w / 2;
    const long long total = M * h * w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const long long t = idx / w;
        const int y = (int)(t % h);
        const long long m = t / h;
        const int y2 = y >> 1, x2 = x >> 1;
        if (y2 < h2 && x2 < w2) gprev[idx] += 0.25f * gcur[(m * h2 + y2) * w2 + x2];
    }
}

// ---- K13 backward.  Pass 1, thread = (b, sub-row i, y, x) like the forward: softmax recomputed, mask gradient written, and the
// partial sums S[b,c,k,i,y,x] = sum_j w_k(i,j) * g_c(i,j) that pass 2 gathers into the data gradient.
__global__ __launch_bounds__(256) void cvx_upsample_bwd_kernel(const float* __restrict__ gup, const float* __restrict__ data,
                                                               const float* __restrict__ mask, float* __restrict__ gmask,
                                                               float* __restrict__ S, int B, int C, int h, int w) {
    const int N = h * w;
    const long long total = (long long)B * 8 * N;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const int bi = (int)(idx / N);
        const int i = bi & 7, b = bi >> 3;
        const int y = n / w, x = n - y * w;
        const float* mb = mask + (long long)b * 576 * N + n;
        float m[9][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k][j] = mb[(long long)(k * 64 + i * 8 + j) * N];
                mx = fmaxf(mx, m[k][j]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k][j] = expf(m[k][j] - mx);
                s += m[k][j];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) m[k][j] = m[k][j] / s;
        }
        float gw[9][8];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) gw[k][j] = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* d = data + ((long long)b * C + c) * N;
            float nb[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    nb[ky * 3 + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? 8.f * d[yy * w + xx] : 0.f;
                }
            const float* gpn = gup + (((long long)b * C + c) * (8 * h) + (8 * y + i)) * (8LL * w) + 8 * x;
            const float4 g0 = *reinterpret_cast<const float4*>(gpn), g1 = *reinterpret_cast<const float4*>(gpn + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    gw[k][j] = fmaf(g[j], nb[k], gw[k][j]);
                    s = fmaf(m[k][j], g[j], s);
                }
                S[((((long long)b * C + c) * 9 + k) * 8 + i) * N + n] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) dot = fmaf(m[k][j], gw[k][j], dot);
#pragma unroll
            for (int k = 0; k < 9; ++k) gmask[((long long)b * 576 + k * 64 + i * 8 + j) * N + n] = m[k][j] * (gw[k][j] - dot);
        }
    }
}

// Pass 2: gdata[b,c,y',x'] = 8 * sum_k sum_i S[b,c,k,i, y'-ky+1, x'-kx+1]   (the low-res pixels whose tap k lands on (y',x'))
__global__ __launch_bounds__(256) void cvx_upsample_bwd_gather_kernel(const float* __restrict__ S, float* __restrict__ gdata, int B,
                                                                      int C, int h, int w) {
    const int N = h * w;
    const long long total = (long long)B * C * N;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const long long bc = idx / N;
        const int y = n / w, x = n - y * w;
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int ys = y - ky + 1, xs = x - kx + 1;
                if (ys < 0 || ys >= h || xs < 0 || xs >= w) continue;
                const float* sp = S + ((bc * 9 + ky * 3 + kx) * 8) * N + ys * w + xs;
                for (int i = 0; i < 8; ++i) acc += sp[(long long)i * N];
            }
        gdata[idx] = 8.f * acc;
    }
}

// ---- masked L1 (utils/losses.py:6-22): acc[0] += sum over valid positions of sum_c |s - t|, acc[1] += #valid positions
__global__ __launch_bounds__(256) void l1_masked_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                        const unsigned char* __restrict__ valid, int B, int C, long long HW,
                                                        double* __restrict__ acc) {
    double s = 0.0, cnt = 0.0;
    const long long total = (long long)B * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        if (valid && !valid[idx]) continue;
        const long long b = idx / HW, n = idx - b * HW;
        float v = 0.f;
        for (int c = 0; c < C; ++c) {
            const long long o = (b * C + c) * HW + n;
            v += fabsf(src[o] - tgt[o]);
        }
        s += (double)v;
        cnt += 1.0;
    }
    __shared__ double red[2][4];
    s = bflow::wave_sum(s);
    cnt = bflow::wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(acc + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// grad (+)= weight * upstream / count * sign(s - t) on valid positions (0 elsewhere); count = acc[1] of the forward (device)
__global__ __launch_bounds__(256) void l1_masked_grad_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                             const unsigned char* __restrict__ valid, int B, int C, long long HW,
                                                             const double* __restrict__ acc, const float* __restrict__ upstream,
                                                             float weight, float* __restrict__ grad) {
    const float scale = weight * (upstream ? upstream[0] : 1.f) / (float)acc[1];
    const long long total = (long long)B * C * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx % HW, b = idx / (HW * C);
        const bool ok = !valid || valid[b * HW + n];
        const float d = src[idx] - tgt[idx];
        grad[idx] = ok ? (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) : 0.f;
    }
}

}  // namespace

extern "C" int bflow_corr_lookup_bezier_bwd(const bflow_plane_t* planes, float* const* grad_planes, int P, const float* params,
                                            const float* coef, int T, int deg, const float* grad_out, float* grad_coords, int B, int h1,
                                            int w1, bflow_stream_t stream) {
    LookupBwdArgs a;
    if (int rc = fill_bwd_args(a, planes, grad_planes, P, T)) return rc;
    BFLOW_REQUIRE(params && coef && grad_out && grad_coords && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_bezier_bwd: bad arguments");
    BFLOW_REQUIRE(deg >= 1 && deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "corr_lookup_bezier_bwd: degree %d", deg);
    a.deg = deg;
    for (int i = 0; i < T * deg; ++i) a.coef[i] = coef[i];
    dim3 grid(bflow::ceil_div((long long)h1 * w1, BT), P, B);
    hipLaunchKernelGGL((corr_lookup_bwd_kernel<true>), grid, dim3(BTHREADS), 0, (hipStream_t)stream, a, params, grad_out, grad_coords, B, h1, w1);
    return bflow::launch_status("corr_lookup_bezier_bwd");
}

extern "C" int bflow_corr_lookup_bwd(const bflow_plane_t* planes, float* const* grad_planes, int P, const float* coords, int T,
                                     const float* grad_out, float* grad_coords, int B, int h1, int w1, bflow_stream_t stream) {
    LookupBwdArgs a;
    if (int rc = fill_bwd_args(a, planes, grad_planes, P, T)) return rc;
    BFLOW_REQUIRE(coords && grad_out && grad_coords && B > 0 && h1 > 0 && w1 > 0, BFLOW_E_ARG, "corr_lookup_bwd: bad arguments");
    dim3 grid(bflow::ceil_div((long long)h1 * w1, BT), P, B);
    hipLaunchKernelGGL((corr_lookup_bwd_kernel<false>), grid, dim3(BTHREADS), 0, (hipStream_t)stream, a, coords, grad_out, grad_coords, B, h1, w1);
    return bflow::launch_status("corr_lookup_bwd");
}

extern "C" int bflow_corr_pool2x2_bwd(const float* grad_cur, float* grad_prev, long long M, int h, int w, bflow_stream_t stream) {
    BFLOW_REQUIRE(grad_cur && grad_prev && M > 0 && h >= 2 && w >= 2, BFLOW_E_ARG, "corr_pool2x2_bwd: bad arguments");
    const long long total = M * h * w;
    hipLaunchKernelGGL(corr_pool2x2_bwd_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, grad_cur, grad_prev, M, h, w);
    return bflow::launch_status("corr_pool2x2_bwd");
}

extern "C" int bflow_cvx_upsample_bwd(const float* grad_up, const float* data, const float* mask, float* grad_data, float* grad_mask,
                                      float* scratch, int B, int C, int h, int w, bflow_stream_t stream) {
    BFLOW_REQUIRE(grad_up && data && mask && grad_data && grad_mask && scratch && B > 0 && C > 0 && h > 0 && w > 0, BFLOW_E_ARG,
                  "cvx_upsample_bwd: bad arguments");
    BFLOW_REQUIRE(((uintptr_t)grad_up & 15) == 0, BFLOW_E_ARG, "cvx_upsample_bwd: grad_up must be 16-byte aligned");
    hipLaunchKernelGGL(cvx_upsample_bwd_kernel, dim3(bflow::stream_grid((long long)B * 8 * h * w, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_up, data, mask, grad_mask, scratch, B, C, h, w);
    if (int rc = bflow::launch_status("cvx_upsample_bwd")) return rc;
    hipLaunchKernelGGL(cvx_upsample_bwd_gather_kernel, dim3(bflow::stream_grid((long long)B * C * h * w, 256)), dim3(256), 0,
                       (hipStream_t)stream, scratch, grad_data, B, C, h, w);
    return bflow::launch_status("cvx_upsample_bwd_gather");
}

extern "C" int bflow_l1_masked_accumulate(const float* src, const float* tgt, const unsigned char* valid, int B, int C, long long HW,
                                          double* acc, bflow_stream_t stream) {
    BFLOW_REQUIRE(src && tgt && acc && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "l1_masked_accumulate: bad arguments");
    hipLaunchKernelGGL(l1_masked_kernel, dim3(bflow::reduce_grid((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, src, tgt,
                       valid, B, C, HW, acc);
    return bflow::launch_status("l1_masked_accumulate");
}

extern "C" int bflow_l1_masked_grad(const float* src, const float* tgt, const unsigned char* valid, int B, int C, long long HW,
                                    const double* acc, const float* upstream, float weight, float* grad, bflow_stream_t stream) {
    BFLOW_REQUIRE(src && tgt && acc && grad && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "l1_masked_grad: bad arguments");
    hipLaunchKernelGGL(l1_masked_grad_kernel, dim3(bflow::stream_grid((long long)B * C * HW, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       tgt, valid, B, C, HW, acc, upstream, weight, grad);
    return bflow::launch_status("l1_masked_grad");
}
