// K15: end-point-error partial sums (reference: epe_masked / EPE metric state, utils/metrics.py:30-49,196-213).
// Streaming reduction: per-pixel sqrt(sum_c d^2) in fp32 (as the reference), accumulated in fp64 with wavefront
// shuffles and one fp64 atomic per block.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void epe_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                  const unsigned char* __restrict__ valid, int B, int C, long long HW, double* acc) {
    __shared__ double sh[2][4];
    double s = 0.0, cnt = 0.0;
    const long long total = (long long)B * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        if (valid && !valid[idx]) continue;
        const long long b = idx / HW, i = idx - b * HW;
        float ss = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = pred[(b * C + c) * HW + i] - gt[(b * C + c) * HW + i];
            ss += d * d;
        }
        s += (double)sqrtf(ss);
        cnt += 1.0;
    }
    s = bflow::wave_sum(s);
    cnt = bflow::wave_sum(cnt);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = s;
        sh[1][wv] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc + 0, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(acc + 1, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

// Validation metrics in one pass (SURVEY f-3; utils/metrics.py:160-193,259-296): per valid pixel
//   epe  = ||p - g||_2                                      -> acc[0] += epe, acc[1] += 1
//   ae   = acos(clamp(<(p,1),(g,1)> / (||(p,1)|| ||(g,1)||), -1, 1))   [radians]   -> acc[2] += ae
//   npe_k: epe > thr[k]  AND  epe / max(||g||, 1e-6) >= 0.05            -> acc[3+k] += 1      (k = 0..2)
// all per-pixel arithmetic in fp32 like the reference, sums in fp64.
__global__ __launch_bounds__(256) void flow_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           const unsigned char* __restrict__ valid, int B, int C, long long HW, float t0,
                                                           float t1, float t2, double* acc) {
#pragma clang fp contract(off)   // the reference multiplies and adds separately (no FMA); acos near 1 amplifies every ulp of the cosine
    __shared__ double sh[6][4];
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const long long total = (long long)B * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        if (valid && !valid[idx]) continue;
        const long long b = idx / HW, i = idx - b * HW;
        float dd = 0.f, gg = 0.f, pp = 0.f, pg = 0.f;
        for (int c = 0; c < C; ++c) {
            const float p = pred[(b * C + c) * HW + i], g = gt[(b * C + c) * HW + i];
            const float d = p - g;
            dd += d * d;
            gg += g * g;
            pp += p * p;
            pg += p * g;
        }
        const float err = sqrtf(dd), gm = sqrtf(gg);
        float cs = (pg + 1.0f) / (sqrtf(pp + 1.0f) * sqrtf(gg + 1.0f));
        cs = fminf(fmaxf(cs, -1.0f), 1.0f);
        const bool rel = err / fmaxf(gm, 1e-6f) >= 0.05f;
        s[0] += (double)err;
        s[1] += 1.0;
        s[2] += (double)acosf(cs);
        s[3] += (err > t0 && rel) ? 1.0 : 0.0;
        s[4] += (err > t1 && rel) ? 1.0 : 0.0;
        s[5] += (err > t2 && rel) ? 1.0 : 0.0;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        s[k] = bflow::wave_sum(s[k]);
        if (lane == 0) sh[k][wv] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(acc + threadIdx.x, sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// EPE_MULTI.compute_traj_len (utils/metrics.py:60-64): length of the ground-truth polyline through M flow fields, per pixel.
__global__ __launch_bounds__(256) void traj_len_kernel(const float* __restrict__ tg, float* __restrict__ out, int M, int B, int C, long long HW) {
    const long long total = (long long)B * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW, i = idx - b * HW;
        float len = 0.f;
        for (int m = 1; m < M; ++m) {
            float dd = 0.f;
            for (int c = 0; c < C; ++c) {
                const float d = tg[(((long long)m * B + b) * C + c) * HW + i] - tg[(((long long)(m - 1) * B + b) * C + c) * HW + i];
                dd += d * d;
            }
            len += sqrtf(dd);
        }
        out[idx] = len;
    }
}

// InputPadder.pad (modules/utils.py:63-78): replicate padding of the last two dimensions.
__global__ __launch_bounds__(256) void pad_replicate_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes, int H, int W,
                                                            int pl, int pt, int Ho, int Wo) {
    const long long total = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long p = idx / ((long long)Ho * Wo);
        const int rem = (int)(idx - p * Ho * Wo);
        const int yo = rem / Wo, xo = rem - yo * Wo;
        const int y = min(max(yo - pt, 0), H - 1), xx = min(max(xo - pl, 0), W - 1);
        out[idx] = x[(p * H + y) * W + xx];
    }
}
}  // namespace

extern "C" int bflow_flow_metrics_accumulate(const float* pred, const float* gt, const unsigned char* valid, int B, int C, long long HW,
                                             float n_pixels0, float n_pixels1, float n_pixels2, double* acc, bflow_stream_t stream) {
    BFLOW_REQUIRE(pred && gt && acc && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "flow_metrics_accumulate: bad arguments");
    hipLaunchKernelGGL(flow_metrics_kernel, dim3(bflow::reduce_grid((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, valid,
                       B, C, HW, n_pixels0, n_pixels1, n_pixels2, acc);
    return bflow::launch_status("flow_metrics_accumulate");
}

extern "C" int bflow_traj_len(const float* targets, float* out, int M, int B, int C, long long HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(targets && out && M >= 1 && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "traj_len: bad arguments");
    hipLaunchKernelGGL(traj_len_kernel, dim3(bflow::stream_grid((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, targets, out, M, B,
                       C, HW);
    return bflow::launch_status("traj_len");
}

extern "C" int bflow_pad_replicate(const float* x, float* out, long long planes, int H, int W, int pad_left, int pad_right, int pad_top,
                                   int pad_bottom, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && out && planes > 0 && H > 0 && W > 0 && pad_left >= 0 && pad_right >= 0 && pad_top >= 0 && pad_bottom >= 0, BFLOW_E_ARG,
                  "pad_replicate: bad arguments");
    const int Ho = H + pad_top + pad_bottom, Wo = W + pad_left + pad_right;
    hipLaunchKernelGGL(pad_replicate_kernel, dim3(bflow::stream_grid(planes * Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream, x, out, planes,
                       H, W, pad_left, pad_top, Ho, Wo);
    return bflow::launch_status("pad_replicate");
}

extern "C" int bflow_epe_accumulate(const float* pred, const float* gt, const unsigned char* valid, int B, int C, long long HW, double* acc,
                                    bflow_stream_t stream) {
    BFLOW_REQUIRE(pred && gt && acc && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "epe_accumulate: bad arguments");
    hipLaunchKernelGGL(epe_kernel, dim3(bflow::reduce_grid((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, valid, B,
                       C, HW, acc);
    return bflow::launch_status("epe_accumulate");
}
