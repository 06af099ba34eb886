// InstanceNorm2d / BatchNorm2d (+ ReLU) of the TRAINING path, forward and backward.
// Reference: the norm layers of BasicEncoder / ResidualBlock (models/raft_utils/extractor.py:5-55,58-125) under autograd --
// torch.nn.InstanceNorm2d (no affine, no running statistics) in the feature encoders, torch.nn.BatchNorm2d (affine, running statistics,
// momentum 0.1) in the context encoder, each followed by ReLU except on the down-sampling shortcut.
//   forward   per-plane (sum, sum of squares) come from bflow_plane_stats (fp64);
//             bflow_norm_train_finalize: statistics of the normalisation set (plane | channel over the batch) -> mean, rstd per plane,
//               y = x * scale + shift coefficients, and BatchNorm's running mean / (unbiased) variance update;
//             bflow_norm_train_apply:    y = [relu](x * scale[plane] + shift[plane])
//   backward  g = dy * [pre-activation > 0]  (recomputed from x: nothing but x, mean, rstd is kept for the backward pass)
//             bflow_norm_train_bwd_stats:    per plane  s1 = sum g,  s2 = sum g * xhat      (fp64)
//             bflow_norm_train_bwd_finalize: k1, k2 per plane (means of s1, s2 over the normalisation set), dgamma, dbeta
//             bflow_norm_train_bwd_apply:    dx = gamma * rstd * (g - k1 - xhat * k2)
// All tensors NCHW fp32 contiguous; one workgroup per plane in the reductions, float4 streams in the element-wise passes when HW % 4 == 0.
#include "common.h"
#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void norm_finalize_kernel(const double* __restrict__ stats, int mode, int B, int C, int HW, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                            float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                                            float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    if (mode == 0) {                                   // InstanceNorm: every plane on its own (biased variance)
        for (int b = 0; b < B; ++b) {
            const double s1 = stats[((long long)b * C + c) * 2], s2 = stats[((long long)b * C + c) * 2 + 1];
            const double m = s1 / HW;
            const double var = fmax(s2 / HW - m * m, 0.0);
            const float r = (float)(1.0 / sqrt(var + (double)eps));
            mean[b * C + c] = (float)m;
            rstd[b * C + c] = r;
            scale[b * C + c] = g * r;
            shift[b * C + c] = bt - (float)m * g * r;
        }
    } else {                                           // BatchNorm (training): the channel over the whole batch
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < B; ++b) {
            s1 += stats[((long long)b * C + c) * 2];
            s2 += stats[((long long)b * C + c) * 2 + 1];
        }
        const double n = (double)B * HW;
        const double m = s1 / n;
        const double var = fmax(s2 / n - m * m, 0.0);
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        for (int b = 0; b < B; ++b) {
            mean[b * C + c] = (float)m;
            rstd[b * C + c] = r;
            scale[b * C + c] = g * r;
            shift[b * C + c] = bt - (float)m * g * r;
        }
        if (run_mean && run_var) {                     // torch.nn.BatchNorm2d: running_var takes the UNBIASED batch variance
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * n / fmax(n - 1.0, 1.0));
        }
    }
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ y, int HW, int relu) {
    const long long plane = blockIdx.y;
    const float sc = scale[plane], sh = shift[plane];
    const float* xp = x + plane * HW;
    float* yp = y + plane * HW;
    if ((HW & 3) == 0 && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0) {
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += gridDim.x * 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xp + i);
            float4 o = make_float4(v.x * sc + sh, v.y * sc + sh, v.z * sc + sh, v.w * sc + sh);
            if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            *reinterpret_cast<float4*>(yp + i) = o;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            const float o = xp[i] * sc + sh;
            yp[i] = relu ? fmaxf(o, 0.f) : o;
        }
    }
}

// one workgroup per plane: s1 = sum g, s2 = sum g * xhat with g = dy * [x * scale + shift > 0] (relu) or dy
__global__ __launch_bounds__(256) void norm_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, double* __restrict__ sums, int HW, int relu) {
    __shared__ double sh[2][4];
    const long long plane = blockIdx.x;
    const float m = mean[plane], r = rstd[plane], sc = scale[plane], sf = shift[plane];
    const float* xp = x + plane * HW;
    const float* gp = dy + plane * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float xv = xp[i];
        const float g = (relu && !(xv * sc + sf > 0.f)) ? 0.f : gp[i];
        s1 += g;
        s2 += (double)g * ((xv - m) * r);
    }
    s1 = bflow::wave_sum(s1);
    s2 = bflow::wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s1;
        sh[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 2) sums[plane * 2 + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(const double* __restrict__ sums, int mode, int B, int C, int HW, float* __restrict__ k1,
                                                                float* __restrict__ k2, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (mode == 0) {
        for (int b = 0; b < B; ++b) {
            k1[b * C + c] = (float)(sums[((long long)b * C + c) * 2] / HW);
            k2[b * C + c] = (float)(sums[((long long)b * C + c) * 2 + 1] / HW);
        }
    } else {
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < B; ++b) {
            s1 += sums[((long long)b * C + c) * 2];
            s2 += sums[((long long)b * C + c) * 2 + 1];
        }
        const double n = (double)B * HW;
        for (int b = 0; b < B; ++b) {
            k1[b * C + c] = (float)(s1 / n);
            k2[b * C + c] = (float)(s2 / n);
        }
        if (dgamma) dgamma[c] = (float)s2;             // d/dgamma sum g * (gamma * xhat + beta) = sum g * xhat
        if (dbeta) dbeta[c] = (float)s1;
    }
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ k1,
                                                             const float* __restrict__ k2, float* __restrict__ dx, int HW, int relu) {
    const long long plane = blockIdx.y;
    const float m = mean[plane], r = rstd[plane], sc = scale[plane], sf = shift[plane], a1 = k1[plane], a2 = k2[plane];
    const float* xp = x + plane * HW;
    const float* gp = dy + plane * HW;
    float* op = dx + plane * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float xv = xp[i];
        const float g = (relu && !(xv * sc + sf > 0.f)) ? 0.f : gp[i];
        op[i] = sc * (g - a1 - (xv - m) * r * a2);       // scale = gamma * rstd
    }
}

}  // namespace

extern "C" int bflow_norm_train_finalize(const double* stats, int mode, int B, int C, int HW, float eps, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float* mean, float* rstd, float* scale, float* shift,
                                         bflow_stream_t stream) {
    BFLOW_REQUIRE(stats && mean && rstd && scale && shift && (mode == 0 || mode == 1) && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "norm_train_finalize: bad arguments");
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(bflow::ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, mode, B, C, HW, eps, gamma, beta,
                       running_mean, running_var, momentum, mean, rstd, scale, shift);
    return bflow::launch_status("norm_train_finalize");
}

extern "C" int bflow_norm_train_apply(const float* x, const float* scale, const float* shift, float* y, long long planes, int HW, int relu,
                                      bflow_stream_t stream) {
    BFLOW_REQUIRE(x && scale && shift && y && planes > 0 && planes <= 65535 && HW > 0, BFLOW_E_ARG, "norm_train_apply: bad arguments");
    dim3 grid(std::max(1, std::min(64, bflow::ceil_div(HW, 4096))), (unsigned)planes);
    hipLaunchKernelGGL(norm_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, HW, relu);
    return bflow::launch_status("norm_train_apply");
}

extern "C" int bflow_norm_train_bwd_stats(const float* dy, const float* x, const float* mean, const float* rstd, const float* scale, const float* shift,
                                          double* sums, long long planes, int HW, int relu, bflow_stream_t stream) {
    BFLOW_REQUIRE(dy && x && mean && rstd && scale && shift && sums && planes > 0 && planes < (1LL << 31) && HW > 0, BFLOW_E_ARG,
                  "norm_train_bwd_stats: bad arguments");
    hipLaunchKernelGGL(norm_bwd_stats_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, scale, shift, sums, HW, relu);
    return bflow::launch_status("norm_train_bwd_stats");
}

extern "C" int bflow_norm_train_bwd_finalize(const double* sums, int mode, int B, int C, int HW, float* k1, float* k2, float* dgamma, float* dbeta,
                                             bflow_stream_t stream) {
    BFLOW_REQUIRE(sums && k1 && k2 && (mode == 0 || mode == 1) && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "norm_train_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(bflow::ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, mode, B, C, HW, k1, k2, dgamma, dbeta);
    return bflow::launch_status("norm_train_bwd_finalize");
}

extern "C" int bflow_norm_train_bwd_apply(const float* dy, const float* x, const float* mean, const float* rstd, const float* scale, const float* shift,
                                          const float* k1, const float* k2, float* dx, long long planes, int HW, int relu, bflow_stream_t stream) {
    BFLOW_REQUIRE(dy && x && mean && rstd && scale && shift && k1 && k2 && dx && planes > 0 && planes <= 65535 && HW > 0, BFLOW_E_ARG,
                  "norm_train_bwd_apply: bad arguments");
    dim3 grid(std::max(1, std::min(64, bflow::ceil_div(HW, 1024))), (unsigned)planes);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, scale, shift, k1, k2, dx, HW, relu);
    return bflow::launch_status("norm_train_bwd_apply");
}
