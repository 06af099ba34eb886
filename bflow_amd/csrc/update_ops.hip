// Element-wise pieces of the GRU / Bezier update loop (K8, K9, K10, K12).  All are HBM-streaming kernels over
// (B, C, HW) channel slices of larger NCHW buffers: float4 accesses when HW % 4 == 0 and the slices are 16-B
// aligned, grid-stride, 256-thread blocks.  Conv biases and activations are folded in here so that the MIOpen
// convolutions run bias-free and no torch.cat / add / relu / sigmoid / tanh launch remains in the loop.
// Reference: models/raft_spline/update.py:33-48,88-97,116-126; bezier.py:137-139,185-216; raft.py:145-147,181.
#include <initializer_list>
#include "common.h"

namespace {

using bflow::sigmoidf_;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float act1(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : v; }
__device__ __forceinline__ float4 add4(float4 v, float b) { return make_float4(v.x + b, v.y + b, v.z + b, v.w + b); }
__device__ __forceinline__ float4 act4(float4 v, int act) {
    return make_float4(act1(v.x, act), act1(v.y, act), act1(v.z, act), act1(v.w, act));
}

// Generic driver: f(b, c, hw_index) over (B, C, HW) in steps of VEC pixels.
template <int VEC, class F>
__global__ __launch_bounds__(256) void slice_kernel(int B, int C, int HW, F f) {
    const int hwv = HW / VEC;
    const long long total = (long long)B * C * hwv;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(idx % hwv);
        const long long bc = idx / hwv;
        const int c = (int)(bc % C);
        const int b = (int)(bc / C);
        f(b, c, i * VEC);
    }
}

template <class F1, class F4>
int run_slices(bool vec, int B, int C, int HW, hipStream_t s, F1 f1, F4 f4, const char* what) {
    if (vec) {
        const long long total = (long long)B * C * (HW / 4);
        hipLaunchKernelGGL((slice_kernel<4, F4>), dim3(bflow::stream_grid(total, 256)), dim3(256), 0, s, B, C, HW, f4);
    } else {
        const long long total = (long long)B * C * HW;
        hipLaunchKernelGGL((slice_kernel<1, F1>), dim3(bflow::stream_grid(total, 256)), dim3(256), 0, s, B, C, HW, f1);
    }
    return bflow::launch_status(what);
}

inline bool vec_ok(int HW, std::initializer_list<const void*> ptrs, std::initializer_list<long long> strides) {
    if (HW % 4) return false;
    for (auto p : ptrs)
        if (p && ((uintptr_t)p & 15)) return false;
    for (auto s : strides)
        if (s % 4) return false;
    return true;
}

}  // namespace

extern "C" int bflow_concat2_act(const float* a, long long a_bs, int Ca, const float* bias_a, int act_a, const float* b_, long long b_bs,
                                 int Cb, const float* bias_b, int act_b, float* dst1, long long dst1_bs, float* dst2, long long dst2_bs,
                                 int B, int HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(a && dst1 && B > 0 && HW > 0 && Ca > 0 && Cb >= 0 && (Cb == 0 || b_), BFLOW_E_ARG, "concat2_act: bad arguments");
    const bool vec = vec_ok(HW, {a, b_, dst1, dst2}, {a_bs, b_bs, dst1_bs, dst2 ? dst2_bs : 0});
    auto f1 = [=] __device__(int b, int c, int i) {
        float v;
        if (c < Ca) v = act1(a[b * a_bs + (long long)c * HW + i] + (bias_a ? bias_a[c] : 0.f), act_a);
        else v = act1(b_[b * b_bs + (long long)(c - Ca) * HW + i] + (bias_b ? bias_b[c - Ca] : 0.f), act_b);
        dst1[b * dst1_bs + (long long)c * HW + i] = v;
        if (dst2) dst2[b * dst2_bs + (long long)c * HW + i] = v;
    };
    auto f4 = [=] __device__(int b, int c, int i) {
        float4 v;
        if (c < Ca) v = act4(add4(ld4(a + b * a_bs + (long long)c * HW + i), bias_a ? bias_a[c] : 0.f), act_a);
        else v = act4(add4(ld4(b_ + b * b_bs + (long long)(c - Ca) * HW + i), bias_b ? bias_b[c - Ca] : 0.f), act_b);
        st4(dst1 + b * dst1_bs + (long long)c * HW + i, v);
        if (dst2) st4(dst2 + b * dst2_bs + (long long)c * HW + i, v);
    };
    return run_slices(vec, B, Ca + Cb, HW, (hipStream_t)stream, f1, f4, "concat2_act");
}

extern "C" int bflow_bias_act_inplace(float* x, long long x_bs, const float* bias, int act, int B, int C, int HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(x && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "bias_act_inplace: bad arguments");
    const bool vec = vec_ok(HW, {x}, {x_bs});
    auto f1 = [=] __device__(int b, int c, int i) {
        float* p = x + b * x_bs + (long long)c * HW + i;
        *p = act1(*p + (bias ? bias[c] : 0.f), act);
    };
    auto f4 = [=] __device__(int b, int c, int i) {
        float* p = x + b * x_bs + (long long)c * HW + i;
        st4(p, act4(add4(ld4(p), bias ? bias[c] : 0.f), act));
    };
    return run_slices(vec, B, C, HW, (hipStream_t)stream, f1, f4, "bias_act_inplace");
}

extern "C" int bflow_gru_rh(const float* r_pre, long long r_bs, const float* bias_r, const float* h, long long h_bs, float* rh,
                            long long rh_bs, int B, int C, int HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(r_pre && h && rh && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "gru_rh: bad arguments");
    const bool vec = vec_ok(HW, {r_pre, h, rh}, {r_bs, h_bs, rh_bs});
    auto f1 = [=] __device__(int b, int c, int i) {
        const long long o = (long long)c * HW + i;
        rh[b * rh_bs + o] = sigmoidf_(r_pre[b * r_bs + o] + (bias_r ? bias_r[c] : 0.f)) * h[b * h_bs + o];
    };
    auto f4 = [=] __device__(int b, int c, int i) {
        const long long o = (long long)c * HW + i;
        const float4 r = add4(ld4(r_pre + b * r_bs + o), bias_r ? bias_r[c] : 0.f), hv = ld4(h + b * h_bs + o);
        st4(rh + b * rh_bs + o, make_float4(sigmoidf_(r.x) * hv.x, sigmoidf_(r.y) * hv.y, sigmoidf_(r.z) * hv.z, sigmoidf_(r.w) * hv.w));
    };
    return run_slices(vec, B, C, HW, (hipStream_t)stream, f1, f4, "gru_rh");
}

namespace {
__device__ __forceinline__ float blend1(float z_pre, float q_pre, float h) {
    const float z = sigmoidf_(z_pre);
    return (1.f - z) * h + z * tanhf(q_pre);
}
}  // namespace

extern "C" int bflow_gru_blend(const float* z_pre, long long z_bs, const float* bias_z, const float* q_pre, long long q_bs,
                               const float* bias_q, float* h, long long h_bs, float* h2, long long h2_bs, int B, int C, int HW,
                               bflow_stream_t stream) {
    BFLOW_REQUIRE(z_pre && q_pre && h && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "gru_blend: bad arguments");
    const bool vec = vec_ok(HW, {z_pre, q_pre, h, h2}, {z_bs, q_bs, h_bs, h2 ? h2_bs : 0});
    auto f1 = [=] __device__(int b, int c, int i) {
        const long long o = (long long)c * HW + i;
        const float v = blend1(z_pre[b * z_bs + o] + (bias_z ? bias_z[c] : 0.f), q_pre[b * q_bs + o] + (bias_q ? bias_q[c] : 0.f),
                               h[b * h_bs + o]);
        h[b * h_bs + o] = v;
        if (h2) h2[b * h2_bs + o] = v;
    };
    auto f4 = [=] __device__(int b, int c, int i) {
        const long long o = (long long)c * HW + i;
        const float4 z = add4(ld4(z_pre + b * z_bs + o), bias_z ? bias_z[c] : 0.f);
        const float4 q = add4(ld4(q_pre + b * q_bs + o), bias_q ? bias_q[c] : 0.f);
        const float4 hv = ld4(h + b * h_bs + o);
        const float4 v = make_float4(blend1(z.x, q.x, hv.x), blend1(z.y, q.y, hv.y), blend1(z.z, q.z, hv.z), blend1(z.w, q.w, hv.w));
        st4(h + b * h_bs + o, v);
        if (h2) st4(h2 + b * h2_bs + o, v);
    };
    return run_slices(vec, B, C, HW, (hipStream_t)stream, f1, f4, "gru_blend");
}

extern "C" int bflow_tanh_relu_split(const float* cnet, long long cnet_bs, const float* bias, int C_h, int C_i, float* net, long long net_bs,
                                     float* inp, long long inp_bs, int B, int HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(cnet && net && inp && B > 0 && HW > 0 && C_h > 0 && C_i > 0, BFLOW_E_ARG, "tanh_relu_split: bad arguments");
    const bool vec = vec_ok(HW, {cnet, net, inp}, {cnet_bs, net_bs, inp_bs});
    auto f1 = [=] __device__(int b, int c, int i) {
        const float v = cnet[b * cnet_bs + (long long)c * HW + i] + (bias ? bias[c] : 0.f);
        if (c < C_h) net[b * net_bs + (long long)c * HW + i] = tanhf(v);
        else inp[b * inp_bs + (long long)(c - C_h) * HW + i] = fmaxf(v, 0.f);
    };
    auto f4 = [=] __device__(int b, int c, int i) {
        const float4 v = add4(ld4(cnet + b * cnet_bs + (long long)c * HW + i), bias ? bias[c] : 0.f);
        if (c < C_h) st4(net + b * net_bs + (long long)c * HW + i, make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)));
        else st4(inp + b * inp_bs + (long long)(c - C_h) * HW + i, act4(v, 1));
    };
    return run_slices(vec, B, C_h + C_i, HW, (hipStream_t)stream, f1, f4, "tanh_relu_split");
}

extern "C" int bflow_add_delta(float* params, const float* delta, const float* bias, int B, int C, int HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(params && delta && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "add_delta: bad arguments");
    const long long bs = (long long)C * HW;
    const bool vec = vec_ok(HW, {params, delta}, {});
    auto f1 = [=] __device__(int b, int c, int i) {
        const long long o = b * bs + (long long)c * HW + i;
        params[o] = params[o] + (delta[o] + (bias ? bias[c] : 0.f));
    };
    auto f4 = [=] __device__(int b, int c, int i) {
        const long long o = b * bs + (long long)c * HW + i;
        const float4 p = ld4(params + o), d = add4(ld4(delta + o), bias ? bias[c] : 0.f);
        st4(params + o, make_float4(p.x + d.x, p.y + d.y, p.z + d.z, p.w + d.w));
    };
    return run_slices(vec, B, C, HW, (hipStream_t)stream, f1, f4, "add_delta");
}

namespace {
struct CoefArg {
    float c[BFLOW_MAX_TARGETS * BFLOW_MAX_DEGREE];
};

__global__ __launch_bounds__(256) void bezier_eval_kernel(const float* __restrict__ params, CoefArg coef, int T, int deg, int B, int h,
                                                          int w, int add_coords0, float* __restrict__ out) {
    const int N = h * w;
    const long long total = (long long)T * B * N;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const long long tb = idx / N;
        const int b = (int)(tb % B), t = (int)(tb / B);
        const float* pp = params + (long long)b * 2 * deg * N + n;
        float fx = 0.f, fy = 0.f;
        for (int i = 0; i < deg; ++i) {
            fx = fmaf(pp[(long long)i * N], coef.c[t * deg + i], fx);
            fy = fmaf(pp[(long long)(deg + i) * N], coef.c[t * deg + i], fy);
        }
        if (add_coords0) {
            const int y = n / w, x = n - y * w;
            fx += (float)x;
            fy += (float)y;
        }
        float* o = out + tb * 2 * N + n;
        o[0] = fx;
        o[N] = fy;
    }
}
}  // namespace

extern "C" int bflow_bezier_eval(const float* params, const float* coef, int T, int deg, int B, int h, int w, int add_coords0,
                                 float* out, bflow_stream_t stream) {
    BFLOW_REQUIRE(params && coef && out && T > 0 && B > 0 && h > 0 && w > 0, BFLOW_E_ARG, "bezier_eval: bad arguments");
    BFLOW_REQUIRE(T <= BFLOW_MAX_TARGETS && deg >= 1 && deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "bezier_eval: T=%d deg=%d", T, deg);
    CoefArg c;
    for (int i = 0; i < T * deg; ++i) c.c[i] = coef[i];
    const long long total = (long long)T * B * h * w;
    hipLaunchKernelGGL(bezier_eval_kernel, dim3(bflow::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, params, c, T, deg, B,
                       h, w, add_coords0, out);
    return bflow::launch_status("bezier_eval");
}
