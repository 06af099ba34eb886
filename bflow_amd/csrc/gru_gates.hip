// SepConvGRU gate arithmetic of the TRAINING path, forward and backward, as four element-wise kernels.
// Reference: SepConvGRU.forward (models/raft_spline/update.py:33-48) under autograd:
//     z = sigmoid(convz(hx)), r = sigmoid(convr(hx)), q = tanh(convq(cat(r*h, x))), h' = (1-z)*h + z*q
// The reference leaves ~10 element-wise launches per GRU half to autograd (and as many again in the backward pass); at batch 3 and
// 36 x 48 pixels each of them is a ~5 us node of the training graph for a few microseconds of HBM traffic.  Here:
//   bflow_gru_zr_fwd     zr_pre (B, 2C, HW) [z | r pre-activations of ONE merged convolution], h (B, C, HW) -> z, r, rh = r*h
//   bflow_gru_zr_bwd     dz, drh, z, r, h -> dzr_pre (B, 2C, HW), dh  [dh = drh * r: only the part through r*h]
//   bflow_gru_blend_fwd  q_pre, z, h -> q = tanh(q_pre), h' = (1-z)*h + z*q
//   bflow_gru_blend_bwd  dh', q, z, h -> dq_pre = dh'*z*(1-q^2), dz = dh'*(q-h), dh = dh'*(1-z)
// (the inference product path has the same arithmetic in the conv engine's epilogues: conv_split.hip GATE_ZR / GATE_BLEND).
// All tensors fp32, contiguous NCHW; C*HW % 4 == 0 and 16-B alignment are required (float4 streams).
#include "common.h"
#include <initializer_list>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
#define EACH4(EXPR_X, EXPR_Y, EXPR_Z, EXPR_W) make_float4(EXPR_X, EXPR_Y, EXPR_Z, EXPR_W)

// i indexes float4s of a (B, C, HW) tensor; the matching z / r pre-activations live at (b, c) and (b, C + c) of (B, 2C, HW)
__global__ __launch_bounds__(256) void gru_zr_fwd_kernel(const float* __restrict__ zr, const float* __restrict__ h, float* __restrict__ z,
                                                         float* __restrict__ r, float* __restrict__ rh, long long n4, long long chw4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long b = i / chw4, e = i - b * chw4;
        const float4 zp = ld4(zr + (2 * b * chw4 + e) * 4), rp = ld4(zr + ((2 * b + 1) * chw4 + e) * 4), hv = ld4(h + i * 4);
        const float4 zv = EACH4(bflow::sigmoidf_(zp.x), bflow::sigmoidf_(zp.y), bflow::sigmoidf_(zp.z), bflow::sigmoidf_(zp.w));
        const float4 rv = EACH4(bflow::sigmoidf_(rp.x), bflow::sigmoidf_(rp.y), bflow::sigmoidf_(rp.z), bflow::sigmoidf_(rp.w));
        st4(z + i * 4, zv);
        st4(r + i * 4, rv);
        st4(rh + i * 4, EACH4(rv.x * hv.x, rv.y * hv.y, rv.z * hv.z, rv.w * hv.w));
    }
}

__global__ __launch_bounds__(256) void gru_zr_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ drh, const float* __restrict__ z,
                                                         const float* __restrict__ r, const float* __restrict__ h, float* __restrict__ dzr,
                                                         float* __restrict__ dh, long long n4, long long chw4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long b = i / chw4, e = i - b * chw4;
        const float4 g = dz ? ld4(dz + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f), gr = ld4(drh + i * 4);
        const float4 zv = ld4(z + i * 4), rv = ld4(r + i * 4), hv = ld4(h + i * 4);
        // d sigmoid = s * (1 - s)  (the form autograd uses: sigmoid_backward(grad, out) = grad * (1 - out) * out)
        st4(dzr + (2 * b * chw4 + e) * 4, EACH4(g.x * (1.f - zv.x) * zv.x, g.y * (1.f - zv.y) * zv.y, g.z * (1.f - zv.z) * zv.z, g.w * (1.f - zv.w) * zv.w));
        st4(dzr + ((2 * b + 1) * chw4 + e) * 4, EACH4(gr.x * hv.x * (1.f - rv.x) * rv.x, gr.y * hv.y * (1.f - rv.y) * rv.y,
                                                      gr.z * hv.z * (1.f - rv.z) * rv.z, gr.w * hv.w * (1.f - rv.w) * rv.w));
        st4(dh + i * 4, EACH4(gr.x * rv.x, gr.y * rv.y, gr.z * rv.z, gr.w * rv.w));
    }
}

__global__ __launch_bounds__(256) void gru_blend_fwd_kernel(const float* __restrict__ qp, const float* __restrict__ z, const float* __restrict__ h,
                                                            float* __restrict__ q, float* __restrict__ hn, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 p = ld4(qp + i * 4), zv = ld4(z + i * 4), hv = ld4(h + i * 4);
        const float4 qv = EACH4(tanhf(p.x), tanhf(p.y), tanhf(p.z), tanhf(p.w));
        st4(q + i * 4, qv);
        st4(hn + i * 4, EACH4((1.f - zv.x) * hv.x + zv.x * qv.x, (1.f - zv.y) * hv.y + zv.y * qv.y, (1.f - zv.z) * hv.z + zv.z * qv.z,
                              (1.f - zv.w) * hv.w + zv.w * qv.w));
    }
}

__global__ __launch_bounds__(256) void gru_blend_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ q, const float* __restrict__ z,
                                                            const float* __restrict__ h, float* __restrict__ dqp, float* __restrict__ dz,
                                                            float* __restrict__ dh, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 g = ld4(dhn + i * 4), qv = ld4(q + i * 4), zv = ld4(z + i * 4), hv = ld4(h + i * 4);
        st4(dqp + i * 4, EACH4(g.x * zv.x * (1.f - qv.x * qv.x), g.y * zv.y * (1.f - qv.y * qv.y), g.z * zv.z * (1.f - qv.z * qv.z),
                               g.w * zv.w * (1.f - qv.w * qv.w)));
        st4(dz + i * 4, EACH4(g.x * (qv.x - hv.x), g.y * (qv.y - hv.y), g.z * (qv.z - hv.z), g.w * (qv.w - hv.w)));
        st4(dh + i * 4, EACH4(g.x * (1.f - zv.x), g.y * (1.f - zv.y), g.z * (1.f - zv.z), g.w * (1.f - zv.w)));
    }
}

bool aligned16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
    return true;
}

}  // namespace

#define GATE_COMMON(NAME)                                                                                                                   \
    BFLOW_REQUIRE(B > 0 && C > 0 && HW > 0 && ((long long)C * HW) % 4 == 0, BFLOW_E_ARG, NAME ": C*HW must be a positive multiple of 4");   \
    const long long chw4 = (long long)C * HW / 4, n4 = chw4 * B;                                                                            \
    const int grid = bflow::stream_grid(n4, 256);

extern "C" int bflow_gru_zr_fwd(const float* zr_pre, const float* h, float* z, float* r, float* rh, int B, int C, long long HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(zr_pre && h && z && r && rh && aligned16({zr_pre, h, z, r, rh}), BFLOW_E_ARG, "gru_zr_fwd: null or unaligned pointer");
    GATE_COMMON("gru_zr_fwd")
    hipLaunchKernelGGL(gru_zr_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, zr_pre, h, z, r, rh, n4, chw4);
    return bflow::launch_status("gru_zr_fwd");
}

extern "C" int bflow_gru_zr_bwd(const float* dz, const float* drh, const float* z, const float* r, const float* h, float* dzr_pre, float* dh, int B, int C,
                                long long HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(drh && z && r && h && dzr_pre && dh && aligned16({dz, drh, z, r, h, dzr_pre, dh}), BFLOW_E_ARG, "gru_zr_bwd: null or unaligned pointer");
    GATE_COMMON("gru_zr_bwd")
    hipLaunchKernelGGL(gru_zr_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dz, drh, z, r, h, dzr_pre, dh, n4, chw4);
    return bflow::launch_status("gru_zr_bwd");
}

extern "C" int bflow_gru_blend_fwd(const float* q_pre, const float* z, const float* h, float* q, float* h_new, int B, int C, long long HW,
                                   bflow_stream_t stream) {
    BFLOW_REQUIRE(q_pre && z && h && q && h_new && aligned16({q_pre, z, h, q, h_new}), BFLOW_E_ARG, "gru_blend_fwd: null or unaligned pointer");
    GATE_COMMON("gru_blend_fwd")
    (void)chw4;
    hipLaunchKernelGGL(gru_blend_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, q_pre, z, h, q, h_new, n4);
    return bflow::launch_status("gru_blend_fwd");
}

extern "C" int bflow_gru_blend_bwd(const float* dh_new, const float* q, const float* z, const float* h, float* dq_pre, float* dz, float* dh, int B, int C,
                                   long long HW, bflow_stream_t stream) {
    BFLOW_REQUIRE(dh_new && q && z && h && dq_pre && dz && dh && aligned16({dh_new, q, z, h, dq_pre, dz, dh}), BFLOW_E_ARG,
                  "gru_blend_bwd: null or unaligned pointer");
    GATE_COMMON("gru_blend_bwd")
    (void)chw4;
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dh_new, q, z, h, dq_pre, dz, dh, n4);
    return bflow::launch_status("gru_blend_bwd");
}
