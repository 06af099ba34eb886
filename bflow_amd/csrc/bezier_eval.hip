// K8: Bezier curve evaluation at the look-up / output times (reference: BezierCurves.get_flow_from_reference,
// models/raft_spline/bezier.py:185,188-216; coords0 of raft.py:181 optionally added).
#include "common.h"

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
