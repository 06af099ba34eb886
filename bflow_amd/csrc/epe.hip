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
}  // namespace

extern "C" int bflow_epe_accumulate(const float* pred, const float* gt, const unsigned char* valid, int B, int C, long long HW, double* acc,
                                    bflow_stream_t stream) {
    BFLOW_REQUIRE(pred && gt && acc && B > 0 && C > 0 && HW > 0, BFLOW_E_ARG, "epe_accumulate: bad arguments");
    hipLaunchKernelGGL(epe_kernel, dim3(bflow::stream_grid((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, valid, B,
                       C, HW, acc);
    return bflow::launch_status("epe_accumulate");
}
