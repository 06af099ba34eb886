// Library-level entry points: version, last-error string, host-side Bezier coefficients.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include "common.h"

namespace bflow {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
}  // namespace bflow

extern "C" int bflow_version(void) { return BFLOW_ABI_VERSION; }

extern "C" const char* bflow_last_error_string(void) { return bflow::g_err; }

namespace {
__global__ void clock_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

// s_memtime is a counter PER CU (tools/micro/memtime_domains: offsets of 10^6..10^12 cycles between the CUs of one XCD, < 120 cycles inside
// a CU), so two stamps are only comparable CU by CU: 2048 one-wave workgroups cover the chip, each writes its CU's two counters into the
// CU's row of a (1024, 2) table, row = (XCD * 8 + shader engine) * 16 + CU (later writers of a CU overwrite earlier ones: same instant).
__global__ void shader_clock_stamp_kernel(unsigned long long* table) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        unsigned long long cyc, rt;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc), "=s"(rt));
        const unsigned row = (((xcc & 7) * 8 + ((hw >> 13) & 7)) * 16 + ((hw >> 8) & 15)) & (BFLOW_CLOCK_TABLE_ROWS - 1);
        table[row * 2] = cyc;
        table[row * 2 + 1] = rt;
    }
}
}  // namespace

extern "C" int bflow_clock_stamp(unsigned long long* slot, bflow_stream_t stream) {
    BFLOW_REQUIRE(slot, BFLOW_E_ARG, "clock_stamp: null slot");
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    return bflow::launch_status("clock_stamp");
}

extern "C" int bflow_shader_clock_stamp(unsigned long long* table, bflow_stream_t stream) {
    BFLOW_REQUIRE(table, BFLOW_E_ARG, "shader_clock_stamp: null table");
    hipLaunchKernelGGL(shader_clock_stamp_kernel, dim3(2048), dim3(64), 0, (hipStream_t)stream, table);
    return bflow::launch_status("shader_clock_stamp");
}

// models/raft_spline/bezier.py:141-180: binom(deg, i) * (1-t)^(deg-i) * t^i in float64, then cast to fp32.
extern "C" int bflow_bezier_coeffs(const double* times, int T, int deg, float* coef_out) {
    BFLOW_REQUIRE(times && coef_out && T > 0 && deg >= 1, BFLOW_E_ARG, "bezier_coeffs: bad arguments");
    BFLOW_REQUIRE(deg <= BFLOW_MAX_DEGREE, BFLOW_E_LIMIT, "bezier_coeffs: degree %d > %d", deg, BFLOW_MAX_DEGREE);
    for (int t = 0; t < T; ++t) {
        const double tm = times[t];
        BFLOW_REQUIRE(tm >= 0.0 && tm <= 1.0, BFLOW_E_ARG, "bezier_coeffs: time %g outside [0,1]", tm);
        double binom = 1.0;  // C(deg, i), exact for these small integers
        for (int i = 1; i <= deg; ++i) {
            binom = binom * (double)(deg - i + 1) / (double)i;
            const double c = std::round(binom) * (std::pow(1.0 - tm, (double)(deg - i)) * std::pow(tm, (double)i));
            coef_out[t * deg + (i - 1)] = (float)c;
        }
    }
    return 0;
}
