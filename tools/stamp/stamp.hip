// Debug-only (tools/): one-thread kernel that writes the 100 MHz wall clock into a slot -- timeline of a hipGraph replay
// without the profiler.  Not part of the product library.
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
extern "C" int stamp(unsigned long long* slot, void* stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    return (int)hipGetLastError();
}
