// A kernel that holds `blocks` workgroups on the chip for a given time (test harness for CU contention: stands for a
// long-running kernel of another stream, such as a collective).  Time-bounded: every wave leaves after `us` microseconds.
#include <hip/hip_runtime.h>
__global__ void hog_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); ++n; }
    if (n < 0) *sink = n;
}
extern "C" int hog_launch(int blocks, int threads, long long us, void* stream) {
    int rate = 0;  // kHz
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    if (rate <= 0) rate = 100000;
    hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, us * rate / 1000, (int*)nullptr);
    return (int)hipGetLastError();
}
