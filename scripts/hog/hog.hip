// A kernel that holds `blocks` workgroups on the chip for a given time (test harness for CU contention: stands for a
// long-running kernel of another stream, such as a collective).  Time-bounded: every wave leaves after `us` microseconds.
#include <hip/hip_runtime.h>
__global__ void hog_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); ++n; }
    if (n < 0) *sink = n;
}
// The same with about a hundred live vector registers per lane and 16 KB of LDS per workgroup, the footprint of a collective's
// kernel rather than of an idle one: a workgroup that needs whole SIMDs' register files cannot be placed beside it.
__global__ __launch_bounds__(256) void hog_fat_kernel(long long ticks, float* sink) {
    __shared__ float lds[4096];
    float r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = r[5];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = r[i] * 1.0001f + lds[(threadIdx.x + i) & 4095];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += r[i];
    if (s == 12345.678f) *sink = s;
}
extern "C" int hog_fat_launch(int blocks, int threads, long long us, void* stream) {
    int rate = 0;  // kHz
    if (hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0) != hipSuccess || rate <= 0) rate = 100000;
    hipLaunchKernelGGL(hog_fat_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, us * rate / 1000, (float*)nullptr);
    return (int)hipGetLastError();
}
extern "C" int hog_launch(int blocks, int threads, long long us, void* stream) {
    int rate = 0;  // kHz
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    if (rate <= 0) rate = 100000;
    hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, us * rate / 1000, (int*)nullptr);
    return (int)hipGetLastError();
}
