// Shared device/host helpers for the MultiMAE gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mmae.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MMAE_WAVE 64

// ---- error reporting (thread-local message, negative return codes) -------------------
void mmae_set_error(const char* fmt, ...);

#define MMAE_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            mmae_set_error(__VA_ARGS__);         \
            return (code);                       \
        }                                        \
    } while (0)

#define MMAE_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            mmae_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MMAE_ELAUNCH;                                                  \
        }                                                                         \
    } while (0)

// ---- device helpers --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over a 256..1024-thread block; every thread gets the result. `red` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// erf-form GELU for GEMM epilogues.  erf(z) = sign(z) * (1 - P(t) * exp(-z^2)), t = 1 / (1 + p|z|)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7, far below the bf16 rounding of the result): one rcp, one exp and
// six FMAs per element instead of the ~100-instruction branchy libm erff, which made the epilogue longer than the main loop.
// `e` returns exp(-x^2/2) (= exp(-z^2) with z = x/sqrt(2)) for the derivative.
__device__ __forceinline__ float erf_as(float z, float& e) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    e = __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float r = 1.0f - p * t * e;
    return z < 0.f ? -r : r;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float e;
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e));
    return cdf + x * 0.39894228040143267794f * e;
}

// GELU and its derivative for eight values at once on the packed fp32 ALU, without transcendentals.  Both
// Phi(x) - 1/2 and gelu'(x) - 1/2 are odd functions: x * P(u) with u = x^2 scaled to [-1, 1] (least-squares fits on
// Chebyshev nodes), saturating to +-1/2 beyond |x| = 4 (4.5).  Absolute error <= 3.2e-5 (7e-5), below the bf16 rounding of
// the stored result for every value that matters; a GEMM epilogue with the rcp/exp2 form spent a quarter of the tile time here.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 odd_poly2(f32x2 x, const float* c, int n, float inv_half_range, float xmax) {
    const f32x2 u = x * x * inv_half_range - 1.0f;
    f32x2 acc = {c[n - 1], c[n - 1]};
#pragma unroll
    for (int k = n - 2; k >= 0; --k) acc = acc * u + c[k];
    f32x2 r = x * acc;
#pragma unroll
    for (int e = 0; e < 2; ++e) r[e] = fabsf(x[e]) >= xmax ? copysignf(0.5f, x[e]) : r[e];
    return r;
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {  // x * Phi(x)
    const float c[9] = {0.1759490966796875f, -0.08430877327919006f, 0.05595685914158821f, -0.03706042096018791f,
                        0.022424334660172462f, -0.011718817986547947f, 0.006272502709180117f, -0.0038783808704465628f,
                        0.0013566274428740144f};
    return x * (odd_poly2(x, c, 9, 2.0f / 16.0f, 4.0f) + 0.5f);
}
__device__ __forceinline__ f32x2 gelu_grad_poly2(f32x2 x) {  // Phi(x) + x * phi(x)
    const float c[11] = {0.15943185985088348f, -0.09001392871141434f, 0.08697269856929779f, -0.09390359371900558f,
                         0.09800127893686295f, -0.09394010901451111f, 0.06953548640012741f, -0.03601986914873123f,
                         0.02863761968910694f, -0.029474526643753052f, 0.011901799589395523f};
    return odd_poly2(x, c, 11, 2.0f / 20.25f, 4.5f) + 0.5f;
}

// async 16-byte global -> LDS copy; `lds_wave_base` must be wave-uniform, lane i lands at +16*i
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware block id remap: blocks id, id+8, ... share an XCD (and its L2); give each of the
// 8 groups a contiguous run of logical ids so neighbouring tiles reuse operand panels in L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, j = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}
