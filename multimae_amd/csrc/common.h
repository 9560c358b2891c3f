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

// acc + sum of the eight bf16 values of a packed fragment: four v_dot2c_f32_bf16 against (1, 1) instead of eight
// widen-and-add pairs (the row sum of dS has to be taken over the rounded values the MFMA sees)
__device__ __forceinline__ float add_bf16x8(float acc, const bf16x8& v) {
    typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
    const bf16x2 ones = {(bf16)1.0f, (bf16)1.0f};
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 0, 1), ones, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 2, 3), ones, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 4, 5), ones, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 6, 7), ones, acc, false);
    return acc;
}

// GELU and its derivative on the packed fp32 ALU, without transcendentals.  Both Phi(x) - 1/2 and gelu'(x) - 1/2 are odd
// functions: x * P(u) with u = x^2 scaled to [-1, 1] (least squares on Chebyshev nodes, scripts/fit_gelu_poly.py), fitted
// under the condition that the value at the end of the range is exactly 1/2: the argument is clamped to the range and the
// saturated branches (+-1/2 beyond |x| = 4, 4.5) need no select.  Absolute error <= 3.8e-5 (6.9e-5), below the bf16
// rounding of the stored result for every value that matters; a GEMM epilogue with the rcp/exp2 form spent a quarter of the
// tile time here.  NP pairs are evaluated side by side, coefficient by coefficient: a dependent chain of packed FMAs costs
// a wait state per link, independent chains fill them.
typedef __attribute__((ext_vector_type(2))) float f32x2;
// r = xc * P(w) + 1/2 with xc = clamp(x, -xmax, xmax) and w = xc^2 * scale + shift (scale = 1, shift = 0: P is given in x^2
// itself, one instruction fewer; accurate enough in fp32 for the degree-7 polynomial of GELU, not for the degree-9 one of
// its derivative, which keeps the argument scaled to [-1, 1])
template <int NP, int NC, bool SCALED>
__device__ __forceinline__ void odd_poly2_half(const f32x2 (&x)[NP], f32x2 (&r)[NP], const float (&c)[NC], float scale, float xmax) {
    f32x2 xc[NP], w[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        xc[i] = (f32x2){__builtin_amdgcn_fmed3f(x[i][0], -xmax, xmax), __builtin_amdgcn_fmed3f(x[i][1], -xmax, xmax)};
        w[i] = xc[i] * xc[i];
        if (SCALED) w[i] = w[i] * scale - 1.0f;
        r[i] = (f32x2){c[NC - 1], c[NC - 1]};
    }
#pragma unroll
    for (int k = NC - 2; k >= 0; --k)
#pragma unroll
        for (int i = 0; i < NP; ++i) r[i] = r[i] * w[i] + c[k];
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = xc[i] * r[i] + 0.5f;
}
template <int NP>
__device__ __forceinline__ void gelu_poly2(const f32x2 (&x)[NP], f32x2 (&g)[NP]) {  // x * Phi(x)
    // in x^2 (scripts/fit_gelu_poly.py: the fit in u = x^2 / 8 - 1 re-expanded in double precision)
    const float c[8] = {0.398851215839386f, -0.066166453063488f, 0.009651601314544678f, -0.0010426313383504748f,
                        7.960928633110598e-05f, -3.9993565223994665e-06f, 1.1691168566585475e-07f, -1.4918947188746756e-09f};
    odd_poly2_half<NP, 8, false>(x, g, c, 1.0f, 4.0f);
#pragma unroll
    for (int i = 0; i < NP; ++i) g[i] = x[i] * g[i];
}
template <int NP>
__device__ __forceinline__ void gelu_grad_poly2(const f32x2 (&x)[NP], f32x2 (&d)[NP]) {  // Phi(x) + x * phi(x)
    const float c[10] = {0.15944573283195496f, -0.08991280198097229f, 0.0861779972910881f, -0.09547753632068634f, 0.10513326525688171f, -0.08748847991228104f, 0.04747903347015381f, -0.04563887044787407f, 0.05615197867155075f, -0.024759210646152496f};
    odd_poly2_half<NP, 10, true>(x, d, c, 2.0f / 20.25f, 4.5f);
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
