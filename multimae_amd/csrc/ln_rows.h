// LayerNorm forward over rows of width 256 * NV, one wave per row group: the loop shared by the stand-alone kernel
// (norm.hip) and by the workgroups that run it beside the tail tiles of a residual GEMM (gemm.hip).
#pragma once
#include "common.h"

// Rows row0 = first, first + stride, ... (R consecutive rows each) of x f32 [rows, 256 * NV] -> y bf16, mean, rstd.
// Every load of the R rows is issued before the first of the two wave reductions (the second needs the first: mean,
// then variance around it), so the reductions of one pass overlap with the loads of the rows in flight instead of
// serialising one row's round trip per wave; gamma and beta stay in registers across rows.  The arithmetic of a row
// does not depend on which wave takes it, so results are bitwise the same for every split of the rows over waves.
template <int NV, int R>
__device__ __forceinline__ void ln_fwd_rows(const float* __restrict__ x, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, bf16* __restrict__ y, float* __restrict__ mean,
                                            float* __restrict__ rstd, int rows, float eps, int first, int stride, int lane) {
    constexpr int dim = 256 * NV;
    f32x4 gm[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gm[i] = ((const f32x4*)gamma)[lane + i * 64];
        bt[i] = ((const f32x4*)beta)[lane + i * 64];
    }
    for (int row0 = first; row0 < rows; row0 += stride) {
        f32x4 v[R][NV];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[r][i] = ((const f32x4*)(x + (long)row * dim))[lane + i * 64];
        }
        float s[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s[r] += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] = s[r] / dim;  // the mean
            q[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                v[r][i] = v[r][i] - s[r];
                // explicit fused multiply-adds in a fixed order: left to the compiler, the contraction of these products
                // differs between instantiations (R), and the statistics must not depend on who computes a row
#pragma unroll
                for (int e = 0; e < 4; ++e) q[r] = __builtin_fmaf(v[r][i][e], v[r][i][e], q[r]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row0 + r >= rows) break;
            const float rs = rsqrtf(q[r] / dim + eps);
            if (lane == 0) { mean[row0 + r] = s[r]; rstd[row0 + r] = rs; }
            bf16* yr = y + (long)(row0 + r) * dim;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(v[r][i][e] * rs, gm[i][e], bt[i][e]);
                bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                *(bf16x4*)(yr + (lane + i * 64) * 4) = ob;
            }
        }
    }
}

// LayerNorm forward launch on `stream` (norm.hip): the body of mmae_layernorm_fwd without its argument checks.
int mmae_ln_fwd_launch(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int dim,
                       float eps, hipStream_t stream);
