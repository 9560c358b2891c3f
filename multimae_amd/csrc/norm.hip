// LayerNorm forward / backward for the fp32 residual stream (gfx950). One wave per row, the row held
// in registers; HBM-bound: forward moves 4*dim (read) + 2*dim (bf16 write) bytes per row.
#include "common.h"
#include "ln_rows.h"
#include <stdlib.h>

#define LN_MAXV 4  // float4 per lane: dim <= 64 * 4 * LN_MAXV = 1024

// YT = bf16 (the operand type of the product that follows) or float (decoders that run outside the low-precision regime)
template <typename YT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, YT* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                            int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = dim >> 2;
    const f32x4* xr = (const f32x4*)(x + (long)row * dim);
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            v[i] = xr[c];
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mu = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 d = v[i] - mu;
            q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
    }
    const float rs = rsqrtf(wave_sum(q) / dim + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    YT* yr = y + (long)row * dim;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 g = ((const f32x4*)gamma)[c], b = ((const f32x4*)beta)[c];
            const f32x4 o = (v[i] - mu) * rs * g + b;
            if constexpr (sizeof(YT) == 4) {
                *(f32x4*)(yr + c * 4) = o;
            } else {
                bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                *(bf16x4*)(yr + c * 4) = ob;
            }
        }
    }
}

// Forward for dim == 256 * NV, R rows per wave and pass, persistent waves (the row loop is ln_fwd_rows, ln_rows.h).
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, bf16* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                                 float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    ln_fwd_rows<NV, R>(x, gamma, beta, y, mean, rstd, rows, eps, (blockIdx.x * 4 + w) * R, gridDim.x * 4 * R, lane);
}

// Backward. Each wave walks rows `wave_global, wave_global + total_waves, ...`, keeps its dgamma / dbeta
// partial sums in registers, and the block writes one partial row pair to `part` at the end.
template <typename DT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const DT* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dx, bf16* __restrict__ dxb,
                                                            float* __restrict__ part, int rows, int dim) {
    __shared__ float red[4][2][64 * 4];  // per wave, per (dgamma|dbeta), one float4 column group at a time
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nv = dim >> 2;
    f32x4 dg[LN_MAXV], db[LN_MAXV], gm[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int c = lane + i * 64;
        gm[i] = c < nv ? ((const f32x4*)gamma)[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const f32x4* xr = (const f32x4*)(x + (long)row * dim);
        const DT* dyr = dy + (long)row * dim;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                f32x4 d;
                if constexpr (sizeof(DT) == 4) {
                    d = *(const f32x4*)(dyr + c * 4);
                } else {
                    const bf16x4 d4 = *(const bf16x4*)(dyr + c * 4);
                    d = (f32x4){(float)d4[0], (float)d4[1], (float)d4[2], (float)d4[3]};
                }
                xh[i] = (xr[c] - mu) * rs;
                g[i] = d * gm[i];
                dg[i] += d * xh[i];
                db[i] += d;
                s1 += g[i][0] + g[i][1] + g[i][2] + g[i][3];
                const f32x4 t = g[i] * xh[i];
                s2 += t[0] + t[1] + t[2] + t[3];
            }
        }
        const float c1 = wave_sum(s1) / dim, c2 = wave_sum(s2) / dim;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                f32x4 o = (g[i] - c1 - xh[i] * c2) * rs;
                if (dres) o += ((const f32x4*)(dres + (long)row * dim))[c];
                ((f32x4*)(dx + (long)row * dim))[c] = o;
                if (dxb) {
                    bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                    *(bf16x4*)(dxb + (long)row * dim + c * 4) = ob;
                }
            }
        }
    }
    // block-level reduction of the four waves' partial sums, one 64-float4 column group at a time
    float* prow = part + (long)blockIdx.x * 2 * dim;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        __syncthreads();
        *(f32x4*)&red[w][0][lane * 4] = dg[i];
        *(f32x4*)&red[w][1][lane * 4] = db[i];
        __syncthreads();
        if (w < 2) {
            const int c = lane + i * 64;
            if (c < nv) {
                f32x4 t = *(f32x4*)&red[0][w][lane * 4];
                t += *(f32x4*)&red[1][w][lane * 4];
                t += *(f32x4*)&red[2][w][lane * 4];
                t += *(f32x4*)&red[3][w][lane * 4];
                *(f32x4*)(prow + w * dim + c * 4) = t;
            }
        }
    }
}

// Backward for dim == 256 * NV, R rows per wave and pass.  Every global load of the R rows (dy, x, the incoming residual
// gradient, the row statistics) is issued before the first reduction, so a wave keeps R * 12 * dim bytes in flight: with
// one row per pass the loads of a row wait behind the two wave reductions of the previous one and the kernel runs on
// latency, not bandwidth.  Measured cold on MI355X: 768-wide rows 76 -> 54 us (one row per pass is enough once the loads
// lead), 256-wide decoder rows 59 -> 34 us with two rows per pass; more rows or more blocks change nothing (4.4 TB/s,
// the rate the forward kernel streams at).  Rows past the end are clamped for the loads and masked after.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ dres,
                                                                 float* __restrict__ dx, bf16* __restrict__ dxb,
                                                                 float* __restrict__ part, int rows) {
    __shared__ float red[4][2][64 * 4];
    constexpr int dim = 256 * NV;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 dg[NV], db[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gm[i] = ((const f32x4*)gamma)[lane + i * 64];
    }
    const int stride = gridDim.x * 4 * R;
    for (int row0 = (blockIdx.x * 4 + w) * R; row0 < rows; row0 += stride) {
        bf16x4 d4[R][NV];
        f32x4 xv[R][NV], dr[R][NV];
        float mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r < rows ? row0 + r : rows - 1;
            mu[r] = mean[row];
            rs[r] = rstd[row];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                d4[r][i] = *(const bf16x4*)(dy + (long)row * dim + c * 4);
                xv[r][i] = ((const f32x4*)(x + (long)row * dim))[c];
                if (dres) dr[r][i] = ((const f32x4*)(dres + (long)row * dim))[c];
            }
        }
        float s1[R], s2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool live = row0 + r < rows;  // wave-uniform
            s1[r] = s2[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const f32x4 d = {(float)d4[r][i][0], (float)d4[r][i][1], (float)d4[r][i][2], (float)d4[r][i][3]};
                const f32x4 xh = (xv[r][i] - mu[r]) * rs[r];
                const f32x4 g = d * gm[i];
                if (live) {
                    dg[i] += d * xh;
                    db[i] += d;
                }
                s1[r] += g[0] + g[1] + g[2] + g[3];
                const f32x4 t = g * xh;
                s2[r] += t[0] + t[1] + t[2] + t[3];
                xv[r][i] = xh;  // keep x-hat and gamma * dy in place of the loaded values
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)  // the 2 R reductions side by side
#pragma unroll
            for (int r = 0; r < R; ++r) {
                s1[r] += __shfl_xor(s1[r], o, 64);
                s2[r] += __shfl_xor(s2[r], o, 64);
            }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row0 + r >= rows) break;
            const long base = (long)(row0 + r) * dim;
            const float c1 = s1[r] / dim, c2 = s2[r] / dim;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                const f32x4 d = {(float)d4[r][i][0], (float)d4[r][i][1], (float)d4[r][i][2], (float)d4[r][i][3]};
                f32x4 o = (d * gm[i] - c1 - xv[r][i] * c2) * rs[r];
                if (dres) o += dr[r][i];
                ((f32x4*)(dx + base))[c] = o;
                if (dxb) {
                    bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                    *(bf16x4*)(dxb + base + c * 4) = ob;
                }
            }
        }
    }
    float* prow = part + (long)blockIdx.x * 2 * dim;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        __syncthreads();
        *(f32x4*)&red[w][0][lane * 4] = dg[i];
        *(f32x4*)&red[w][1][lane * 4] = db[i];
        __syncthreads();
        if (w < 2) {
            f32x4 t = *(f32x4*)&red[0][w][lane * 4];
            t += *(f32x4*)&red[1][w][lane * 4];
            t += *(f32x4*)&red[2][w][lane * 4];
            t += *(f32x4*)&red[3][w][lane * 4];
            *(f32x4*)(prow + w * dim + (lane + i * 64) * 4) = t;
        }
    }
}

// column sums of the per-block partial rows: 32 columns x 32 row groups per block, fixed order
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ part, int nblocks, int dim,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float t = 0.f;
    if (c < 2 * dim) {
        int b = rg;
        for (; b + 7 * 32 < nblocks; b += 8 * 32) {  // eight loads in flight, additions in order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(long)(b + j * 32) * 2 * dim + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) t += v[j];
        }
        for (; b < nblocks; b += 32) t += part[(long)b * 2 * dim + c];
    }
    red[rg][cl] = t;
    __syncthreads();
    if (rg == 0 && c < 2 * dim) {
        t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][cl];
        float* dst = c < dim ? dgamma + c : dbeta + (c - dim);
        *dst = accumulate ? *dst + t : t;
    }
}

// 0: the multi-row kernel where dim is 256 / 768 / 1024, 1: the generic one-row-per-pass kernel everywhere (for tests and timing)
static thread_local int g_ln_bwd_variant = 0;
extern "C" int mmae_layernorm_bwd_set_variant(int v) {
    MMAE_REQUIRE(v == 0 || v == 1, MMAE_EINVAL, "layernorm_bwd_set_variant: %d is not 0 or 1", v);
    g_ln_bwd_variant = v;
    return MMAE_OK;
}

static int ln_bwd_blocks(int rows) {
    int b = (rows + 3) / 4;
    return b > 512 ? 512 : b;
}

extern "C" int mmae_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                  int rows, int dim, float eps, hipStream_t stream) {
    MMAE_REQUIRE(x && gamma && beta && y && mean && rstd, MMAE_EINVAL, "layernorm_fwd: null pointer");
    MMAE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAXV, MMAE_ESHAPE,
                 "layernorm_fwd: dim=%d must be a multiple of 4 and <= %d", dim, 256 * LN_MAXV);
    return mmae_ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, dim, eps, stream);
}

int mmae_ln_fwd_launch(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int dim,
                       float eps, hipStream_t stream) {
    static int env_rows = -1;  // tuning: MMAE_LN_FWD_ROWS = rows per wave and pass of the multi-row kernel (0 = the one-row kernel)
    if (env_rows < 0) {
        const char* e = getenv("MMAE_LN_FWD_ROWS");
        env_rows = e ? atoi(e) : 2;
    }
#define LN_FWD_ROWS(NV, R)                                                                                              \
    {                                                                                                                   \
        int nb = ((rows + R - 1) / R + 3) / 4;                                                                          \
        nb = nb > 1024 ? 1024 : nb;                                                                                     \
        hipLaunchKernelGGL((layernorm_fwd_rows_kernel<NV, R>), dim3(nb), dim3(256), 0, stream, x, gamma, beta, (bf16*)y, mean, \
                           rstd, rows, eps);                                                                            \
    }
#define LN_FWD_DIM(NV)                        \
    if (env_rows >= 4) LN_FWD_ROWS(NV, 4)     \
    else if (env_rows >= 2) LN_FWD_ROWS(NV, 2) \
    else LN_FWD_ROWS(NV, 1)
    if (env_rows > 0 && dim == 256) { LN_FWD_DIM(1) }
    else if (env_rows > 0 && dim == 768) { LN_FWD_DIM(3) }
    else if (env_rows > 0 && dim == 1024) { LN_FWD_DIM(4) }
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, (bf16*)y, mean, rstd,
                           rows, dim, eps);
#undef LN_FWD_DIM
#undef LN_FWD_ROWS
    MMAE_CHECK_LAUNCH("layernorm_fwd");
    return MMAE_OK;
}

extern "C" size_t mmae_layernorm_bwd_workspace_bytes(int rows, int dim) {
    return (size_t)ln_bwd_blocks(rows) * 2 * dim * sizeof(float);
}

extern "C" int mmae_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                  const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, int accumulate,
                                  int rows, int dim, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(dy && x && gamma && mean && rstd && dx && workspace, MMAE_EINVAL, "layernorm_bwd: null pointer");
    MMAE_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), MMAE_EINVAL, "layernorm_bwd: give both parameter gradients or neither");
    MMAE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAXV, MMAE_ESHAPE,
                 "layernorm_bwd: dim=%d must be a multiple of 4 and <= %d", dim, 256 * LN_MAXV);
    MMAE_REQUIRE(workspace_bytes >= mmae_layernorm_bwd_workspace_bytes(rows, dim), MMAE_EWORKSPACE, "layernorm_bwd: workspace too small");
    const int nb = ln_bwd_blocks(rows);
#define LN_BWD_ROWS(NV, R)                                                                                                  \
    hipLaunchKernelGGL((layernorm_bwd_rows_kernel<NV, R>), dim3(nb), dim3(256), 0, stream, (const bf16*)dy, x, gamma, mean, rstd, \
                       dres, dx, (bf16*)dx_bf16, (float*)workspace, rows)
    if (g_ln_bwd_variant != 1 && dim == 256) LN_BWD_ROWS(1, 2);
    else if (g_ln_bwd_variant != 1 && dim == 768) LN_BWD_ROWS(3, 1);
    else if (g_ln_bwd_variant != 1 && dim == 1024) LN_BWD_ROWS(4, 1);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, stream, (const bf16*)dy, x, gamma, mean, rstd, dres, dx,
                           (bf16*)dx_bf16, (float*)workspace, rows, dim);
#undef LN_BWD_ROWS
    MMAE_CHECK_LAUNCH("layernorm_bwd");
    if (dgamma) return mmae_layernorm_param_reduce(workspace, workspace_bytes, rows, dim, dgamma, dbeta, accumulate, stream);
    return MMAE_OK;
}

// fp32 in, fp32 out (and an fp32 incoming gradient): the LayerNorms of a decoder that runs in fp32 end to end
// (fp32_output_adapters, multimae/multimae.py:426-436 of the reference)
extern "C" int mmae_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                      int rows, int dim, float eps, hipStream_t stream) {
    MMAE_REQUIRE(x && gamma && beta && y && mean && rstd, MMAE_EINVAL, "layernorm_fwd_f32: null pointer");
    MMAE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAXV, MMAE_ESHAPE,
                 "layernorm_fwd_f32: dim=%d must be a multiple of 4 and <= %d", dim, 256 * LN_MAXV);
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, rows, dim, eps);
    MMAE_CHECK_LAUNCH("layernorm_fwd_f32");
    return MMAE_OK;
}

extern "C" int mmae_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                      float* dx, float* dgamma, float* dbeta, int accumulate, int rows, int dim, void* workspace,
                                      size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, MMAE_EINVAL, "layernorm_bwd_f32: null pointer");
    MMAE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAXV, MMAE_ESHAPE,
                 "layernorm_bwd_f32: dim=%d must be a multiple of 4 and <= %d", dim, 256 * LN_MAXV);
    MMAE_REQUIRE(workspace_bytes >= mmae_layernorm_bwd_workspace_bytes(rows, dim), MMAE_EWORKSPACE, "layernorm_bwd_f32: workspace too small");
    hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(ln_bwd_blocks(rows)), dim3(256), 0, stream, dy, x, gamma, mean, rstd,
                       (const float*)nullptr, dx, (bf16*)nullptr, (float*)workspace, rows, dim);
    MMAE_CHECK_LAUNCH("layernorm_bwd_f32");
    return mmae_layernorm_param_reduce(workspace, workspace_bytes, rows, dim, dgamma, dbeta, accumulate, stream);
}

extern "C" int mmae_layernorm_param_reduce(const void* workspace, size_t workspace_bytes, int rows, int dim, float* dgamma,
                                           float* dbeta, int accumulate, hipStream_t stream) {
    MMAE_REQUIRE(workspace && dgamma && dbeta, MMAE_EINVAL, "layernorm_param_reduce: null pointer");
    MMAE_REQUIRE(rows > 0 && dim > 0 && workspace_bytes >= mmae_layernorm_bwd_workspace_bytes(rows, dim), MMAE_EWORKSPACE,
                 "layernorm_param_reduce: workspace too small");
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * dim + 31) / 32), dim3(1024), 0, stream, (const float*)workspace,
                       ln_bwd_blocks(rows), dim, dgamma, dbeta, accumulate);
    MMAE_CHECK_LAUNCH("ln_param_reduce");
    return MMAE_OK;
}
