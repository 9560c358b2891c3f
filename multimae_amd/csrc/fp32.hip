// fp32 path of the output adapters that the caller lists in fp32_output_adapters (multimae/multimae.py:426-436 of the
// reference runs those decoders outside autocast: fp32 weights, activations and products end to end).
//
//  * Products: an fp32 operand is split into two bf16 terms, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa
//    bits together), and a product a w^T is taken as a_hi w_hi^T + a_hi w_lo^T + a_lo w_hi^T (the lo x lo term is below
//    2^-16 of the result) by ONE call of the bf16 MFMA GEMMs over operands concatenated along the reduction index:
//    [a_hi | a_hi | a_lo] x [w_hi | w_lo | w_hi]^T.  All three terms accumulate in the fp32 MFMA accumulators.
//    split_f32_kernel writes those concatenations (column blocks for gemm_nt, row blocks for gemm_tn).
//  * GELU (exact erf form) and its derivative, elementwise in fp32.
//  * Attention in fp32 on the vector ALU: the decoders' sequences are short (99 / 196 tokens, head_dim 32), one thread per
//    query (or key) row, the other side's rows broadcast from LDS.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// split: src f32 [rows, cols] (row stride lds) -> dst bf16, three blocks in the order given by `pattern`
//   pattern 0: hi, hi, lo     pattern 1: hi, lo, hi
//   by_rows 0: blocks side by side, dst [rows, 3 * block_cols];  by_rows 1: blocks stacked, dst [3 * rows, block_cols];
//   block_cols >= cols: columns cols .. block_cols - 1 of every block are zero (so that 3 * block_cols suits the GEMM's k-tile)
// ---------------------------------------------------------------------------------------------
__global__ void split_f32_kernel(const float* __restrict__ src, long lds, int rows, int cols, bf16* __restrict__ dst, int pattern,
                                 int by_rows, int block_cols) {
    const int q = block_cols / 4;  // quads per row of a block, padding included
    const long n4 = (long)rows * q;
    const long ldd = by_rows ? block_cols : 3L * block_cols;
    const long blk = by_rows ? (long)rows * block_cols : block_cols;  // distance between blocks in dst
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / q;
        const int c = (int)(i - r * q) * 4;
        bf16x4 hi = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, lo = hi;
        if (c < cols) {
            const f32x4 v = *(const f32x4*)(src + r * lds + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (bf16)v[e];
                lo[e] = (bf16)(v[e] - (float)hi[e]);
            }
        }
        bf16* d = dst + r * ldd + c;
        *(bf16x4*)d = hi;
        *(bf16x4*)(d + blk) = pattern == 0 ? hi : lo;
        *(bf16x4*)(d + 2 * blk) = pattern == 0 ? lo : hi;
    }
}

extern "C" int mmae_split_f32(const float* src, long ld_src, int rows, int cols, void* dst_bf16, int pattern, int by_rows,
                              int block_cols, hipStream_t stream) {
    MMAE_REQUIRE(src && dst_bf16, MMAE_EINVAL, "split_f32: null pointer");
    if (block_cols == 0) block_cols = cols;
    MMAE_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld_src >= cols && ld_src % 4 == 0 && (pattern == 0 || pattern == 1) &&
                 block_cols >= cols && block_cols % 4 == 0, MMAE_ESHAPE,
                 "split_f32: cols=%d, the row stride %ld and the block width %d must be multiples of 4, pattern 0 or 1", cols, ld_src, block_cols);
    MMAE_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst_bf16 % 8 == 0, MMAE_EINVAL, "split_f32: unaligned pointer");
    const long n4 = (long)rows * (block_cols / 4);
    int blocks = (int)((n4 + 255) / 256); blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(split_f32_kernel, dim3(blocks), dim3(256), 0, stream, src, ld_src, rows, cols, (bf16*)dst_bf16, pattern, by_rows,
                       block_cols);
    MMAE_CHECK_LAUNCH("split_f32");
    return MMAE_OK;
}

// ---------------------------------------------------------------------------------------------
// GELU (erf form, nn.GELU default) and the product of an incoming gradient with its derivative
// ---------------------------------------------------------------------------------------------
__global__ void gelu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)x)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
        ((f32x4*)y)[i] = o;
    }
}
__global__ void gelu_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)x)[i], g = ((const f32x4*)dy)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float cdf = 0.5f * (1.0f + erff(v[e] * 0.70710678118654752f));
            const float pdf = 0.39894228040143268f * __expf(-0.5f * v[e] * v[e]);
            o[e] = g[e] * (cdf + v[e] * pdf);
        }
        ((f32x4*)dx)[i] = o;
    }
}
extern "C" int mmae_gelu_f32(const float* x, float* y, long n, hipStream_t stream) {
    MMAE_REQUIRE(x && y && n > 0 && n % 4 == 0, MMAE_EINVAL, "gelu_f32: bad argument (n must be a multiple of 4)");
    int blocks = (int)((n / 4 + 255) / 256); blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(gelu_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, y, n / 4);
    MMAE_CHECK_LAUNCH("gelu_f32");
    return MMAE_OK;
}
extern "C" int mmae_gelu_bwd_f32(const float* x, const float* dy, float* dx, long n, hipStream_t stream) {
    MMAE_REQUIRE(x && dy && dx && n > 0 && n % 4 == 0, MMAE_EINVAL, "gelu_bwd_f32: bad argument (n must be a multiple of 4)");
    int blocks = (int)((n / 4 + 255) / 256); blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, dy, dx, n / 4);
    MMAE_CHECK_LAUNCH("gelu_bwd_f32");
    return MMAE_OK;
}

// ---------------------------------------------------------------------------------------------
// fp32 attention, head_dim D in {32, 64}.  q / k / v / o rows are addressed as base + (b * len + i) * ld + h * D, like the
// bf16 kernels, so packed qkv / kv buffers are read in place.
//   forward:  one workgroup per (batch, head, block of 256 queries); K and V of the head in LDS ([kv_len][D] floats),
//             one thread per query: scores against the keys (every lane reads the same key row: a broadcast LDS read of 16
//             bytes), running maximum and sum rescaled once per chunk of keys, o in registers; writes o and lse (natural log).
//   backward: delta = rowsum(dO * O) per query by the dQ kernel, which walks the keys like the forward (p recomputed from
//             lse); the dK / dV kernel is the mirror image: one thread per key, Q / dO / lse / delta of the head in LDS.
// ---------------------------------------------------------------------------------------------
struct AttnF32 {
    const float* q; long ldq; const float* k; long ldk; const float* v; long ldv;
    float* o; long ldo; float* lse;
    const float* dO; long lddo; float* dq; long lddq; float* dk; long lddk; float* dv; long lddv; float* delta;
    int B, H, q_len, kv_len; float scale;
};

// rows of one head into LDS, [n][D] floats (every lane of a wavefront later reads the SAME row, a broadcast: no padding needed)
template <int D>
__device__ __forceinline__ void stage_rows_f32(const float* src, long ld, int n, float* img, int tid, int nthreads, float mul = 1.0f) {
    for (int i = tid; i < n * (D / 4); i += nthreads) {
        const int r = i / (D / 4), c = (i - r * (D / 4)) * 4;
        *(f32x4*)(img + r * D + c) = *(const f32x4*)(src + (long)r * ld + c) * mul;
    }
}

// dot product of a register row with an LDS row, 16 bytes per LDS read
template <int D>
__device__ __forceinline__ float dot_row(const float (&a)[D], const float* row, float acc = 0.f) {
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 t = *(const f32x4*)(row + c);
        acc = fmaf(a[c], t[0], acc); acc = fmaf(a[c + 1], t[1], acc); acc = fmaf(a[c + 2], t[2], acc); acc = fmaf(a[c + 3], t[3], acc);
    }
    return acc;
}
// acc += w * row
template <int D>
__device__ __forceinline__ void axpy_row(float (&acc)[D], float w, const float* row) {
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 t = *(const f32x4*)(row + c);
        acc[c] = fmaf(w, t[0], acc[c]); acc[c + 1] = fmaf(w, t[1], acc[c + 1]); acc[c + 2] = fmaf(w, t[2], acc[c + 2]); acc[c + 3] = fmaf(w, t[3], acc[c + 3]);
    }
}

#define AF32_CHUNK 8  // keys per rescaling step of the running softmax

template <int D>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(AttnF32 p) {
    extern __shared__ float f32smem[];
    float* Ks = f32smem;
    float* Vs = Ks + p.kv_len * D;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    stage_rows_f32<D>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.kv_len, Ks, threadIdx.x, 256);
    stage_rows_f32<D>(p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len, Vs, threadIdx.x, 256);
    __syncthreads();
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= p.q_len) return;
    float qr[D], acc[D];
    const float* qp = p.q + ((long)b * p.q_len + i) * p.ldq + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 t = *(const f32x4*)(qp + c);
        qr[c] = t[0] * p.scale; qr[c + 1] = t[1] * p.scale; qr[c + 2] = t[2] * p.scale; qr[c + 3] = t[3] * p.scale;
        acc[c] = acc[c + 1] = acc[c + 2] = acc[c + 3] = 0.f;
    }
    // running softmax, rescaled once per chunk of keys: the scores of a chunk first (independent dot products), their
    // maximum, one correction of the accumulator, then the weighted value rows
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < p.kv_len; j0 += AF32_CHUNK) {
        float s[AF32_CHUNK];
        float mc = m;
#pragma unroll
        for (int u = 0; u < AF32_CHUNK; ++u) {
            s[u] = j0 + u < p.kv_len ? dot_row<D>(qr, Ks + (j0 + u) * D) : -INFINITY;  // block-uniform condition
            mc = fmaxf(mc, s[u]);
        }
        const float corr = __expf(m - mc);
        l *= corr;
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] *= corr;
#pragma unroll
        for (int u = 0; u < AF32_CHUNK; ++u) {
            if (j0 + u < p.kv_len) {
                const float e = __expf(s[u] - mc);
                l += e;
                axpy_row<D>(acc, e, Vs + (j0 + u) * D);
            }
        }
        m = mc;
    }
    const float inv = 1.0f / l;
    float* op = p.o + ((long)b * p.q_len + i) * p.ldo + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4) *(f32x4*)(op + c) = (f32x4){acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv};
    p.lse[((long)b * p.H + h) * p.q_len + i] = m + logf(l);
}

template <int D>
__global__ __launch_bounds__(256) void attn_f32_bwd_dq_kernel(AttnF32 p) {
    extern __shared__ float f32smem[];
    float* Ks = f32smem;
    float* Vs = Ks + p.kv_len * D;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    stage_rows_f32<D>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.kv_len, Ks, threadIdx.x, 256);
    stage_rows_f32<D>(p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len, Vs, threadIdx.x, 256);
    __syncthreads();
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= p.q_len) return;
    const long row = (long)b * p.q_len + i;
    float qr[D], dor[D], acc[D];
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 t = *(const f32x4*)(p.q + row * p.ldq + h * D + c);
        const f32x4 g = *(const f32x4*)(p.dO + row * p.lddo + h * D + c);
        const f32x4 o = *(const f32x4*)(p.o + row * p.ldo + h * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qr[c + e] = t[e] * p.scale; dor[c + e] = g[e]; acc[c + e] = 0.f;
            delta = fmaf(g[e], o[e], delta);
        }
    }
    const long si = ((long)b * p.H + h) * p.q_len + i;
    p.delta[si] = delta;
    const float lse = p.lse[si];
    for (int j = 0; j < p.kv_len; ++j) {
        const float s = dot_row<D>(qr, Ks + j * D), dp = dot_row<D>(dor, Vs + j * D);
        axpy_row<D>(acc, __expf(s - lse) * (dp - delta), Ks + j * D);
    }
    float* op = p.dq + row * p.lddq + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4)
        *(f32x4*)(op + c) = (f32x4){acc[c] * p.scale, acc[c + 1] * p.scale, acc[c + 2] * p.scale, acc[c + 3] * p.scale};
}

template <int D>
__global__ __launch_bounds__(256) void attn_f32_bwd_dkv_kernel(AttnF32 p) {
    extern __shared__ float f32smem[];
    float* Qs = f32smem;                 // [q_len][D], scaled
    float* dOs = Qs + p.q_len * D;
    float* ls = dOs + p.q_len * D;       // lse [q_len]
    float* dl = ls + p.q_len;            // delta [q_len]
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    stage_rows_f32<D>(p.q + (long)b * p.q_len * p.ldq + h * D, p.ldq, p.q_len, Qs, threadIdx.x, 256, p.scale);
    stage_rows_f32<D>(p.dO + (long)b * p.q_len * p.lddo + h * D, p.lddo, p.q_len, dOs, threadIdx.x, 256);
    for (int i = threadIdx.x; i < p.q_len; i += 256) {
        ls[i] = p.lse[((long)b * p.H + h) * p.q_len + i];
        dl[i] = p.delta[((long)b * p.H + h) * p.q_len + i];
    }
    __syncthreads();
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= p.kv_len) return;
    const long row = (long)b * p.kv_len + j;
    float kr[D], vr[D], dk[D], dv[D];
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 t = *(const f32x4*)(p.k + row * p.ldk + h * D + c);
        const f32x4 u = *(const f32x4*)(p.v + row * p.ldv + h * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { kr[c + e] = t[e]; vr[c + e] = u[e]; dk[c + e] = 0.f; dv[c + e] = 0.f; }
    }
    for (int i = 0; i < p.q_len; ++i) {
        const float* qr = Qs + i * D;     // already times scale: s = (scale q) . k, and dk = sum ds (scale q)
        const float* gr = dOs + i * D;
        const float pr = __expf(dot_row<D>(kr, qr) - ls[i]);
        const float ds = pr * (dot_row<D>(vr, gr) - dl[i]);
        axpy_row<D>(dv, pr, gr);
        axpy_row<D>(dk, ds, qr);
    }
    float* okp = p.dk + row * p.lddk + h * D;
    float* ovp = p.dv + row * p.lddv + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        *(f32x4*)(okp + c) = (f32x4){dk[c], dk[c + 1], dk[c + 2], dk[c + 3]};
        *(f32x4*)(ovp + c) = (f32x4){dv[c], dv[c + 1], dv[c + 2], dv[c + 3]};
    }
}

static int attn_f32_check(const char* name, int B, int H, int q_len, int kv_len, int D, size_t lds_bytes) {
    MMAE_REQUIRE(B > 0 && H > 0 && q_len > 0 && kv_len > 0, MMAE_EINVAL, "%s: empty problem", name);
    MMAE_REQUIRE(D == 32 || D == 64, MMAE_ESHAPE, "%s: head_dim %d is not 32 or 64", name, D);
    MMAE_REQUIRE(lds_bytes <= 160 * 1024, MMAE_ESHAPE, "%s: the rows of one head (%zu bytes) do not fit the 160 KiB LDS", name, lds_bytes);
    return MMAE_OK;
}

template <typename KernelT>
static int attn_f32_reserve(KernelT kernel, size_t lds) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MMAE_REQUIRE(e == hipSuccess, MMAE_ELAUNCH, "attn_f32: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    return MMAE_OK;
}

extern "C" int mmae_attn_f32_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo,
                                 float* lse, int B, int H, int q_len, int kv_len, int head_dim, float scale, hipStream_t stream) {
    MMAE_REQUIRE(q && k && v && o && lse, MMAE_EINVAL, "attn_f32_fwd: null pointer");
    const size_t lds = (size_t)2 * kv_len * head_dim * sizeof(float);
    int rc = attn_f32_check("attn_f32_fwd", B, H, q_len, kv_len, head_dim, lds);
    if (rc) return rc;
    MMAE_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, MMAE_ESHAPE, "attn_f32_fwd: row strides must be multiples of 4");
    AttnF32 p = {};
    p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.o = o; p.ldo = ldo; p.lse = lse;
    p.B = B; p.H = H; p.q_len = q_len; p.kv_len = kv_len; p.scale = scale;
    const dim3 grid(B * H, (q_len + 255) / 256);
    if (head_dim == 32) {
        if ((rc = attn_f32_reserve(attn_f32_fwd_kernel<32>, lds))) return rc;
        hipLaunchKernelGGL(attn_f32_fwd_kernel<32>, grid, dim3(256), lds, stream, p);
    } else {
        if ((rc = attn_f32_reserve(attn_f32_fwd_kernel<64>, lds))) return rc;
        hipLaunchKernelGGL(attn_f32_fwd_kernel<64>, grid, dim3(256), lds, stream, p);
    }
    MMAE_CHECK_LAUNCH("attn_f32_fwd");
    return MMAE_OK;
}

extern "C" size_t mmae_attn_f32_bwd_workspace_bytes(int B, int H, int q_len) { return (size_t)B * H * q_len * sizeof(float); }

extern "C" int mmae_attn_f32_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const float* o, long ldo,
                                 const float* d_o, long lddo, const float* lse, float* dq, long lddq, float* dk, long lddk, float* dv,
                                 long lddv, int B, int H, int q_len, int kv_len, int head_dim, float scale, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv && workspace, MMAE_EINVAL, "attn_f32_bwd: null pointer");
    MMAE_REQUIRE(workspace_bytes >= mmae_attn_f32_bwd_workspace_bytes(B, H, q_len), MMAE_EWORKSPACE, "attn_f32_bwd: workspace too small");
    const size_t lds_q = (size_t)2 * kv_len * head_dim * sizeof(float);
    const size_t lds_k = ((size_t)2 * q_len * head_dim + 2 * (size_t)q_len) * sizeof(float);
    int rc = attn_f32_check("attn_f32_bwd", B, H, q_len, kv_len, head_dim, lds_q > lds_k ? lds_q : lds_k);
    if (rc) return rc;
    MMAE_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
                 MMAE_ESHAPE, "attn_f32_bwd: row strides must be multiples of 4");
    AttnF32 p = {};
    p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.o = const_cast<float*>(o); p.ldo = ldo;
    p.lse = const_cast<float*>(lse); p.dO = d_o; p.lddo = lddo; p.dq = dq; p.lddq = lddq; p.dk = dk; p.lddk = lddk; p.dv = dv; p.lddv = lddv;
    p.delta = (float*)workspace; p.B = B; p.H = H; p.q_len = q_len; p.kv_len = kv_len; p.scale = scale;
    const dim3 gq(B * H, (q_len + 255) / 256), gk(B * H, (kv_len + 255) / 256);
    if (head_dim == 32) {
        if ((rc = attn_f32_reserve(attn_f32_bwd_dq_kernel<32>, lds_q))) return rc;
        if ((rc = attn_f32_reserve(attn_f32_bwd_dkv_kernel<32>, lds_k))) return rc;
        hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<32>, gq, dim3(256), lds_q, stream, p);
        hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<32>, gk, dim3(256), lds_k, stream, p);
    } else {
        if ((rc = attn_f32_reserve(attn_f32_bwd_dq_kernel<64>, lds_q))) return rc;
        if ((rc = attn_f32_reserve(attn_f32_bwd_dkv_kernel<64>, lds_k))) return rc;
        hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<64>, gq, dim3(256), lds_q, stream, p);
        hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<64>, gk, dim3(256), lds_k, stream, p);
    }
    MMAE_CHECK_LAUNCH("attn_f32_bwd");
    return MMAE_OK;
}
