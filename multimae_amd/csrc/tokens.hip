// Token plumbing around the encoder and the decoders (gfx950): HBM-bound gather / scatter kernels that
// replace the reference's cat / repeat / gather-with-expanded-int64-index sequences.
#include "common.h"

#define TOK_MAX_TASKS 8

// ---------------------------------------------------------------------------------------------
// pack the kept patches:  row (b*k + j) of A[B*k, Ktot] (bf16) =
//   [ zeros | patch of token ids_keep[b,j] in (c, ph, pw) order at its domain's column block | zeros | one-hot(domain) | zeros ]
// The trailing one-hot columns (at Kpatch + t) make the weight-gradient GEMM also produce the per-domain bias gradient.
// Every domain has the same NH x NW token grid; its patch edge P_t (and so its image size NH*P_t x NW*P_t) is its own.
// A class-map domain (class_emb != null) holds int64 class ids; its "pixels" are rows of the class-embedding table.
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const void* img[TOK_MAX_TASKS];
    const float* class_emb[TOK_MAX_TASKS];
    int ncls[TOK_MAX_TASKS];
    int chan[TOK_MAX_TASKS];
    int patch[TOK_MAX_TASKS];
    int koff[TOK_MAX_TASKS + 1];
    int T, B, k, Ktot, ntok, nw;
    const int* ids_keep;
    bf16* out;
};

__global__ __launch_bounds__(128) void pack_patches_kernel(PackArgs a) {
    const int row = blockIdx.x;
    const int b = row / a.k;
    const int id = a.ids_keep[row];
    const int t = id / a.ntok, pos = id - t * a.ntok;
    const int P = a.patch[t], pp = P * P;
    const int H = (a.ntok / a.nw) * P, W = a.nw * P;
    const int ph0 = (pos / a.nw) * P, pw0 = (pos % a.nw) * P;
    const int Kpatch = a.koff[a.T];
    const float* emb = a.class_emb[t];
    bf16* orow = a.out + (long)row * a.Ktot;
    for (int g = threadIdx.x; g < a.Ktot / 8; g += blockDim.x) {
        const int kk = g * 8;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
        if (kk >= a.koff[t] && kk < a.koff[t + 1]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // two runs of 4 pixels, each inside one patch row (P % 4 == 0)
                const int l = kk - a.koff[t] + 4 * h;
                const int c = l / pp, rem = l - c * pp, ph = rem / P, pw = rem - ph * P;
                f32x4 v;
                if (emb) {
                    const int64_t* cm = (const int64_t*)a.img[t] + ((long)b * H + ph0 + ph) * W + pw0 + pw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        long cls = cm[e];
                        cls = cls < 0 ? 0 : (cls >= a.ncls[t] ? a.ncls[t] - 1 : cls);  // out-of-range ids are clamped, never read out of bounds
                        v[e] = emb[cls * a.chan[t] + c];
                    }
                } else {
                    v = *(const f32x4*)((const float*)a.img[t] + (((long)b * a.chan[t] + c) * H + ph0 + ph) * W + pw0 + pw);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * h + e] = (bf16)v[e];
            }
        } else if (kk == Kpatch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)(e == t ? 1.f : 0.f);
        }
        *(bf16x8*)(orow + kk) = o;
    }
}

extern "C" int mmae_pack_patches(const void* const* images_host, const int* channels_host, const int* patch_host,
                                 const float* const* class_emb_host, const int* num_classes_host, int T, int B, int NH, int NW,
                                 const int* ids_keep_i32, int k, void* out_bf16, int Ktot, hipStream_t stream) {
    MMAE_REQUIRE(images_host && channels_host && patch_host && ids_keep_i32 && out_bf16, MMAE_EINVAL, "pack_patches: null pointer");
    MMAE_REQUIRE(T >= 1 && T <= TOK_MAX_TASKS && NH > 0 && NW > 0, MMAE_ESHAPE, "pack_patches: T=%d grid %dx%d", T, NH, NW);
    if (B * k == 0) return MMAE_OK;
    PackArgs a = {};
    a.koff[0] = 0;
    for (int t = 0; t < T; ++t) {
        MMAE_REQUIRE(images_host[t], MMAE_EINVAL, "pack_patches: image %d is null", t);
        MMAE_REQUIRE(patch_host[t] >= 4 && patch_host[t] % 4 == 0 && channels_host[t] > 0, MMAE_ESHAPE,
                     "pack_patches: patch edge %d of domain %d must be a multiple of 4", patch_host[t], t);
        a.img[t] = images_host[t]; a.chan[t] = channels_host[t]; a.patch[t] = patch_host[t];
        a.class_emb[t] = class_emb_host ? class_emb_host[t] : nullptr;
        a.ncls[t] = (a.class_emb[t] && num_classes_host) ? num_classes_host[t] : 0;
        MMAE_REQUIRE(!a.class_emb[t] || a.ncls[t] > 0, MMAE_EINVAL, "pack_patches: class-map domain %d needs its class count", t);
        a.koff[t + 1] = a.koff[t] + channels_host[t] * patch_host[t] * patch_host[t];
    }
    MMAE_REQUIRE(Ktot % 64 == 0 && Ktot >= a.koff[T] + 8, MMAE_ESHAPE, "pack_patches: Ktot=%d must be a multiple of 64 and >= %d", Ktot, a.koff[T] + 8);
    a.T = T; a.B = B; a.k = k; a.Ktot = Ktot; a.nw = NW; a.ntok = NH * NW;
    a.ids_keep = ids_keep_i32; a.out = (bf16*)out_bf16;
    hipLaunchKernelGGL(pack_patches_kernel, dim3(B * k), dim3(128), 0, stream, a);
    MMAE_CHECK_LAUNCH("pack_patches");
    return MMAE_OK;
}

// ---------------------------------------------------------------------------------------------
// class-embedding gradient of a class-map domain:
//   d_emb[cls, c] = sum over kept tokens of the domain, over their P*P pixels with class id cls, of dA[row, (c, ph, pw)]
// where dA = d_tokens * W_proj is the gradient of the packed patch rows.  No float atomics: thread (c, g) owns column c of
// private table g in LDS and walks rows g, g+G, ... of its block's row range in order; tables, blocks and the final sum are
// added in a fixed order, so the result is bitwise reproducible.
// ---------------------------------------------------------------------------------------------
struct ClsBwdArgs {
    const bf16* dA; long ldd;
    const int64_t* cmap; const int* ids_keep;
    int M, k, tok_start, ntok, nw, P, ncls, dim, G, rows_per_block;
    float* part;  // [nblocks][ncls][dim]
};
__global__ void class_emb_bwd_kernel(ClsBwdArgs a) {
    extern __shared__ float tab[];  // [G][ncls][dim]
    const int nthr = blockDim.x;
    const int tsize = a.ncls * a.dim;
    for (int i = threadIdx.x; i < a.G * tsize; i += nthr) tab[i] = 0.f;
    __syncthreads();
    const int c = threadIdx.x % a.dim, g = threadIdx.x / a.dim;
    const int pp = a.P * a.P;
    const int H = (a.ntok / a.nw) * a.P, W = a.nw * a.P;
    const int r0 = blockIdx.x * a.rows_per_block;
    const int r1 = min(a.M, r0 + a.rows_per_block);
    float* mine = tab + (long)g * tsize;
    for (int row = r0 + g; row < r1; row += a.G) {
        const int id = a.ids_keep[row] - a.tok_start;
        if (id < 0 || id >= a.ntok) continue;
        const int b = row / a.k;
        const int ph0 = (id / a.nw) * a.P, pw0 = (id % a.nw) * a.P;
        const bf16* src = a.dA + (long)row * a.ldd + (long)c * pp;
        for (int q = 0; q < pp; q += 4) {
            const int ph = q / a.P, pw = q - ph * a.P;
            const bf16x4 v = *(const bf16x4*)(src + q);
            const int64_t* cm = a.cmap + ((long)b * H + ph0 + ph) * W + pw0 + pw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                long cls = cm[e];
                cls = cls < 0 ? 0 : (cls >= a.ncls ? a.ncls - 1 : cls);
                mine[cls * a.dim + c] += (float)v[e];
            }
        }
    }
    __syncthreads();
    float* out = a.part + (long)blockIdx.x * tsize;
    for (int i = threadIdx.x; i < tsize; i += nthr) {
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < a.G; ++q) s += tab[(long)q * tsize + i];
        out[i] = s;
    }
}
__global__ void class_emb_reduce_kernel(const float* __restrict__ part, int nblocks, int n, float* __restrict__ d_emb, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < nblocks; ++b) s += part[(long)b * n + i];
    d_emb[i] = accumulate ? d_emb[i] + s : s;
}

#define CLS_BWD_BLOCKS 512
#define CLS_BWD_LDS (152 * 1024)
extern "C" size_t mmae_class_emb_bwd_workspace_bytes(int num_classes, int dim_emb) {
    return (size_t)CLS_BWD_BLOCKS * num_classes * dim_emb * sizeof(float);
}
extern "C" int mmae_class_emb_bwd(const void* dA_bf16, long ldd, const int64_t* class_map, const int* ids_keep_i32, int B, int k,
                                  int tok_start, int NH, int NW, int P, int num_classes, int dim_emb, float* d_emb, int accumulate,
                                  void* workspace, size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(dA_bf16 && class_map && ids_keep_i32 && d_emb && workspace, MMAE_EINVAL, "class_emb_bwd: null pointer");
    MMAE_REQUIRE(P >= 4 && P % 4 == 0 && dim_emb >= 1 && dim_emb <= 1024 && num_classes >= 1 && ldd >= (long)dim_emb * P * P && ldd % 4 == 0 &&
                 (uintptr_t)dA_bf16 % 8 == 0, MMAE_ESHAPE, "class_emb_bwd: bad shape (P=%d dim=%d classes=%d ldd=%ld)", P, dim_emb, num_classes, ldd);
    const size_t tbytes = (size_t)num_classes * dim_emb * sizeof(float);
    MMAE_REQUIRE(tbytes <= CLS_BWD_LDS, MMAE_ESHAPE, "class_emb_bwd: table %d x %d does not fit LDS", num_classes, dim_emb);
    MMAE_REQUIRE(workspace_bytes >= mmae_class_emb_bwd_workspace_bytes(num_classes, dim_emb), MMAE_EWORKSPACE, "class_emb_bwd: workspace too small");
    const int M = B * k;
    const int n = num_classes * dim_emb;
    if (M == 0) {
        if (!accumulate) { hipError_t e = hipMemsetAsync(d_emb, 0, tbytes, stream); MMAE_REQUIRE(e == hipSuccess, MMAE_ELAUNCH, "class_emb_bwd: memset failed"); }
        return MMAE_OK;
    }
    int G = (int)(CLS_BWD_LDS / tbytes);
    if (G > 4) G = 4;
    if (G * dim_emb > 1024) G = 1024 / dim_emb;
    if (G < 1) G = 1;
    ClsBwdArgs a = {};
    a.dA = (const bf16*)dA_bf16; a.ldd = ldd; a.cmap = class_map; a.ids_keep = ids_keep_i32; a.M = M; a.k = k; a.tok_start = tok_start;
    a.ntok = NH * NW; a.nw = NW; a.P = P; a.ncls = num_classes; a.dim = dim_emb; a.G = G;
    a.rows_per_block = (M + CLS_BWD_BLOCKS - 1) / CLS_BWD_BLOCKS;
    if (a.rows_per_block < 4 * G) a.rows_per_block = 4 * G;
    const int nblocks = (M + a.rows_per_block - 1) / a.rows_per_block;
    a.part = (float*)workspace;
    const size_t lds = (size_t)G * tbytes;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)class_emb_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MMAE_REQUIRE(e == hipSuccess, MMAE_ELAUNCH, "class_emb_bwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(class_emb_bwd_kernel, dim3(nblocks), dim3(G * dim_emb), lds, stream, a);
    MMAE_CHECK_LAUNCH("class_emb_bwd");
    hipLaunchKernelGGL(class_emb_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)workspace, nblocks, n, d_emb, accumulate);
    MMAE_CHECK_LAUNCH("class_emb_reduce");
    return MMAE_OK;
}

// add_tab[t*ntok + pos][n] = pos_emb_t[n*ntok + pos] + bias_t[n]
struct TabArgs { const float* pos[TOK_MAX_TASKS]; const float* bias[TOK_MAX_TASKS]; int T, ntok, dim; float* tab; };
__global__ void build_token_table_kernel(TabArgs a) {
    const long n_all = (long)a.T * a.ntok * a.dim;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % a.dim);
        const int tp = (int)(i / a.dim);
        const int t = tp / a.ntok, pos = tp - t * a.ntok;
        a.tab[i] = a.pos[t][(long)n * a.ntok + pos] + (a.bias[t] ? a.bias[t][n] : 0.f);
    }
}
extern "C" int mmae_build_token_table(const float* const* pos_emb_host, const float* const* bias_host, int T, int ntok, int dim,
                                      float* add_tab, hipStream_t stream) {
    MMAE_REQUIRE(pos_emb_host && bias_host && add_tab && T >= 1 && T <= TOK_MAX_TASKS, MMAE_EINVAL, "build_token_table: bad argument");
    TabArgs a = {};
    for (int t = 0; t < T; ++t) { a.pos[t] = pos_emb_host[t]; a.bias[t] = bias_host[t]; MMAE_REQUIRE(a.pos[t], MMAE_EINVAL, "build_token_table: null table"); }
    a.T = T; a.ntok = ntok; a.dim = dim; a.tab = add_tab;
    hipLaunchKernelGGL(build_token_table_kernel, dim3(512), dim3(256), 0, stream, a);
    MMAE_CHECK_LAUNCH("build_token_table");
    return MMAE_OK;
}

__global__ void append_global_kernel(const float* __restrict__ g, float* __restrict__ x, int B, int k, int G, int dim) {
    const long n_all = (long)B * G * dim;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % dim);
        const int bg = (int)(i / dim);
        const int b = bg / G, gi = bg - b * G;
        x[((long)b * (k + G) + k + gi) * dim + d] = g[(long)gi * dim + d];
    }
}
extern "C" int mmae_append_global(const float* global_tokens, float* x, int B, int k, int G, int dim, hipStream_t stream) {
    MMAE_REQUIRE(global_tokens && x, MMAE_EINVAL, "append_global: null pointer");
    if (B * G * dim == 0) return MMAE_OK;
    hipLaunchKernelGGL(append_global_kernel, dim3(256), dim3(256), 0, stream, global_tokens, x, B, k, G, dim);
    MMAE_CHECK_LAUNCH("append_global");
    return MMAE_OK;
}

// d_global[g,:] = sum_b dx[b,k+g,:]  and  rows_bf16[b*k + j,:] = bf16(dx[b,j,:])
__global__ void embed_bwd_kernel(const float* __restrict__ dx, int B, int k, int G, int dim, float* __restrict__ d_global,
                                 bf16* __restrict__ rows) {
    const int seq = k + G;
    const long n_rows4 = (long)B * k * (dim / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (dim / 4));
        const long bj = i / (dim / 4);
        const int b = (int)(bj / k), j = (int)(bj - (long)b * k);
        const f32x4 v = *(const f32x4*)(dx + ((long)b * seq + j) * dim + c * 4);
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *(bf16x4*)(rows + bj * dim + c * 4) = o;
    }
    // global-token gradient: one thread per (g, d), serial over the batch (B*G*dim is tiny)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)G * dim; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % dim), gi = (int)(i / dim);
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s += dx[((long)b * seq + k + gi) * dim + d];
        d_global[i] = s;
    }
}
extern "C" int mmae_embed_bwd_reduce(const float* dx, const int* ids_keep_i32, int B, int k, int G, int dim, int T, int ntok,
                                     float* d_global, float* const* d_bias_host, void* dx_bf16_rows, hipStream_t stream) {
    (void)ids_keep_i32; (void)T; (void)ntok; (void)d_bias_host;  // bias gradients come out of the weight-gradient GEMM
    MMAE_REQUIRE(dx && d_global && dx_bf16_rows && dim % 4 == 0, MMAE_EINVAL, "embed_bwd_reduce: bad argument");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(1024), dim3(256), 0, stream, dx, B, k, G, dim, d_global, (bf16*)dx_bf16_rows);
    MMAE_CHECK_LAUNCH("embed_bwd");
    return MMAE_OK;
}

// ---------------------------------------------------------------------------------------------
// decoder assembly
// ---------------------------------------------------------------------------------------------
struct AsmArgs {
    const void* ctx; int ctx_f32; const float* mask_token; const float* emb; const int* ids_keep; const int* ids_restore;
    int B, k, G, total, q_start, q_len, Dd;
    float* queries; float* context;
};
// four consecutive values of row `row` of the projected context (bf16, or f32 for decoders that run in fp32)
__device__ __forceinline__ f32x4 asm_ctx4(const AsmArgs& a, long row, int c) {
    if (a.ctx_f32) return *(const f32x4*)((const float*)a.ctx + row * a.Dd + c * 4);
    const bf16x4 s = *(const bf16x4*)((const bf16*)a.ctx + row * a.Dd + c * 4);
    return (f32x4){(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
}
// one block per output row (queries first, then context rows); Dd/4 <= blockDim
__global__ __launch_bounds__(64) void decoder_assemble_fwd_kernel(AsmArgs a) {
    const int rows_per_b = a.q_len + a.k + a.G;
    const int b = blockIdx.x / rows_per_b, rr = blockIdx.x - b * rows_per_b;
    const int seq = a.k + a.G;
    for (int c = threadIdx.x; c < a.Dd / 4; c += blockDim.x) {
        f32x4 v;
        if (rr < a.q_len) {
            const int pos = a.q_start + rr;
            const int r = a.ids_restore[(long)b * a.total + pos];
            if (r < a.k) {
                v = asm_ctx4(a, (long)b * seq + r, c);
            } else {
                v = ((const f32x4*)a.mask_token)[c];
            }
            v += ((const f32x4*)(a.emb + (long)pos * a.Dd))[c];
            ((f32x4*)(a.queries + ((long)b * a.q_len + rr) * a.Dd))[c] = v;
        } else {
            const int j = rr - a.q_len;
            v = asm_ctx4(a, (long)b * seq + j, c);
            if (j < a.k) {
                // gather(full, ids_keep): full[pos] = (ids_restore[pos] < k ? ctx[ids_restore[pos]] : mask_token) + emb[pos]
                const int pos = a.ids_keep[(long)b * a.k + j];
                const int r = a.ids_restore[(long)b * a.total + pos];
                if (r != j) {
                    if (r < a.k) {
                        v = asm_ctx4(a, (long)b * seq + r, c);
                    } else {
                        v = ((const f32x4*)a.mask_token)[c];
                    }
                }
                v += ((const f32x4*)(a.emb + (long)pos * a.Dd))[c];
            }
            ((f32x4*)(a.context + ((long)b * seq + j) * a.Dd))[c] = v;
        }
    }
}

extern "C" int mmae_decoder_assemble_fwd(const void* ctx_bf16, int ctx_is_f32, const float* mask_token, const float* emb,
                                         const int* ids_keep_i32, const int* ids_restore_i32, int B, int k, int G, int total,
                                         int q_start, int q_len, int Dd, float* queries, float* context, hipStream_t stream) {
    MMAE_REQUIRE(ctx_bf16 && mask_token && emb && ids_keep_i32 && ids_restore_i32 && queries && context, MMAE_EINVAL,
                 "decoder_assemble_fwd: null pointer");
    MMAE_REQUIRE(Dd % 4 == 0 && q_start >= 0 && q_start + q_len <= total && k <= total, MMAE_ESHAPE, "decoder_assemble_fwd: bad shape");
    if (B == 0) return MMAE_OK;
    AsmArgs a = {ctx_bf16, ctx_is_f32, mask_token, emb, ids_keep_i32, ids_restore_i32, B, k, G, total, q_start, q_len, Dd, queries, context};
    hipLaunchKernelGGL(decoder_assemble_fwd_kernel, dim3(B * (q_len + k + G)), dim3(64), 0, stream, a);
    MMAE_CHECK_LAUNCH("decoder_assemble_fwd");
    return MMAE_OK;
}

// backward: d_ctx (gather form, no atomics) and per-sample partial sums for mask_token / task embeddings
struct AsmBwdArgs {
    const float* dq; const float* dc; const int* ids_keep; const int* ids_restore;
    int B, k, G, total, q_start, q_len, Dd, T, ntok;
    void* d_ctx; int d_ctx_f32; long ldc; float* part;  // d_ctx (bf16 or f32) rows are ldc elements apart; part[b][T+1][Dd]
};
// grid (B, column splits): a workgroup owns nv = Dd / 4 / gridDim.y column quads of its sample, so that a sample's work (a
// few hundred KB read more than once) is spread over several CUs instead of one
__global__ __launch_bounds__(1024) void decoder_assemble_bwd_kernel(AsmBwdArgs a) {
    const int b = blockIdx.x;
    const int seq = a.k + a.G;
    const int nv = a.Dd / 4 / gridDim.y, c0 = blockIdx.y * nv;
    // d_ctx rows
    for (int i = threadIdx.x; i < seq * nv; i += blockDim.x) {
        const int j = i / nv, c = c0 + i - j * nv;
        f32x4 v = ((const f32x4*)(a.dc + ((long)b * seq + j) * a.Dd))[c];
        if (j < a.k) {
            const int pos = a.ids_keep[(long)b * a.k + j];
            if (pos >= a.q_start && pos < a.q_start + a.q_len)
                v += ((const f32x4*)(a.dq + ((long)b * a.q_len + pos - a.q_start) * a.Dd))[c];
        }
        if (a.d_ctx_f32) {
            *(f32x4*)((float*)a.d_ctx + ((long)b * seq + j) * a.ldc + c * 4) = v;
        } else {
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4*)((bf16*)a.d_ctx + ((long)b * seq + j) * a.ldc + c * 4) = o;
        }
    }
    // partial sums over positions: slot t < T collects d_emb of task t, slot T the mask token.
    // threads are laid out as (column group c, position lane pl); each walks positions pl, pl+PL, ...
    __shared__ float red[1024 * 4];
    const int PL = blockDim.x / nv;  // position lanes (Dd = 256 in four column splits: nv = 16 -> 64 lanes)
    const int cl = threadIdx.x % nv, c = c0 + cl, pl = threadIdx.x / nv;
    for (int slot = 0; slot <= a.T; ++slot) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (pl < PL) {
            const int p0 = slot < a.T ? slot * a.ntok : 0;
            const int p1 = slot < a.T ? p0 + a.ntok : a.total;
            // four positions per trip: their indices first, then all their rows, then the additions in position order
            // (one position per trip is a chain of two dependent loads per addition)
            for (int pos0 = p0 + pl; pos0 < p1; pos0 += 4 * PL) {
                int r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = pos0 + j * PL;
                    r[j] = pos < p1 ? a.ids_restore[(long)b * a.total + pos] : -1;
                }
                f32x4 vq[4], vc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = pos0 + j * PL;
                    const bool live = pos < p1;
                    const bool in_q = live && pos >= a.q_start && pos < a.q_start + a.q_len;
                    const bool take_q = in_q && (slot < a.T || r[j] >= a.k);
                    const bool take_c = live && slot < a.T && r[j] < a.k;
                    vq[j] = take_q ? ((const f32x4*)(a.dq + ((long)b * a.q_len + pos - a.q_start) * a.Dd))[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    vc[j] = take_c ? ((const f32x4*)(a.dc + ((long)b * seq + r[j]) * a.Dd))[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { s += vq[j]; s += vc[j]; }
            }
        }
        __syncthreads();
        *(f32x4*)&red[threadIdx.x * 4] = s;
        __syncthreads();
        if (pl == 0) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < PL; ++q) t += *(f32x4*)&red[(q * nv + cl) * 4];
            *(f32x4*)(a.part + ((long)b * (a.T + 1) + slot) * a.Dd + c * 4) = t;
        }
    }
}
struct AsmRedArgs { const float* part; int B, T, Dd; float* d_mask_token; float* d_task[TOK_MAX_TASKS]; int accumulate; };
__global__ void decoder_assemble_reduce_kernel(AsmRedArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (a.T + 1) * a.Dd) return;
    const int slot = i / a.Dd, d = i - slot * a.Dd;
    float s = 0.f;
    const float* __restrict__ part = a.part + (long)slot * a.Dd + d;
    const long stride = (long)(a.T + 1) * a.Dd;
    int b = 0;
    for (; b + 16 <= a.B; b += 16) {  // sixteen loads in flight, additions in order
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = part[(b + j) * stride];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; b < a.B; ++b) s += part[b * stride];
    float* dst = slot < a.T ? a.d_task[slot] : a.d_mask_token;
    if (dst) dst[d] = a.accumulate ? dst[d] + s : s;
}

extern "C" size_t mmae_decoder_assemble_bwd_workspace_bytes(int B, int T, int Dd) { return (size_t)B * (T + 1) * Dd * sizeof(float); }

extern "C" int mmae_decoder_assemble_bwd(const float* d_queries, const float* d_context, const int* ids_keep_i32,
                                         const int* ids_restore_i32, int B, int k, int G, int T, int ntok, int q_start, int q_len,
                                         int Dd, void* d_ctx_bf16, int d_ctx_is_f32, long ld_d_ctx, float* d_mask_token, float* const* d_task_emb_host,
                                         void* workspace, size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(d_queries && d_context && ids_keep_i32 && ids_restore_i32 && d_ctx_bf16 && d_mask_token && d_task_emb_host && workspace,
                 MMAE_EINVAL, "decoder_assemble_bwd: null pointer");
    MMAE_REQUIRE(T >= 1 && T <= TOK_MAX_TASKS && Dd % 4 == 0 && Dd / 4 <= 256, MMAE_ESHAPE, "decoder_assemble_bwd: bad shape (T=%d Dd=%d)", T, Dd);
    MMAE_REQUIRE(workspace_bytes >= mmae_decoder_assemble_bwd_workspace_bytes(B, T, Dd), MMAE_EWORKSPACE, "decoder_assemble_bwd: workspace too small");
    if (ld_d_ctx == 0) ld_d_ctx = Dd;
    MMAE_REQUIRE(ld_d_ctx >= Dd && ld_d_ctx % 4 == 0 && (uintptr_t)d_ctx_bf16 % (d_ctx_is_f32 ? 16 : 8) == 0, MMAE_ESHAPE,
                 "decoder_assemble_bwd: d_ctx row stride %ld must be >= Dd = %d and a multiple of 4 (rows 8-byte aligned)", ld_d_ctx, Dd);
    if (B == 0) return MMAE_OK;
    AsmBwdArgs a = {d_queries, d_context, ids_keep_i32, ids_restore_i32, B, k, G, T * ntok, q_start, q_len, Dd, T, ntok,
                    d_ctx_bf16, d_ctx_is_f32, ld_d_ctx, (float*)workspace};
    const int splits = (Dd / 4) % 4 == 0 && Dd >= 128 ? 4 : 1;  // column groups of a sample, one workgroup each
    hipLaunchKernelGGL(decoder_assemble_bwd_kernel, dim3(B, splits), dim3(1024), 0, stream, a);
    MMAE_CHECK_LAUNCH("decoder_assemble_bwd");
    AsmRedArgs r = {};
    r.part = (const float*)workspace; r.B = B; r.T = T; r.Dd = Dd; r.d_mask_token = d_mask_token; r.accumulate = 0;
    for (int t = 0; t < T; ++t) r.d_task[t] = d_task_emb_host[t];
    hipLaunchKernelGGL(decoder_assemble_reduce_kernel, dim3(((T + 1) * Dd + 255) / 256), dim3(256), 0, stream, r);
    MMAE_CHECK_LAUNCH("decoder_assemble_reduce");
    return MMAE_OK;
}

// emb[t*ntok + pos][d] = task_emb_t[d] + pos_emb[d*ntok + pos]
struct EmbArgs { const float* task[TOK_MAX_TASKS]; const float* pos; int T, ntok, Dd; float* emb; };
__global__ void build_decoder_emb_kernel(EmbArgs a) {
    const long n_all = (long)a.T * a.ntok * a.Dd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % a.Dd);
        const int tp = (int)(i / a.Dd);
        const int t = tp / a.ntok, pos = tp - t * a.ntok;
        a.emb[i] = (a.task[t] ? a.task[t][d] : 0.f) + a.pos[(long)d * a.ntok + pos];
    }
}
extern "C" int mmae_build_decoder_emb(const float* const* task_emb_host, const float* pos_emb, int T, int ntok, int Dd, float* emb,
                                      hipStream_t stream) {
    MMAE_REQUIRE(task_emb_host && pos_emb && emb && T >= 1 && T <= TOK_MAX_TASKS, MMAE_EINVAL, "build_decoder_emb: bad argument");
    EmbArgs a = {};
    for (int t = 0; t < T; ++t) a.task[t] = task_emb_host[t];
    a.pos = pos_emb; a.T = T; a.ntok = ntok; a.Dd = Dd; a.emb = emb;
    hipLaunchKernelGGL(build_decoder_emb_kernel, dim3(256), dim3(256), 0, stream, a);
    MMAE_CHECK_LAUNCH("build_decoder_emb");
    return MMAE_OK;
}

// image (bf16 or f32) [B,C,NH*P,NW*P] -> rows bf16 [B*NH*NW, C*P*P] ((c, ph, pw) inner order)
// One workgroup per (sample, row of patches): its C * P image rows are contiguous runs of NW * P pixels, its NW output rows
// contiguous runs of C * P * P.  A thread moves four pixels at a time, consecutive threads along the output row (so the
// stores of a wave are contiguous and its loads are P-pixel segments of P consecutive image rows); all index arithmetic is
// 32-bit and per workgroup (the grid-stride form spent its time on 64-bit divisions: 3.1 TB/s).
template <bool F32IN>
__global__ __launch_bounds__(256) void patchify_bf16_kernel(const void* __restrict__ img, int C, int NH, int NW, int P, bf16* __restrict__ rows) {
    const int nh = blockIdx.x, b = blockIdx.y;
    const int W = NW * P, H = NH * P, PP = P * P, g_per_tok = C * PP / 4, qp = P / 4;
    const long img_base = ((long)b * C * H + (long)nh * P) * W;     // (b, c = 0, row nh * P, col 0)
    bf16* out = rows + ((long)b * NH + nh) * NW * (long)(C * PP);
    for (int i = threadIdx.x; i < NW * g_per_tok; i += 256) {
        const int nw = i / g_per_tok, g = i - nw * g_per_tok;       // token of the strip, quad inside its output row
        const int c = g / (PP / 4), rem = g - c * (PP / 4), ph = rem / qp, q = rem - ph * qp;
        const long src = img_base + ((long)c * H + ph) * W + nw * P + q * 4;
        bf16x4 o;
        if (F32IN) {
            const f32x4 v = *(const f32x4*)((const float*)img + src);
            o = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        } else {
            o = *(const bf16x4*)((const bf16*)img + src);
        }
        *(bf16x4*)(out + (long)nw * (C * PP) + g * 4) = o;
    }
}
// The same through LDS, for strips whose NW output rows fit there (C * P * P * NW bf16 <= 64 KiB): image rows are read as the
// contiguous runs they are (a thread takes four pixels, consecutive threads along the image row), converted and placed in an
// LDS image of the NW output rows, which is then written out as the one contiguous block it is.  With small patches (a class
// map at stride 4: P = 4, 133 channels) the direct form reads 16-byte pieces of 2,128 different image rows per output row and
// ran at 1.1 TB/s.
template <bool F32IN>
__global__ __launch_bounds__(256) void patchify_lds_kernel(const void* __restrict__ img, int C, int NH, int NW, int P, bf16* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* strip = (bf16*)smem;  // [NW][C * P * P]
    const int nh = blockIdx.x, b = blockIdx.y;
    const int W = NW * P, H = NH * P, PP = P * P, row_len = C * PP, qw = W / 4;
    const long img_base = ((long)b * C * H + (long)nh * P) * W;
    // four quads per thread and trip, all requested before the first is placed (one trip to memory for the four)
    constexpr int U = 4;
    const int n_items = C * P * qw;
    for (int i0 = threadIdx.x; i0 < n_items; i0 += U * 256) {
        f32x4 vf[U];
        bf16x4 vb[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256 < n_items ? i0 + u * 256 : n_items - 1;
            const int cr = i / qw, qx = i - cr * qw;    // image row (channel c, row ph of the strip), quad along it
            const int c = cr / P, ph = cr - c * P;
            const int px = qx * 4, nw = px / P, pw = px - nw * P;
            const long src = img_base + ((long)c * H + ph) * W + px;
            dst[u] = nw * row_len + c * PP + ph * P + pw;
            if (F32IN) vf[u] = *(const f32x4*)((const float*)img + src);
            else vb[u] = *(const bf16x4*)((const bf16*)img + src);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * 256 < n_items) {
                if (F32IN) vb[u] = (bf16x4){(bf16)vf[u][0], (bf16)vf[u][1], (bf16)vf[u][2], (bf16)vf[u][3]};
                *(bf16x4*)(strip + dst[u]) = vb[u];
            }
        }
    }
    __syncthreads();
    bf16* out = rows + ((long)b * NH + nh) * NW * (long)row_len;
    for (int i = threadIdx.x; i < NW * row_len / 8; i += 256) *(bf16x8*)(out + i * 8) = *(const bf16x8*)(strip + i * 8);
}

extern "C" int mmae_patchify_bf16(const void* image, int image_is_f32, int B, int C, int NH, int NW, int P, void* rows,
                                  hipStream_t stream) {
    MMAE_REQUIRE(image && rows && P % 4 == 0, MMAE_EINVAL, "patchify: bad argument");
    if (B == 0) return MMAE_OK;
    const size_t strip_bytes = sizeof(bf16) * (size_t)NW * C * P * P;
    if (strip_bytes <= 65536 && (C * P * P) % 8 == 0) {
        if (image_is_f32) hipLaunchKernelGGL(patchify_lds_kernel<true>, dim3(NH, B), dim3(256), strip_bytes, stream, image, C, NH, NW, P, (bf16*)rows);
        else hipLaunchKernelGGL(patchify_lds_kernel<false>, dim3(NH, B), dim3(256), strip_bytes, stream, image, C, NH, NW, P, (bf16*)rows);
    } else if (image_is_f32) {
        hipLaunchKernelGGL(patchify_bf16_kernel<true>, dim3(NH, B), dim3(256), 0, stream, image, C, NH, NW, P, (bf16*)rows);
    } else {
        hipLaunchKernelGGL(patchify_bf16_kernel<false>, dim3(NH, B), dim3(256), 0, stream, image, C, NH, NW, P, (bf16*)rows);
    }
    MMAE_CHECK_LAUNCH("patchify");
    return MMAE_OK;
}
