/* libmmae_hip.so -- C ABI of the MI355X (gfx950) kernels behind the MultiMAE pretraining step.
 *
 * The reference (thinh-re/multimae) is pure PyTorch: its hot path is sequences of ATen ops inside
 * nn.Modules.  Each entry point below replaces one such sequence; the comment on each cites the
 * reference lines it stands for (paths relative to the reference tree).  The Python package
 * multimae_amd binds these with ctypes and wraps them in autograd Functions behind the
 * reference's module API.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - calls only enqueue work on `stream`; they never allocate, free or synchronise;
 *   - return 0 (MMAE_OK) or a negative MMAE_E* code; mmae_last_error() gives the message of the
 *     calling thread's last failure;
 *   - bf16 tensors are row-major with explicit row strides counted in elements; 16-byte aligned.
 */
#ifndef MMAE_H
#define MMAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mmae_stream_t; /* == hipStream_t */

enum {
    MMAE_OK = 0,
    MMAE_EINVAL = -1,     /* null pointer / bad argument */
    MMAE_ESHAPE = -2,     /* shape or stride the kernels do not support */
    MMAE_EWORKSPACE = -3, /* workspace too small */
    MMAE_ELAUNCH = -4,    /* HIP reported a launch error */
};

const char* mmae_last_error(void);
int mmae_version(void);

/* ---- GEMM family --------------------------------------------------------------------------
 * mmae_gemm_nt: C[M,N] = A[M,K] x W[N,K]^T with a fused epilogue.  Replaces nn.Linear forward
 * (multimae/multimae_utils.py:163,167,194,205,229,234,245; output_adapters.py:341,361) and, fed with a
 * transposed weight copy, the dA = dC x W half of its backward.
 *   epi 0  out(bf16) = acc + bias                               (qkv, q, kv, proj_context, dA)
 *   epi 1  out(bf16) = h = acc + bias, out2(bf16) = gelu_erf(h) (Mlp.fc1 + nn.GELU, :163-164)
 *   epi 2  out(f32)  = aux(f32) + acc + bias                    (attn.proj / Mlp.fc2 + residual, :283-284)
 *   epi 3  out(bf16) = acc * gelu_erf'(aux(bf16))               (backward through nn.GELU)
 *   epi 4  out(f32)  = acc + bias
 * K and N must be multiples of 8; bias may be NULL for epi 0/3/4. */
int mmae_gemm_nt(int epi, const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                 void* out, long ldo, void* out2, const void* aux, long ldaux, mmae_stream_t stream);

/* The residual product of a transformer block fused with the LayerNorm that follows it (multimae/multimae_utils.py:283-284:
 * x = x + proj(attn) ... norm2(x); x = x + fc2(...) ... the next block's norm1): out(f32) [M,N] = aux + A W^T + bias (rows of
 * out are N apart), then ln(bf16) [M,N] = LayerNorm(out) * gamma + beta, mean / rstd [M].  The same kernels as mmae_gemm_nt(epi
 * 2) followed by mmae_layernorm_fwd and bitwise the same results; where the product ends in a tail launch that leaves CUs
 * idle, those CUs normalise the already complete rows while the tail runs.  N <= 1024. */
int mmae_gemm_nt_resid_ln(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias, float* out,
                          const float* aux, long ldaux, const float* gamma, const float* beta, float eps, void* ln_bf16,
                          float* mean, float* rstd, mmae_stream_t stream);

/* Kernel choice for mmae_gemm_nt and its two siblings below: 0 = by problem size (default), 1 = always the 128x128
 * two-workgroups-per-CU kernel, 4 = persistent 128x128 (2-stage ring), 5 = 256x256 eight-wave ping-pong kernel (K a
 * multiple of 128, else as 0), 7 = 128x256 four-wave kernel, two workgroups per CU (K a multiple of 64, else as 0).  Like every *_set_* switch of this library it applies to the calling thread's later
 * calls only; for tests and tuning.  NOTE: PyTorch runs the backward nodes of a CUDA graph on its autograd engine's device
 * thread, not on the thread that called backward(): a switch set from the Python main thread reaches the kernels of the
 * forward pass and of direct calls, but NOT those launched inside loss.backward(), which keep the defaults.  To measure a
 * variant in backward, set it from inside a backward hook (that runs on the engine's thread) or call the op directly. */
void mmae_gemm_nt_set_variant(int variant);
/* The 256x256 kernel may cut the last row panels into 128-row half tiles so that a thin last round of tiles is shared by
 * twice as many CUs (default on); 0 switches that off.  Per calling thread; for tests and tuning. */
void mmae_gemm_nt_set_half_tiles(int on);  /* 0: none; 1: 128- or 64-row tail tiles by cost (default); 2: 128-row tail tiles only */
/* For timing the two launches of the 256x256 path separately: the calling thread's next mmae_gemm_nt calls run
 * 1 = the full-tile launch only, 2 = the half-tile launch only, 0 = both (default); other kernels ignore it.
 * mmae_gemm_nt_last_full_rows: rows the full tiles covered in the calling thread's last call (M when there was no
 * half-tile launch). */
void mmae_gemm_nt_set_phase(int phase);
int mmae_gemm_nt_last_full_rows(void);
/* rows per tile of the second launch of that call (128 or 64), 0 when there was none */
int mmae_gemm_nt_last_tail_rows(void);
/* Kernel used by the calling thread's last mmae_gemm_nt / patch-embed / un-patchify call (codes as above, never 0). */
int mmae_gemm_nt_last_kernel(void);

/* Patch projection of the kept tokens only: out[(m/seg_in)*seg_out + m%seg_in] = A[m] x W^T + add_tab[add_idx[m]]
 * where A holds the packed patches (mmae_pack_patches), W the per-domain Conv2d weights concatenated along K
 * and add_tab = pos_emb + conv bias per (domain, position).  Replaces Conv2d(k=s=P) + rearrange +
 * pos-emb add + gather(ids_keep) (multimae/input_adapters.py:126-135, multimae/multimae.py:393-402). */
int mmae_patch_embed_gemm(const void* A, const void* W, int M, int N, int K, const int* add_idx, const float* add_tab,
                          int seg_in, int seg_out, float* out, long ldo, mmae_stream_t stream);

/* out_proj + "b (nh nw) (c ph pw) -> b c (nh ph) (nw pw)" store (multimae/output_adapters.py:361-372);
 * out_image is bf16 or f32 (out_is_f32) [M/(NH*NW), C, NH*P, NW*P]; P = patch_size_full / stride_level, a multiple of 4. */
int mmae_unpatchify_gemm(const void* A, long lda, const void* W, int M, int N, int K, const float* bias,
                         void* out_image, int out_is_f32, int C, int P, int NH, int NW, mmae_stream_t stream);

/* mmae_gemm_tn: out[N,K] (f32) (+)= X[M,N]^T x Y[M,K]: the weight-gradient half of nn.Linear backward
 * (dW = dC^T x A).  Split over M into fp32 slabs held in `workspace`, then reduced in a fixed order
 * (bitwise reproducible).  If colsum_out != NULL it also receives (+)= the column sums of X (the bias
 * gradient), computed by the same kernel. */
/* Kernel choice for mmae_gemm_tn: 0 = by problem size (default), 1 = 128x128 kernel, 2 = 256x256 eight-wave kernel.
 * Per calling thread; for tests and tuning.  Set it before asking for the workspace size (the split count depends on it). */
void mmae_gemm_tn_set_variant(int variant);
int mmae_gemm_tn_last_kernel(void);
/* For timing the two launches of mmae_gemm_tn separately: the calling thread's next calls run 1 = the split GEMM only
 * (slabs stay in the workspace), 2 = the slab reduction only (same arguments), 0 = both (default). */
void mmae_gemm_tn_set_phase(int phase);
size_t mmae_gemm_tn_workspace_bytes(int M, int N, int K);
int mmae_gemm_tn(const void* X, long ldx, const void* Y, long ldy, int M, int N, int K, float* out, int accumulate,
                 float* colsum_out, int colsum_accumulate, void* workspace, size_t workspace_bytes, mmae_stream_t stream);

/* Column sums of a bf16 matrix: bias gradient of nn.Linear. */
size_t mmae_colsum_workspace_bytes(int M, int N);
int mmae_colsum_bf16(const void* X, long ldx, int M, int N, float* out, int accumulate, void* workspace,
                     size_t workspace_bytes, mmae_stream_t stream);

/* ---- LayerNorm (eps 1e-6; multimae/multimae_utils.py:283-284, output_adapters.py:350-353) ---------
 * fwd: y(bf16) = LN(x(f32)) * gamma + beta; saves mean and rstd per row.
 * bwd: dx(f32) = (dres ? dres : 0) + LN'(dy(bf16)); optional bf16 copy of dx; dgamma/dbeta are
 *      written as per-block partial rows into `workspace` and summed by a second kernel.  With dgamma = dbeta = NULL
 *      the partial rows stay in `workspace` for a later mmae_layernorm_param_reduce (e.g. on another stream). */
int mmae_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                       int rows, int dim, float eps, mmae_stream_t stream);
size_t mmae_layernorm_bwd_workspace_bytes(int rows, int dim);
int mmae_layernorm_bwd(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                       const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, int accumulate,
                       int rows, int dim, void* workspace, size_t workspace_bytes, mmae_stream_t stream);

/* Kernel choice for mmae_layernorm_bwd: 0 = several rows per wave and pass where dim is 256, 768 or 1024 (default),
 * 1 = the generic one-row kernel everywhere.  Per calling thread; for tests and tuning. */
int mmae_layernorm_bwd_set_variant(int variant);
int mmae_layernorm_param_reduce(const void* workspace, size_t workspace_bytes, int rows, int dim, float* dgamma, float* dbeta,
                                int accumulate, mmae_stream_t stream);

/* ---- attention (multimae/multimae_utils.py:200-204 and :240-244) ----------------------------
 * softmax(q k^T * scale) v per (batch, head) without materialising the scores.  q/k/v/o rows are
 * addressed as base + (b*len + i)*ld + h*head_dim, so the packed [B,N,3,H,d] qkv buffer and the
 * [B,M,2,H,d] kv buffer are read in place.  head_dim in {32, 64}.  fwd takes any kv_len: the K / V rows of a head stay
 * in LDS where they fit 160 KiB (padded to a multiple of 32 rows of head_dim + 8 elements, or unpadded and swizzled where
 * only that leaves room for a second workgroup on the CU) and stream through it in 64-key tiles beyond.  bwd keeps K,V
 * (dQ kernel) and Q,dO plus two floats per query (dK/dV kernel) of a head in LDS: q_len and kv_len <= 544 at head_dim 64
 * and <= 960 at head_dim 32 per call (mmae_attn_max_len), longer sequences return MMAE_ESHAPE (callers chunk them,
 * mmae_attn_bwd_chunk).
 * fwd also writes lse[B,H,q_len] (f32, natural log) for the backward.  bwd = one kernel per (batch, head) with Q, dO, K, V
 * in LDS when two such blocks fit a CU (the pretraining lengths), else two kernels (dQ per query block; dK,dV per key
 * tile); accumulation in registers, no atomics; workspace holds rowsum(dO*O) [B,H,q_len] f32 for the two-kernel form. */
int mmae_attn_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                  float* lse, int B, int H, int q_len, int kv_len, int head_dim, float scale, mmae_stream_t stream);
/* Kernel choice for mmae_attn_fwd: 0 = by size (default), 1 = K / V of a head resident in LDS in padded images (kv_len up
 * to mmae_attn_max_len(head_dim, 0), MMAE_ESHAPE beyond), 2 = keys streamed through LDS in 64-key tiles with a running softmax
 * (default where the resident images do not fit), 3 = resident in unpadded swizzled images (default where the padded images
 * take more than half of the LDS and these do not).  mmae_attn_fwd_set_one_pass: passes of the resident kernels over the keys,
 * 0 / 1 = one pass with a lazily moved reference point of the exponentials (default), 2 = two passes (row maximum first).
 * Per calling thread; for tests and tuning. */
void mmae_attn_fwd_set_variant(int variant);
void mmae_attn_fwd_set_one_pass(int on);
/* Longest sequence the LDS-resident kernels take at this head_dim (a multiple of 32; 0 for an unsupported head_dim):
 * backward = 0: kv_len up to which mmae_attn_fwd keeps K / V resident (it streams them beyond; q_len is not limited);
 * backward = 1: q_len and kv_len of one mmae_attn_bwd call.  Longer backward sequences are served by the caller in chunks of
 * queries x keys (partial gradients summed), as multimae_amd/ops.py does. */
int mmae_attn_max_len(int head_dim, int backward);
size_t mmae_attn_bwd_workspace_bytes(int B, int H, int q_len);
int mmae_attn_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o, long ldo,
                  const void* d_o, long lddo, const float* lse, void* dq, long lddq, void* dk, long lddk, void* dv,
                  long lddv, int B, int H, int q_len, int kv_len, int head_dim, float scale, void* workspace,
                  size_t workspace_bytes, mmae_stream_t stream);
/* mmae_attn_bwd for one (chunk of queries, chunk of keys) of rows longer than one call takes: q / o / d_o / lse / dq are the
 * chunk's queries with the log-sum-exp and output of their WHOLE rows, k / v / dk / dv the chunk's keys; dq is this key chunk's
 * share (the caller sums the shares over the key chunks, and dk / dv over the query chunks).  mean_key [B,H,head_dim] f32: the
 * mean over ALL keys of the row (the kernels take the rounding residue of dS's row sum out of dQ along the mean key; the
 * residue only cancels over the whole row, so every chunk has to use the same vector); NULL = the mean of this call's keys,
 * which makes the call mmae_attn_bwd. */
int mmae_attn_bwd_chunk(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o, long ldo,
                        const void* d_o, long lddo, const float* lse, const float* mean_key, void* dq, long lddq, void* dk,
                        long lddk, void* dv, long lddv, int B, int H, int q_len, int kv_len, int head_dim, float scale,
                        void* workspace, size_t workspace_bytes, mmae_stream_t stream);
/* Kernel choice for mmae_attn_bwd: 0 = by size (default), 1 = two kernels with padded LDS images, 2 = one kernel whenever
 * Q, dO, K, V of a head fit the 160 KiB LDS, 3 = two kernels with unpadded swizzled images wherever those fit half of the LDS
 * (the default at head_dim 32 where the padded images take more than half).  Per calling thread; for tests and tuning. */
void mmae_attn_bwd_set_variant(int variant);

/* ---- masking (multimae/multimae.py:218-253) --------------------------------------------------
 * One workgroup per sample.  task_dist [B,T] f32 is the Dirichlet sample, noise [B,total] f32 the
 * per-task noise tensors concatenated in task order, noise_all [B,total] f32 the second draw.
 * Outputs: ids_keep [B,k] i64, ids_restore [B,total] i64, mask [B,total] i64 (0 keep / 1 masked,
 * split per task by the caller), plus i32 copies of ids_keep / ids_restore for the gather kernels.
 * Ties sort lower index first.  T <= 8, tokens per task <= 1024, total <= 2048. */
int mmae_mask_generate(const float* task_dist, const float* noise, const float* noise_all, const int* tokens_per_task_host,
                       int T, int B, int num_encoded, int64_t* ids_keep, int64_t* ids_restore, int64_t* mask,
                       int* ids_keep_i32, int* ids_restore_i32, mmae_stream_t stream);

/* ---- token plumbing ------------------------------------------------------------------------- */
/* Pack the kept patches as bf16 rows of the K-concatenated patch matrix:
 * row (b*k + j) = [0 .. | patch (c,ph,pw) of token ids_keep[b,j] in its domain's column block | .. 0].
 * Every domain shares the NH x NW token grid; domain t has its own patch edge patch_host[t] (patch_size_full /
 * stride_level, multimae/input_adapters.py:64-65; a multiple of 4) and image size NH*P_t x NW*P_t.
 * images_host: T device pointers passed by value in a host array: f32 [B,C_t,H_t,W_t], or, for a class-map domain
 * (class_emb_host[t] != NULL: f32 [num_classes_host[t], C_t] embedding table; the call site of the reference's
 * SemSegInputAdapter is run_pretraining_multimae.py:72-85), i64 class ids [B,H_t,W_t] whose pixels are table rows.
 * Class ids outside [0, num_classes) are clamped.  class_emb_host / num_classes_host may be NULL (no class maps). */
int mmae_pack_patches(const void* const* images_host, const int* channels_host, const int* patch_host,
                      const float* const* class_emb_host, const int* num_classes_host, int T, int B, int NH, int NW,
                      const int* ids_keep_i32, int k, void* out_bf16, int Ktot, mmae_stream_t stream);
/* The column block after the patches holds one-hot(domain) so that the weight-gradient GEMM over the packed
 * rows also yields the per-domain Conv2d bias gradient; Ktot >= sum_t C_t*P*P + 8, a multiple of 64. */

/* add_tab[t*ntok + pos][:] = pos_emb_t[:, pos] + bias_t   (multimae/input_adapters.py:129-135) */
int mmae_build_token_table(const float* const* pos_emb_host, const float* const* bias_host, int T, int ntok, int dim,
                           float* add_tab, mmae_stream_t stream);
/* x[b, k + g, :] = global_tokens[g, :]  (multimae/multimae.py:405-406); rows are seq = k + G long */
int mmae_append_global(const float* global_tokens, float* x, int B, int k, int G, int dim, mmae_stream_t stream);
/* d_global[g,:] = sum_b dx[b, k+g, :] and dx_bf16_rows[b*k + j, :] = bf16(dx[b, j, :]) (input of the weight-gradient GEMM) */
int mmae_embed_bwd_reduce(const float* dx, const int* ids_keep_i32, int B, int k, int G, int dim, int T, int ntok,
                          float* d_global, float* const* d_bias_host, void* dx_bf16_rows, mmae_stream_t stream);

/* Decoder input assembly (multimae/output_adapters.py:195-314): from ctx = proj_context(encoder tokens)
 * (bf16 [B, k+G, Dd]; f32 with ctx_is_f32, for a decoder that runs in fp32) build queries f32 [B, ntok, Dd] (own-task slice of the un-shuffled, mask-token-filled,
 * embedded sequence) and context f32 [B, k+G, Dd] (kept tokens + embeddings, global tokens appended raw).
 * emb is f32 [total, Dd] = task_embedding[t] + pos_emb[pos]. */
int mmae_decoder_assemble_fwd(const void* ctx_bf16, int ctx_is_f32, const float* mask_token, const float* emb,
                              const int* ids_keep_i32, const int* ids_restore_i32, int B, int k, int G, int total, int q_start,
                              int q_len, int Dd, float* queries, float* context, mmae_stream_t stream);
/* Gradients of the assembly: d_ctx bf16 (f32 with d_ctx_is_f32) [B,k+G,Dd], rows ld_d_ctx elements apart (0 = Dd; a larger stride lets several
 * decoders write their gradients side by side as the column blocks of one matrix); d_mask_token f32 [Dd]; d_task_emb_host[t]
 * f32 [Dd] (sum over the positions of task t of the embedding gradient; pos_emb is frozen). Tasks have ntok tokens each. */
size_t mmae_decoder_assemble_bwd_workspace_bytes(int B, int T, int Dd);
int mmae_decoder_assemble_bwd(const float* d_queries, const float* d_context, const int* ids_keep_i32,
                              const int* ids_restore_i32, int B, int k, int G, int T, int ntok, int q_start, int q_len,
                              int Dd, void* d_ctx_bf16, int d_ctx_is_f32, long ld_d_ctx, float* d_mask_token, float* const* d_task_emb_host,
                              void* workspace, size_t workspace_bytes, mmae_stream_t stream);
/* emb[t*ntok + pos][:] = task_emb_t[:] + pos_emb[:, pos]  (multimae/output_adapters.py:195-228) */
int mmae_build_decoder_emb(const float* const* task_emb_host, const float* pos_emb, int T, int ntok, int Dd, float* emb,
                           mmae_stream_t stream);

/* image (bf16 or f32 [B,C,H,W]) -> token rows (bf16 [B*NH*NW, C*P*P]): the adjoint of the un-patchify store */
int mmae_patchify_bf16(const void* image, int image_is_f32, int B, int C, int NH, int NW, int P, void* rows, mmae_stream_t stream);

/* Class-embedding gradient of a class-map input domain (nn.Embedding backward of the SemSegInputAdapter named at
 * run_pretraining_multimae.py:72-85): d_emb[cls, c] (+)= sum over the domain's kept tokens (ids_keep in
 * [tok_start, tok_start + NH*NW)) and their P*P pixels of class cls of dA[row, c*P*P + ph*P + pw], where dA (bf16,
 * row stride ldd) is the gradient of the packed patch rows restricted to this domain's columns.  Fixed summation
 * order (bitwise reproducible). */
size_t mmae_class_emb_bwd_workspace_bytes(int num_classes, int dim_emb);
int mmae_class_emb_bwd(const void* dA_bf16, long ldd, const int64_t* class_map, const int* ids_keep_i32, int B, int k,
                       int tok_start, int NH, int NW, int P, int num_classes, int dim_emb, float* d_emb, int accumulate,
                       void* workspace, size_t workspace_bytes, mmae_stream_t stream);

/* ---- masked losses (multimae/criterion.py:96-130, :161-195) ----------------------------------
 * kind 0 = MSE, 1 = L1; norm_pix standardises each target patch over (p1 p2 c) with unbiased variance.
 * pred is bf16 or f32 [B,C,H,W] (pred_is_bf16), target f32, mask i64 [B, nh*nw] or NULL.
 * Writes loss (f32 scalar, nanmean over samples of masked-mean error) and, if d_pred != NULL, the
 * gradient of the loss w.r.t. pred in pred's dtype.  workspace: mmae_masked_loss_workspace_bytes(B). */
size_t mmae_masked_loss_workspace_bytes(int B, int C, int H, int W, int patch);
int mmae_masked_loss(int kind, int norm_pix, const void* pred, int pred_is_bf16, const float* target, const int64_t* mask,
                     int B, int C, int H, int W, int patch, float* loss, void* d_pred, void* workspace,
                     size_t workspace_bytes, mmae_stream_t stream);

/* mmae_masked_loss without a workspace, for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The per-sample mask counts and
 * per-patch-row partial sums (B * (1 + H / patch) floats) live in a fixed array of MMAE_MASKED_LOSS_STATIC_BYTES inside the
 * library's device image, one copy per GPU, held until the process ends; larger inputs return MMAE_EWORKSPACE.  The call
 * allocates and frees nothing and enqueues the same kernels as mmae_masked_loss, so results are bitwise the same.  Calls
 * on one GPU share the array: calls that overlap (different streams without an ordering between them) race.
 * Precondition: the GPU that owns the tensors (and `stream`) is the calling thread's current device; the call checks the
 * device of `loss` against it and returns MMAE_EINVAL on a mismatch (or when `loss` is not device memory). */
#define MMAE_MASKED_LOSS_STATIC_BYTES 262144
int mmae_masked_loss_static(int kind, int norm_pix, const void* pred, int pred_is_bf16, const float* target,
                            const int64_t* mask, int B, int C, int H, int W, int patch, float* loss, void* d_pred,
                            mmae_stream_t stream);

/* Masked cross entropy over a class map (multimae/criterion.py:39-65): logits bf16 or f32 [B,C,H,W], target i64 [B,H,W]
 * (-100 = ignored pixel), mask i64 [B, (H/patch)*(W/patch)] or NULL (plain mean), patch = patch_size / stride.
 * Writes the f32 scalar loss and, if d_logits != NULL, its gradient in the logits' dtype. */
size_t mmae_masked_ce_workspace_bytes(int B, int H, int W);
int mmae_masked_ce(const void* logits, int logits_is_bf16, const int64_t* target, const int64_t* mask, int B, int C, int H,
                   int W, int patch, float label_smoothing, float* loss, void* d_logits, void* workspace,
                   size_t workspace_bytes, mmae_stream_t stream);

/* Truncated depth standardisation (run_pretraining_multimae.py:448-459): per sample (B rows of n values),
 * out = (x - mean) / sqrt(var + eps) with mean / unbiased var over the sorted values of ranks
 * [int(lo_frac * n), int(hi_frac * n)) (0.1, 0.9, eps 1e-6 in the reference).  Order statistics by radix select; no sort.
 * out may alias depth. */
int mmae_standardize_depth(const float* depth, float* out, int B, long n, double lo_frac, double hi_frac, float eps,
                           mmae_stream_t stream);

/* ---- optimiser step over flat buffers (AdamW; run_pretraining_multimae_v2.py:339-344) -----------
 * p, g, m, v are f32 [n]; elements [0, n_decay) use weight_decay, the rest none.
 * grad_scale multiplies g first (1/world_size for the data-parallel mean).
 * If w_bf16 != NULL the updated parameters are also written there as bf16.
 * Gradient clipping / skipping of NativeScalerWithGradNormCount (utils/native_scaler.py:33-51) without a host round trip:
 * grad_sumsq (device, from mmae_sumsq over g; NULL = neither) gives norm = sqrt(sumsq) * grad_scale; max_norm > 0 scales
 * the gradient by min(1, max_norm / (norm + 1e-6)); skip_norm > 0 leaves p, m, v untouched when norm >= skip_norm.
 * step_dev (device int, optional): number of updates applied so far; when given, the bias corrections use
 * step_dev[0] + 1 instead of `step`, and the count advances only if the update was applied. */
int mmae_adamw_flat(float* p, const float* g, float* m, float* v, long n, long n_decay, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void* w_bf16,
                    const float* grad_sumsq, float max_norm, float skip_norm, int* step_dev, mmae_stream_t stream);
/* sum of squares of f32 [n] into out[0] (f32); partials in workspace (>= 4096 bytes) */
/* mmae_adamw_flat with two by-products of its read of the gradients, for the update that runs range by range beside backward
 * (no clipping / skipping): clear_grad != 0 writes zeros over g behind the read (the optimizer.zero_grad() of the next
 * iteration, run_pretraining_multimae.py:507), and sumsq_part (or NULL) receives mmae_adamw_flat_parts(n) partial sums of g^2 over
 * the range (raw gradients, before grad_scale): the squared norm of utils/native_scaler.py:63-83 is mmae_sum_partials over
 * the parts of all ranges, added in a fixed order. */
int mmae_adamw_flat_parts(long n);
int mmae_adamw_flat_ex(float* p, float* g, float* m, float* v, long n, long n_decay, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int step, float grad_scale, void* w_bf16, const float* grad_sumsq, float max_norm,
                       float skip_norm, int* step_dev, int clear_grad, float* sumsq_part, mmae_stream_t stream);
int mmae_sum_partials(const float* part, int nparts, float* out, mmae_stream_t stream);
int mmae_sumsq(const float* x, long n, float* out, void* workspace, size_t workspace_bytes, mmae_stream_t stream);
/* dst(bf16)[rows, cols] = src(f32)[rows, cols]; dst_t(bf16)[cols, rows] = transpose (either may be NULL) */
int mmae_cast_transpose(const float* src, int rows, int cols, void* dst_bf16, void* dst_t_bf16, mmae_stream_t stream);
/* Transposed copies of many 2-D bf16 tensors in one launch: table[t] = {element offset, rows, cols, first 64x64 tile};
 * tensor t is [rows, cols] at src + offset and is written as [cols, rows] at dst + offset (device int32 table). */
int mmae_transpose_multi(const void* src_bf16, void* dst_bf16, const int* table, int n_tensors, int total_tiles,
                         mmae_stream_t stream);
/* y(bf16) = x(f32) elementwise */
int mmae_cast_bf16(const float* x, void* y_bf16, long n, mmae_stream_t stream);
/* ---- fp32 path of the output adapters listed in fp32_output_adapters (multimae/multimae.py:426-436: the reference runs
 * those decoders outside autocast, fp32 end to end) ----------------------------------------------------------------
 * mmae_split_f32: src f32 [rows, cols] (row stride ld_src) -> three bf16 blocks hi = bf16(x), lo = bf16(x - hi) in the order
 * pattern 0: hi, hi, lo / pattern 1: hi, lo, hi; side by side (dst [rows, 3 * block_cols]) or, with by_rows, stacked (dst
 * [3 * rows, block_cols]); block_cols (0 = cols) >= cols pads every block with zero columns.  A pattern-0 operand against a pattern-1 operand in mmae_gemm_nt (blocks along K) or mmae_gemm_tn (blocks along M)
 * gives a_hi w_hi + a_hi w_lo + a_lo w_hi in the fp32 accumulators: nn.Linear with fp32-faithful operands (16 mantissa
 * bits each) on the bf16 matrix cores. */
int mmae_split_f32(const float* src, long ld_src, int rows, int cols, void* dst_bf16, int pattern, int by_rows, int block_cols,
                   mmae_stream_t stream);
/* nn.GELU (erf form) and dx = dy * gelu'(x), elementwise in fp32; n a multiple of 4 (multimae/multimae_utils.py:164) */
int mmae_gelu_f32(const float* x, float* y, long n, mmae_stream_t stream);
int mmae_gelu_bwd_f32(const float* x, const float* dy, float* dx, long n, mmae_stream_t stream);
/* nn.LayerNorm with fp32 output / fp32 incoming gradient (multimae/output_adapters.py:350-353 outside autocast) */
int mmae_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int rows,
                           int dim, float eps, mmae_stream_t stream);
int mmae_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                           float* dgamma, float* dbeta, int accumulate, int rows, int dim, void* workspace,
                           size_t workspace_bytes, mmae_stream_t stream);
/* softmax(q k^T scale) v in fp32 on the vector ALU (multimae/multimae_utils.py:200-204, :240-244 outside autocast); rows
 * addressed like mmae_attn_fwd; head_dim 32 or 64; the K / V rows (forward, dQ) and the Q / dO rows (dK / dV) of one head must
 * fit the LDS as f32.  workspace: mmae_attn_f32_bwd_workspace_bytes (rowsum(dO * O)). */
int mmae_attn_f32_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo, float* lse,
                      int B, int H, int q_len, int kv_len, int head_dim, float scale, mmae_stream_t stream);
size_t mmae_attn_f32_bwd_workspace_bytes(int B, int H, int q_len);
int mmae_attn_f32_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const float* o, long ldo,
                      const float* d_o, long lddo, const float* lse, float* dq, long lddq, float* dk, long lddk, float* dv,
                      long lddv, int B, int H, int q_len, int kv_len, int head_dim, float scale, void* workspace,
                      size_t workspace_bytes, mmae_stream_t stream);

/* ---- data-parallel gradient exchange over RCCL ------------------------------------------------------
 * The reference wraps the model in DistributedDataParallel (run_pretraining_multimae.py:224-226: rank-0 parameter
 * broadcast at construction, bucketed gradient all-reduce during backward) over torch.distributed's NCCL backend
 * (utils/dist.py:94-107).  These entries are that exchange on raw device pointers: one communicator per process and GPU,
 * created from a 128-byte id that rank 0 draws (mmae_comm_unique_id) and hands to the other ranks by any host channel;
 * mmae_comm_init is collective (every rank calls it, with its GPU current); all-reduce (SUM, in place) and broadcast only
 * enqueue on `stream`.  librccl is resolved at the first call (the copy already loaded in the process, else librccl.so). */
#define MMAE_COMM_ID_BYTES 128
enum { MMAE_COMM_F32 = 0, MMAE_COMM_BF16 = 1 };
int mmae_comm_unique_id(void* id_host);
int mmae_comm_init(const void* id_host, int rank, int world, void** handle);
int mmae_comm_allreduce_bucket(void* handle, void* ptr, size_t count, int dtype, mmae_stream_t stream);
int mmae_comm_broadcast(void* handle, void* ptr, size_t count, int dtype, int root, mmae_stream_t stream);
/* rank and rank count as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount) */
int mmae_comm_info(void* handle, int* rank_host, int* world_host);
int mmae_comm_destroy(void* handle);

/* One idle workgroup for `us` microseconds on `stream` (no reference counterpart).  The host side times pairs of these on
 * two streams to learn whether the streams share a hardware queue (a process has 4; streams on one queue do not overlap). */
/* CUs that the one-workgroup-per-CU GEMM kernels plan their tiles, splits and grids for (no reference counterpart: the
 * reference leaves CU sharing between DDP's NCCL kernels and its matrix products to the libraries).  n = 0: all CUs (default);
 * 32 <= n < CUs: leave the others free, e.g. for the channels of a collective that runs beside the products (a workgroup
 * of these kernels cannot share a CU with another kernel's resident workgroup and would otherwise start a tile time
 * late).  Process-wide; returns the count in effect.  Env MMAE_CU_BUDGET sets the initial value. */
int mmae_set_cu_budget(int n);
int mmae_get_cu_budget(void);  /* the count in effect */
int mmae_spin_us(int us, mmae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMAE_H */
