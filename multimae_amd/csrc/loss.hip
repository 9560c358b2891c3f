// Masked MSE / L1 loss with optional per-patch target standardisation (multimae/criterion.py:96-130, :161-195 of
// the reference), forward value and gradient w.r.t. the prediction in one pass.  HBM-bound:
// per image it reads pred + target once (target three times from L2 when norm_pix) and writes d_pred.
//
//   loss_b = sum_{masked patches} sum_{pixels} (1/C) sum_c err  /  (P*P * nmask_b)
//   loss   = mean of loss_b over the samples with nmask_b > 0 (nanmean); 0 if no token is masked at all
//   d_pred = mask * err' / (C * P*P * nmask_b * nvalid)      (NaN for a sample with nmask_b == 0, as in the reference)
#include "common.h"
#include <mutex>

struct LossArgs {
    const void* pred; const float* target; const int64_t* mask;
    int B, C, H, W, P, nh, nw;
    int kind, norm_pix, pred_bf16;
    float* nmask;     // [B]
    float* perr;      // [B][nh] partial error sums (already masked, divided by C)
    float* loss; void* d_pred;
};

__global__ void loss_count_kernel(LossArgs a) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    float s = 0.f;
    const int n = a.nh * a.nw;
    if (a.mask) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) s += (float)a.mask[(long)b * n + i];
    } else {
        s = threadIdx.x == 0 ? (float)n : 0.f;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) a.nmask[b] = s;
}

template <bool BF16>
__device__ __forceinline__ f32x4 load_pred4(const void* p, long off) {
    if (BF16) {
        const bf16x4 v = *(const bf16x4*)((const bf16*)p + off);
        return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    return *(const f32x4*)((const float*)p + off);
}

// grid (nh, B); block (W/4, RR): thread (qx, ry) owns pixel quad qx of rows ry, ry+RR, ... of the C*P rows of this patch row.
// NR > 0: the thread owns at most NR rows and keeps their prediction and target quads in registers: every load of the block is
// issued before the first reduction (one trip to memory instead of one per row: the stores to d_pred keep the compiler from
// moving later rows' loads ahead of them), and the standardisation statistics (norm_pix: mean, then centred sum of squares) are
// taken from the registers instead of two more reads of the target.  NR == 0: any number of rows, re-read per pass.
// Both forms add in the same order, so their results are bitwise equal.
template <bool BF16, int NR>
__global__ __launch_bounds__(NR > 0 ? 256 : 1024) void loss_main_kernel(LossArgs a) {
    __shared__ float s_part[1024];                    // one partial per thread
    __shared__ float s_sum[64], s_sq[64], s_err[64];  // per patch column (nw <= 64)
    __shared__ float s_nvalid;
    const int b = blockIdx.y, ih = blockIdx.x;
    const int qx = threadIdx.x, ry = threadIdx.y, RR = blockDim.y;
    const int tid = ry * blockDim.x + qx, nthreads = blockDim.x * blockDim.y;
    const int pw_idx = (qx * 4) / a.P;  // patch column of this quad
    const int rows = a.C * a.P;
    const float n_el = (float)(a.C * a.P * a.P);
    constexpr int NRA = NR > 0 ? NR : 1;

    auto off_of = [&](int r) {
        const int c = r / a.P, ph = r - c * a.P;
        return (((long)b * a.C + c) * a.H + ih * a.P + ph) * a.W + qx * 4;
    };
    // rows past the thread's last are fetched from its last row and skipped in the sums: no branch between the loads
    f32x4 tv[NRA], pv[NRA];
    if (NR > 0) {
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int r = ry + i * RR;
            const long off = off_of(r < rows ? r : rows - 1);
            tv[i] = *(const f32x4*)(a.target + off);
            pv[i] = load_pred4<BF16>(a.pred, off);
        }
    }

    // fixed-order sum of the partials of every patch column (bitwise reproducible)
    auto patch_reduce = [&](float v, float* dst) {
        __syncthreads();
        s_part[tid] = v;
        __syncthreads();
        if (tid < a.nw) {
            const int q0 = tid * (a.P / 4), q1 = q0 + a.P / 4;
            float t = 0.f;
            for (int y = 0; y < RR; ++y)
                for (int q = q0; q < q1; ++q) t += s_part[y * blockDim.x + q];
            dst[tid] = t;
        }
        __syncthreads();
    };
    {   // samples with at least one masked token: counted by all threads (one thread walking the batch was most of this
        // kernel's time), combined by a tree; the terms are small integers, so the order does not matter
        float nvp = 0.f;
        for (int i = tid; i < a.B; i += nthreads) nvp += a.nmask[i] > 0.f ? 1.f : 0.f;
        s_part[tid] = nvp;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if (tid < off && tid + off < nthreads) s_part[tid] += s_part[tid + off];
            __syncthreads();
        }
        if (tid == 0) s_nvalid = s_part[0];
        __syncthreads();
    }

    float mean = 0.f, inv_std = 1.f;
    if (a.norm_pix) {
        float s = 0.f;
        if (NR > 0) {
#pragma unroll
            for (int i = 0; i < NRA; ++i)
                if (ry + i * RR < rows) s += tv[i][0] + tv[i][1] + tv[i][2] + tv[i][3];
        } else {
            for (int r = ry; r < rows; r += RR) {
                const f32x4 t = *(const f32x4*)(a.target + off_of(r));
                s += t[0] + t[1] + t[2] + t[3];
            }
        }
        patch_reduce(s, s_sum);
        mean = s_sum[pw_idx] / n_el;
        float q = 0.f;
        if (NR > 0) {
#pragma unroll
            for (int i = 0; i < NRA; ++i) {
                const f32x4 t = tv[i] - mean;
                if (ry + i * RR < rows) q += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
            }
        } else {
            for (int r = ry; r < rows; r += RR) {
                const f32x4 t = *(const f32x4*)(a.target + off_of(r)) - mean;
                q += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
            }
        }
        patch_reduce(q, s_sq);
        inv_std = 1.0f / sqrtf(s_sq[pw_idx] / (n_el - 1.0f) + 1e-6f);  // unbiased variance, eps 1e-6
    }

    const float mk = a.mask ? (float)a.mask[(long)b * a.nh * a.nw + ih * a.nw + pw_idx] : 1.f;
    const float nm = a.nmask[b];
    // gradient scale: 1 / (C * P*P * nmask_b * nvalid); 0 * inf = NaN when this sample has no masked token
    // and exactly 0 everywhere when no sample has one (the reference then returns a constant 0)
    const float gscale = s_nvalid > 0.f ? (nm > 0.f ? 1.f / s_nvalid : 0.f) * (1.0f / ((float)(a.P * a.P) * nm)) / (float)a.C : 0.f;
    float e = 0.f;
    auto one_row = [&](long off, const f32x4& p4, const f32x4& traw) {
        const f32x4 t4 = (traw - mean) * inv_std;
        const f32x4 d = p4 - t4;
        f32x4 g;
        if (a.kind == 0) {
            e += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            g = 2.f * d;
        } else {
            e += fabsf(d[0]) + fabsf(d[1]) + fabsf(d[2]) + fabsf(d[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = d[j] > 0.f ? 1.f : (d[j] < 0.f ? -1.f : 0.f);
        }
        if (a.d_pred) {
            g = g * (mk * gscale);
            if (BF16) {
                bf16x4 o = {(bf16)g[0], (bf16)g[1], (bf16)g[2], (bf16)g[3]};
                *(bf16x4*)((bf16*)a.d_pred + off) = o;
            } else {
                *(f32x4*)((float*)a.d_pred + off) = g;
            }
        }
    };
    if (NR > 0) {
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int r = ry + i * RR;
            if (r < rows) one_row(off_of(r), pv[i], tv[i]);
        }
    } else {
        for (int r = ry; r < rows; r += RR) {
            const long off = off_of(r);
            one_row(off, load_pred4<BF16>(a.pred, off), *(const f32x4*)(a.target + off));
        }
    }
    patch_reduce(e * mk, s_err);
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < a.nw; ++i) t += s_err[i];
        a.perr[(long)b * a.nh + ih] = t / (float)a.C;
    }
}

__global__ void loss_final_kernel(LossArgs a) {
    __shared__ float red[16];
    float tot = 0.f, cnt = 0.f;
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
        const float nm = a.nmask[b];
        if (nm > 0.f) {
            float s = 0.f;
#pragma unroll 8
            for (int i = 0; i < a.nh; ++i) s += a.perr[(long)b * a.nh + i];
            tot += s / ((float)(a.P * a.P) * nm);
            cnt += 1.f;
        }
    }
    tot = block_sum(tot, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) a.loss[0] = cnt > 0.f ? tot / cnt : 0.f;
}

extern "C" size_t mmae_masked_loss_workspace_bytes(int B, int C, int H, int W, int patch) {
    (void)C; (void)W;
    return (size_t)B * sizeof(float) * (1 + (size_t)(H / (patch > 0 ? patch : 1)));
}

// Argument checks and the three launches shared by the two entry points; `ws` holds B * (1 + H / patch) floats.
static int masked_loss_launch(int kind, int norm_pix, const void* pred, int pred_is_bf16, const float* target,
                              const int64_t* mask, int B, int C, int H, int W, int patch, float* loss, void* d_pred,
                              float* ws, hipStream_t stream) {
    LossArgs a = {};
    a.pred = pred; a.target = target; a.mask = mask; a.B = B; a.C = C; a.H = H; a.W = W; a.P = patch;
    a.nh = H / patch; a.nw = W / patch; a.kind = kind; a.norm_pix = norm_pix; a.pred_bf16 = pred_is_bf16;
    a.nmask = ws; a.perr = a.nmask + B; a.loss = loss; a.d_pred = d_pred;
    hipLaunchKernelGGL(loss_count_kernel, dim3(B), dim3(64), 0, stream, a);
    MMAE_CHECK_LAUNCH("loss_count");
    const int qx = W / 4;
    int rr = 256 / qx; if (rr < 1) rr = 1; if (rr > C * patch) rr = C * patch;
    const int per_thread = (C * patch + rr - 1) / rr;  // rows of the patch row that one thread owns
#define LOSS_MAIN(NR)                                                                                                      \
    do {                                                                                                                   \
        if (pred_is_bf16) hipLaunchKernelGGL((loss_main_kernel<true, NR>), dim3(a.nh, B), dim3(qx, rr), 0, stream, a);      \
        else hipLaunchKernelGGL((loss_main_kernel<false, NR>), dim3(a.nh, B), dim3(qx, rr), 0, stream, a);                  \
    } while (0)
    if (qx * rr > 256) LOSS_MAIN(0);            // (the register forms are built for blocks of up to 256 threads)
    else if (per_thread <= 4) LOSS_MAIN(4);     // one channel at 224^2
    else if (per_thread <= 12) LOSS_MAIN(12);   // three channels at 224^2
    else if (per_thread <= 24) LOSS_MAIN(24);   // three channels at 384^2
    else LOSS_MAIN(0);
#undef LOSS_MAIN
    MMAE_CHECK_LAUNCH("loss_main");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, a);
    MMAE_CHECK_LAUNCH("loss_final");
    return MMAE_OK;
}

#define MASKED_LOSS_CHECK_ARGS(name)                                                                                         \
    MMAE_REQUIRE(pred && target && loss, MMAE_EINVAL, name ": null pointer");                                                 \
    MMAE_REQUIRE(kind == 0 || kind == 1, MMAE_EINVAL, name ": kind %d not in {0 (MSE), 1 (L1)}", kind);                       \
    MMAE_REQUIRE(B > 0 && C > 0 && patch >= 4 && patch % 4 == 0 && H % patch == 0 && W % patch == 0, MMAE_ESHAPE,             \
                 name ": image %dx%d must be a multiple of the patch %d (itself a multiple of 4)", H, W, patch);              \
    MMAE_REQUIRE(W / 4 <= 1024 && W / patch <= 64, MMAE_ESHAPE, name ": width %d too large", W)

extern "C" int mmae_masked_loss(int kind, int norm_pix, const void* pred, int pred_is_bf16, const float* target,
                                const int64_t* mask, int B, int C, int H, int W, int patch, float* loss, void* d_pred,
                                void* workspace, size_t workspace_bytes, hipStream_t stream) {
    MASKED_LOSS_CHECK_ARGS("masked_loss");
    MMAE_REQUIRE(workspace, MMAE_EINVAL, "masked_loss: null pointer");
    MMAE_REQUIRE(workspace_bytes >= mmae_masked_loss_workspace_bytes(B, C, H, W, patch), MMAE_EWORKSPACE, "masked_loss: workspace too small");
    return masked_loss_launch(kind, norm_pix, pred, pred_is_bf16, target, mask, B, C, H, W, patch, loss, d_pred,
                              (float*)workspace, stream);
}

// The same loss for callers that cannot allocate (HIP graph capture, serving loops): the per-sample mask counts and the
// per-patch-row partial sums live in a fixed array in the library's own device memory instead of a caller workspace.
// Every GPU holds its own copy of the array; host code reaches it only through the address hipGetSymbolAddress returns
// for the current device (the array's name is not a device address on the host side).  The kernels overwrite every entry
// they later read, so nothing needs clearing between calls; two calls in flight on one GPU share the array and race.
#define MMAE_LOSS_STATIC_FLOATS (MMAE_MASKED_LOSS_STATIC_BYTES / 4)  /* B * (1 + H / patch) floats, e.g. 2048 images of 24 patch rows */
#ifndef MMAE_MASKED_LOSS_STATIC_BYTES
#error "include/mmae.h must define MMAE_MASKED_LOSS_STATIC_BYTES"
#endif
static __device__ __align__(256) float scratch[MMAE_LOSS_STATIC_FLOATS];

#define MMAE_MAX_DEVICES 64
static float* g_scratch_ptr[MMAE_MAX_DEVICES];  // per device; written once under the mutex
static std::mutex g_scratch_mutex;

static int static_scratch(float** out, const void* device_ptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    MMAE_REQUIRE(e == hipSuccess && dev >= 0 && dev < MMAE_MAX_DEVICES, MMAE_EINVAL, "masked_loss_static: no current device (%s)",
                 hipGetErrorString(e));
    // The array is per GPU and its address is taken for the calling thread's current device: the tensors (and so the
    // stream the kernels run on) must live on that device, else the kernels would be handed another GPU's address.
    hipPointerAttribute_t attr;
    e = hipPointerGetAttributes(&attr, device_ptr);
    MMAE_REQUIRE(e == hipSuccess, MMAE_EINVAL, "masked_loss_static: the loss pointer is not device memory (%s)", hipGetErrorString(e));
    MMAE_REQUIRE(attr.device == dev, MMAE_EINVAL, "masked_loss_static: tensors live on device %d but the current device is %d "
                 "(make the tensors' device current before the call)", attr.device, dev);
    std::lock_guard<std::mutex> guard(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        MMAE_REQUIRE(e == hipSuccess && p, MMAE_ELAUNCH, "masked_loss_static: hipGetSymbolAddress failed on device %d: %s", dev,
                     hipGetErrorString(e));
        g_scratch_ptr[dev] = (float*)p;
    }
    *out = g_scratch_ptr[dev];
    return MMAE_OK;
}

extern "C" int mmae_masked_loss_static(int kind, int norm_pix, const void* pred, int pred_is_bf16, const float* target,
                                       const int64_t* mask, int B, int C, int H, int W, int patch, float* loss, void* d_pred,
                                       hipStream_t stream) {
    MASKED_LOSS_CHECK_ARGS("masked_loss_static");
    const size_t need = mmae_masked_loss_workspace_bytes(B, C, H, W, patch);
    MMAE_REQUIRE(need <= sizeof(float) * MMAE_LOSS_STATIC_FLOATS, MMAE_EWORKSPACE,
                 "masked_loss_static: %d images x %d patch rows need %zu bytes of partial sums, the fixed array holds %zu "
                 "(use mmae_masked_loss with a workspace)", B, H / patch, need, sizeof(float) * MMAE_LOSS_STATIC_FLOATS);
    float* ws = nullptr;
    const int rc = static_scratch(&ws, loss);
    if (rc != MMAE_OK) return rc;
    return masked_loss_launch(kind, norm_pix, pred, pred_is_bf16, target, mask, B, C, H, W, patch, loss, d_pred, ws, stream);
}

// ---------------------------------------------------------------------------------------------
// Masked cross entropy over a class map (multimae/criterion.py:39-65 of the reference): per-pixel
//   ce = (1 - eps) * (lse - x_target) + eps * (lse - mean_c x_c)          (eps = label smoothing; 0 for target == -100)
// masked mean per sample over the pixels of masked patches, then mean over the samples that have any (nanmean).
// One thread per pixel; classes are strided by H*W so each class step is a coalesced row of pixels.  The second sweep
// (softmax - target distribution, scaled) re-reads the logits, mostly from L2.
// ---------------------------------------------------------------------------------------------
struct CeArgs {
    const void* logits; const int64_t* target; const int64_t* mask;
    int B, C, H, W, P, nh, nw, nblk;
    float eps;
    float* nmask; float* perr; float* loss; void* d_logits;
};

template <bool BF16>
__global__ __launch_bounds__(256) void ce_main_kernel(CeArgs a) {
    __shared__ float red[16];
    __shared__ float s_nvalid;
    const int b = blockIdx.y;
    const long HW = (long)a.H * a.W;
    const long px = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = px < HW;
    {   // samples with at least one masked token (small integers: any order)
        float nvp = 0.f;
        for (int i = threadIdx.x; i < a.B; i += 256) nvp += a.nmask[i] > 0.f ? 1.f : 0.f;
        nvp = block_sum(nvp, red);
        if (threadIdx.x == 0) s_nvalid = nvp;
    }
    __syncthreads();
    float e = 0.f;
    if (live) {
        const int y = (int)(px / a.W), x = (int)(px - (long)y * a.W);
        const float mk = a.mask ? (float)a.mask[(long)b * a.nh * a.nw + (y / a.P) * a.nw + x / a.P] : 1.f;
        const long t = a.target[(long)b * HW + px];
        const bool ignored = t == -100;
        const long base = (long)b * a.C * HW + px;
        auto ld = [&](int c) -> float {
            return BF16 ? (float)((const bf16*)a.logits)[base + (long)c * HW] : ((const float*)a.logits)[base + (long)c * HW];
        };
        float m = -INFINITY, s = 0.f, sumx = 0.f, xt = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float v = ld(c);
            sumx += v;
            if (c == t) xt = v;
            if (v > m) { s = s * __expf(m - v) + 1.f; m = v; } else { s += __expf(v - m); }
        }
        const float lse = m + logf(s);
        const float ce = ignored ? 0.f : (1.f - a.eps) * (lse - xt) + a.eps * (lse - sumx / (float)a.C);
        e = ce * mk;
        if (a.d_logits) {
            const float nm = a.nmask[b];
            // 1 / (P*P * nmask_b * nvalid); 0 * inf = NaN for a sample without masked tokens (as the reference's autograd)
            float g = s_nvalid > 0.f ? (nm > 0.f ? 1.f / s_nvalid : 0.f) * (1.0f / ((float)(a.P * a.P) * nm)) * mk : 0.f;
            if (ignored) g = 0.f;
            const float uni = a.eps / (float)a.C;
            for (int c = 0; c < a.C; ++c) {
                const float p = __expf(ld(c) - lse);
                const float d = (p - uni - (c == t ? 1.f - a.eps : 0.f)) * g;
                if (BF16) ((bf16*)a.d_logits)[base + (long)c * HW] = (bf16)d;
                else ((float*)a.d_logits)[base + (long)c * HW] = d;
            }
        }
    }
    e = block_sum(e, red);
    if (threadIdx.x == 0) a.perr[(long)b * a.nblk + blockIdx.x] = e;
}

// The same with the pixel's logits held in registers (C <= CP): every class row is requested before the first one is used,
// the logits are read once (the sweep above re-reads them for the gradient), and the log-sum-exp is taken against the exact
// maximum.  133 classes at 56^2 x 256 samples in fp32: 0.85 GB moved per launch.
template <bool BF16, int CP>
__global__ __launch_bounds__(256) void ce_main_regs_kernel(CeArgs a) {
    __shared__ float red[16];
    __shared__ float s_nvalid;
    const int b = blockIdx.y;
    const long HW = (long)a.H * a.W;
    const long px = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = px < HW;
    const long base = (long)b * a.C * HW + (live ? px : 0);
    float v[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const long off = base + (long)(c < a.C ? c : a.C - 1) * HW;  // past the last class: its row again, ignored below
        v[c] = BF16 ? (float)((const bf16*)a.logits)[off] : ((const float*)a.logits)[off];
    }
    {   // samples with at least one masked token (small integers: any order)
        float nvp = 0.f;
        for (int i = threadIdx.x; i < a.B; i += 256) nvp += a.nmask[i] > 0.f ? 1.f : 0.f;
        nvp = block_sum(nvp, red);
        if (threadIdx.x == 0) s_nvalid = nvp;
    }
    __syncthreads();
    float e = 0.f;
    if (live) {
        const int y = (int)(px / a.W), x = (int)(px - (long)y * a.W);
        const float mk = a.mask ? (float)a.mask[(long)b * a.nh * a.nw + (y / a.P) * a.nw + x / a.P] : 1.f;
        const long t = a.target[(long)b * HW + px];
        const bool ignored = t == -100;
        float m = -INFINITY, sumx = 0.f, xt = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            if (c < a.C) {
                m = fmaxf(m, v[c]);
                sumx += v[c];
                if (c == t) xt = v[c];
            }
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            v[c] = c < a.C ? __expf(v[c] - m) : 0.f;  // the un-normalised probabilities replace the logits
            s += v[c];
        }
        const float lse = m + logf(s);
        const float ce = ignored ? 0.f : (1.f - a.eps) * (lse - xt) + a.eps * (lse - sumx / (float)a.C);
        e = ce * mk;
        if (a.d_logits) {
            const float nm = a.nmask[b];
            // 1 / (P*P * nmask_b * nvalid); 0 * inf = NaN for a sample without masked tokens (as the reference's autograd)
            float g = s_nvalid > 0.f ? (nm > 0.f ? 1.f / s_nvalid : 0.f) * (1.0f / ((float)(a.P * a.P) * nm)) * mk : 0.f;
            if (ignored) g = 0.f;
            const float uni = a.eps / (float)a.C, inv_s = 1.f / s;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                if (c < a.C) {
                    const float d = (v[c] * inv_s - uni - (c == t ? 1.f - a.eps : 0.f)) * g;
                    if (BF16) ((bf16*)a.d_logits)[base + (long)c * HW] = (bf16)d;
                    else ((float*)a.d_logits)[base + (long)c * HW] = d;
                }
            }
        }
    }
    e = block_sum(e, red);
    if (threadIdx.x == 0) a.perr[(long)b * a.nblk + blockIdx.x] = e;
}

__global__ void ce_final_kernel(CeArgs a) {
    __shared__ float red[16];
    float tot = 0.f, cnt = 0.f;
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
        const float nm = a.nmask[b];
        if (nm > 0.f) {
            float s = 0.f;
#pragma unroll 8
            for (int i = 0; i < a.nblk; ++i) s += a.perr[(long)b * a.nblk + i];
            tot += s / ((float)(a.P * a.P) * nm);
            cnt += 1.f;
        }
    }
    tot = block_sum(tot, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) a.loss[0] = cnt > 0.f ? tot / cnt : 0.f;
}

extern "C" size_t mmae_masked_ce_workspace_bytes(int B, int H, int W) {
    return (size_t)B * sizeof(float) * (1 + (size_t)(((long)H * W + 255) / 256));
}

extern "C" int mmae_masked_ce(const void* logits, int logits_is_bf16, const int64_t* target, const int64_t* mask, int B, int C,
                              int H, int W, int patch, float label_smoothing, float* loss, void* d_logits, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(logits && target && loss && workspace, MMAE_EINVAL, "masked_ce: null pointer");
    MMAE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, MMAE_ESHAPE, "masked_ce: empty input");
    if (!mask) patch = 1;  // plain mean over all pixels
    MMAE_REQUIRE(patch >= 1 && H % patch == 0 && W % patch == 0, MMAE_ESHAPE, "masked_ce: map %dx%d must be a multiple of the patch %d", H, W, patch);
    MMAE_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, MMAE_EINVAL, "masked_ce: label_smoothing %f", label_smoothing);
    MMAE_REQUIRE(workspace_bytes >= mmae_masked_ce_workspace_bytes(B, H, W), MMAE_EWORKSPACE, "masked_ce: workspace too small");
    CeArgs a = {};
    a.logits = logits; a.target = target; a.mask = mask; a.B = B; a.C = C; a.H = H; a.W = W; a.P = patch;
    a.nh = H / patch; a.nw = W / patch; a.nblk = (int)(((long)H * W + 255) / 256); a.eps = label_smoothing;
    a.nmask = (float*)workspace; a.perr = a.nmask + B; a.loss = loss; a.d_logits = d_logits;
    LossArgs c = {};
    c.mask = mask; c.nh = a.nh; c.nw = a.nw; c.nmask = a.nmask;
    hipLaunchKernelGGL(loss_count_kernel, dim3(B), dim3(64), 0, stream, c);
    MMAE_CHECK_LAUNCH("ce_count");
    if (C <= 136) {  // the pixel's logits fit its registers (133 classes in the reference's configuration)
        if (logits_is_bf16) hipLaunchKernelGGL((ce_main_regs_kernel<true, 136>), dim3(a.nblk, B), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((ce_main_regs_kernel<false, 136>), dim3(a.nblk, B), dim3(256), 0, stream, a);
    } else if (logits_is_bf16) {
        hipLaunchKernelGGL(ce_main_kernel<true>, dim3(a.nblk, B), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(ce_main_kernel<false>, dim3(a.nblk, B), dim3(256), 0, stream, a);
    }
    MMAE_CHECK_LAUNCH("ce_main");
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(256), 0, stream, a);
    MMAE_CHECK_LAUNCH("ce_final");
    return MMAE_OK;
}
