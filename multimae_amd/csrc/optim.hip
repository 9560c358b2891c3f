// Optimiser-side kernels over flat parameter storage (gfx950): fused AdamW, gradient norm, bf16 weight copies.
// HBM-bound: AdamW moves 28 bytes per parameter (p, g, m, v read; p, m, v written) (+2 for the bf16 copy).
#include "common.h"

// Clipping / skipping without a host round trip: `sumsq` (device) holds sum(g^2) of the raw gradient buffer, the true gradient
// is g * gscale.  max_norm > 0 scales by min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_); skip_norm > 0
// leaves every buffer untouched when norm >= skip_norm.  `step_dev` (device, optional) is the count of updates applied so far:
// the bias corrections then use step_dev[0] + 1 and a skipped update does not advance it.
struct AdamArgs {
    float* p; const float* g; float* m; float* v; long n4, n_decay;
    float lr, b1, b2, eps, wd, gscale;
    int step_host;
    const float* sumsq; float max_norm, skip_norm;
    int* step_dev;
    bf16* wb;
    float* gclear;      // not null: the gradient range, cleared behind the read (the next step accumulates into it again)
    float* sumsq_part;  // not null: [gridDim.x] partial sums of g^2 over the range (raw gradients, before gscale)
};

__device__ __forceinline__ bool adam_plan(const AdamArgs& a, float& gs, int& step) {
    gs = a.gscale;
    step = a.step_dev ? a.step_dev[0] + 1 : a.step_host;
    if (a.sumsq) {
        const float nrm = sqrtf(a.sumsq[0]) * a.gscale;
        if (a.skip_norm > 0.f && nrm >= a.skip_norm) return false;
        if (a.max_norm > 0.f) gs *= fminf(1.0f, a.max_norm / (nrm + 1e-6f));
    }
    return true;
}

__global__ void adamw_flat_kernel(AdamArgs a) {
    // PyTorch AdamW: p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    //                p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
    float gscale; int t;
    if (!adam_plan(a, gscale, t)) return;
    const float bc1 = 1.0f - powf(a.b1, (float)t);
    const float bc2_sqrt = sqrtf(1.0f - powf(a.b2, (float)t));
    const float step = a.lr / bc1, lr = a.lr, wd = a.wd, b1 = a.b1, b2 = a.b2, eps = a.eps;
    float* p = a.p; float* m = a.m; float* v = a.v; const float* g = a.g; bf16* wb = a.wb;
    __shared__ float red[16];
    float ss = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = ((f32x4*)p)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 graw = ((const f32x4*)g)[i];
        if (a.sumsq_part) ss += graw[0] * graw[0] + graw[1] * graw[1] + graw[2] * graw[2] + graw[3] * graw[3];
        if (a.gclear) ((f32x4*)a.gclear)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 gg = graw * gscale;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = pp[e];
            if (i * 4 + e < a.n_decay) x *= 1.0f - lr * wd;
            const float m1 = b1 * mm[e] + (1.0f - b1) * gg[e];
            const float v1 = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
            x -= step * m1 / (sqrtf(v1) / bc2_sqrt + eps);
            pp[e] = x; mm[e] = m1; vv[e] = v1;
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
        if (wb) {
            bf16x4 o = {(bf16)pp[0], (bf16)pp[1], (bf16)pp[2], (bf16)pp[3]};
            ((bf16x4*)wb)[i] = o;
        }
    }
    if (a.sumsq_part) {  // fixed order: a thread's elements in index order, then the block's threads (block_sum)
        ss = block_sum(ss, red);
        if (threadIdx.x == 0) a.sumsq_part[blockIdx.x] = ss;
    }
}
// runs after the update kernel on the same stream: advance the device step count unless the update was skipped
__global__ void adamw_advance_kernel(AdamArgs a) {
    float gs; int t;
    if (threadIdx.x == 0 && blockIdx.x == 0 && adam_plan(a, gs, t)) a.step_dev[0] = t;
}

static int adamw_blocks(long n) { long b = (n / 4 + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
extern "C" int mmae_adamw_flat_parts(long n) { return adamw_blocks(n); }

extern "C" int mmae_adamw_flat_ex(float* p, float* g, float* m, float* v, long n, long n_decay, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, float grad_scale, void* w_bf16, const float* grad_sumsq,
                                  float max_norm, float skip_norm, int* step_dev, int clear_grad, float* sumsq_part,
                                  hipStream_t stream) {
    MMAE_REQUIRE(p && g && m && v, MMAE_EINVAL, "adamw_flat: null pointer");
    MMAE_REQUIRE(n >= 0 && n % 4 == 0 && n_decay >= 0 && n_decay <= n, MMAE_ESHAPE, "adamw_flat: n=%ld must be a multiple of 4, n_decay=%ld within it", n, n_decay);
    MMAE_REQUIRE(step_dev || step >= 1, MMAE_EINVAL, "adamw_flat: step counts from 1");
    MMAE_REQUIRE(grad_sumsq || (max_norm <= 0.f && skip_norm <= 0.f), MMAE_EINVAL, "adamw_flat: clipping / skipping needs the gradient sum of squares");
    // a skipped update returns before the loop: its gradients would be neither summed nor cleared
    MMAE_REQUIRE(!(clear_grad || sumsq_part) || !(skip_norm > 0.f), MMAE_EINVAL, "adamw_flat: clear_grad / sumsq_part cannot be combined with skipping");
    if (n == 0) return MMAE_OK;
    AdamArgs a = {p, g, m, v, n / 4, n_decay, lr, beta1, beta2, eps, weight_decay, grad_scale, step, grad_sumsq, max_norm, skip_norm,
                  step_dev, (bf16*)w_bf16, clear_grad ? g : nullptr, sumsq_part};
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(adamw_blocks(n)), dim3(256), 0, stream, a);
    MMAE_CHECK_LAUNCH("adamw_flat");
    if (step_dev) {
        hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(64), 0, stream, a);
        MMAE_CHECK_LAUNCH("adamw_advance");
    }
    return MMAE_OK;
}

extern "C" int mmae_adamw_flat(float* p, const float* g, float* m, float* v, long n, long n_decay, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale, void* w_bf16,
                               const float* grad_sumsq, float max_norm, float skip_norm, int* step_dev, hipStream_t stream) {
    return mmae_adamw_flat_ex(p, (float*)g, m, v, n, n_decay, lr, beta1, beta2, eps, weight_decay, step, grad_scale, w_bf16, grad_sumsq,
                              max_norm, skip_norm, step_dev, 0, nullptr, stream);
}

#define SUMSQ_BLOCKS 1024
__global__ void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)x)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += x[i] * x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sumsq_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
// out = sum of part[0 .. nparts) in a fixed order (the partial sums mmae_adamw_flat_ex leaves, over all of a step's ranges)
extern "C" int mmae_sum_partials(const float* part, int nparts, float* out, hipStream_t stream) {
    MMAE_REQUIRE(part && out && nparts > 0, MMAE_EINVAL, "sum_partials: bad argument");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, part, nparts, out);
    MMAE_CHECK_LAUNCH("sum_partials");
    return MMAE_OK;
}
extern "C" int mmae_sumsq(const float* x, long n, float* out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(x && out && workspace, MMAE_EINVAL, "sumsq: null pointer");
    MMAE_REQUIRE(workspace_bytes >= SUMSQ_BLOCKS * sizeof(float), MMAE_EWORKSPACE, "sumsq: workspace must hold %d floats", SUMSQ_BLOCKS);
    MMAE_REQUIRE(((uintptr_t)x % 16) == 0, MMAE_EINVAL, "sumsq: x must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, x, n, (float*)workspace);
    MMAE_CHECK_LAUNCH("sumsq_partial");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, SUMSQ_BLOCKS, out);
    MMAE_CHECK_LAUNCH("sumsq_final");
    return MMAE_OK;
}

// dst[r][c] = bf16(src[r][c]); dst_t[c][r] = bf16(src[r][c]); 64x64 tiles through LDS
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, int rows, int cols, bf16* __restrict__ dst,
                                                             bf16* __restrict__ dst_t) {
    __shared__ bf16 tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        bf16 v = (bf16)0.f;
        if (r < rows && c < cols) {
            v = (bf16)src[(long)r * cols + c];
            if (dst) dst[(long)r * cols + c] = v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (dst_t) {
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;
            if (r < rows && c < cols) dst_t[(long)c * rows + r] = tile[tx][i];
        }
    }
}
extern "C" int mmae_cast_transpose(const float* src, int rows, int cols, void* dst_bf16, void* dst_t_bf16, hipStream_t stream) {
    MMAE_REQUIRE(src && (dst_bf16 || dst_t_bf16) && rows > 0 && cols > 0, MMAE_EINVAL, "cast_transpose: bad argument");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, stream, src, rows, cols,
                       (bf16*)dst_bf16, (bf16*)dst_t_bf16);
    MMAE_CHECK_LAUNCH("cast_transpose");
    return MMAE_OK;
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)x)[i];
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        ((bf16x4*)y)[i] = o;
    }
    if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) y[i] = (bf16)x[i];
}
extern "C" int mmae_cast_bf16(const float* x, void* y_bf16, long n, hipStream_t stream) {
    MMAE_REQUIRE(x && y_bf16 && n >= 0, MMAE_EINVAL, "cast_bf16: bad argument");
    if (n == 0) return MMAE_OK;
    int blocks = (int)((n / 4 + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, stream, x, (bf16*)y_bf16, n);
    MMAE_CHECK_LAUNCH("cast_bf16");
    return MMAE_OK;
}

// ---------------------------------------------------------------------------------------------
// Transposed bf16 copies of many 2-D weights in one launch.  src and dst are flat bf16 buffers with the same element
// offsets; table[t] = {offset, rows, cols, first tile} describes tensor t ([rows, cols] at src + offset, transposed to
// [cols, rows] at dst + offset); tiles are 64x64 and numbered tensor by tensor.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_multi_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                              const int4* __restrict__ table, int n_tensors) {
    __shared__ bf16 tile[64][66];
    // binary search for the tensor that owns this tile
    int lo = 0, hi = n_tensors - 1;
    const int tid_blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].w <= tid_blk) lo = mid; else hi = mid - 1;
    }
    const int4 d = table[lo];
    const int rows = d.y, cols = d.z;
    const int tiles_c = (cols + 63) / 64;
    const int t = tid_blk - d.w;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const bf16* s = src + d.x;
    bf16* o = dst + d.x;
    if ((rows & 7) == 0 && (cols & 7) == 0) {
        // 16-byte global accesses: a thread moves 8 elements of a row in, 8 elements of a column out (tensor offsets are
        // multiples of 8 elements, so both sides are aligned)
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int rr = i >> 3, ch = i & 7;
            const int r = r0 + rr, c = c0 + ch * 8;
            bf16x8 v;
            if (r < rows && c < cols) v = *(const bf16x8*)(s + (long)r * cols + c);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[rr][ch * 8 + e] = v[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int cc = i >> 3, ch = i & 7;
            const int c = c0 + cc, r = r0 + ch * 8;
            if (c < cols && r < rows) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[ch * 8 + e][cc];
                *(bf16x8*)(o + (long)c * rows + r) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? s[(long)r * cols + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) o[(long)c * rows + r] = tile[tx][i];
    }
}

extern "C" int mmae_transpose_multi(const void* src_bf16, void* dst_bf16, const int* table, int n_tensors, int total_tiles,
                                    hipStream_t stream) {
    MMAE_REQUIRE(src_bf16 && dst_bf16 && table, MMAE_EINVAL, "transpose_multi: null pointer");
    if (n_tensors <= 0 || total_tiles <= 0) return MMAE_OK;
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(total_tiles), dim3(256), 0, stream, (const bf16*)src_bf16, (bf16*)dst_bf16,
                       (const int4*)table, n_tensors);
    MMAE_CHECK_LAUNCH("transpose_multi");
    return MMAE_OK;
}


// A kernel that occupies one workgroup for `us` microseconds and does nothing else (time-bounded: the wave leaves when the
// constant-rate clock says so).  The host uses pairs of these to find out whether two streams share a hardware queue:
// kernels of streams on one queue run one after the other, on different queues side by side.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int mmae_spin_us(int us, hipStream_t stream) {
    MMAE_REQUIRE(us > 0 && us <= 100000, MMAE_EINVAL, "spin_us: %d us is outside (0, 100000]", us);
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        khz = 100000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, (long long)us * khz / 1000);
    MMAE_CHECK_LAUNCH("spin_us");
    return MMAE_OK;
}
