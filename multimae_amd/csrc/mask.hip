// Random-mask generation for one sample per workgroup (multimae/multimae.py:218-253 of the reference):
// per-task budgets from the Dirichlet sample, per-task noise argsort, global argsort of
// (preliminary mask + noise), inverse permutation, final mask.  All sorts are LDS bitonic sorts of
// 64-bit composite keys (fp32 bits of the non-negative key in the high word, index in the low word),
// which makes them stable (ties resolve to the lower index).
#include "common.h"

#define MASK_MAX_TOTAL 2048
#define MASK_MAX_TASKS 8

struct MaskArgs {
    const float* task_dist; const float* noise; const float* noise_all;
    int tokens[MASK_MAX_TASKS];
    int T, B, total, num_encoded;
    int64_t* ids_keep; int64_t* ids_restore; int64_t* mask;
    int* ids_keep32; int* ids_restore32;
};

__device__ __forceinline__ void bitonic_sort(unsigned long long* a, int n, int tid, int nt) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n >> 1); t += nt) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

__global__ __launch_bounds__(256) void mask_generate_kernel(MaskArgs a) {
    __shared__ unsigned long long keys[MASK_MAX_TOTAL];
    __shared__ unsigned char prelim[MASK_MAX_TOTAL];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const float* noise = a.noise + (long)b * a.total;
    const float* noise_all = a.noise_all + (long)b * a.total;

    int off = 0;
    for (int t = 0; t < a.T; ++t) {
        const int n = a.tokens[t];
        const int np = next_pow2(n);
        // budget of this task: fp32 product, round half to even (multimae.py:218)
        const long long kt = (long long)rintf(a.task_dist[(long)b * a.T + t] * (float)a.num_encoded);
        __syncthreads();
        for (int i = tid; i < np; i += nt)
            keys[i] = i < n ? (((unsigned long long)__float_as_uint(noise[off + i])) << 32) | (unsigned)i : ~0ull;
        bitonic_sort(keys, np, tid, nt);
        // position i of the sorted order holds original index keys[i] & 0xffffffff; the reference marks
        // POSITION i as kept iff that index is below the budget (multimae.py:230-233)
        for (int i = tid; i < n; i += nt) prelim[off + i] = ((long long)(unsigned)(keys[i] & 0xffffffffu) < kt) ? 0 : 1;
        off += n;
    }
    __syncthreads();
    const int np = next_pow2(a.total);
    for (int i = tid; i < np; i += nt) {
        if (i < a.total) {
            const float key = (float)prelim[i] + noise_all[i];  // fp32 add as in multimae.py:237
            keys[i] = (((unsigned long long)__float_as_uint(key)) << 32) | (unsigned)i;
        } else {
            keys[i] = ~0ull;
        }
    }
    bitonic_sort(keys, np, tid, nt);
    for (int rank = tid; rank < a.total; rank += nt) {
        const int idx = (int)(keys[rank] & 0xffffffffu);
        a.ids_restore[(long)b * a.total + idx] = rank;
        a.mask[(long)b * a.total + idx] = rank >= a.num_encoded ? 1 : 0;
        if (a.ids_restore32) a.ids_restore32[(long)b * a.total + idx] = rank;
        if (rank < a.num_encoded) {
            a.ids_keep[(long)b * a.num_encoded + rank] = idx;
            if (a.ids_keep32) a.ids_keep32[(long)b * a.num_encoded + rank] = idx;
        }
    }
}

extern "C" int mmae_mask_generate(const float* task_dist, const float* noise, const float* noise_all,
                                  const int* tokens_per_task_host, int T, int B, int num_encoded, int64_t* ids_keep,
                                  int64_t* ids_restore, int64_t* mask, int* ids_keep_i32, int* ids_restore_i32,
                                  hipStream_t stream) {
    MMAE_REQUIRE(task_dist && noise && noise_all && tokens_per_task_host && ids_restore && mask && (ids_keep || num_encoded == 0),
                 MMAE_EINVAL, "mask_generate: null pointer");
    MMAE_REQUIRE(T >= 1 && T <= MASK_MAX_TASKS, MMAE_ESHAPE, "mask_generate: T=%d not in [1, %d]", T, MASK_MAX_TASKS);
    if (B == 0) return MMAE_OK;
    MMAE_REQUIRE(B > 0, MMAE_EINVAL, "mask_generate: negative batch");
    MaskArgs a = {};
    int total = 0;
    for (int t = 0; t < T; ++t) {
        MMAE_REQUIRE(tokens_per_task_host[t] > 0, MMAE_ESHAPE, "mask_generate: task %d has no tokens", t);
        a.tokens[t] = tokens_per_task_host[t];
        total += tokens_per_task_host[t];
    }
    MMAE_REQUIRE(total <= MASK_MAX_TOTAL, MMAE_ESHAPE, "mask_generate: %d tokens > %d", total, MASK_MAX_TOTAL);
    MMAE_REQUIRE(num_encoded >= 0 && num_encoded <= total, MMAE_ESHAPE, "mask_generate: num_encoded=%d outside [0, %d]", num_encoded, total);
    a.task_dist = task_dist; a.noise = noise; a.noise_all = noise_all; a.T = T; a.B = B; a.total = total;
    a.num_encoded = num_encoded; a.ids_keep = ids_keep; a.ids_restore = ids_restore; a.mask = mask;
    a.ids_keep32 = ids_keep_i32; a.ids_restore32 = ids_restore_i32;
    hipLaunchKernelGGL(mask_generate_kernel, dim3(B), dim3(256), 0, stream, a);
    MMAE_CHECK_LAUNCH("mask_generate");
    return MMAE_OK;
}
