// Random-mask generation, one workgroup of four wavefronts per sample (multimae/multimae.py:218-253 of the reference):
// per-task budgets from the Dirichlet sample, per-task noise argsort, global argsort of (preliminary mask + noise), inverse
// permutation, final mask.
//
// A sequence is sorted in two steps.  Each wavefront sorts its quarter with a bitonic network that never leaves the wave:
// key i of the quarter lives in register i / 64 of lane i % 64, so a compare-exchange at a distance below 64 is a lane
// exchange and one at a larger distance is between two registers of a lane (no LDS traffic of the keys, no barrier).  The four
// sorted runs are then staged in LDS, and every key finds its place in the whole order as its position in its own run plus,
// by binary search, the number of smaller keys in each of the other three: one barrier per sequence instead of one per
// stage of a workgroup-wide network.  Keys are 64-bit composites (fp32 bits of the non-negative key in the high word, index
// in the low word): unique, so the ranks are a permutation, and ties resolve to the lower index (a stable argsort); padding
// entries are all-ones and sort last.
#include "common.h"

#define MASK_MAX_TOTAL 2048
#define MASK_MAX_TASKS 8
#define MASK_WAVES 4

typedef unsigned long long u64;

struct MaskArgs {
    const float* task_dist; const float* noise; const float* noise_all;
    int tokens[MASK_MAX_TASKS];
    int T, B, total, num_encoded;
    int64_t* ids_keep; int64_t* ids_restore; int64_t* mask;
    int* ids_keep32; int* ids_restore32;
};

// compare-exchange between registers j and j + JS of every lane (distance 64 * JS)
template <int E, int JS>
__device__ __forceinline__ void exchange_registers(u64 (&k)[E], int size, int lane) {
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if ((j & JS) == 0 && j + JS < E) {
            const bool up = ((j * 64 + lane) & size) == 0;
            const u64 x = k[j], y = k[j + JS];
            if ((x > y) == up) { k[j] = y; k[j + JS] = x; }
        }
    }
}

// ascending bitonic sort of the 64 * E keys of one wavefront (key i in k[i / 64] of lane i % 64)
template <int E>
__device__ __forceinline__ void wave_bitonic_sort(u64 (&k)[E], int lane) {
    // the stage loops stay rolled: a stage is E independent exchanges already
#pragma nounroll
    for (int size = 2; size <= 64 * E; size <<= 1) {
#pragma nounroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride < 64) {
                // every lane exchange of the stage is requested before the first result is used: one trip through the LDS
                // crossbar per stage instead of one per key
                const bool lower = (lane & stride) == 0;
                const int peer = (lane ^ stride) << 2;
                unsigned other_lo[E], other_hi[E];
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    other_lo[j] = (unsigned)__builtin_amdgcn_ds_bpermute(peer, (int)(unsigned)(k[j] & 0xffffffffu));
                    other_hi[j] = (unsigned)__builtin_amdgcn_ds_bpermute(peer, (int)(unsigned)(k[j] >> 32));
                }
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const bool up = ((j * 64 + lane) & size) == 0;
                    const u64 mine = k[j], other = ((u64)other_hi[j] << 32) | other_lo[j];
                    const u64 lo = mine < other ? mine : other, hi = mine < other ? other : mine;
                    k[j] = (lower == up) ? lo : hi;
                }
            } else {
                switch (stride >> 6) {
                    case 1: exchange_registers<E, 1>(k, size, lane); break;
                    case 2: exchange_registers<E, 2>(k, size, lane); break;
                    default: exchange_registers<E, 4>(k, size, lane); break;
                }
            }
        }
    }
}

// number of keys below `key` in a sorted run whose length is a power of two
__device__ __forceinline__ int count_below(const u64* run, int len, u64 key) {
    int pos = 0;
    for (int step = len >> 1; step > 0; step >>= 1) pos += run[pos + step - 1] < key ? step : 0;
    return pos + (run[pos] < key ? 1 : 0);
}

// Argsort of up to 4 * 64 * E keys by the workgroup: make_key(i) is the composite key of element i (all-ones past the end);
// use(index, rank) is called once for every real element with its original index (the key's low word) and its position in
// the sorted order.  `runs` is the LDS staging area (4 * 64 * E keys); ends with a barrier, so it can be reused at once.
template <int E, typename MakeKey, typename Use>
__device__ __forceinline__ void workgroup_argsort(u64* runs, MakeKey make_key, Use use, int lane, int wave) {
    constexpr int RUN = 64 * E;
    u64 k[E];
#pragma unroll
    for (int j = 0; j < E; ++j) k[j] = make_key(wave * RUN + j * 64 + lane);
    wave_bitonic_sort<E>(k, lane);
#pragma unroll
    for (int j = 0; j < E; ++j) runs[wave * RUN + j * 64 + lane] = k[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (k[j] != ~0ull) {
            int rank = j * 64 + lane;
#pragma unroll
            for (int w = 0; w < MASK_WAVES; ++w)
                if (w != wave) rank += count_below(runs + w * RUN, RUN, k[j]);
            use((int)(k[j] & 0xffffffffu), rank);
        }
    }
    __syncthreads();
}

// registers per lane for a sequence of n keys spread over the four waves: 256 * E >= n, E a power of two
__device__ __forceinline__ int regs_per_lane(int n) {
    int e = 1;
    while (256 * e < n) e <<= 1;
    return e;
}

template <typename MakeKey, typename Use>
__device__ __forceinline__ void argsort_any(int n, u64* runs, MakeKey make_key, Use use, int lane, int wave) {
    switch (regs_per_lane(n)) {  // workgroup-uniform
        case 1: workgroup_argsort<1>(runs, make_key, use, lane, wave); break;   // up to 256 keys (196 tokens of a task)
        case 2: workgroup_argsort<2>(runs, make_key, use, lane, wave); break;   // up to 512 (2 x 196)
        case 4: workgroup_argsort<4>(runs, make_key, use, lane, wave); break;   // up to 1024 (3 x 196; 576 of a task at 384^2)
        default: workgroup_argsort<8>(runs, make_key, use, lane, wave); break;  // up to 2048 (2 x 576)
    }
}

__global__ __launch_bounds__(64 * MASK_WAVES) void mask_generate_kernel(MaskArgs a) {
    __shared__ u64 runs[MASK_MAX_TOTAL];
    __shared__ unsigned char prelim[MASK_MAX_TOTAL];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int off = 0;
    for (int t = 0; t < a.T; ++t) {
        const int n = a.tokens[t];
        // budget of this task: fp32 product, round half to even (multimae.py:218)
        const long long kt = (long long)rintf(a.task_dist[(long)b * a.T + t] * (float)a.num_encoded);
        const float* noise = a.noise + (long)b * a.total + off;
        unsigned char* pre = prelim + off;
        // the reference marks POSITION r of the task's sorted order as kept iff the index found there is below the budget
        // (multimae.py:230-233)
        argsort_any(n, runs,
                    [&](int i) { return i < n ? (((u64)__float_as_uint(noise[i])) << 32) | (unsigned)i : ~0ull; },
                    [&](int idx, int rank) { pre[rank] = (long long)idx < kt ? 0 : 1; }, lane, wave);
        off += n;
    }
    const float* noise_all = a.noise_all + (long)b * a.total;
    argsort_any(a.total, runs,
                [&](int i) {
                    if (i >= a.total) return ~0ull;
                    const float key = (float)prelim[i] + noise_all[i];  // fp32 add as in multimae.py:237
                    return (((u64)__float_as_uint(key)) << 32) | (unsigned)i;
                },
                [&](int idx, int rank) {
                    a.ids_restore[(long)b * a.total + idx] = rank;
                    a.mask[(long)b * a.total + idx] = rank >= a.num_encoded ? 1 : 0;
                    if (a.ids_restore32) a.ids_restore32[(long)b * a.total + idx] = rank;
                    if (rank < a.num_encoded) {
                        a.ids_keep[(long)b * a.num_encoded + rank] = idx;
                        if (a.ids_keep32) a.ids_keep32[(long)b * a.num_encoded + rank] = idx;
                    }
                }, lane, wave);
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
    hipLaunchKernelGGL(mask_generate_kernel, dim3(B), dim3(64 * MASK_WAVES), 0, stream, a);
    MMAE_CHECK_LAUNCH("mask_generate");
    return MMAE_OK;
}
