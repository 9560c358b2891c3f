// Truncated depth standardisation (run_pretraining_multimae.py:448-459 of the reference): per sample, sort the n depth
// values, keep ranks [int(0.1 n), int(0.9 n)), and normalise the whole map by the mean and the unbiased variance of that
// slice.  A full sort is not needed: two order statistics (radix select over the monotone integer image of the floats,
// 8 bits per round, both ranks in the same sweeps) bound the slice, and the slice's moments follow from the values
// strictly between them plus the right number of copies of the two boundary values.  One block per sample; the map
// (n * 4 bytes, 196 KiB at 224^2) is re-read from L2 for each of the 7 sweeps.
#include "common.h"
#include <stdlib.h>

#define DS_THREADS 1024

__device__ __forceinline__ unsigned ds_key(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // increasing in x; +NaN last, like a sort
}

__device__ __forceinline__ float ds_value(unsigned k) {  // inverse of ds_key
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// The bin of a 256-bin histogram that holds rank `rank` (counting from 0) and the rank inside that bin, by one wavefront:
// lane l sums bins 4l .. 4l + 3, an inclusive scan over the lanes finds the lane, its four bins the bin.  (One thread
// walking the bins is 255 dependent LDS reads per round.)  Every lane returns the same result.
__device__ __forceinline__ int ds_select_bin(const unsigned* h, long rank, long& rank_in_bin) {
    const int lane = threadIdx.x & 63;
    const unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
    const unsigned own = c0 + c1 + c2 + c3;
    unsigned incl = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const unsigned excl = incl - own;
    const bool mine = (long)excl <= rank && rank < (long)incl;
    unsigned long long who = __ballot(mine);
    // rank beyond the total (cannot happen for ranks < n): the last bin, as the serial walk would end there
    const int src = who ? __ffsll((long long)who) - 1 : 63;
    long r = rank - (long)excl;
    int b = 4 * lane;
    if (r >= (long)c0) { r -= c0; ++b; if (r >= (long)c1) { r -= c1; ++b; if (r >= (long)c2) { r -= c2; ++b; } } }
    if (!who) { b = 255; r = rank; }
    const int bin = __shfl(b, src, 64);
    const int rlo = __shfl((int)(r & 0xffffffff), src, 64), rhi = __shfl((int)(r >> 32), src, 64);
    rank_in_bin = ((long)rhi << 32) | (unsigned)rlo;
    return bin;
}

struct DsArgs { const float* x; float* out; long n; long k_lo, k_hi; float eps; };

__global__ __launch_bounds__(DS_THREADS) void depth_standardize_kernel(DsArgs a) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned sel_prefix[2];
    __shared__ long sel_rank[2];
    __shared__ float red[16];
    __shared__ long lred[3][16];
    const int tid = threadIdx.x;
    const float* x = a.x + (long)blockIdx.x * a.n;
    float* out = a.out + (long)blockIdx.x * a.n;

    // ---- radix select of the keys at ranks k_lo and k_hi - 1 ----
    if (tid == 0) { sel_prefix[0] = sel_prefix[1] = 0; sel_rank[0] = a.k_lo; sel_rank[1] = a.k_hi - 1; }
    unsigned mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 512) hist[tid >> 8][tid & 255] = 0;
        __syncthreads();
        const unsigned p0 = sel_prefix[0], p1 = sel_prefix[1];
        const bool shared_prefix = p0 == p1;  // both ranks still in the same bin: one histogram serves both
        for (long i = tid; i < a.n; i += DS_THREADS) {
            const unsigned k = ds_key(x[i]);
            const unsigned digit = (k >> shift) & 255;
            if ((k & mask) == p0) atomicAdd(&hist[0][digit], 1u);
            if (!shared_prefix && (k & mask) == p1) atomicAdd(&hist[1][digit], 1u);
        }
        __syncthreads();
        if (tid < 128) {  // wavefront 0 places the lower rank, wavefront 1 the upper one
            const int which = tid >> 6;
            long rank_in_bin;
            const int b = ds_select_bin(hist[shared_prefix ? 0 : which], sel_rank[which], rank_in_bin);
            if ((tid & 63) == 0) {
                sel_rank[which] = rank_in_bin;
                sel_prefix[which] |= (unsigned)b << shift;
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned key_lo = sel_prefix[0], key_hi = sel_prefix[1];

    // ---- sums over the slice: values strictly inside (key_lo, key_hi) plus copies of the boundary values ----
    long c_lt_lo = 0, c_le_lo = 0, c_lt_hi = 0;
    float s = 0.f;
    for (long i = tid; i < a.n; i += DS_THREADS) {
        const float v = x[i];
        const unsigned k = ds_key(v);
        c_lt_lo += k < key_lo; c_le_lo += k <= key_lo; c_lt_hi += k < key_hi;
        if (k > key_lo && k < key_hi) s += v;
    }
    // block reductions (fixed order)
    auto lsum = [&](long v, int slot) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0) lred[slot][tid >> 6] = v;
    };
    lsum(c_lt_lo, 0); lsum(c_le_lo, 1); lsum(c_lt_hi, 2);
    s = block_sum(s, red);
    long n_lt_lo = 0, n_le_lo = 0, n_lt_hi = 0;
    for (int w = 0; w < DS_THREADS / 64; ++w) { n_lt_lo += lred[0][w]; n_le_lo += lred[1][w]; n_lt_hi += lred[2][w]; }
    const float b_lo = ds_value(key_lo), b_hi = ds_value(key_hi);
    const long n_trim = a.k_hi - a.k_lo;
    // copies of the boundary values inside ranks [k_lo, k_hi)
    const long cp_lo = (n_le_lo < a.k_hi ? n_le_lo : a.k_hi) - a.k_lo;
    const long cp_hi = key_hi > key_lo ? a.k_hi - n_lt_hi : 0;
    const float mean = (s + (float)cp_lo * b_lo + (float)cp_hi * b_hi) / (float)n_trim;

    float q = 0.f;
    for (long i = tid; i < a.n; i += DS_THREADS) {
        const float v = x[i];
        const unsigned k = ds_key(v);
        if (k > key_lo && k < key_hi) { const float d = v - mean; q += d * d; }
    }
    q = block_sum(q, red);
    q += (float)cp_lo * (b_lo - mean) * (b_lo - mean) + (float)cp_hi * (b_hi - mean) * (b_hi - mean);
    const float var = q / (float)(n_trim - 1);  // unbiased, as Tensor.var
    const float rstd = 1.0f / sqrtf(var + a.eps);
    for (long i = tid; i < a.n; i += DS_THREADS) out[i] = (x[i] - mean) * rstd;
}

// The same selection with the sample held in registers (n <= DS_REG_KEYS * DS_THREADS values: 224^2 maps are 49 per
// thread): one read of the map, every radix round and both moment sums from registers, one write (256 maps of 224^2:
// 77 us, of which 47 are the LDS atomics of the histograms, against 217 us for the memory form above, which re-reads the
// map seven times with one dependent load per loop trip).
#define DS_REG_KEYS 64  // most keys per thread; the kernel is instantiated for a few sizes so that the usual maps do not spill
template <int KEYS>
__global__ __launch_bounds__(DS_THREADS) void depth_standardize_regs_kernel(DsArgs a) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned sel_prefix[2];
    __shared__ long sel_rank[2];
    __shared__ float red[16];
    __shared__ long lred[3][16];
    const int tid = threadIdx.x;
    const float* x = a.x + (long)blockIdx.x * a.n;
    float* out = a.out + (long)blockIdx.x * a.n;
    const int kpt = (int)((a.n + DS_THREADS - 1) / DS_THREADS);  // keys per thread (block-uniform), <= KEYS
    // element j * DS_THREADS + tid exists for j < kpt - 1, and for j == kpt - 1 where tail_ok
    const int last = kpt - 1;
    const bool tail_ok = (long)last * DS_THREADS + tid < a.n;
    unsigned key[KEYS];  // the sortable image of each value (ds_value gives the value back): one register per element
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
        key[j] = ds_key((j < last || (j == last && tail_ok)) ? x[j * DS_THREADS + tid] : 0.f);
    }
    if (tid == 0) { sel_prefix[0] = sel_prefix[1] = 0; sel_rank[0] = a.k_lo; sel_rank[1] = a.k_hi - 1; }
    unsigned mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 512) hist[tid >> 8][tid & 255] = 0;
        __syncthreads();
        const unsigned p0 = sel_prefix[0], p1 = sel_prefix[1];
        const bool shared_prefix = p0 == p1;
#pragma unroll
        for (int j = 0; j < KEYS; ++j) {
            if (j < last || (j == last && tail_ok)) {
                // plain LDS atomics: settling the lanes of a wavefront that share a bin by ballots first (the leading
                // digits of a depth map sit in two or three bins) measured slower than letting the LDS unit take them
                const unsigned k = key[j];
                const unsigned digit = (k >> shift) & 255;
                if ((k & mask) == p0) atomicAdd(&hist[0][digit], 1u);
                if (!shared_prefix && (k & mask) == p1) atomicAdd(&hist[1][digit], 1u);
            }
        }
        __syncthreads();
        if (tid < 128) {  // wavefront 0 places the lower rank, wavefront 1 the upper one
            const int which = tid >> 6;
            long rank_in_bin;
            const int b = ds_select_bin(hist[shared_prefix ? 0 : which], sel_rank[which], rank_in_bin);
            if ((tid & 63) == 0) {
                sel_rank[which] = rank_in_bin;
                sel_prefix[which] |= (unsigned)b << shift;
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned key_lo = sel_prefix[0], key_hi = sel_prefix[1];
    long c_le_lo = 0, c_lt_hi = 0;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
        if (j < last || (j == last && tail_ok)) {
            const unsigned k = key[j];
            c_le_lo += k <= key_lo; c_lt_hi += k < key_hi;
            if (k > key_lo && k < key_hi) s += ds_value(k);
        }
    }
    auto lsum = [&](long val, int slot) {
        for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
        if ((tid & 63) == 0) lred[slot][tid >> 6] = val;
    };
    lsum(c_le_lo, 1); lsum(c_lt_hi, 2);
    s = block_sum(s, red);
    long n_le_lo = 0, n_lt_hi = 0;
    for (int w = 0; w < DS_THREADS / 64; ++w) { n_le_lo += lred[1][w]; n_lt_hi += lred[2][w]; }
    const float b_lo = ds_value(key_lo), b_hi = ds_value(key_hi);
    const long n_trim = a.k_hi - a.k_lo;
    const long cp_lo = (n_le_lo < a.k_hi ? n_le_lo : a.k_hi) - a.k_lo;
    const long cp_hi = key_hi > key_lo ? a.k_hi - n_lt_hi : 0;
    const float mean = (s + (float)cp_lo * b_lo + (float)cp_hi * b_hi) / (float)n_trim;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
        if (j < last || (j == last && tail_ok)) {
            const unsigned k = key[j];
            if (k > key_lo && k < key_hi) { const float d = ds_value(k) - mean; q += d * d; }
        }
    }
    q = block_sum(q, red);
    q += (float)cp_lo * (b_lo - mean) * (b_lo - mean) + (float)cp_hi * (b_hi - mean) * (b_hi - mean);
    const float var = q / (float)(n_trim - 1);
    const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
        if (j < last || (j == last && tail_ok)) out[j * DS_THREADS + tid] = (ds_value(key[j]) - mean) * rstd;
    }
}

extern "C" int mmae_standardize_depth(const float* depth, float* out, int B, long n, double lo_frac, double hi_frac, float eps,
                                      hipStream_t stream) {
    MMAE_REQUIRE(depth && out, MMAE_EINVAL, "standardize_depth: null pointer");
    MMAE_REQUIRE(B >= 0 && n >= 1 && lo_frac >= 0.0 && hi_frac <= 1.0 && lo_frac < hi_frac, MMAE_EINVAL, "standardize_depth: bad argument");
    const long k_lo = (long)(lo_frac * (double)n), k_hi = (long)(hi_frac * (double)n);  // int(0.1 * n), int(0.9 * n)
    MMAE_REQUIRE(k_hi - k_lo >= 2, MMAE_ESHAPE, "standardize_depth: the kept slice [%ld, %ld) needs at least two values", k_lo, k_hi);
    if (B == 0) return MMAE_OK;
    DsArgs a = {depth, out, n, k_lo, k_hi, eps};
    const long kpt = (n + DS_THREADS - 1) / DS_THREADS;
    if (kpt <= 16)
        hipLaunchKernelGGL(depth_standardize_regs_kernel<16>, dim3(B), dim3(DS_THREADS), 0, stream, a);
    else if (kpt <= 32)
        hipLaunchKernelGGL(depth_standardize_regs_kernel<32>, dim3(B), dim3(DS_THREADS), 0, stream, a);
    else if (kpt <= 49)  // 224 x 224
        hipLaunchKernelGGL(depth_standardize_regs_kernel<49>, dim3(B), dim3(DS_THREADS), 0, stream, a);
    else if (kpt <= DS_REG_KEYS)
        hipLaunchKernelGGL(depth_standardize_regs_kernel<DS_REG_KEYS>, dim3(B), dim3(DS_THREADS), 0, stream, a);
    else
        hipLaunchKernelGGL(depth_standardize_kernel, dim3(B), dim3(DS_THREADS), 0, stream, a);
    MMAE_CHECK_LAUNCH("depth_standardize");
    return MMAE_OK;
}
