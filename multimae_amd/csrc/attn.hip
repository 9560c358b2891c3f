// Fused attention for short sequences (whole K/V -- or Q/dO -- of one (batch, head) resident in LDS), gfx950.
//
// All MFMAs are v_mfma_f32_32x32x16_bf16 with the KEY (forward, dQ kernel) or the QUERY (dK/dV kernel) index of the
// score tile on the accumulator rows, so that a score tile, converted to bf16 in registers, is directly the B operand
// of the products that reduce over that index (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS): no LDS
// round trip for P / dS.  The A operands of those products are columns of the row-major LDS images, fetched with the
// transposing LDS read (ds_read_b64_tr_b16): one image serves both the row reads and the column reads.
//
// Accumulator layout of a 32x32 tile: column = lane & 31, row(i) = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).
// Registers 8s..8s+7 of a tile, packed to bf16, are k-step s of the next product, whose k slot j of lane half h is
// tile row 16s + 8(j>>2) + 4h + (j&3); the A operand is fetched in that k order.
#include "common.h"

#define LOG2E 1.4426950408889634f

struct AttnArgs {
    const bf16* q; long ldq;
    const bf16* k; long ldk;
    const bf16* v; long ldv;
    bf16* o; long ldo;
    float* lse;
    // backward only
    const bf16* dO; long lddo;
    bf16* dq; long lddq;
    bf16* dk; long lddk;
    bf16* dv; long lddv;
    float* delta;  // [B,H,q_len] workspace: rowsum(dO * O), written by the dQ kernel, read by the dK/dV kernel
    const float* mean_key;  // [B,H,D] or null: the vector dQ's residual row sum is taken out along, when the keys of this call
                            // are only a chunk of the row (the mean over all of the row's keys; null: the mean of the call's keys)
    int B, H, q_len, kv_len;
    float scale;
};

__device__ __forceinline__ int acc_row(int i, int h2) { return (i & 3) + 8 * (i >> 2) + 4 * h2; }

__device__ __forceinline__ bf16x8 pack8(const f32x16& t, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)t[8 * s + j];
    return r;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 a;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
    return a;
}

// A-operand fragment X^T[dd][n] from the row-major image X[n][ldr]: lane (r = lane & 31 -> column dd0 + r, h2 = lane >> 5)
// gets rows n0 + 4*h2 + {0..3} and n0 + 8 + 4*h2 + {0..3}.  Each 16-lane group reads a 4-row x 16-column block; lane 4q+p
// of the group supplies the address of row q, columns 4p..4p+3 and receives column (lane & 15) of the four rows.
// Needs all 64 lanes active.
__device__ __forceinline__ bf16x8 tr_frag(const bf16* X, int ldr, int n0, int dd0, int lane) {
    const int i = lane & 15, h2 = lane >> 5, rhalf = (lane >> 4) & 1;
    const bf16* p = X + (n0 + 4 * h2 + (i >> 2)) * ldr + dd0 + 16 * rhalf + 4 * (i & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 8 * ldr));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Two [len, D] slices that share the row count (K and V, or Q and dO; row strides lda / ldb) into row-major LDS images
// [n_pad][D + 8] with a zero-filled tail: U chunks of each requested before the first
// LDS store and no branch between the requests (rows past the end are fetched from the last row and zeroed afterwards): one
// trip to memory per U * nthreads chunks instead of one per chunk and slice
template <int D, int U>
__device__ __forceinline__ void stage_rows2(const bf16* a, long lda, const bf16* b, long ldb, int n_valid, int n_pad, bf16* img_a, bf16* img_b,
                                            int tid, int nthreads) {
    constexpr int CH = D / 8, LDR = D + 8;
    for (int base = tid; base < n_pad * CH; base += U * nthreads) {
        bf16x8 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * nthreads, n = idx / CH, c = idx - n * CH;
            const long nc = n < n_valid ? n : n_valid - 1;
            va[u] = *(const bf16x8*)(a + nc * lda + c * 8);
            vb[u] = *(const bf16x8*)(b + nc * ldb + c * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * nthreads, n = idx / CH, c = idx - n * CH;
            if (idx < n_pad * CH) {
                const bool ok = n < n_valid;
#pragma unroll
                for (int e = 0; e < 8; ++e) { va[u][e] = ok ? va[u][e] : (bf16)0.f; vb[u][e] = ok ? vb[u][e] : (bf16)0.f; }
                *(bf16x8*)(img_a + n * LDR + c * 8) = va[u];
                *(bf16x8*)(img_b + n * LDR + c * 8) = vb[u];
            }
        }
    }
}

// U chosen so that a round does not fetch more than it stores: two chunks per thread and slice where that covers the image
template <int D>
__device__ __forceinline__ void stage_pair(const bf16* a, long lda, const bf16* b, long ldb, int n_valid, int n_pad, bf16* img_a, bf16* img_b,
                                           int tid, int nthreads) {
    if (n_pad * (D / 8) <= 2 * nthreads) stage_rows2<D, 2>(a, lda, b, ldb, n_valid, n_pad, img_a, img_b, tid, nthreads);
    else stage_rows2<D, 4>(a, lda, b, ldb, n_valid, n_pad, img_a, img_b, tid, nthreads);
}

// row fragment (A or B operand with the row index on the lane) of a row-major image
template <int D>
__device__ __forceinline__ bf16x8 row_frag(const bf16* X, int row, int ks, int h2) {
    return *(const bf16x8*)(X + row * (D + 8) + ks * 16 + h2 * 8);
}

// The same two images without the 8 padding columns: row stride D, the 16-byte chunk c of row n stored at chunk
// c ^ swz<D>(n), which spreads the rows a fragment read touches over the banks as the padding does.  Used where the padding
// alone decides whether two workgroups share a CU (289 keys at head_dim 64: 2 x 320 x 72 x 2 = 92 KiB padded, 75 KiB here).
template <int D>
__device__ __forceinline__ int swz(int row) { return D == 64 ? (row & 7) : ((row >> 1) & 3); }

template <int D>
__device__ __forceinline__ bf16x8 row_frag_sw(const bf16* X, int row, int ks, int h2) {
    return *(const bf16x8*)(X + row * D + (((ks * 2 + h2) ^ swz<D>(row)) << 3));
}

template <int D>
__device__ __forceinline__ bf16x8 tr_frag_sw(const bf16* X, int n0, int dd0, int lane) {
    const int i = lane & 15, h2 = lane >> 5, rhalf = (lane >> 4) & 1;
    const int row = n0 + 4 * h2 + (i >> 2), col = dd0 + 16 * rhalf + 4 * (i & 3);
    // rows n and n + 8 share the swizzle (8 is a multiple of its period)
    const bf16* p = X + row * D + (((col >> 3) ^ swz<D>(row)) << 3) + (col & 7);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 8 * D));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// K rows [0, kv_len) and V rows [0, v_rows) (zero from kv_len on) into swizzled images; U chunks of each in flight
template <int D, int U>
__device__ __forceinline__ void stage_rows2_sw(const bf16* a, long lda, const bf16* b, long ldb, int n_valid, int v_rows, bf16* img_a, bf16* img_b,
                                               int tid, int nthreads) {
    constexpr int CH = D / 8;
    for (int base = tid; base < v_rows * CH; base += U * nthreads) {
        bf16x8 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * nthreads, n = idx / CH, c = idx - n * CH;
            const long nc = n < n_valid ? n : n_valid - 1;
            va[u] = *(const bf16x8*)(a + nc * lda + c * 8);
            vb[u] = *(const bf16x8*)(b + nc * ldb + c * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * nthreads, n = idx / CH, c = idx - n * CH;
            if (idx < v_rows * CH) {
                const bool ok = n < n_valid;
                const int off = n * D + ((c ^ swz<D>(n)) << 3);
#pragma unroll
                for (int e = 0; e < 8; ++e) vb[u][e] = ok ? vb[u][e] : (bf16)0.f;
                if (ok) *(bf16x8*)(img_a + off) = va[u];
                *(bf16x8*)(img_b + off) = vb[u];
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// forward: two passes over the key tiles per 32-query block (row maximum first, then exp / P V), so only one score
// tile is live at a time; the second Q K^T costs D/16 MFMAs per tile.
// --------------------------------------------------------------------------------------------
#define ONE_TAU 8.0f  // see attn_fwd_kernel<..., ONE>

template <int D, bool SWZ = false>
__device__ __forceinline__ f32x16 score_tile(const bf16* Ks, int kt, int r, int h2, const bf16x8* qf) {
    f32x16 a = zero16();
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SWZ ? row_frag_sw<D>(Ks, kt * 32 + r, ks, h2) : row_frag<D>(Ks, kt * 32 + r, ks, h2), qf[ks], a,
                                                    0, 0, 0);
    return a;
}

// SWZ: unpadded swizzled images, K rows [0, kv_len) followed by V rows [0, v_rows) where v_rows covers the rows the last
// tile's P V steps read (16 or 32 past the last full tile; zero from kv_len on).  The last key tile's score MFMAs then read
// K rows past kv_len out of the V image (the launcher sizes the allocation so that they stay inside it): those score rows
// are replaced by -inf / 0 through selects before any arithmetic, as the padded layout's zero rows are.
template <int D, int NW, bool SWZ, int WPE, bool ONE>
__global__ __launch_bounds__(NW * 64, WPE) void attn_fwd_kernel(AttnArgs p, int nkp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDR = SWZ ? D : D + 8, KS = D / 16, DT = D / 32;
    bf16* Ks = (bf16*)smem;                                  // [nkp][LDR]   (SWZ: [kv_len][D])
    bf16* Vs = Ks + (SWZ ? p.kv_len : nkp) * LDR;            // [nkp][LDR]   (SWZ: [v_rows][D])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;

    // every kernel argument in registers before the first branch (otherwise the pointers are fetched behind it: a second
    // scalar round trip before the first load of a workgroup that lives about 15 us)
    asm volatile("" ::"s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.lse), "s"(p.ldq), "s"(p.ldk), "s"(p.ldv), "s"(p.ldo));
    const int nkt = nkp / 32;
    const int tail = p.kv_len - (nkt - 1) * 32;  // valid keys in the last tile (1..32)
    if (SWZ)
        stage_rows2_sw<D, 4>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len,
                             nkp - (tail <= 16 ? 16 : 0), Ks, Vs, tid, NW * 64);
    else
        stage_pair<D>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len, nkp, Ks, Vs,
                      tid, NW * 64);
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nqt = (p.q_len + 31) / 32;
    for (int qb = wave; qb < nqt; qb += NW) {
        const int qrow = qb * 32 + r;
        const int qc = qrow < p.q_len ? qrow : p.q_len - 1;
        const bf16* qp = p.q + ((long)b * p.q_len + qc) * p.ldq + h * D + h2 * 8;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);

        float m = -INFINITY;
        if (!ONE) {
            for (int kt = 0; kt < nkt; ++kt) {
                const f32x16 a = score_tile<D, SWZ>(Ks, kt, r, h2, qf);
                // only the last key tile has padded keys to mask (workgroup-uniform test; the vector ALU bounds these kernels)
                if (kt == nkt - 1 && tail < 32) {
                    // accumulator registers 4g..4g+3 hold key rows 8g..8g+7 of the tile: groups past the last key are skipped
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (8 * g < tail) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) m = fmaxf(m, acc_row(4 * g + e, h2) < tail ? a[4 * g + e] : -INFINITY);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = fmaxf(m, a[i]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
        }
        float mc = -m * c;

        float l = 0.f;
        f32x16 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt] = zero16();
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 a = score_tile<D, SWZ>(Ks, kt, r, h2, qf);
            if (ONE) {
                // one pass: the reference point m of the exponentials follows the row maximum lazily: it moves (and the sums
                // so far are rescaled) only when a tile's maximum exceeds it by more than 2^ONE_TAU after scaling, so that the
                // exponentials stay below 2^ONE_TAU; after the first tiles that is rare and the rescale is skipped wave-wide
                float mt = -INFINITY;
                if (kt == nkt - 1 && tail < 32) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (8 * g < tail) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) mt = fmaxf(mt, acc_row(4 * g + e, h2) < tail ? a[4 * g + e] : -INFINITY);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) mt = fmaxf(mt, a[i]);
                }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const bool move = (mt - m) * c > ONE_TAU;  // true on the first tile (m = -inf)
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const float mn = move ? mt : m;
                    const float corr = __builtin_amdgcn_exp2f((m - mn) * c);  // 1 where m stays, 0 on the first tile
                    m = mn;
                    mc = -mn * c;
                    l *= corr;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) oacc[dt][i] *= corr;
                }
            }
            const bool short_tail = kt == nkt - 1 && tail <= 16;  // the second k-step of P V would multiply zeros
            if (kt == nkt - 1 && tail < 32) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (8 * g < tail) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const int i = 4 * g + e4;
                            const float e = acc_row(i, h2) < tail ? __builtin_amdgcn_exp2f(fmaf(a[i], c, mc)) : 0.f;
                            a[i] = e;
                            l += e;
                        }
                    } else {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) a[4 * g + e4] = 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(a[i], c, mc));
                    a[i] = e;
                    l += e;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 == 1 && short_tail) continue;
                const bf16x8 pf = pack8(a, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SWZ ? tr_frag_sw<D>(Vs, kt * 32 + 16 * s2, dt * 32, lane) : tr_frag(Vs, LDR, kt * 32 + 16 * s2, dt * 32, lane), pf,
                                                                       oacc[dt], 0, 0, 0);
            }
        }
        l += __shfl_xor(l, 32, 64);
        if (h2 == 0 && qrow < p.q_len) p.lse[((long)b * p.H + h) * p.q_len + qrow] = m * p.scale + logf(l);
        const float inv = 1.0f / l;
        if (qrow < p.q_len) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16* op = p.o + ((long)b * p.q_len + qrow) * p.ldo + h * D + dt * 32 + 4 * h2;
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    bf16x4 ov = {(bf16)(oacc[dt][4 * i4] * inv), (bf16)(oacc[dt][4 * i4 + 1] * inv),
                                 (bf16)(oacc[dt][4 * i4 + 2] * inv), (bf16)(oacc[dt][4 * i4 + 3] * inv)};
                    *(bf16x4*)(op + 8 * i4) = ov;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// forward for key sequences whose whole K / V image would take more than half of the LDS (one workgroup per CU, a handful
// of waves to hide every latency behind): the keys stream through a ring of two 64-key tiles instead.  A workgroup owns
// NWQ consecutive 32-query blocks of one (batch, head), one per wave; all waves walk the key tiles together: the next
// tile's K / V chunks are requested into registers before the current tile is computed and stored to the other ring slot
// after it, one barrier per tile.  One pass with a running maximum: per 64-key tile the two score tiles, their maximum,
// one correction of the output accumulators (exp2 of the maximum's change), then exp / P V as in the resident kernel.
// LDS: 2 slots x (K tile + V tile) x 64 rows x (D + 8) bf16 = 36 KiB at head_dim 64, so four workgroups share a CU.
// --------------------------------------------------------------------------------------------
template <int D, int NWQ>
__global__ __launch_bounds__(NWQ * 64, 4) void attn_fwd_stream_kernel(AttnArgs p, int qblocks_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDR = D + 8, KS = D / 16, DT = D / 32, CH = D / 8, NT = NWQ * 64;
    constexpr int TILE = 64 * LDR;               // elements of one K (or V) tile image
    constexpr int PER = (64 * CH + NT - 1) / NT;  // 16-byte chunks per thread, operand and tile
    bf16* ring = (bf16*)smem;                    // [2 slots][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int qb = blockIdx.y * qblocks_per_wg + wave;   // this wave's 32-query block
    const int nqt = (p.q_len + 31) / 32, nt = (p.kv_len + 63) / 64;
    const bool active = wave < qblocks_per_wg && qb < nqt;  // idle waves still stage and meet the barriers
    asm volatile("" ::"s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.lse), "s"(p.ldq), "s"(p.ldk), "s"(p.ldv), "s"(p.ldo));
    const bf16* kbase = p.k + (long)b * p.kv_len * p.ldk + h * D;
    const bf16* vbase = p.v + (long)b * p.kv_len * p.ldv + h * D;

    bf16x8 kreg[PER], vreg[PER];
    auto fetch = [&](int t) {  // tile t -> registers (rows past the end: the last row, zeroed when stored)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * NT, n = idx / CH, c = idx - n * CH;
            const int key = t * 64 + n;
            const long kc = key < p.kv_len ? key : p.kv_len - 1;
            if (idx < 64 * CH) {
                kreg[u] = *(const bf16x8*)(kbase + kc * p.ldk + c * 8);
                vreg[u] = *(const bf16x8*)(vbase + kc * p.ldv + c * 8);
            }
        }
    };
    auto store = [&](int t) {
        bf16* Kt = ring + (t & 1) * 2 * TILE;
        bf16* Vt = Kt + TILE;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * NT, n = idx / CH, c = idx - n * CH;
            if (idx < 64 * CH) {
                const bool ok = t * 64 + n < p.kv_len;
                bf16x8 kv = kreg[u], vv = vreg[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) { kv[e] = ok ? kv[e] : (bf16)0.f; vv[e] = ok ? vv[e] : (bf16)0.f; }
                *(bf16x8*)(Kt + n * LDR + c * 8) = kv;
                *(bf16x8*)(Vt + n * LDR + c * 8) = vv;
            }
        }
    };
    fetch(0);
    // the wave's query fragments, requested beside the first tile
    const int qrow = qb * 32 + r;
    const int qc = qrow < p.q_len ? qrow : p.q_len - 1;
    bf16x8 qf[KS];
    {
        const bf16* qp = p.q + ((long)b * p.q_len + (active ? qc : 0)) * p.ldq + h * D + h2 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    store(0);
    __syncthreads();

    const float c = p.scale * LOG2E;
    float m = -INFINITY, l = 0.f;
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = zero16();
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) fetch(t + 1);
        const bf16* Kt = ring + (t & 1) * 2 * TILE;
        const bf16* Vt = Kt + TILE;
        if (active) {
            const int valid = p.kv_len - t * 64;          // keys of this tile (1..64 for the last, >= 64 before)
            const bool two = valid > 32;                 // the second 32-key half holds keys at all
            f32x16 a0 = score_tile<D>(Kt, 0, r, h2, qf), a1 = zero16();
            if (two) a1 = score_tile<D>(Kt, 1, r, h2, qf);
            // maximum over the tile's valid keys
            float mt = -INFINITY;
            const int v0 = valid < 32 ? valid : 32, v1 = valid - 32 < 32 ? valid - 32 : 32;
            if (v0 < 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) mt = fmaxf(mt, acc_row(i, h2) < v0 ? a0[i] : -INFINITY);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) mt = fmaxf(mt, a0[i]);
            }
            if (two) {
                if (v1 < 32) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) mt = fmaxf(mt, acc_row(i, h2) < v1 ? a1[i] : -INFINITY);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) mt = fmaxf(mt, a1[i]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float mn = fmaxf(m, mt);
            const float corr = __builtin_amdgcn_exp2f((m - mn) * c);  // 0 on the first tile (m = -inf)
            m = mn;
            const float mc = -mn * c;
            l *= corr;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[dt][i] *= corr;
            // exp / P V, half by half
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && !two) continue;
                f32x16& a = half == 0 ? a0 : a1;
                const int vh = half == 0 ? v0 : v1;
                if (vh < 32) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float e = acc_row(i, h2) < vh ? __builtin_amdgcn_exp2f(fmaf(a[i], c, mc)) : 0.f;
                        a[i] = e;
                        l += e;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(a[i], c, mc));
                        a[i] = e;
                        l += e;
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (s2 == 1 && vh <= 16) continue;  // the second k-step would multiply zeros
                    const bf16x8 pf = pack8(a, s2);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt, LDR, half * 32 + 16 * s2, dt * 32, lane), pf, oacc[dt], 0, 0, 0);
                }
            }
        }
        if (t + 1 < nt) store(t + 1);  // the other slot: every wave left it at the previous barrier
        __syncthreads();
    }
    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    if (h2 == 0 && qrow < p.q_len) p.lse[((long)b * p.H + h) * p.q_len + qrow] = m * p.scale + logf(l);
    const float inv = 1.0f / l;
    if (qrow < p.q_len) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16* op = p.o + ((long)b * p.q_len + qrow) * p.ldo + h * D + dt * 32 + 4 * h2;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                bf16x4 ov = {(bf16)(oacc[dt][4 * i4] * inv), (bf16)(oacc[dt][4 * i4 + 1] * inv),
                             (bf16)(oacc[dt][4 * i4 + 2] * inv), (bf16)(oacc[dt][4 * i4 + 3] * inv)};
                *(bf16x4*)(op + 8 * i4) = ov;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// backward, kernel 1: dQ per 32-query block.  K and V of the head live in LDS; Q, dO, O rows come straight from
// global memory as fragments.  P is recomputed from the forward's log-sum-exp; delta = rowsum(dO * O) is computed from
// the fragments and also written out for kernel 2.
//
// In exact arithmetic every row of dS sums to zero, so dQ = dS K = dS (K - 1 kbar^T) for any vector kbar.  In bf16 it
// does not (delta comes from the rounded O, dS is rounded again for the MFMA), and the residual row sum times the
// component all keys share lands in dQ: with near-uniform attention over similar keys that was a 15 % error.  The kernel
// therefore accumulates rs = rowsum(bf16(dS)) and returns dS K - rs kbar^T with kbar = mean key, i.e. the product with
// the centred keys, which brings dQ to the level of the other bf16 roundings (0.5 %).
// --------------------------------------------------------------------------------------------
// SWZ: unpadded swizzled images (see attn_fwd_kernel), V rows [0, kv_len) first, then K rows [0, nkp) with zeros from kv_len on:
// the last key tile's dP MFMAs read V rows past kv_len out of the K image (finite values, multiplied by probabilities that are
// exactly 0), K itself keeps its zero rows because the dS K product runs over all of the tile's rows.
template <int D, int NW, bool SWZ, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void attn_bwd_dq_kernel(AttnArgs p, int nkp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDR = SWZ ? D : D + 8, KS = D / 16, DT = D / 32;
    constexpr bool RS_DOT2 = D == 32;  // as in attn_bwd_fused_kernel
    bf16* Ks = SWZ ? (bf16*)smem + p.kv_len * D : (bf16*)smem;
    bf16* Vs = SWZ ? (bf16*)smem : Ks + nkp * LDR;
    float* kbar = (float*)((bf16*)smem + (SWZ ? p.kv_len + nkp : 2 * nkp) * LDR);  // [D] mean key
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;

    if (SWZ)
        stage_rows2_sw<D, 4>(p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.kv_len, nkp, Vs,
                             Ks, tid, NW * 64);
    else
        stage_pair<D>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len, nkp, Ks, Vs,
                      tid, NW * 64);
    __syncthreads();
    if (p.mean_key) {
        if (tid < D) kbar[tid] = p.mean_key[((long)b * p.H + h) * D + tid];
    } else {
        // mean key: every wave sums the rows n = wave * RL + (lane / CH) (mod NW * RL) of its lanes' 8-column chunks, folds the
        // RL row lanes together, and the NW per-wave sums are added in wave order (one thread walking the 289 rows of a ViT-L
        // head kept the other waves at the barrier for a twentieth of the workgroup's life: dQ + dK / dV 715 -> 680 us)
        constexpr int CH = D / 8, RL = 64 / CH;
        float* part = kbar + D;  // [NW][D]
        float acc8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc8[e] = 0.f;
        const int ch = lane % CH;
        for (int n = wave * RL + lane / CH; n < p.kv_len; n += NW * RL) {
            const bf16x8 kv = *(const bf16x8*)(Ks + (SWZ ? n * D + ((ch ^ swz<D>(n)) << 3) : n * LDR + ch * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc8[e] += (float)kv[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int off = CH; off < 64; off <<= 1) acc8[e] += __shfl_xor(acc8[e], off, 64);
        if (lane < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) part[wave * D + ch * 8 + e] = acc8[e];
        }
        __syncthreads();
        if (tid < D) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += part[w * D + tid];
            kbar[tid] = sum / (float)p.kv_len;
        }
    }
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nqt = (p.q_len + 31) / 32, nkt = nkp / 32;
    const int tail = p.kv_len - (nkt - 1) * 32;
    for (int qb = wave; qb < nqt; qb += NW) {
        const int qrow = qb * 32 + r;
        const bool qvalid = qrow < p.q_len;
        const int qc = qvalid ? qrow : p.q_len - 1;
        const long rowoff = (long)b * p.q_len + qc;
        bf16x8 qf[KS], dof[KS];
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *(const bf16x8*)(p.q + rowoff * p.ldq + h * D + h2 * 8 + ks * 16);
            dof[ks] = *(const bf16x8*)(p.dO + rowoff * p.lddo + h * D + h2 * 8 + ks * 16);
            const bf16x8 of = *(const bf16x8*)(p.o + rowoff * p.ldo + h * D + h2 * 8 + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) dpart += (float)dof[ks][e] * (float)of[e];
        }
        const float my_delta = dpart + __shfl_xor(dpart, 32, 64);
        // padded query rows: lse = +big makes every recomputed probability 0
        const float my_lse = qvalid ? p.lse[((long)b * p.H + h) * p.q_len + qrow] * LOG2E : 1e30f;
        if (h2 == 0 && qvalid) p.delta[((long)b * p.H + h) * p.q_len + qrow] = my_delta;

        f32x16 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();
        float rs = 0.f;  // this lane's share of rowsum(bf16(dS)) of query r
        const float nds = -my_delta * p.scale;  // dS = P * fma(dP, scale, nds)
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? row_frag_sw<D>(Ks, kt * 32 + r, ks, h2) : row_frag<D>(Ks, kt * 32 + r, ks, h2)), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? row_frag_sw<D>(Vs, kt * 32 + r, ks, h2) : row_frag<D>(Vs, kt * 32 + r, ks, h2)), dof[ks], dp, 0, 0, 0);
            }
            // only the last key tile has padded keys to mask (workgroup-uniform test; the vector ALU bounds these kernels)
            if (kt == nkt - 1 && tail < 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pr = acc_row(i, h2) < tail ? __builtin_amdgcn_exp2f(s[i] * c - my_lse) : 0.f;
                    s[i] = pr * fmaf(dp[i], p.scale, nds);
                    if (!RS_DOT2) rs += (float)(bf16)s[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[i] = __builtin_amdgcn_exp2f(s[i] * c - my_lse) * fmaf(dp[i], p.scale, nds);
                    if (!RS_DOT2) rs += (float)(bf16)s[i];
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 dsf = pack8(s, s2);
                if (RS_DOT2) rs = add_bf16x8(rs, dsf);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? tr_frag_sw<D>(Ks, kt * 32 + 16 * s2, dt * 32, lane) : tr_frag(Ks, LDR, kt * 32 + 16 * s2, dt * 32, lane)), dsf, dq[dt], 0, 0, 0);
            }
        }
        rs += __shfl_xor(rs, 32, 64);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[dt][i] -= rs * kbar[dt * 32 + acc_row(i, h2)];
        if (qvalid) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16* op = p.dq + ((long)b * p.q_len + qrow) * p.lddq + h * D + dt * 32 + 4 * h2;
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    bf16x4 ov = {(bf16)dq[dt][4 * i4], (bf16)dq[dt][4 * i4 + 1], (bf16)dq[dt][4 * i4 + 2], (bf16)dq[dt][4 * i4 + 3]};
                    *(bf16x4*)(op + 8 * i4) = ov;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// backward, kernel 2: dK, dV per 32-key tile.  Q and dO of the head live in LDS (+ lse, delta per query); the K and V
// rows of the tile come from global memory as fragments; accumulation over the query blocks stays in registers.
// --------------------------------------------------------------------------------------------
// SWZ: unpadded swizzled images, dO rows [0, q_len) first, then Q rows [0, nqp) with zeros from q_len on: padded queries have
// zero scores and lse = +big, hence probabilities and dS that are exactly 0, and the dO rows read for them out of the Q image
// are finite.
template <int D, int NW, bool SWZ, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void attn_bwd_dkv_kernel(AttnArgs p, int nqp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDR = SWZ ? D : D + 8, KS = D / 16, DT = D / 32;
    bf16* Qs = SWZ ? (bf16*)smem + p.q_len * D : (bf16*)smem;   // [nqp][LDR]
    bf16* dOs = SWZ ? (bf16*)smem : Qs + nqp * LDR;             // [nqp][LDR]   (SWZ: [q_len][D])
    float* lse2 = (float*)((bf16*)smem + (SWZ ? p.q_len + nqp : 2 * nqp) * LDR);   // [nqp]
    float* delta = lse2 + nqp;                 // [nqp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;

    if (SWZ)
        stage_rows2_sw<D, 4>(p.dO + (long)b * p.q_len * p.lddo + h * D, p.lddo, p.q + (long)b * p.q_len * p.ldq + h * D, p.ldq, p.q_len, nqp, dOs,
                             Qs, tid, NW * 64);
    else
        stage_pair<D>(p.q + (long)b * p.q_len * p.ldq + h * D, p.ldq, p.dO + (long)b * p.q_len * p.lddo + h * D, p.lddo, p.q_len, nqp, Qs, dOs,
                      tid, NW * 64);
    for (int n = tid; n < nqp; n += NW * 64) {
        const bool ok = n < p.q_len;
        lse2[n] = ok ? p.lse[((long)b * p.H + h) * p.q_len + n] * LOG2E : 1e30f;
        delta[n] = ok ? p.delta[((long)b * p.H + h) * p.q_len + n] : 0.f;
    }
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nqt = nqp / 32, nkt = (p.kv_len + 31) / 32;
    for (int kt = wave; kt < nkt; kt += NW) {
        const int key = kt * 32 + r;
        const bool key_valid = key < p.kv_len;
        const long krow = (long)b * p.kv_len + (key_valid ? key : p.kv_len - 1);
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *(const bf16x8*)(p.k + krow * p.ldk + h * D + h2 * 8 + ks * 16);
            vf[ks] = *(const bf16x8*)(p.v + krow * p.ldv + h * D + h2 * 8 + ks * 16);
        }
        f32x16 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }
        for (int qb = 0; qb < nqt; ++qb) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? row_frag_sw<D>(Qs, qb * 32 + r, ks, h2) : row_frag<D>(Qs, qb * 32 + r, ks, h2)), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? row_frag_sw<D>(dOs, qb * 32 + r, ks, h2) : row_frag<D>(dOs, qb * 32 + r, ks, h2)), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 lv = *(const f32x4*)(lse2 + qb * 32 + 8 * i4 + 4 * h2);
                const f32x4 dl = *(const f32x4*)(delta + qb * 32 + 8 * i4 + 4 * h2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * i4 + e;
                    const float pr = __builtin_amdgcn_exp2f(s[i] * c - lv[e]);  // padded keys: columns that are never stored
                    s[i] = pr;
                    dp[i] = pr * (dp[i] - dl[e]) * p.scale;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(s, s2), dsf = pack8(dp, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? tr_frag_sw<D>(dOs, qb * 32 + 16 * s2, dt * 32, lane) : tr_frag(dOs, LDR, qb * 32 + 16 * s2, dt * 32, lane)), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((SWZ ? tr_frag_sw<D>(Qs, qb * 32 + 16 * s2, dt * 32, lane) : tr_frag(Qs, LDR, qb * 32 + 16 * s2, dt * 32, lane)), dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        if (key_valid) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16* kp = p.dk + ((long)b * p.kv_len + key) * p.lddk + h * D + dt * 32 + 4 * h2;
                bf16* vp = p.dv + ((long)b * p.kv_len + key) * p.lddv + h * D + dt * 32 + 4 * h2;
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    bf16x4 a = {(bf16)dk[dt][4 * i4], (bf16)dk[dt][4 * i4 + 1], (bf16)dk[dt][4 * i4 + 2], (bf16)dk[dt][4 * i4 + 3]};
                    bf16x4 c2 = {(bf16)dv[dt][4 * i4], (bf16)dv[dt][4 * i4 + 1], (bf16)dv[dt][4 * i4 + 2], (bf16)dv[dt][4 * i4 + 3]};
                    *(bf16x4*)(kp + 8 * i4) = a;
                    *(bf16x4*)(vp + 8 * i4) = c2;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// backward, both kernels in one for sequences short enough that Q, dO, K and V of a head fit LDS together: every operand
// is read from HBM once (the two-kernel form reads Q, dO, K, V twice and passes delta through a workspace), which is what
// bounds attention at these lengths (99-token encoder: 15 GFLOP against 0.3 GB per layer).  After staging (delta =
// rowsum(dO * O) is formed while dO is staged and stays in LDS) the block's work items are the 32-query dQ tiles followed
// by the 32-key dK / dV tiles, dealt round-robin to the waves: same arithmetic as the two kernels above, every fragment
// from LDS.
// --------------------------------------------------------------------------------------------
#ifdef MMAE_ATTN_TRACE
// timing build only: per workgroup (first 1024) and wave, 100 MHz timestamps: 0 start, 1 staged, 2 items done,
// 3 loads issued, 4 loads landed and stored, 5 past the first barrier
__device__ unsigned long long g_attn_trace[1024 * 8 * 8];
extern "C" int mmae_debug_attn_trace(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_trace), sizeof(g_attn_trace)) == hipSuccess ? 0 : 1;
}
#define ATTN_TRACE(slot) do { if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_attn_trace[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define ATTN_TRACE(slot) do {} while (0)
#endif
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 4) void attn_bwd_fused_kernel(AttnArgs p, int nqp, int nkp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDR = D + 8, KS = D / 16, DT = D / 32, CH = D / 8;
    // row sum of bf16(dS): at head_dim 32 from the packed fragments with dot2 instructions (dec self 104 -> 98 us); at 64 the
    // per-score widen-and-add stays (the dot2 form spills there: 124 -> 126 us)
    constexpr bool RS_DOT2 = D == 32;
    bf16* Qs = (bf16*)smem;                    // [nqp][LDR]
    bf16* dOs = Qs + nqp * LDR;                // [nqp][LDR]
    bf16* Ks = dOs + nqp * LDR;                // [nkp][LDR]
    bf16* Vs = Ks + nkp * LDR;                 // [nkp][LDR]
    float* lse2 = (float*)(Vs + nkp * LDR);    // [nqp]
    float* delta = lse2 + nqp;                 // [nqp]
    float* kbar = delta + nqp;                 // [D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    ATTN_TRACE(0);

    // Staging: every 16-byte chunk of Q, dO, O, K and V that this thread moves in a round (two per operand: one round for
    // up to 128 rows at head_dim 64 with 512 threads) is requested before the first one is stored to LDS, so a round costs
    // one trip to memory instead of one per operand and loop iteration (ten dependent trips at the pretraining shapes,
    // about half of the workgroup's lifetime).  CH adjacent lanes hold one row (nqp * CH and nkp * CH are multiples of 64,
    // so whole waves run each round); delta = rowsum(dO * O) is formed on the way; the column sums of K (for kbar) are
    // taken from the chunks in registers: a thread's chunk column is the same in every round.
    {
        constexpr int U = 2, NT = NW * 64;
        static_assert(NT % CH == 0, "a thread keeps its chunk column across rounds");
        const bf16* qg = p.q + (long)b * p.q_len * p.ldq + h * D;
        const bf16* kg = p.k + (long)b * p.kv_len * p.ldk + h * D;
        const bf16* vg = p.v + (long)b * p.kv_len * p.ldv + h * D;
        const bf16* dog = p.dO + (long)b * p.q_len * p.lddo + h * D;
        const bf16* og = p.o + (long)b * p.q_len * p.ldo + h * D;
        const int ch = tid % CH;
        float ksum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ksum[e] = 0.f;
        const int nmax = nqp > nkp ? nqp : nkp;
        for (int base = tid; base < nmax * CH; base += U * NT) {
            bf16x8 vq[U], vdo[U], vo[U], vk[U], vv[U];
            float vl[U];
            // rows past the end are fetched from the last row and zeroed afterwards: no branch between the loads, so all ten
            // are in flight together (a branch join waits for every load issued before it)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = (base + u * NT) / CH;
                const long nq_c = n < p.q_len ? n : p.q_len - 1, nk_c = n < p.kv_len ? n : p.kv_len - 1;
                vk[u] = *(const bf16x8*)(kg + nk_c * p.ldk + ch * 8);
                vv[u] = *(const bf16x8*)(vg + nk_c * p.ldv + ch * 8);
                vq[u] = *(const bf16x8*)(qg + nq_c * p.ldq + ch * 8);
                vdo[u] = *(const bf16x8*)(dog + nq_c * p.lddo + ch * 8);
                vo[u] = *(const bf16x8*)(og + nq_c * p.ldo + ch * 8);
                vl[u] = p.lse[((long)b * p.H + h) * p.q_len + nq_c];  // with the operands: fetched after them it is a second trip
            }
            ATTN_TRACE(3);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = (base + u * NT) / CH;
                const bool qv = n < p.q_len, kv = n < p.kv_len;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    vk[u][e] = kv ? vk[u][e] : (bf16)0.f; vv[u][e] = kv ? vv[u][e] : (bf16)0.f;
                    vq[u][e] = qv ? vq[u][e] : (bf16)0.f; vdo[u][e] = qv ? vdo[u][e] : (bf16)0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = (base + u * NT) / CH;
                if (n < nkp) {
                    *(bf16x8*)(Ks + n * LDR + ch * 8) = vk[u];
                    *(bf16x8*)(Vs + n * LDR + ch * 8) = vv[u];
#pragma unroll
                    for (int e = 0; e < 8; ++e) ksum[e] += (float)vk[u][e];
                }
                if (n < nqp) {  // wave-uniform: n differs by at most 64 / CH rows inside a wave and nqp is a multiple of 32 >= that
                    *(bf16x8*)(Qs + n * LDR + ch * 8) = vq[u];
                    *(bf16x8*)(dOs + n * LDR + ch * 8) = vdo[u];
                    float part = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) part += (float)vdo[u][e] * (float)vo[u][e];
#pragma unroll
                    for (int m = 1; m < CH; m <<= 1) part += __shfl_xor(part, m, 64);
                    if (ch == 0) {
                        delta[n] = -part * p.scale;  // kept negated and scaled: dS = P * fma(dP, scale, delta), one instruction fewer per score
                        lse2[n] = n < p.q_len ? vl[u] * LOG2E : 1e30f;  // padded rows: every probability 0
                    }
                }
            }
        }
        ATTN_TRACE(4);
        // kbar: lanes with the same chunk column (lane % CH) across the wave, then across the waves through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int m = CH; m < 64; m <<= 1) ksum[e] += __shfl_xor(ksum[e], m, 64);
        }
        float* kpart = (float*)(kbar + D);  // [NW][D] partial sums
        if (lane < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) kpart[wave * D + lane * 8 + e] = ksum[e];
        }
        __syncthreads();
        ATTN_TRACE(5);
        if (tid < D) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += kpart[w * D + tid];
            kbar[tid] = p.mean_key ? p.mean_key[((long)b * p.H + h) * D + tid] : sum / (float)p.kv_len;
        }
        __syncthreads();
    }

    ATTN_TRACE(1);
    const float c = p.scale * LOG2E;
    const int nqt = nqp / 32, nkt = nkp / 32;
    const int tail = p.kv_len - (nkt - 1) * 32;

    for (int item = wave; item < nqt + nkt; item += NW) {
        // lane-derived indices are rebuilt from an opaque copy per item: carried across the items they are spilled
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int r = ln & 31, h2 = ln >> 5, lane = ln;
        if (item < nqt) {
            // ---- dQ of a 32-query block (see attn_bwd_dq_kernel) ----
            const int qrow = item * 32 + r;
            bf16x8 qf[KS], dof[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qf[ks] = row_frag<D>(Qs, qrow, ks, h2);
                dof[ks] = row_frag<D>(dOs, qrow, ks, h2);
            }
            const float my_delta = delta[qrow], my_lse = lse2[qrow];
            f32x16 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();
            float rs = 0.f;
            for (int kt = 0; kt < nkt; ++kt) {
                f32x16 s = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, kt * 32 + r, ks, h2), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Vs, kt * 32 + r, ks, h2), dof[ks], dp, 0, 0, 0);
                }
                // the vector ALU bounds this kernel: only the last key tile has padded keys to mask (workgroup-uniform test)
                // (skipping padded groups and k-steps pays at head_dim 64 only: 124 -> 122 us there, 66 -> 70 us at 32 with 99 keys)
                const bool short_tail = D == 64 && kt == nkt - 1 && tail <= 16;  // the second k-step of dQ would multiply zeros
                if (kt == nkt - 1 && tail < 32) {
                    // accumulator registers 4g..4g+3 hold key rows 8g..8g+7 of the tile: groups past the last key are skipped
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (D != 64 || 8 * g < tail) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int i = 4 * g + e;
                                const float pr = acc_row(i, h2) < tail ? __builtin_amdgcn_exp2f(s[i] * c - my_lse) : 0.f;
                                s[i] = pr * fmaf(dp[i], p.scale, my_delta);
                                if (!RS_DOT2) rs += (float)(bf16)s[i];
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[4 * g + e] = 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[i] = __builtin_amdgcn_exp2f(s[i] * c - my_lse) * fmaf(dp[i], p.scale, my_delta);
                        if (!RS_DOT2) rs += (float)(bf16)s[i];
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (s2 == 1 && short_tail) continue;
                    const bf16x8 dsf = pack8(s, s2);
                    if (RS_DOT2) rs = add_bf16x8(rs, dsf);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ks, LDR, kt * 32 + 16 * s2, dt * 32, lane), dsf, dq[dt], 0, 0, 0);
                }
            }
            rs += __shfl_xor(rs, 32, 64);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) dq[dt][i] -= rs * kbar[dt * 32 + acc_row(i, h2)];
            if (qrow < p.q_len) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16* op = p.dq + ((long)b * p.q_len + qrow) * p.lddq + h * D + dt * 32 + 4 * h2;
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        bf16x4 ov = {(bf16)dq[dt][4 * i4], (bf16)dq[dt][4 * i4 + 1], (bf16)dq[dt][4 * i4 + 2], (bf16)dq[dt][4 * i4 + 3]};
                        *(bf16x4*)(op + 8 * i4) = ov;
                    }
                }
            }
        } else {
            // ---- dK, dV of a 32-key tile (see attn_bwd_dkv_kernel) ----
            const int key = (item - nqt) * 32 + r;
            const bool key_valid = key < p.kv_len;
            // the tile's own K / V fragments are re-read from LDS per query block (opaque row index, so that the reads
            // are not hoisted): holding them would take 2 * D / 2 more registers than four waves per SIMD leave
            f32x16 dk[DT], dv[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }
            for (int qb = 0; qb < nqt; ++qb) {
                int krow = key;
                asm volatile("" : "+v"(krow));
                f32x16 s = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Qs, qb * 32 + r, ks, h2), row_frag<D>(Ks, krow, ks, h2), s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(dOs, qb * 32 + r, ks, h2), row_frag<D>(Vs, krow, ks, h2), dp, 0, 0, 0);
                }
                const int qleft = p.q_len - qb * 32;  // queries of this block; registers 4g..4g+3 hold query rows 8g..8g+7
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    if (D == 64 && 8 * i4 >= qleft) {  // padded queries only (wave-uniform): their probabilities are zero
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[4 * i4 + e] = dp[4 * i4 + e] = 0.f;
                        continue;
                    }
                    const f32x4 lv = *(const f32x4*)(lse2 + qb * 32 + 8 * i4 + 4 * h2);
                    const f32x4 dl = *(const f32x4*)(delta + qb * 32 + 8 * i4 + 4 * h2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * i4 + e;
                        // no mask for padded keys: a key is a column of these tiles (one lane), its probabilities feed only
                        // its own dK / dV rows, and those are not stored (K and V rows past the end are zero: finite values)
                        const float pr = __builtin_amdgcn_exp2f(s[i] * c - lv[e]);
                        s[i] = pr;
                        dp[i] = pr * fmaf(dp[i], p.scale, dl[e]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (D == 64 && s2 == 1 && qleft <= 16) continue;  // the second k-step would multiply zeros
                    const bf16x8 pf = pack8(s, s2), dsf = pack8(dp, s2);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(dOs, LDR, qb * 32 + 16 * s2, dt * 32, lane), pf, dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, LDR, qb * 32 + 16 * s2, dt * 32, lane), dsf, dk[dt], 0, 0, 0);
                    }
                }
            }
            if (key_valid) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16* kp = p.dk + ((long)b * p.kv_len + key) * p.lddk + h * D + dt * 32 + 4 * h2;
                    bf16* vp = p.dv + ((long)b * p.kv_len + key) * p.lddv + h * D + dt * 32 + 4 * h2;
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        bf16x4 a = {(bf16)dk[dt][4 * i4], (bf16)dk[dt][4 * i4 + 1], (bf16)dk[dt][4 * i4 + 2], (bf16)dk[dt][4 * i4 + 3]};
                        bf16x4 c2 = {(bf16)dv[dt][4 * i4], (bf16)dv[dt][4 * i4 + 1], (bf16)dv[dt][4 * i4 + 2], (bf16)dv[dt][4 * i4 + 3]};
                        *(bf16x4*)(kp + 8 * i4) = a;
                        *(bf16x4*)(vp + 8 * i4) = c2;
                    }
                }
            }
        }
    }
    ATTN_TRACE(2);
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
#define ATTN_MAX_LDS (160 * 1024)

template <typename K>
static int launch_with_lds(K kern, const char* name, size_t bytes, int blocks, int threads, hipStream_t s, const AttnArgs& a, int npad,
                           bool& reserved) {
    MMAE_REQUIRE(bytes <= ATTN_MAX_LDS, MMAE_ESHAPE, "%s: sequence needs %zu bytes of LDS (> 160 KiB)", name, bytes);
    if (bytes > 65536 && !reserved) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_MAX_LDS);
        if (e != hipSuccess) {
            mmae_set_error("%s: cannot reserve LDS: %s", name, hipGetErrorString(e));
            return MMAE_ELAUNCH;
        }
        reserved = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), bytes, s, a, npad);
    MMAE_CHECK_LAUNCH(name);
    return MMAE_OK;
}

static inline int pad32(int n) { return (n + 31) / 32 * 32; }

// Passes over the keys in the resident forward: 0 or 1 = one pass with a lazily moved reference point (measured against two
// passes: 236 -> 207 us at 289 keys / head_dim 64 with swizzled images, 179 -> 173 us at 576 / 32, 37.0 -> 34.8 us at 196 / 32,
// 37.3 -> 34.0 us at 99 / 64 with 4 waves once the kernel is bounded to 128 registers: unbounded the compiler spread that
// instantiation over 168 and it lost), 2 = two passes, row maximum first
static thread_local int g_fwd_one_pass = 0;
extern "C" void mmae_attn_fwd_set_one_pass(int v) { g_fwd_one_pass = v; }
template <int D, int NW>
static int launch_fwd(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    const int nkp = pad32(a.kv_len);
    if (g_fwd_one_pass != 2) {
        static bool reserved1 = false;
        return launch_with_lds(attn_fwd_kernel<D, NW, false, 4, true>, "attn_fwd", sizeof(bf16) * 2 * (size_t)nkp * (D + 8), a.B * a.H, NW * 64, s, a,
                               nkp, reserved1);
    }
    return launch_with_lds(attn_fwd_kernel<D, NW, false, 1, false>, "attn_fwd", sizeof(bf16) * 2 * (size_t)nkp * (D + 8), a.B * a.H, NW * 64, s, a,
                           nkp, reserved);
}
// bytes of the unpadded swizzled images: K rows [0, kv_len), V rows up to the last row a P V step reads, and at least the
// nkp rows the score MFMAs of the last key tile address from the start of the K image
static size_t fwd_swizzled_bytes(int kv_len, int D) {
    const int nkp = pad32(kv_len), tail = kv_len - (nkp - 32);
    const int v_rows = nkp - (tail <= 16 ? 16 : 0);
    const int rows = kv_len + v_rows > nkp ? kv_len + v_rows : nkp;
    return sizeof(bf16) * (size_t)rows * D;
}
// two workgroups per CU: NW * 2 / 4 waves per SIMD
template <int D, int NW>
static int launch_fwd_sw(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    if (g_fwd_one_pass != 2) {
        static bool reserved1 = false;
        return launch_with_lds(attn_fwd_kernel<D, NW, true, NW * 2 / 4, true>, "attn_fwd", fwd_swizzled_bytes(a.kv_len, D), a.B * a.H, NW * 64, s, a,
                               pad32(a.kv_len), reserved1);
    }
    return launch_with_lds(attn_fwd_kernel<D, NW, true, NW * 2 / 4, false>, "attn_fwd", fwd_swizzled_bytes(a.kv_len, D), a.B * a.H, NW * 64, s, a,
                           pad32(a.kv_len), reserved);
}
// streaming forward: NWQ query blocks per workgroup, chosen so that the blocks of a head divide evenly over its workgroups
template <int D, int NWQ>
static int launch_fwd_stream_n(const AttnArgs& a, int per_wg, hipStream_t s) {
    const int nqt = (a.q_len + 31) / 32;
    const size_t bytes = sizeof(bf16) * 2 * 2 * 64 * (size_t)(D + 8);
    hipLaunchKernelGGL((attn_fwd_stream_kernel<D, NWQ>), dim3(a.B * a.H, (nqt + per_wg - 1) / per_wg), dim3(NWQ * 64), bytes, s, a, per_wg);
    MMAE_CHECK_LAUNCH("attn_fwd_stream");
    return MMAE_OK;
}
template <int D>
static int launch_fwd_stream(const AttnArgs& a, hipStream_t s) {
    const int nqt = (a.q_len + 31) / 32;
    // 4 to 6 query blocks per workgroup with the fewest idle waves (10 blocks: 2 x 5; 18: 3 x 6; 7: 2 x 4)
    int best = 4, best_idle = 1 << 30;
    for (int n = 6; n >= 4; --n) {
        const int idle = (nqt + n - 1) / n * n - nqt;
        if (idle < best_idle) { best = n; best_idle = idle; }
    }
    if (best == 4) return launch_fwd_stream_n<D, 4>(a, 4, s);
    if (best == 5) return launch_fwd_stream_n<D, 5>(a, 5, s);
    return launch_fwd_stream_n<D, 6>(a, 6, s);
}
template <int D, int NW>
static int launch_dq(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    const int nkp = pad32(a.kv_len);
    return launch_with_lds(attn_bwd_dq_kernel<D, NW, false, 1>, "attn_bwd_dq", sizeof(bf16) * 2 * (size_t)nkp * (D + 8) + sizeof(float) * D * (1 + NW), a.B * a.H, NW * 64, s, a, nkp, reserved);
}
template <int D, int NW>
static int launch_dkv(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    const int nqp = pad32(a.q_len);
    return launch_with_lds(attn_bwd_dkv_kernel<D, NW, false, 1>, "attn_bwd_dkv", sizeof(bf16) * 2 * (size_t)nqp * (D + 8) + sizeof(float) * 2 * nqp,
                           a.B * a.H, NW * 64, s, a, nqp, reserved);
}

// swizzled images, two workgroups per CU
static size_t dq_swizzled_bytes(int kv_len, int D) { return sizeof(bf16) * (size_t)(kv_len + pad32(kv_len)) * D + sizeof(float) * D * (1 + 8); }  // 8 waves
static size_t dkv_swizzled_bytes(int q_len, int D) { return sizeof(bf16) * (size_t)(q_len + pad32(q_len)) * D + sizeof(float) * 2 * pad32(q_len); }
template <int D, int NW>
static int launch_dq_sw(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    return launch_with_lds(attn_bwd_dq_kernel<D, NW, true, NW * 2 / 4>, "attn_bwd_dq", dq_swizzled_bytes(a.kv_len, D), a.B * a.H, NW * 64, s, a,
                           pad32(a.kv_len), reserved);
}
template <int D, int NW>
static int launch_dkv_sw(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    return launch_with_lds(attn_bwd_dkv_kernel<D, NW, true, NW * 2 / 4>, "attn_bwd_dkv", dkv_swizzled_bytes(a.q_len, D), a.B * a.H, NW * 64, s, a,
                           pad32(a.q_len), reserved);
}

// Waves per workgroup for `tiles` 32-row work items (query tiles of the forward and dQ kernels, key tiles of the dK / dV
// kernel), dealt round-robin to the waves: 4 waves for up to four items, else 8.  One exception, measured: at head_dim 64
// with LDS images too large for a second workgroup on the CU (ViT-L/16 at 384^2: 289 encoder tokens = 10 items) the
// workgroup's waves are all the CU has, and 10 or 12 waves that finish in fewer rounds pay (forward 293 -> 265 us, dK / dV
// 388 -> 347 us, dQ unchanged).  At head_dim 32 (576 decoder queries = 18 items) 9 waves in two rounds were slower than
// 8 in three (forward 223 -> 249 us): those kernels are bound by what the CU's vector ALU and LDS deliver in total, not by
// the rounds.
static int pick_waves(int tiles, size_t lds_bytes, int head_dim) {
    if (tiles <= 4) return 4;
    if (head_dim != 64 || lds_bytes <= (size_t)80 * 1024) return 8;
    int best = 8, best_rounds = (tiles + 7) / 8, best_idle = best_rounds * 8 - tiles;
    const int cand[2] = {10, 12};
    for (int nw : cand) {
        const int rounds = (tiles + nw - 1) / nw, idle = rounds * nw - tiles;
        if (rounds < best_rounds || (rounds == best_rounds && idle < best_idle)) { best = nw; best_rounds = rounds; best_idle = idle; }
    }
    return best;
}
#define ATTN_BY_WAVES(launch, D, nw, a, stream)                                                    \
    ((nw) == 4 ? launch<D, 4>(a, stream) : (nw) == 8 ? launch<D, 8>(a, stream) : (nw) == 10 ? launch<D, 10>(a, stream) : launch<D, 12>(a, stream))

static thread_local int g_fwd_variant = 0;  // 0 = by size, 1 = K / V of the head resident in LDS (padded rows), 2 = keys streamed in 64-key tiles, 3 = resident, swizzled rows
extern "C" void mmae_attn_fwd_set_variant(int v) { g_fwd_variant = v; }
static thread_local int g_bwd_variant = 0;  // 0 = by size, 1 = two kernels (padded images), 2 = one kernel whenever the head fits LDS, 3 = two kernels, swizzled images wherever they fit
extern "C" void mmae_attn_bwd_set_variant(int v) { g_bwd_variant = v; }

static size_t fused_lds_bytes(const AttnArgs& a, int D) {
    // Q, dO, K, V images; lse and delta per query; the mean key and its per-wave partial sums (8 waves)
    return sizeof(bf16) * 2 * (size_t)(pad32(a.q_len) + pad32(a.kv_len)) * (D + 8) + sizeof(float) * (2 * pad32(a.q_len) + D + 8 * D);
}
template <int D, int NW>
static int launch_fused(const AttnArgs& a, hipStream_t s) {
    static bool reserved = false;
    const size_t bytes = fused_lds_bytes(a, D);
    if (bytes > 65536 && !reserved) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<D, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_MAX_LDS);
        MMAE_REQUIRE(e == hipSuccess, MMAE_ELAUNCH, "attn_bwd: cannot reserve LDS: %s", hipGetErrorString(e));
        reserved = true;
    }
    hipLaunchKernelGGL((attn_bwd_fused_kernel<D, NW>), dim3(a.B * a.H), dim3(NW * 64), bytes, s, a, pad32(a.q_len), pad32(a.kv_len));
    MMAE_CHECK_LAUNCH("attn_bwd_fused");
    return MMAE_OK;
}

static int check_common(const AttnArgs& a, int head_dim, const char* who) {
    MMAE_REQUIRE(a.q && a.k && a.v && a.o && a.lse, MMAE_EINVAL, "%s: null pointer", who);
    MMAE_REQUIRE(a.B > 0 && a.H > 0 && a.q_len > 0 && a.kv_len > 0, MMAE_EINVAL, "%s: empty problem", who);
    MMAE_REQUIRE(head_dim == 32 || head_dim == 64, MMAE_ESHAPE, "%s: head_dim %d not in {32, 64}", who, head_dim);
    MMAE_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0, MMAE_ESHAPE,
                 "%s: row strides must be multiples of 8 elements", who);
    MMAE_REQUIRE(((uintptr_t)a.q % 16 == 0) && ((uintptr_t)a.k % 16 == 0) && ((uintptr_t)a.v % 16 == 0) && ((uintptr_t)a.o % 16 == 0),
                 MMAE_EINVAL, "%s: pointers must be 16-byte aligned", who);
    return MMAE_OK;
}

extern "C" int mmae_attn_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                             float* lse, int B, int H, int q_len, int kv_len, int head_dim, float scale, hipStream_t stream) {
    AttnArgs a = {};
    a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.v = (const bf16*)v; a.ldv = ldv;
    a.o = (bf16*)o; a.ldo = ldo; a.lse = lse; a.B = B; a.H = H; a.q_len = q_len; a.kv_len = kv_len; a.scale = scale;
    int rc = check_common(a, head_dim, "attn_fwd");
    if (rc) return rc;
    // keys streamed where the resident images do not fit the LDS at all (no length limit then)
    const size_t resident = sizeof(bf16) * 2 * (size_t)pad32(kv_len) * (head_dim + 8);
    if (g_fwd_variant == 2 || (g_fwd_variant != 1 && resident > (size_t)ATTN_MAX_LDS))
        return head_dim == 64 ? launch_fwd_stream<64>(a, stream) : launch_fwd_stream<32>(a, stream);
    // images without padding columns where only they let two workgroups share a CU
    if ((g_fwd_variant == 3 || (g_fwd_variant == 0 && resident > (size_t)80 * 1024)) && fwd_swizzled_bytes(kv_len, head_dim) <= (size_t)80 * 1024) {
        const int tiles = (q_len + 31) / 32;
        // 8 waves per workgroup, 4 per SIMD with both workgroups resident (a 10-wave workgroup lands 3 + 3 + 2 + 2 on the SIMDs
        // and the second one no longer fits beside it at more than 85 registers: measured no faster than one workgroup)
        if (head_dim == 64) return launch_fwd_sw<64, 8>(a, stream);
        return launch_fwd_sw<32, 8>(a, stream);
    }
    const int nw = pick_waves((q_len + 31) / 32, sizeof(bf16) * 2 * (size_t)pad32(kv_len) * (head_dim + 8), head_dim);
    if (head_dim == 64) return ATTN_BY_WAVES(launch_fwd, 64, nw, a, stream);
    return ATTN_BY_WAVES(launch_fwd, 32, nw, a, stream);
}

extern "C" int mmae_attn_max_len(int head_dim, int backward) {
    if (head_dim != 32 && head_dim != 64) return 0;
    const size_t row = sizeof(bf16) * 2 * (size_t)(head_dim + 8);  // a K row and a V row (or a Q row and a dO row) of the LDS images
    const int kv = (int)((ATTN_MAX_LDS - (backward ? sizeof(float) * head_dim * (1 + 12) : 0)) / row) / 32 * 32;  // attn_fwd / attn_bwd_dq (mean key + up to 12 per-wave sums)
    if (!backward) return kv;
    const int q = (int)(ATTN_MAX_LDS / (row + 2 * sizeof(float))) / 32 * 32;                            // attn_bwd_dkv
    return q < kv ? q : kv;
}

extern "C" size_t mmae_attn_bwd_workspace_bytes(int B, int H, int q_len) { return (size_t)B * H * q_len * sizeof(float); }

extern "C" int mmae_attn_bwd_chunk(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o,
                                   long ldo, const void* d_o, long lddo, const float* lse, const float* mean_key, void* dq,
                                   long lddq, void* dk, long lddk, void* dv, long lddv, int B, int H, int q_len, int kv_len,
                                   int head_dim, float scale, void* workspace, size_t workspace_bytes, hipStream_t stream);

extern "C" int mmae_attn_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o,
                             long ldo, const void* d_o, long lddo, const float* lse, void* dq, long lddq, void* dk, long lddk,
                             void* dv, long lddv, int B, int H, int q_len, int kv_len, int head_dim, float scale,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return mmae_attn_bwd_chunk(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse, nullptr, dq, lddq, dk, lddk, dv, lddv, B, H, q_len, kv_len,
                               head_dim, scale, workspace, workspace_bytes, stream);
}

extern "C" int mmae_attn_bwd_chunk(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o,
                                   long ldo, const void* d_o, long lddo, const float* lse, const float* mean_key, void* dq,
                                   long lddq, void* dk, long lddk, void* dv, long lddv, int B, int H, int q_len, int kv_len,
                                   int head_dim, float scale, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    AttnArgs a = {};
    a.mean_key = mean_key;
    a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.v = (const bf16*)v; a.ldv = ldv;
    a.o = (bf16*)o; a.ldo = ldo; a.lse = (float*)lse; a.dO = (const bf16*)d_o; a.lddo = lddo;
    a.dq = (bf16*)dq; a.lddq = lddq; a.dk = (bf16*)dk; a.lddk = lddk; a.dv = (bf16*)dv; a.lddv = lddv;
    a.delta = (float*)workspace;
    a.B = B; a.H = H; a.q_len = q_len; a.kv_len = kv_len; a.scale = scale;
    int rc = check_common(a, head_dim, "attn_bwd");
    if (rc) return rc;
    MMAE_REQUIRE(d_o && dq && dk && dv && workspace, MMAE_EINVAL, "attn_bwd: null pointer");
    MMAE_REQUIRE(workspace_bytes >= mmae_attn_bwd_workspace_bytes(B, H, q_len), MMAE_EWORKSPACE, "attn_bwd: workspace too small");
    MMAE_REQUIRE(lddo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && (uintptr_t)d_o % 16 == 0, MMAE_ESHAPE,
                 "attn_bwd: bad gradient strides / alignment");
    // one kernel when Q, dO, K, V of a head fit LDS twice per CU (two blocks keep 16 waves resident)
    if (g_bwd_variant != 1 && fused_lds_bytes(a, head_dim) <= (g_bwd_variant == 2 ? (size_t)ATTN_MAX_LDS : (size_t)80 * 1024))
        return head_dim == 64 ? launch_fused<64, 8>(a, stream) : launch_fused<32, 8>(a, stream);
    const size_t dq_padded = sizeof(bf16) * 2 * (size_t)pad32(kv_len) * (head_dim + 8), dkv_padded = sizeof(bf16) * 2 * (size_t)pad32(q_len) * (head_dim + 8);
    const int nw_q = pick_waves((q_len + 31) / 32, dq_padded, head_dim);   // dQ: K, V in LDS
    const int nw_k = pick_waves((kv_len + 31) / 32, dkv_padded, head_dim);   // dK / dV: Q, dO in LDS
    // unpadded swizzled images where only they let two workgroups share a CU (variant 3: wherever they fit half the LDS).
    // By size at head_dim 32 only (576 rows: dQ + dK / dV 667 -> 491 us; 576 queries x 289 keys: 405 -> 322 us): at head_dim 64
    // these kernels need 157 / 166 registers, two 8-wave workgroups leave them 128 and the spills cost more than the second
    // workgroup brings (289 rows: 720 -> 1070 us; with 6 waves and 168 registers 772 us).
    const size_t half = (size_t)80 * 1024;
    const bool by_size = g_bwd_variant == 0 && head_dim == 32;
    const bool sw_q = (g_bwd_variant == 3 || (by_size && dq_padded + sizeof(float) * head_dim * (1 + nw_q) > half)) && dq_swizzled_bytes(kv_len, head_dim) <= half;
    const bool sw_k = (g_bwd_variant == 3 || (by_size && dkv_padded + sizeof(float) * 2 * pad32(q_len) > half)) && dkv_swizzled_bytes(q_len, head_dim) <= half;
    if (head_dim == 64) {
        rc = sw_q ? launch_dq_sw<64, 8>(a, stream) : ATTN_BY_WAVES(launch_dq, 64, nw_q, a, stream);
        if (rc) return rc;
        return sw_k ? launch_dkv_sw<64, 8>(a, stream) : ATTN_BY_WAVES(launch_dkv, 64, nw_k, a, stream);
    }
    rc = sw_q ? launch_dq_sw<32, 8>(a, stream) : ATTN_BY_WAVES(launch_dq, 32, nw_q, a, stream);
    if (rc) return rc;
    return sw_k ? launch_dkv_sw<32, 8>(a, stream) : ATTN_BY_WAVES(launch_dkv, 32, nw_k, a, stream);
}
