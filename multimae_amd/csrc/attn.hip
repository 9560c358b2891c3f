// Fused attention for short sequences (whole K/V of one (batch, head) resident in LDS), gfx950.
//
// All MFMAs are v_mfma_f32_32x32x16_bf16 with the KEY (forward, backward phase 1) or the QUERY
// (backward phase 2) index of the score tile in the accumulator rows, so that a score tile,
// converted to bf16 in registers, is directly the B operand of the products that reduce over
// that index (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS): no LDS round trip
// for P / dS.  The A operands of those products are read from transposed LDS images.
//
// Accumulator layout of a 32x32 tile: column = lane & 31, row(i) = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).
// Registers 8s..8s+7 of a tile, packed to bf16, are k-step s of the next product, whose k slot j of
// lane half h is tile row 16s + 8(j>>2) + 4h + (j&3); the A operand is fetched in that k order.
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

// A-operand fragment from a transposed image T[dd][n] (row stride `ld` elements): the 8 k slots of
// k-step s of tile `nt` for row dd
__device__ __forceinline__ bf16x8 tr_frag(const bf16* T, int ld, int dd, int n0, int h2) {
    const bf16x4 lo = *(const bf16x4*)(T + dd * ld + n0 + 4 * h2);
    const bf16x4 hi = *(const bf16x4*)(T + dd * ld + n0 + 8 + 4 * h2);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// stage rows [0, n_valid) of a [len, D] slice (row stride ld) into a row-major LDS image with padded
// rows (zero-filled up to n_pad rows) and optionally a transposed image T[D][n_pad + 4].
template <int D>
__device__ __forceinline__ void stage_rows(const bf16* src, long ld, int n_valid, int n_pad, bf16* rowimg, bf16* trimg,
                                           int tid, int nthreads) {
    constexpr int CH = D / 8, LDR = D + 8;
    const int ldt = n_pad + 4;
    for (int idx = tid; idx < n_pad * CH; idx += nthreads) {
        const int n = idx / CH, c = idx - n * CH;
        bf16x8 v;
        if (n < n_valid) v = *(const bf16x8*)(src + (long)n * ld + c * 8);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
        }
        if (rowimg) *(bf16x8*)(rowimg + n * LDR + c * 8) = v;
        if (trimg) {
#pragma unroll
            for (int e = 0; e < 8; ++e) trimg[(c * 8 + e) * ldt + n] = v[e];
        }
    }
}

// --------------------------------------------------------------------------------------------
// forward: two passes over the key tiles per 32-query block (row maximum first, then exp / P V), so only
// one score tile is live at a time; the second Q K^T costs D/16 MFMAs per tile.
// --------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ f32x16 score_tile(const bf16* Ks, int kt, int r, int h2, const bf16x8* qf) {
    constexpr int LDR = D + 8, KS = D / 16;
    f32x16 a;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 32 + r) * LDR + ks * 16 + h2 * 8);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], a, 0, 0, 0);
    }
    return a;
}

template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs p, int nkp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDR = D + 8, KS = D / 16, DT = D / 32;
    const int LDT = nkp + 4;
    bf16* Ks = (bf16*)smem;                 // [nkp][LDR]
    bf16* Vt = Ks + nkp * LDR;              // [D][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;

    stage_rows<D>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.kv_len, nkp, Ks, nullptr, tid, NW * 64);
    stage_rows<D>(p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len, nkp, nullptr, Vt, tid, NW * 64);
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nqt = (p.q_len + 31) / 32, nkt = nkp / 32;
    const int tail = p.kv_len - (nkt - 1) * 32;  // valid keys in the last tile (1..32)
    for (int qb = wave; qb < nqt; qb += NW) {
        const int qrow = qb * 32 + r;
        const int qc = qrow < p.q_len ? qrow : p.q_len - 1;
        const bf16* qp = p.q + ((long)b * p.q_len + qc) * p.ldq + h * D + h2 * 8;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);

        float m = -INFINITY;
        for (int kt = 0; kt < nkt; ++kt) {
            const f32x16 a = score_tile<D>(Ks, kt, r, h2, qf);
            const int lim = kt == nkt - 1 ? tail : 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, acc_row(i, h2) < lim ? a[i] : -INFINITY);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));

        float l = 0.f;
        f32x16 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 a = score_tile<D>(Ks, kt, r, h2, qf);
            const int lim = kt == nkt - 1 ? tail : 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = acc_row(i, h2) < lim ? exp2f((a[i] - m) * c) : 0.f;
                a[i] = e;
                l += e;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(a, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = tr_frag(Vt, LDT, dt * 32 + r, kt * 32 + 16 * s2, h2);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
                }
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
// backward: phase 1 per 32-query block (dQ), phase 2 per 32-key tile (dK, dV); P is recomputed
// from the forward's log-sum-exp, delta = rowsum(dO * O).
// --------------------------------------------------------------------------------------------
template <int D, int NQT, int NKT>
struct BwdLds {
    static constexpr int LDR = D + 8, NQP = NQT * 32, NKP = NKT * 32, LDTQ = NQP + 4, LDTK = NKP + 4;
    static constexpr size_t bytes =
        sizeof(bf16) * (size_t)(2 * NQP * LDR + 2 * NKP * LDR + D * LDTK + 2 * D * LDTQ) + sizeof(float) * 2 * NQP;
};

template <int D, int NQT, int NKT, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = BwdLds<D, NQT, NKT>;
    constexpr int LDR = L::LDR, NQP = L::NQP, NKP = L::NKP, LDTQ = L::LDTQ, LDTK = L::LDTK, KS = D / 16, DT = D / 32, CH = D / 8;
    bf16* Qs = (bf16*)smem;            // [NQP][LDR]
    bf16* dOs = Qs + NQP * LDR;        // [NQP][LDR]
    bf16* Ks = dOs + NQP * LDR;        // [NKP][LDR]
    bf16* Vs = Ks + NKP * LDR;         // [NKP][LDR]
    bf16* Kt = Vs + NKP * LDR;         // [D][LDTK]
    bf16* Qt = Kt + D * LDTK;          // [D][LDTQ]
    bf16* dOt = Qt + D * LDTQ;         // [D][LDTQ]
    float* lse2 = (float*)(dOt + D * LDTQ);  // [NQP]
    float* delta = lse2 + NQP;               // [NQP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    constexpr int NT = NW * 64;

    stage_rows<D>(p.q + (long)b * p.q_len * p.ldq + h * D, p.ldq, p.q_len, NQP, Qs, Qt, tid, NT);
    stage_rows<D>(p.dO + (long)b * p.q_len * p.lddo + h * D, p.lddo, p.q_len, NQP, dOs, dOt, tid, NT);
    stage_rows<D>(p.k + (long)b * p.kv_len * p.ldk + h * D, p.ldk, p.kv_len, NKP, Ks, Kt, tid, NT);
    stage_rows<D>(p.v + (long)b * p.kv_len * p.ldv + h * D, p.ldv, p.kv_len, NKP, Vs, nullptr, tid, NT);
    // delta[q] = sum_d dO[q,d] * O[q,d]; CH consecutive threads share a query
    for (int idx = tid; idx < NQP * CH; idx += NT) {
        const int n = idx / CH, cc = idx - n * CH;
        float part = 0.f;
        if (n < p.q_len) {
            const bf16x8 a = *(const bf16x8*)(p.dO + ((long)b * p.q_len + n) * p.lddo + h * D + cc * 8);
            const bf16x8 o = *(const bf16x8*)(p.o + ((long)b * p.q_len + n) * p.ldo + h * D + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)a[e] * (float)o[e];
        }
#pragma unroll
        for (int off = 1; off < CH; off <<= 1) part += __shfl_xor(part, off, 64);
        if (cc == 0) {
            delta[n] = part;
            lse2[n] = n < p.q_len ? p.lse[((long)b * p.H + h) * p.q_len + n] * LOG2E : 1e30f;
        }
    }
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nqt = (p.q_len + 31) / 32, nkt = (p.kv_len + 31) / 32;
    const int tail = p.kv_len - (nkt - 1) * 32;

    // ---- phase 1: dQ ---------------------------------------------------------------------
    for (int qb = wave; qb < nqt; qb += NW) {
        bf16x8 qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *(const bf16x8*)(Qs + (qb * 32 + r) * LDR + ks * 16 + h2 * 8);
            dof[ks] = *(const bf16x8*)(dOs + (qb * 32 + r) * LDR + ks * 16 + h2 * 8);
        }
        const float my_lse = lse2[qb * 32 + r], my_delta = delta[qb * 32 + r];
        f32x16 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[dt][i] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 32 + r) * LDR + ks * 16 + h2 * 8);
                const bf16x8 vf = *(const bf16x8*)(Vs + (kt * 32 + r) * LDR + ks * 16 + h2 * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool valid = acc_row(i, h2) < (kt == nkt - 1 ? tail : 32);
                const float pr = valid ? exp2f(s[i] * c - my_lse) : 0.f;
                s[i] = pr * (dp[i] - my_delta) * p.scale;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 dsf = pack8(s, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 ktf = tr_frag(Kt, LDTK, dt * 32 + r, kt * 32 + 16 * s2, h2);
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[dt], 0, 0, 0);
                }
            }
        }
        const int qrow = qb * 32 + r;
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
    }

    // ---- phase 2: dK, dV -------------------------------------------------------------------
    for (int kt = wave; kt < nkt; kt += NW) {
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *(const bf16x8*)(Ks + (kt * 32 + r) * LDR + ks * 16 + h2 * 8);
            vf[ks] = *(const bf16x8*)(Vs + (kt * 32 + r) * LDR + ks * 16 + h2 * 8);
        }
        const bool key_valid = kt * 32 + r < p.kv_len;
        f32x16 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk[dt][i] = 0.f; dv[dt][i] = 0.f; }
        for (int qb = 0; qb < nqt; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 qa = *(const bf16x8*)(Qs + (qb * 32 + r) * LDR + ks * 16 + h2 * 8);
                const bf16x8 da = *(const bf16x8*)(dOs + (qb * 32 + r) * LDR + ks * 16 + h2 * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 lv = *(const f32x4*)(lse2 + qb * 32 + 8 * i4 + 4 * h2);
                const f32x4 dl = *(const f32x4*)(delta + qb * 32 + 8 * i4 + 4 * h2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * i4 + e;
                    const float pr = key_valid ? exp2f(s[i] * c - lv[e]) : 0.f;
                    s[i] = pr;
                    dp[i] = pr * (dp[i] - dl[e]) * p.scale;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(s, s2), dsf = pack8(dp, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 dot = tr_frag(dOt, LDTQ, dt * 32 + r, qb * 32 + 16 * s2, h2);
                    const bf16x8 qt = tr_frag(Qt, LDTQ, dt * 32 + r, qb * 32 + 16 * s2, h2);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        const int key = kt * 32 + r;
        if (key < p.kv_len) {
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
// host side
// --------------------------------------------------------------------------------------------
template <typename K>
static int set_lds(K kern, size_t bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            mmae_set_error("attention: cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
            return MMAE_ELAUNCH;
        }
    }
    return MMAE_OK;
}

template <int D, int NW>
static int launch_fwd(const AttnArgs& a, hipStream_t s) {
    const int nkp = (a.kv_len + 31) / 32 * 32;
    const size_t bytes = sizeof(bf16) * ((size_t)nkp * (D + 8) + (size_t)D * (nkp + 4));
    MMAE_REQUIRE(bytes <= 160 * 1024, MMAE_ESHAPE, "attn_fwd: kv_len %d needs %zu bytes of LDS (> 160 KiB)", a.kv_len, bytes);
    static size_t reserved = 0;  // largest dynamic LDS size requested so far for this instantiation
    if (bytes > 65536 && bytes > reserved) {
        int rc = set_lds(attn_fwd_kernel<D, NW>, 160 * 1024);
        if (rc) return rc;
        reserved = 160 * 1024;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<D, NW>), dim3(a.B * a.H), dim3(NW * 64), bytes, s, a, nkp);
    MMAE_CHECK_LAUNCH("attn_fwd");
    return MMAE_OK;
}

template <int D, int NQT, int NKT, int NW>
static int launch_bwd(const AttnArgs& a, hipStream_t s) {
    constexpr size_t bytes = BwdLds<D, NQT, NKT>::bytes;
    static_assert(bytes <= 160 * 1024, "backward LDS image exceeds one CU");
    static bool reserved = false;
    if (!reserved) {
        int rc = set_lds(attn_bwd_kernel<D, NQT, NKT, NW>, bytes);
        if (rc) return rc;
        reserved = true;
    }
    hipLaunchKernelGGL((attn_bwd_kernel<D, NQT, NKT, NW>), dim3(a.B * a.H), dim3(NW * 64), bytes, s, a);
    MMAE_CHECK_LAUNCH("attn_bwd");
    return MMAE_OK;
}

static int check_common(const AttnArgs& a, int head_dim, const char* who) {
    MMAE_REQUIRE(a.q && a.k && a.v && a.o && a.lse, MMAE_EINVAL, "%s: null pointer", who);
    MMAE_REQUIRE(a.B > 0 && a.H > 0 && a.q_len > 0 && a.kv_len > 0, MMAE_EINVAL, "%s: empty problem", who);
    MMAE_REQUIRE(head_dim == 32 || head_dim == 64, MMAE_ESHAPE, "%s: head_dim %d not in {32, 64}", who, head_dim);
    MMAE_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0, MMAE_ESHAPE,
                 "%s: row strides must be multiples of 8 elements", who);
    MMAE_REQUIRE(((uintptr_t)a.q % 16 == 0) && ((uintptr_t)a.k % 16 == 0) && ((uintptr_t)a.v % 16 == 0) && ((uintptr_t)a.o % 8 == 0),
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
    const int nqt = (q_len + 31) / 32;
    if (head_dim == 64) return nqt > 4 ? launch_fwd<64, 8>(a, stream) : launch_fwd<64, 4>(a, stream);
    return nqt > 4 ? launch_fwd<32, 8>(a, stream) : launch_fwd<32, 4>(a, stream);
}

extern "C" int mmae_attn_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o,
                             long ldo, const void* d_o, long lddo, const float* lse, void* dq, long lddq, void* dk, long lddk,
                             void* dv, long lddv, int B, int H, int q_len, int kv_len, int head_dim, float scale,
                             hipStream_t stream) {
    AttnArgs a = {};
    a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.v = (const bf16*)v; a.ldv = ldv;
    a.o = (bf16*)o; a.ldo = ldo; a.lse = (float*)lse; a.dO = (const bf16*)d_o; a.lddo = lddo;
    a.dq = (bf16*)dq; a.lddq = lddq; a.dk = (bf16*)dk; a.lddk = lddk; a.dv = (bf16*)dv; a.lddv = lddv;
    a.B = B; a.H = H; a.q_len = q_len; a.kv_len = kv_len; a.scale = scale;
    int rc = check_common(a, head_dim, "attn_bwd");
    if (rc) return rc;
    MMAE_REQUIRE(d_o && dq && dk && dv, MMAE_EINVAL, "attn_bwd: null pointer");
    MMAE_REQUIRE(lddo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0, MMAE_ESHAPE, "attn_bwd: bad gradient strides");
    const int nkt = (kv_len + 31) / 32, nqt = (q_len + 31) / 32;
    if (head_dim == 64) {
        MMAE_REQUIRE(nqt <= 4 && nkt <= 4, MMAE_ESHAPE,
                     "attn_bwd: head_dim 64 supports q_len, kv_len <= 128 (got %d, %d)", q_len, kv_len);
        return launch_bwd<64, 4, 4, 4>(a, stream);
    }
    MMAE_REQUIRE(nqt <= 7 && nkt <= 7, MMAE_ESHAPE, "attn_bwd: head_dim 32 supports q_len, kv_len <= 224 (got %d, %d)", q_len, kv_len);
    if (nqt <= 4 && nkt <= 4) return launch_bwd<32, 4, 4, 4>(a, stream);
    if (nkt <= 4) return launch_bwd<32, 7, 4, 8>(a, stream);
    return launch_bwd<32, 7, 7, 8>(a, stream);
}
