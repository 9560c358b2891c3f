// bf16 MFMA GEMMs for the MultiMAE step (gfx950).
//
//  gemm_nt : C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)      forward Linear, dA = dC * W via W^T copy
//  gemm_tn : D[N,K] = X[M,N]^T * Y[M,K]                           dW = dC^T * A, split over M into fp32 slabs
//
// Both use 128x128 output tiles, four waves (2x2, 64x64 each), v_mfma_f32_16x16x32_bf16,
// 16-byte global->LDS DMA into a double-buffered, XOR-swizzled LDS image.
#include "common.h"
#include "ln_rows.h"
#include <type_traits>
#include <stdlib.h>
#ifndef MMAE_LN_TAIL_R
#define MMAE_LN_TAIL_R 8  // rows per wave and pass of the LayerNorm workgroups beside a tail launch (eight waves per CU)
#endif
#ifndef MMAE_RES_AHEAD
#define MMAE_RES_AHEAD 3  // pieces of the residual fetched ahead of the stores in the row-major epilogue (tuning builds override it)
#endif

#define BM 128
#define BN 128
#define BKT 64            // k elements per stage
#define TILE_BYTES 16384  // 128 rows x 64 bf16
#define RASTER_M 8

__device__ uint4 g_zero_page[4];  // 64 bytes of zeros: DMA source for out-of-range rows / k chunks

enum {
    EPI_BF16 = 0,      // out bf16 = acc (+bias)
    EPI_GELU = 1,      // out bf16 = h = acc+bias ; out2 bf16 = gelu(acc+bias) (from the fp32 value, before h is rounded)
    EPI_RESID = 2,     // out f32 = aux_f32 + acc + bias
    EPI_DGELU = 3,     // out bf16 = acc * gelu'(aux_bf16)
    EPI_F32 = 4,       // out f32 = acc (+bias)
    EPI_TOKENS = 5,    // out f32[row_map(m)] = acc + add_tab[add_idx[m]]   (patch embed)
    EPI_UNPATCH = 6,   // out bf16 / f32 image[b,c,nh*P+ph,nw*P+pw] = acc + bias   (out_proj)
};

struct GemmNT {
    const bf16* A; long lda;
    const bf16* W; long ldw;
    int M, N, K;
    const float* bias;
    void* out; long ldo;
    void* out2;
    const void* aux; long ldaux;
    int seg_in, seg_out;        // EPI_TOKENS: out row = (m / seg_in) * seg_out + m % seg_in
    const int* add_idx; const float* add_tab;
    int C, P, NH, NW, out_f32;  // EPI_UNPATCH
};

// LayerNorm rows handed to workgroups that would otherwise idle beside the tail tiles of a residual GEMM (gemm_nt_pp_kernel):
// x f32 [rows, dim] (rows the preceding full-tile launch has already written) -> y bf16, mean, rstd.  rows == 0: no job.
struct LnJob { const float* x; const float* gamma; const float* beta; bf16* y; float* mean; float* rstd; int rows, dim; float eps; };

// Stores one wave's 64x64 accumulator block: (acc[2u][mi], acc[2u+1][mi]) = C[m = mw + mi*16 + fr][n = nw + 32*u + 8*fq + 0..7]
// bias_regs (optional): the lane's 16 bias values, [8*u + e] = bias[nw + 32*u + 8*fq + e], for callers that cannot afford a
// global load in the epilogue (an ordinary load's result makes the compiler drain every LDS-DMA in flight).
// FULL: the caller guarantees the 64x64 block lies inside the matrix; the stores then carry no per-lane condition, so
// the epilogue is straight-line code (at every branch join the compiler waits for all outstanding memory operations,
// the stores just issued included, before touching a loaded value again).
// RALL: the whole residual block (64 registers) is fetched before the first store, for callers with registers to spare
// (one tile per CU in flight: the epilogue is then bound by the latency of each row's round trip).
// NMI: fragment rows of the block (4 = 64 rows; 2 = the 32-row blocks of the 64-row tiles)
template <int EPI, bool FULL = false, bool BREG = false, bool RALL = false, int NMI = 4>
__device__ __forceinline__ void store_wave_tile(const GemmNT& p, const f32x4 (&acc)[4][4], int mw, int nw, int fr, int fq,
                                                const float* bias_regs = nullptr, const bf16x8 (*h_regs)[2] = nullptr) {
    // Epilogue inputs (saved pre-activation) are fetched for the whole 64x64 block before anything is stored:
    // a load issued after a store has to wait for that store as well (one in-order counter), which would serialise
    // sixteen memory round trips.
    // The residual rows run one fragment row ahead of the stores instead (two buffers of 16 registers): the wait for row
    // mi + 1 then leaves the stores of row mi in flight, where a batch fetched after the stores would wait for them too.
    constexpr int MB = EPI == EPI_RESID ? 1 : 4;  // fragment rows per batch (register budget: 16 / 8 registers per row)
    constexpr int RBUF = EPI == EPI_RESID ? (RALL ? 4 : 2) : 1;
    f32x4 res[RBUF][2][2];
    bf16x8 hsv[EPI == EPI_DGELU ? MB : 1][2];
    auto load_res = [&](int mi_) {
        int m = mw + mi_ * 16 + fr;
        m = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int n = nw + 32 * u + 8 * fq;
            n = n < p.N ? n : p.N - 8;
            const float* rp = (const float*)p.aux + (long)m * p.ldaux + n;
            res[mi_ & (RBUF - 1)][u][0] = *(const f32x4*)rp;
            res[mi_ & (RBUF - 1)][u][1] = *(const f32x4*)(rp + 4);
        }
    };
    // the lane's 16 bias values once, ahead of every store (fetched between stores they would wait for those stores)
    constexpr bool USES_BIAS = EPI != EPI_DGELU && EPI != EPI_TOKENS;
    float bl[16];
    if (USES_BIAS && !BREG) {
#pragma unroll
        for (int e = 0; e < 16; ++e) bl[e] = 0.f;
        if (p.bias) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int n = nw + 32 * u + 8 * fq;
                n = n < p.N ? n : p.N - 8;
                const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bl[8 * u + e] = b0[e]; bl[8 * u + 4 + e] = b1[e]; }
            }
        }
    }
    if (EPI == EPI_RESID) {
        load_res(0);
        if (RALL) {
#pragma unroll
            for (int q = 1; q < NMI; ++q) load_res(q);
        }
    }
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
        if (EPI == EPI_RESID && !RALL && mi + 1 < NMI) load_res(mi + 1);
        if (EPI == EPI_DGELU && !h_regs && mi % MB == 0) {
#pragma unroll
            for (int mj = 0; mj < (MB < NMI ? MB : NMI); ++mj) {
                int m = mw + (mi + mj) * 16 + fr;
                m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int n = nw + 32 * u + 8 * fq;
                    n = n < p.N ? n : p.N - 8;
                    hsv[EPI == EPI_DGELU ? mj : 0][u] = *(const bf16x8*)((const bf16*)p.aux + (long)m * p.ldaux + n);
                }
            }
        }
        const int m = mw + mi * 16 + fr;
        if (!FULL && m >= p.M) continue;
        long orow = m;
        const float* tab = nullptr;
        int ib = 0, inh = 0, inw = 0;
        if (EPI == EPI_TOKENS) {
            orow = (long)(m / p.seg_in) * p.seg_out + (m % p.seg_in);
            tab = p.add_tab + (long)p.add_idx[m] * p.N;
        }
        if (EPI == EPI_UNPATCH) {
            const int ntok = p.NH * p.NW;
            ib = m / ntok;
            const int t = m - ib * ntok;
            inh = t / p.NW; inw = t - inh * p.NW;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = nw + 32 * u + 8 * fq;
            if (!FULL && n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[2 * u][mi][e]; v[4 + e] = acc[2 * u + 1][mi][e]; }
            if (USES_BIAS) {
                if (BREG) {  // zeros when there is no bias
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bias_regs[8 * u + e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bl[8 * u + e];
                }
            }
            if (EPI == EPI_BF16) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                *(bf16x8*)((bf16*)p.out + orow * p.ldo + n) = o;
            } else if (EPI == EPI_GELU) {
                bf16x8 h, g;
                f32x2 xv[4], gv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = (f32x2){v[2 * e], v[2 * e + 1]};
                gelu_poly2<4>(xv, gv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[2 * e] = (bf16)v[2 * e]; h[2 * e + 1] = (bf16)v[2 * e + 1];
                    g[2 * e] = (bf16)gv[e][0]; g[2 * e + 1] = (bf16)gv[e][1];
                }
                *(bf16x8*)((bf16*)p.out + orow * p.ldo + n) = h;
                *(bf16x8*)((bf16*)p.out2 + orow * p.ldo + n) = g;
            } else if (EPI == EPI_RESID) {
                const f32x4 r0 = res[mi & (RBUF - 1)][u][0], r1 = res[mi & (RBUF - 1)][u][1];
                float* op = (float*)p.out + orow * p.ldo + n;
                *(f32x4*)op = (f32x4){r0[0] + v[0], r0[1] + v[1], r0[2] + v[2], r0[3] + v[3]};
                *(f32x4*)(op + 4) = (f32x4){r1[0] + v[4], r1[1] + v[5], r1[2] + v[6], r1[3] + v[7]};
            } else if (EPI == EPI_DGELU) {
                const bf16x8 h = h_regs ? h_regs[mi][u] : hsv[EPI == EPI_DGELU ? mi % MB : 0][u];
                bf16x8 o;
                f32x2 xv[4], dv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = (f32x2){(float)h[2 * e], (float)h[2 * e + 1]};
                gelu_grad_poly2<4>(xv, dv);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[2 * e] = (bf16)(v[2 * e] * dv[e][0]); o[2 * e + 1] = (bf16)(v[2 * e + 1] * dv[e][1]); }
                *(bf16x8*)((bf16*)p.out + orow * p.ldo + n) = o;
            } else if (EPI == EPI_F32) {
                float* op = (float*)p.out + orow * p.ldo + n;
                *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};
                *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else if (EPI == EPI_TOKENS) {
                const f32x4 t0 = *(const f32x4*)(tab + n), t1 = *(const f32x4*)(tab + n + 4);
                float* op = (float*)p.out + orow * p.ldo + n;
                *(f32x4*)op = (f32x4){v[0] + t0[0], v[1] + t0[1], v[2] + t0[2], v[3] + t0[3]};
                *(f32x4*)(op + 4) = (f32x4){v[4] + t1[0], v[5] + t1[1], v[6] + t1[2], v[7] + t1[3]};
            } else if (EPI == EPI_UNPATCH) {
                // the 8 columns are two runs of 4 pixels; each run stays inside one patch row because P % 4 == 0
                const int pp = p.P * p.P;
                const long H = (long)p.NH * p.P, Wd = (long)p.NW * p.P;
                if (!p.out_f32 && (p.P & 7) == 0) {  // one 16-byte store
                    const int c = n / pp, rem = n - c * pp, ph = rem / p.P, pw = rem - ph * p.P;
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                    *(bf16x8*)((bf16*)p.out + (((long)ib * p.C + c) * H + inh * p.P + ph) * Wd + inw * p.P + pw) = o;
                    continue;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nn = n + 4 * h;
                    const int c = nn / pp, rem = nn - c * pp, ph = rem / p.P, pw = rem - ph * p.P;
                    const long o = (((long)ib * p.C + c) * H + inh * p.P + ph) * Wd + inw * p.P + pw;
                    if (p.out_f32) {
                        *(f32x4*)((float*)p.out + o) = (f32x4){v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
                    } else {
                        *(bf16x4*)((bf16*)p.out + o) = (bf16x4){(bf16)v[4 * h], (bf16)v[4 * h + 1], (bf16)v[4 * h + 2], (bf16)v[4 * h + 3]};
                    }
                }
            }
        }
    }
}

// The same 64x64 block, full tiles only, with every global access covering whole 128-byte rows.
// In the accumulator layout neighbouring lanes hold neighbouring ROWS, so a 16-byte store (or load) per lane touches 64
// separate segments and a CU moves about 33 GB/s however few CUs are active; with eight neighbouring lanes on one 128-byte
// row the same instruction runs at about 130 GB/s per CU.  The block therefore goes through a 2-KiB piece of LDS owned by
// the wave, 16 rows x 128 bytes at a time: written in the accumulator layout (lane (fr, fq) -> row fr, 16-byte chunks
// {fq, 4 + fq} for bf16, {2fq, 2fq + 1} for fp32), read back with lane l on row (l >> 3) + 8j, chunk l & 7.  Chunk c of
// row r sits at c ^ ((r >> 1) & 7): both directions are free of bank conflicts.  DS operations of one wave execute in
// order, so no barrier is involved.
//   bf16 outputs: one piece per fragment row mi (64 columns x 2 bytes); fp32 outputs: one piece per (mi, u) (32 columns)
//   EPI_DGELU: h_rows[mi][j] holds the saved pre-activations in the ROW layout (fetched by the caller ahead of the stores);
//              they pass through the same piece to reach the accumulator layout
//   EPI_RESID: the residual is fetched and added in the row layout, RES_AHEAD pieces ahead of the stores
template <int EPI, int NMI = 4>
__device__ __forceinline__ void store_wave_tile_rows(const GemmNT& p, const f32x4 (&acc)[4][4], int mw, int nw, int lane,
                                                     char* stage, const float* bias_regs, const bf16x8 (*h_rows)[2] = nullptr) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_RESID || EPI == EPI_DGELU || EPI == EPI_F32, "row-major epilogues");
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 3, cc = lane & 7;
    char* const wrow = stage + fr * 128;
    const int wsw = (fr >> 1) & 7;
    // read side: rows rr and rr + 8 -> swizzles (rr >> 1) & 3 and 4 + ((rr >> 1) & 3)
    char* const rd0 = stage + rr * 128 + ((cc ^ ((rr >> 1) & 7)) << 4);
    char* const rd1 = stage + (rr + 8) * 128 + ((cc ^ (((rr + 8) >> 1) & 7)) << 4);
    constexpr bool F32OUT = EPI == EPI_RESID || EPI == EPI_F32;
    if constexpr (!F32OUT) {
        bf16* const o0 = (bf16*)p.out + (long)(mw + rr) * p.ldo + nw + 8 * cc;
        bf16* const o20 = EPI == EPI_GELU ? (bf16*)p.out2 + (long)(mw + rr) * p.ldo + nw + 8 * cc : nullptr;
        char* const w0 = wrow + ((fq ^ wsw) << 4);
        char* const w1 = wrow + (((4 + fq) ^ wsw) << 4);
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            bf16x8 hacc[2];
            if constexpr (EPI == EPI_DGELU) {
                *(bf16x8*)rd0 = h_rows[mi][0];
                *(bf16x8*)rd1 = h_rows[mi][1];
                hacc[0] = *(const bf16x8*)w0;
                hacc[1] = *(const bf16x8*)w1;
            }
            bf16x8 o[2], g[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[2 * u][mi][e]; v[4 + e] = acc[2 * u + 1][mi][e]; }
                if constexpr (EPI != EPI_DGELU) {  // zeros when there is no bias
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bias_regs[8 * u + e];
                }
                if constexpr (EPI == EPI_BF16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[u][e] = (bf16)v[e];
                } else if constexpr (EPI == EPI_GELU) {
                    f32x2 xv[4], gv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = (f32x2){v[2 * e], v[2 * e + 1]};
                    gelu_poly2<4>(xv, gv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[u][2 * e] = (bf16)v[2 * e]; o[u][2 * e + 1] = (bf16)v[2 * e + 1];
                        g[u][2 * e] = (bf16)gv[e][0]; g[u][2 * e + 1] = (bf16)gv[e][1];
                    }
                } else {  // EPI_DGELU
                    f32x2 xv[4], dv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = (f32x2){(float)hacc[u][2 * e], (float)hacc[u][2 * e + 1]};
                    gelu_grad_poly2<4>(xv, dv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[u][2 * e] = (bf16)(v[2 * e] * dv[e][0]); o[u][2 * e + 1] = (bf16)(v[2 * e + 1] * dv[e][1]);
                    }
                }
            }
            *(bf16x8*)w0 = o[0];
            *(bf16x8*)w1 = o[1];
            const bf16x8 r0 = *(const bf16x8*)rd0, r1 = *(const bf16x8*)rd1;
            *(bf16x8*)(o0 + (long)(mi * 16) * p.ldo) = r0;
            *(bf16x8*)(o0 + (long)(mi * 16 + 8) * p.ldo) = r1;
            if constexpr (EPI == EPI_GELU) {
                *(bf16x8*)w0 = g[0];
                *(bf16x8*)w1 = g[1];
                const bf16x8 g0 = *(const bf16x8*)rd0, g1 = *(const bf16x8*)rd1;
                *(bf16x8*)(o20 + (long)(mi * 16) * p.ldo) = g0;
                *(bf16x8*)(o20 + (long)(mi * 16 + 8) * p.ldo) = g1;
            }
        }
    } else {
        // pieces in the order (mi, u): 16 rows x 32 fp32 columns
        constexpr int RES_AHEAD = MMAE_RES_AHEAD;
        float* const o0 = (float*)p.out + (long)(mw + rr) * p.ldo + nw + 4 * cc;
        const float* const a0 = EPI == EPI_RESID ? (const float*)p.aux + (long)(mw + rr) * p.ldaux + nw + 4 * cc : nullptr;
        char* const w0 = wrow + (((2 * fq) ^ wsw) << 4);
        char* const w1 = wrow + (((2 * fq + 1) ^ wsw) << 4);
        f32x4 res[RES_AHEAD + 1][2];
        auto load_res = [&](int piece) {
            const int mi = piece >> 1, u = piece & 1;
            res[piece % (RES_AHEAD + 1)][0] = *(const f32x4*)(a0 + (long)(mi * 16) * p.ldaux + 32 * u);
            res[piece % (RES_AHEAD + 1)][1] = *(const f32x4*)(a0 + (long)(mi * 16 + 8) * p.ldaux + 32 * u);
        };
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int q = 0; q < RES_AHEAD; ++q) load_res(q);
        }
#pragma unroll
        for (int piece = 0; piece < 2 * NMI; ++piece) {
            const int mi = piece >> 1, u = piece & 1;
            if constexpr (EPI == EPI_RESID) {
                if (piece + RES_AHEAD < 2 * NMI) load_res(piece + RES_AHEAD);
            }
            f32x4 v0 = acc[2 * u][mi], v1 = acc[2 * u + 1][mi];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] += bias_regs[8 * u + e]; v1[e] += bias_regs[8 * u + 4 + e]; }
            *(f32x4*)w0 = v0;
            *(f32x4*)w1 = v1;
            f32x4 r0 = *(const f32x4*)rd0, r1 = *(const f32x4*)rd1;
            if constexpr (EPI == EPI_RESID) {
                r0 += res[piece % (RES_AHEAD + 1)][0];
                r1 += res[piece % (RES_AHEAD + 1)][1];
            }
            *(f32x4*)(o0 + (long)(mi * 16) * p.ldo + 32 * u) = r0;
            *(f32x4*)(o0 + (long)(mi * 16 + 8) * p.ldo + 32 * u) = r1;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    const int wg = xcd_remap(blockIdx.x, nblk);
    // walk the tiles in groups of RASTER_M row-tiles x all column-tiles, rows fastest: the workgroups in flight on one
    // XCD then share a few A panels and a few W panels that fit its L2
    const int tiles_m = (p.M + BM - 1) / BM;
    const int per_group = RASTER_M * tiles_n;
    const int grp = wg / per_group, within = wg - grp * per_group;
    int gm_rows = tiles_m - grp * RASTER_M; gm_rows = gm_rows < RASTER_M ? gm_rows : RASTER_M;
    const int m0 = (grp * RASTER_M + within % gm_rows) * BM, n0 = (within / gm_rows) * BN;

    // staging: wave-instruction j of this wave fills LDS rows (j*4+wave)*8 .. +7 (8 lanes per 128-B row).
    // LDS holds 16-B chunk c of row r at position c ^ swz(r); swz differs for the two tiles because their fragment
    // reads walk rows differently (see the fragment row maps below); both make every ds_read_b128 conflict-free.
    const bf16* a_src[4];
    const bf16* w_src[4];
    int ka_src[4], kw_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        const int ca = (lane & 7) ^ ((r >> 1) & 7);
        const int cw = (lane & 7) ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1));
        int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + r; gn = gn < p.N ? gn : p.N - 1;
        a_src[j] = p.A + (long)gm * p.lda + ca * 8;
        w_src[j] = p.W + (long)gn * p.ldw + cw * 8;
        ka_src[j] = ca * 8;
        kw_src[j] = cw * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * 2 * TILE_BYTES;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // chunks past K read zeros (K need not be a multiple of 64)
            const void* as = kt * BKT + ka_src[j] < p.K ? (const void*)(a_src[j] + kt * BKT) : (const void*)g_zero_page;
            const void* ws = kt * BKT + kw_src[j] < p.K ? (const void*)(w_src[j] + kt * BKT) : (const void*)g_zero_page;
            glds16(as, sa + (j * 4 + wave) * 1024);
            glds16(ws, sw + (j * 4 + wave) * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BKT - 1) / BKT;
    const int fr = lane & 15, fq = lane >> 4;
    const int swz = fr >> 1;  // both swizzles reduce to fr >> 1 for the rows this lane reads
    // Fragment row maps.  Activation tile (MFMA B operand, output column m): row = 16*t + fr.
    // Weight tile (MFMA A operand, output row n): fragments 2u and 2u+1 interleave their rows in groups of four,
    // row = 32*u + 8*(fr>>2) + 4*(t&1) + (fr&3), so that the accumulators acc[2u][.] and acc[2u+1][.] of a lane are
    // 8 consecutive output columns n = 32*u + 8*fq + 0..7 (one 16-byte bf16 store / two 16-byte fp32 stores).
    const int wrow = wn * 64 + 8 * (fr >> 2) + (fr & 3);
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * 2 * TILE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ((ks * 4 + fq) ^ swz) << 4;
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = *(const bf16x8*)(sw + (wrow + 32 * (t >> 1) + 4 * (t & 1)) * 128 + chunk);
                af[t] = *(const bf16x8*)(sa + (wm * 64 + t * 16 + fr) * 128 + chunk);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (m0 + BM <= p.M && n0 + BN <= p.N) store_wave_tile<EPI, true, false, true>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fq);
    else store_wave_tile<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fq);
}

// ------------------------------------------------------------------------------------------
// Persistent variant of the 128x128 kernel for problems with many tiles of a short K loop whose K is not a multiple
// of 128 (the 256x256 kernel takes the others): waves 2 x 2 (WM = 2; WM = 4 gives a 256x128 tile), two workgroups per
// CU, an LDS ring of STAGES k-tiles filled ahead by 16-byte LDS-DMA, counted vmcnt + raw s_barrier so the prefetch
// stays in flight across barriers, and the ring runs on across tile boundaries (the next tile's first k-tiles are
// already landing while this tile's epilogue stores).
// ------------------------------------------------------------------------------------------

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

// WM = waves along M (2 -> 128-row tile, 256 threads; 4 -> 256-row tile, 512 threads); two waves along N (128 columns).
template <int EPI, int WM, int STAGES>
__global__ __launch_bounds__(WM * 128, (WM == 4 ? 2 : 2)) void gemm_nt_persist_kernel(GemmNT p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char psmem[];
    constexpr int NWV = WM * 2, TBM = WM * 64, A_BYTES = TBM * 128, STAGE = A_BYTES + 16384;
    constexpr int WI = 8 / WM;                 // weight-tile LDS-DMA instructions per thread and stage
    constexpr int LOADS = 4 + WI;              // per thread and stage
    constexpr int RASTER = 1024 / TBM;         // row-tiles per raster group (1024 rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4, swz = fr >> 1;
    const int wrow = wn * 64 + 8 * (fr >> 2) + (fr & 3);
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int me = xcd_remap(blockIdx.x, G);  // workgroups that share an XCD take neighbouring tiles of each round
    const int nk = (p.K + BKT - 1) / BKT;
    const int my_tiles = me < ntiles ? (ntiles - 1 - me) / G + 1 : 0;
    const int total = my_tiles * nk;

    auto tile_origin = [&](int t, int& m0, int& n0) {  // groups of RASTER row-tiles x all column-tiles, rows fastest
        const int per_group = RASTER * tiles_n;
        const int grp = t / per_group, within = t - grp * per_group;
        int rows = tiles_m - grp * RASTER; rows = rows < RASTER ? rows : RASTER;
        m0 = (grp * RASTER + within % rows) * TBM;
        n0 = (within / rows) * BN;
    };

    // ---- load cursor ----
    const bf16* a_src[4]; const bf16* w_src[WI];
    int ka_src[4], kw_src[WI];
    int ld_tile = me, ld_kt = 0, ld_it = 0;
    auto set_sources = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * NWV + wave) * 8 + (lane >> 3);
            const int ca = (lane & 7) ^ ((r >> 1) & 7);
            int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
            a_src[j] = p.A + (long)gm * p.lda + ca * 8;
            ka_src[j] = ca * 8;
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int r = (j * NWV + wave) * 8 + (lane >> 3);
            const int cw = (lane & 7) ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1));
            int gn = n0 + r; gn = gn < p.N ? gn : p.N - 1;
            w_src[j] = p.W + (long)gn * p.ldw + cw * 8;
            kw_src[j] = cw * 8;
        }
    };
    auto issue = [&]() {  // loads of ring iteration ld_it
        char* sa = psmem + (ld_it % STAGES) * STAGE;
        char* sw = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const void* as = ld_kt * BKT + ka_src[j] < p.K ? (const void*)(a_src[j] + ld_kt * BKT) : (const void*)g_zero_page;
            glds16(as, sa + (j * NWV + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const void* ws = ld_kt * BKT + kw_src[j] < p.K ? (const void*)(w_src[j] + ld_kt * BKT) : (const void*)g_zero_page;
            glds16(ws, sw + (j * NWV + wave) * 1024);
        }
        ++ld_it;
        if (++ld_kt == nk) {
            ld_kt = 0;
            ld_tile += G;
            if (ld_tile < ntiles) set_sources(ld_tile);
        }
    };

    if (total == 0) return;
    set_sources(ld_tile);
    // prologue: fill STAGES-1 stages, wait for the first
    issue();
    if (STAGES == 3 && total > 1) {
        issue();
        wait_vmcnt<LOADS>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int tile = me, kt = 0;
    for (int it = 0; it < total; ++it) {
        const bool more = ld_it < total;
        if (more) issue();  // ring iteration it+STAGES-1 -> the stage last read in iteration it-1
        const char* sa = psmem + (it % STAGES) * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ((ks * 4 + fq) ^ swz) << 4;
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = *(const bf16x8*)(sw + (wrow + 32 * (t >> 1) + 4 * (t & 1)) * 128 + chunk);
                af[t] = *(const bf16x8*)(sa + (wm * 64 + t * 16 + fr) * 128 + chunk);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        if (++kt == nk) {
            int m0, n0;
            tile_origin(tile, m0, n0);
            if (m0 + TBM <= p.M && n0 + BN <= p.N) store_wave_tile<EPI, true>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fq);
            else store_wave_tile<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fq);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kt = 0;
            tile += G;
        }
        // everything older than the newest STAGES-2 stages of loads must have landed (in particular iteration it+1);
        // the LDS reads above were consumed by the MFMAs, so after the barrier stage it % STAGES may be refilled
        if (STAGES == 3 && more) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}


// ------------------------------------------------------------------------------------------
// 256x256x64 persistent kernel, eight waves (2 along M x 4 along N, 128x64 outputs each = 128 accumulator registers),
// one workgroup per CU.  Each k-tile is staged as four 16-KiB half-tiles ("slots") by 16-byte LDS-DMA, two instructions
// per thread and slot, one slot per phase, seven half-tiles ahead of the phase that computes, with a counted vmcnt once
// per k-tile and raw s_barriers so the loads stay in flight across barriers.  A k-tile is computed in four phases, one
// 64x32 quadrant of the wave's block each (16 MFMAs):
//   phase 0: read A rows 0-63 (slot 0, 8 reads) and W rows 0-31 (slot 1, 4 reads)   -> quadrant (A-lo, W-lo)
//   phase 1: read W rows 32-63 (slot 2, 4 reads)                                     -> (A-lo, W-hi)
//   phase 2: read A rows 64-127 (slot 3, 8 reads)                                    -> (A-hi, W-hi)
//   phase 3: no reads (W-lo stays in registers)                                      -> (A-hi, W-lo)
// so slot s is last read in phase <= s and can be refilled from phase s + 1 on: phase q stages slot (q + 3) % 4.
// The two wave groups (wm = 0 / 1, one wave of each per SIMD) run one barrier apart: while one group issues its LDS
// reads and loads, the other runs its MFMA cluster.
// Slot images: 128 rows x 128 B; slot 0 / 3 hold activation rows {wm*128 + (0|64) + 0..63}, slot 1 / 2 weight rows
// {wn*64 + (0|32) + 0..31}; 16-byte chunk c of row r sits at c ^ swz(r) (same two swizzles as the 128x128 kernel).
// Requires K % 128 == 0 (k-tiles come in pairs, one per LDS buffer).
// Tile end: per-tile timestamps (the MMAE_PP_TRACE build, scripts/pp_trace.py) show the time between main loops to be the
// VALU work of the epilogue, not the stores: the counted wait after it costs 0.04 us, and spreading the start of the
// workgroups over a whole tile time (so that CUs do not store together) changes no per-tile figure.  What mattered: the two
// groups' epilogues running side by side (see the barrier at the tile end), and the instruction count per value.
// HALF: 128x256 tiles for the tail of a problem.  A workgroup owns a CU, so a last round with few tiles leaves most CUs
// idle for a whole tile time; the launcher may instead give the last row panels to a second launch of this instance, in
// which slot 0 holds 128 contiguous rows (64 per wave group, one 64x64 block per wave), slot 3 and the second
// accumulator half do not exist and a k-tile has two phases: a tile costs a little over half a tile time and twice as
// many CUs share the tail.
// ------------------------------------------------------------------------------------------
#define PP_H_BYTES 16384
#define PP_KT_BYTES (4 * PP_H_BYTES)
#define PP_LDS (2 * PP_KT_BYTES + 8 * 256 + 8 * 2048)  // + per-wave bias slice (64 floats) + per-wave epilogue piece

#ifdef MMAE_PP_TRACE
// timing build only: per workgroup and wave group, eight 100 MHz timestamps per tile: 0 end of main loop, 1 stores issued,
// then in the next tile's first k-tile: 2 past the first barrier, 3 before / 4 after the counted wait of phase 3, 5 k-tile done;
// 7 (first record only) start of the main loop
__device__ unsigned long long g_pp_trace[256 * 2 * 128];
extern "C" int mmae_debug_pp_trace(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_trace), sizeof(g_pp_trace)) == hipSuccess ? 0 : 1;
}
#define PP_TRACE(slot) do { if (!HALF && lane == 0 && wn == 0 && me < 256 && trace_tile < 16) g_pp_trace[(me * 2 + wm) * 128 + trace_tile * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PP_TRACE(slot) do {} while (0)
#endif
template <int EPI, int TM>
__global__ __launch_bounds__(512, 1) void gemm_nt_pp_kernel(GemmNT p, int tiles_m, int tiles_n, int gemm_blocks, LnJob job) {
    constexpr bool HALF = TM < 256;      // 128- and 64-row tiles: two phases per k-tile, one accumulator block per wave
    constexpr bool QUARTER = TM == 64;   // 64-row tiles: the wave's block is 32 x 64 (two activation fragments)
    constexpr int NMI = QUARTER ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) char ppsmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups past the GEMM's own (tail launches of the residual epilogue only): a tail of 64-row tiles occupies two
    // thirds of the CUs (N = 768 at 25344 rows: 168 of 256), and every workgroup of this kernel owns a CU through its LDS
    // reservation, so the rest of the grid lands on the idle CUs and normalises rows that the full-tile launch before
    // this one has already written (the LayerNorm that follows the product in the block), eight waves per CU.
    if constexpr (EPI == EPI_RESID && HALF) {
        if ((int)blockIdx.x >= gemm_blocks) {
            const int first = ((int)blockIdx.x - gemm_blocks) * 8 + wave, nwaves = ((int)gridDim.x - gemm_blocks) * 8;
            if (job.dim == 768) ln_fwd_rows<3, MMAE_LN_TAIL_R>(job.x, job.gamma, job.beta, job.y, job.mean, job.rstd, job.rows, job.eps, first * MMAE_LN_TAIL_R, nwaves * MMAE_LN_TAIL_R, lane);
            else if (job.dim == 1024) ln_fwd_rows<4, MMAE_LN_TAIL_R * 3 / 4>(job.x, job.gamma, job.beta, job.y, job.mean, job.rstd, job.rows, job.eps, first * (MMAE_LN_TAIL_R * 3 / 4), nwaves * (MMAE_LN_TAIL_R * 3 / 4), lane);
            return;
        }
    }
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4, swz = fr >> 1;
    const int ntiles = tiles_m * tiles_n, G = gemm_blocks;
    const int me = xcd_remap(blockIdx.x, G);
    const int nk = p.K >> 6;
    if (me >= ntiles) return;
    const int my_tiles = (ntiles - 1 - me) / G + 1;
    const int total_kt = my_tiles * nk;
    float* bias_lds = (float*)(ppsmem + 2 * PP_KT_BYTES) + wave * 64;
    if (!p.bias) bias_lds[lane] = 0.f;

    auto tile_origin = [&](int t, int& m0, int& n0) {
        if (HALF) {  // column tiles fastest: neighbouring workgroups share the 128 activation rows
            const int hp = t / tiles_n;
            m0 = hp * TM;
            n0 = (t - hp * tiles_n) * 256;
            return;
        }
        // groups of 4 row-tiles x all column-tiles, rows fastest
        const int per_group = 4 * tiles_n;
        const int grp = t / per_group, within = t - grp * per_group;
        int rows = tiles_m - grp * 4; rows = rows < 4 ? rows : 4;
        m0 = (grp * 4 + within % rows) * 256;
        n0 = (within / rows) * 256;
    };

    // ---- staging cursors: one per slot (slot 3 runs one k-tile behind slots 0-2) ----
    const bf16* src[4][2];
    int s_kt[4], s_tile[4];
    auto set_slot = [&](auto slot_c, int t) {
        constexpr int s = decltype(slot_c)::value;
        int m0, n0;
        tile_origin(t, m0, n0);
        // the lane's row terms are recomputed from an opaque copy of the lane id at every tile change: kept across the main
        // loop they cost registers the loop does not have, and a spilled one comes back through a scratch load whose
        // wait drains the LDS-DMA in flight
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (j * 8 + wave) * 8 + (ln >> 3);
            if (s == 0 || s == 3) {
                // 64-row tiles: the second copy instruction of the slot fetches the same 64 rows again (into image rows that
                // are never read), so that every slot stays two instructions and the counted waits hold for every tile size
                int g = m0 + (QUARTER ? (r & 63) : HALF ? r : (r >> 6) * 128 + (s == 3 ? 64 : 0) + (r & 63));
                g = g < p.M ? g : p.M - 1;
                src[s][j] = p.A + (long)g * p.lda + (((ln & 7) ^ ((r >> 1) & 7)) << 3);
            } else {
                int g = n0 + (r >> 5) * 64 + (s == 2 ? 32 : 0) + (r & 31);
                g = g < p.N ? g : p.N - 1;
                src[s][j] = p.W + (long)g * p.ldw + (((ln & 7) ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1))) << 3);
            }
        }
    };
    auto stage_slot = [&](auto slot_c) {
        constexpr int s = decltype(slot_c)::value;
        if (HALF && s == 3) return;
        char* dst = ppsmem + (s_kt[s] & 1) * PP_KT_BYTES + s * PP_H_BYTES + wave * 1024;
        glds16(src[s][0], dst);
        glds16(src[s][1], dst + 8 * 1024);
        if (s_tile[s] < ntiles) {
            src[s][0] += 64; src[s][1] += 64;
            if (++s_kt[s] == nk) {
                s_kt[s] = 0;
                s_tile[s] += G;
                if (s_tile[s] < ntiles) {
                    set_slot(slot_c, s_tile[s]);
                } else {  // end of this workgroup's stream: the run-ahead copies read the zero page (never consumed)
                    src[s][0] = src[s][1] = (const bf16*)g_zero_page;
                }
            }
        } else {
            s_kt[s] ^= 1;  // keep alternating buffers so the write lands in the slot that is free
        }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
#pragma unroll
    for (int s = 0; s < 4; ++s) { s_kt[s] = 0; s_tile[s] = me; }
    set_slot(S0{}, me); set_slot(S1{}, me); set_slot(S2{}, me);
    if (!HALF) set_slot(S3{}, me);

    // ---- prologue: 7 half-tiles in flight, the first k-tile landed (HALF: 5 staged, slot 2 one k-tile behind) ----
    stage_slot(S0{}); stage_slot(S1{}); stage_slot(S2{}); stage_slot(S3{});
    stage_slot(S0{}); stage_slot(S1{});
    if (HALF) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        stage_slot(S2{});
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    f32x4 acc[HALF ? 1 : 2][4][4];  // [A half][weight fragment][activation fragment]
#pragma unroll
    for (int h = 0; h < (HALF ? 1 : 2); ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 af[4][2], wlo[2][2], whi[2][2];
    const int a_off = (wm * (QUARTER ? 32 : 64) + fr) * 128;                // + mi * 2048
    const int w_off = (wn * 32 + 8 * (fr >> 2) + (fr & 3)) * 128;           // + t * 512
    const int ch0 = ((fq ^ swz) << 4), ch1 = (((4 + fq) ^ swz) << 4);

    auto read_a = [&](const char* slot) {
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            af[mi][0] = *(const bf16x8*)(slot + a_off + mi * 2048 + ch0);
            af[mi][1] = *(const bf16x8*)(slot + a_off + mi * 2048 + ch1);
        }
    };
    auto read_w = [&](const char* slot, bf16x8 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            w[t][0] = *(const bf16x8*)(slot + w_off + t * 512 + ch0);
            w[t][1] = *(const bf16x8*)(slot + w_off + t * 512 + ch1);
        }
    };
    auto mfma_quad = [&](f32x4 (&c)[4][4], int nbase, const bf16x8 (&w)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
                    c[nbase + t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][ks], af[mi][ks], c[nbase + t][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    if (wm == 1) __builtin_amdgcn_s_barrier();  // second wave group runs one barrier behind

    int tile = me, kt = 0;
#ifdef MMAE_PP_TRACE
    int trace_tile = 0;
    PP_TRACE(7);
    ++trace_tile;
#endif
    for (int T = 0; T < total_kt; T += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const char* buf = ppsmem + b * PP_KT_BYTES;
            if constexpr (HALF) {
                // Two phases per k-tile.  Every LDS read is complete before the phase's first barrier, so a slot may be
                // refilled in the phase after the one that reads it although the wave groups run one barrier apart:
                // phase 1 refills slots 0 / 1 of this buffer (k-tile + 2), the next phase 0 refills slot 2.  Each
                // phase waits for all but its three newest half-tiles: what the next phase reads has then landed.
                // (A third k-tile buffer, i.e. one more k-tile of run-ahead, measured no faster: the phases are bound by
                // LDS reads, 16 per wave and k-tile for half the MFMAs of a full tile.)
                // ---- phase 0 ----
                read_a(buf);
                read_w(buf + PP_H_BYTES, wlo);
                if (kt == 0 && b == 0 && p.bias) {
                    int m0, n0;
                    tile_origin(tile, m0, n0);
                    int n = n0 + wn * 64 + lane; n = n < p.N ? n : p.N - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                                     (__attribute__((address_space(3))) void*)bias_lds, 4, 0, 0);
                }
                stage_slot(S2{});
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                mfma_quad(acc[0], 0, wlo);
                __builtin_amdgcn_s_barrier();
                // ---- phase 1 ----
                read_w(buf + 2 * PP_H_BYTES, whi);
                stage_slot(S0{});
                stage_slot(S1{});
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                mfma_quad(acc[0], 2, whi);
                __builtin_amdgcn_s_barrier();
                continue;
            }
            // ---- phase 0 ----
            read_a(buf);
            read_w(buf + PP_H_BYTES, wlo);
            if (kt == 0 && b == 0 && p.bias) {  // this tile's bias slice for the wave (256 B), older than every wait below
                int m0, n0;
                tile_origin(tile, m0, n0);
                int n = n0 + wn * 64 + lane; n = n < p.N ? n : p.N - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                                 (__attribute__((address_space(3))) void*)bias_lds, 4, 0, 0);
            }
            stage_slot(S3{});
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // the slot-0 reads are done before the barrier: slot 0 is refilled next phase
            __builtin_amdgcn_s_barrier();
#ifdef MMAE_PP_TRACE
            if (kt == 0 && b == 0) PP_TRACE(-6);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[0], 0, wlo);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1 ----
            read_w(buf + 2 * PP_H_BYTES, whi);
            stage_slot(S0{});
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[0], 2, whi);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2 ----
            read_a(buf + 3 * PP_H_BYTES);
            stage_slot(S1{});
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[HALF ? 0 : 1], 2, whi);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3 ----
            stage_slot(S2{});
#ifdef MMAE_PP_TRACE
            if (kt == 0 && b == 0) PP_TRACE(-5);
#endif
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the newest three half-tiles: the next k-tile has landed
#ifdef MMAE_PP_TRACE
            if (kt == 0 && b == 0) PP_TRACE(-4);
#endif
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[HALF ? 0 : 1], 0, wlo);
            __builtin_amdgcn_s_barrier();
#ifdef MMAE_PP_TRACE
            if (kt == 0 && b == 0) PP_TRACE(-3);  // slot 5 of the previous tile's record
#endif
        }
        kt += 2;
        if (kt == nk) {
            // The second wave group is one barrier behind: its last MFMA cluster of the tile sits between two barriers that
            // this group would only reach after its epilogue (the first of the next tile), and this group would then wait
            // in the next tile's first phase for the second group's epilogue: two epilogues per tile, one after the other.
            // One more barrier on each side (here, and after the second group's epilogue) releases the second group at once,
            // so the two epilogues run side by side, and leaves the groups one barrier apart again for the next tile.
            if (wm == 0) __builtin_amdgcn_s_barrier();
            PP_TRACE(0);
            int m0, n0;
            tile_origin(tile, m0, n0);
            float bv[16];  // the wave's bias slice (zeros without a bias: written once at the start)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 b0 = *(const f32x4*)(bias_lds + 32 * u + 8 * fq), b1 = *(const f32x4*)(bias_lds + 32 * u + 8 * fq + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[8 * u + e] = b0[e]; bv[8 * u + 4 + e] = b1[e]; }
            }
            const bool full = m0 + TM <= p.M && n0 + 256 <= p.N;  // workgroup-uniform
            const int rb = m0 + wm * (TM / 2);                    // first row of this wave's block
            constexpr int NH2 = HALF ? 1 : 2;
#ifdef MMAE_PP_ACC_EPILOGUE  // timing builds: stores straight from the accumulator layout
            constexpr bool ROWS = false;
#else
            constexpr bool ROWS = EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_RESID || EPI == EPI_DGELU || EPI == EPI_F32;
#endif
            if constexpr (ROWS) {
                if (full) {  // whole rows per access, through the wave's LDS piece
                    char* stage = ppsmem + 2 * PP_KT_BYTES + 8 * 256 + wave * 2048;
                    if constexpr (EPI == EPI_DGELU) {
                        // the saved pre-activations of the whole 128x64 block first, in the row layout: one wait for them
                        // (and for the loads still in flight ahead of them) instead of one per piece
                        bf16x8 hs[NH2][4][2];
                        const bf16* hp = (const bf16*)p.aux + (long)(rb + (lane >> 3)) * p.ldaux + n0 + wn * 64 + 8 * (lane & 7);
#pragma unroll
                        for (int h2 = 0; h2 < NH2; ++h2)
#pragma unroll
                            for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    hs[h2][mi][j] = *(const bf16x8*)(hp + (long)(h2 * 64 + mi * 16 + 8 * j) * p.ldaux);
                        store_wave_tile_rows<EPI, NMI>(p, acc[0], rb, n0 + wn * 64, lane, stage, bv, hs[0]);
                        if (!HALF) store_wave_tile_rows<EPI>(p, acc[NH2 - 1], rb + 64, n0 + wn * 64, lane, stage, bv, hs[NH2 - 1]);
                    } else {
                        store_wave_tile_rows<EPI, NMI>(p, acc[0], rb, n0 + wn * 64, lane, stage, bv);
                        if (!HALF) store_wave_tile_rows<EPI>(p, acc[NH2 - 1], rb + 64, n0 + wn * 64, lane, stage, bv);
                    }
                } else {
                    store_wave_tile<EPI, false, !(EPI == EPI_DGELU), false, NMI>(p, acc[0], rb, n0 + wn * 64, fr, fq, bv);
                    if (!HALF) store_wave_tile<EPI, false, !(EPI == EPI_DGELU)>(p, acc[NH2 - 1], rb + 64, n0 + wn * 64, fr, fq, bv);
                }
            } else if (full) {
                store_wave_tile<EPI, true, true, HALF, NMI>(p, acc[0], rb, n0 + wn * 64, fr, fq, bv);
                if (!HALF) store_wave_tile<EPI, true, true>(p, acc[NH2 - 1], rb + 64, n0 + wn * 64, fr, fq, bv);
            } else {
                store_wave_tile<EPI, false, true, false, NMI>(p, acc[0], rb, n0 + wn * 64, fr, fq, bv);
                if (!HALF) store_wave_tile<EPI, false, true>(p, acc[NH2 - 1], rb + 64, n0 + wn * 64, fr, fq, bv);
            }
#pragma unroll
            for (int h = 0; h < NH2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kt = 0;
            tile += G;
            PP_TRACE(1);
            if (wm == 1) __builtin_amdgcn_s_barrier();
#ifdef MMAE_PP_TRACE
            ++trace_tile;
#endif
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // pairs with the extra barrier of the second group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead loads past the end of the stream
}

template <int EPI>
static int launch_nt_pp(const GemmNT& p, hipStream_t s, const LnJob* ln = nullptr, int* ln_rows_done = nullptr);
// For timing the two launches of the ping-pong path separately: 0 = both (default), 1 = the full-tile launch only,
// 2 = the half-tile launch only.  g_nt_full_rows: rows covered by full tiles in this thread's last gemm_nt call.
static thread_local int g_nt_phase = 0, g_nt_full_rows = 0, g_nt_tail_rows = 0;

static int g_num_cus = 0;
static int num_cus();

// Tile plan of the ping-pong kernel: full_panels row panels of 256 rows as 256x256 tiles, then (second launch) the rows
// after them as 128x256 half tiles.  rounds = tile times the two launches take together, a round of half tiles counted
// as PP_HALF_COST (measured 0.72-0.78 at K = 3072: a half tile reads 2/3 of the LDS bytes of a full one for half the
// MFMAs, and the second launch has its own start-up, which is why short reductions, K < 512, keep full tiles).  The last
// row panels go to half tiles when that lowers rounds: typically when the last round of full tiles would occupy at most
// half of the CUs.
#define PP_HALF_COST 0.75
// A round of 64-row tiles (two activation fragments per wave for the same weight reads as a half tile): taken instead of
// half tiles where it lets every remaining row panel run at once on CUs that a round of half tiles would leave idle
// (N = 768 at M = 25344: 168 workgroups of 64 rows instead of 84 of 128).
#define PP_QUARTER_COST 0.55
struct PpPlan { int full_panels, tiles_n, halves, tail_rows; double rounds; };  // tail_rows: 128 or 64 rows per tail tile
static int pp_tails_default() {  // env MMAE_PP_TAILS overrides the default of mmae_gemm_nt_set_half_tiles (same-box comparisons)
    const char* e = getenv("MMAE_PP_TAILS");
    return e ? atoi(e) : 1;
}
static thread_local int g_pp_halves = pp_tails_default();  // 0: no tail tiles; 1: half or quarter tiles by cost; 2: half tiles only
static PpPlan pp_plan(int M, int N, int K) {
    const int G = num_cus();
    const int tm = (M + 255) / 256, tn = (N + 255) / 256;
    const int total = tm * tn;
    PpPlan pl = {tm, tn, 0, 128, (double)((total + G - 1) / G)};
    const int rem = total % G;  // full tiles of the last round (0: that round is full)
    if (!g_pp_halves || rem == 0 || K < 512) return pl;
    int cut = (rem + tn - 1) / tn;  // row panels to cut: the full tiles then end at or before a round boundary
    if (cut > tm) cut = tm;
    const int full_panels = tm - cut;
    const int halves = ((M - full_panels * 256 + 127) / 128) * tn;
    const double base = (double)((full_panels * tn + G - 1) / G);
    const double r = base + PP_HALF_COST * (double)((halves + G - 1) / G);
    if (r < pl.rounds) { pl.full_panels = full_panels; pl.halves = halves; pl.tail_rows = 128; pl.rounds = r; }
    if (g_pp_halves == 1) {
        const int quarters = ((M - full_panels * 256 + 63) / 64) * tn;
        const double rq = base + PP_QUARTER_COST * (double)((quarters + G - 1) / G);
        if (rq < pl.rounds) { pl.full_panels = full_panels; pl.halves = quarters; pl.tail_rows = 64; pl.rounds = rq; }
    }
    return pl;
}
// CUs the one-workgroup-per-CU kernels plan for.  A workgroup of those kernels needs a whole CU (150 KB of LDS, every vector
// register of its SIMDs), so it cannot share one with a resident workgroup of another kernel: while a collective's kernel
// holds R CUs, a grid planned for all CUs starts R workgroups late, and with one round of tiles per launch those finish a
// whole tile time after the rest.  With a budget of (CUs - R) every workgroup of these kernels finds a free CU at once and
// the collective keeps its own (set by the data-parallel host code while a communicator with more than one rank exists;
// env MMAE_CU_BUDGET for experiments).  0 = all CUs.
static int g_cu_budget = -1;
static int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_num_cus = n;
        else
            g_num_cus = 256;
    }
    if (g_cu_budget < 0) {  // first use: the environment (a negative value there counts down from the device's CUs)
        const char* e = getenv("MMAE_CU_BUDGET");
        const int v = e ? atoi(e) : 0;
        g_cu_budget = v < 0 ? (g_num_cus + v > 0 ? g_num_cus + v : 0) : v;
    }
    return (g_cu_budget >= 32 && g_cu_budget < g_num_cus) ? g_cu_budget : g_num_cus;
}
extern "C" int mmae_get_cu_budget(void) { return num_cus(); }
extern "C" int mmae_set_cu_budget(int n) {
    g_cu_budget = n > 0 ? n : 0;
    return num_cus();
}

template <int EPI, int WM, int STAGES>
static int launch_nt_persist(const GemmNT& p, hipStream_t s) {
    static bool reserved = false;
    constexpr int lds = STAGES * (WM * 64 * 128 + 16384);
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_persist_kernel<EPI, WM, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            mmae_set_error("gemm_nt: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
            return MMAE_ELAUNCH;
        }
        reserved = true;
    }
    const int tm = (p.M + WM * 64 - 1) / (WM * 64), tn = (p.N + BN - 1) / BN;
    int grid = tm * tn;
    const int slots = num_cus() * (lds <= 80 * 1024 ? 2 : 1);
    if (grid > slots) grid = slots;
    hipLaunchKernelGGL((gemm_nt_persist_kernel<EPI, WM, STAGES>), dim3(grid), dim3(WM * 128), lds, s, p, tm, tn);
    MMAE_CHECK_LAUNCH("gemm_nt_persist");
    return MMAE_OK;
}

// ln / ln_rows_done (residual epilogue): the LayerNorm of the product's output.  When the plan has a tail launch that leaves
// CUs idle, the rows the full-tile launch covers (or as many of them as the idle CUs get through while the tail runs) are
// normalised by extra workgroups of the tail launch; *ln_rows_done = the number of leading rows taken care of (0: none).
template <int EPI>
static int launch_nt_pp(const GemmNT& p, hipStream_t s, const LnJob* ln, int* ln_rows_done) {
    static bool reserved = false;
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e != hipSuccess) {
            mmae_set_error("gemm_nt: cannot reserve %d bytes of LDS: %s", PP_LDS, hipGetErrorString(e));
            return MMAE_ELAUNCH;
        }
        reserved = true;
    }
    // the row-mapped epilogues address output rows through tables of the whole problem: no second launch on a row offset
    PpPlan pl = pp_plan(p.M, p.N, p.K);
    if (EPI == EPI_TOKENS || EPI == EPI_UNPATCH) pl = {(p.M + 255) / 256, (p.N + 255) / 256, 0, 128, 0.0};
    g_nt_full_rows = pl.halves ? pl.full_panels * 256 : p.M;
    g_nt_tail_rows = pl.halves ? pl.tail_rows : 0;
    if (pl.full_panels > 0 && g_nt_phase != 2) {
        GemmNT q = p;
        if (pl.halves) q.M = pl.full_panels * 256;
        int grid = pl.full_panels * pl.tiles_n;
        if (grid > num_cus()) grid = num_cus();
        hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, 256>), dim3(grid), dim3(512), PP_LDS, s, q, pl.full_panels, pl.tiles_n, grid, LnJob{});
        MMAE_CHECK_LAUNCH("gemm_nt_pp");
    }
    if (ln_rows_done) *ln_rows_done = 0;
    if (pl.halves && g_nt_phase != 1) {
        // the rows after the full panels, as a problem of its own
        const long r0 = (long)pl.full_panels * 256;
        constexpr long out_sz = (EPI == EPI_RESID || EPI == EPI_F32) ? 4 : 2;
        constexpr long aux_sz = EPI == EPI_RESID ? 4 : 2;
        GemmNT q = p;
        q.A = p.A + r0 * p.lda;
        q.M = p.M - (int)r0;
        q.out = (char*)p.out + r0 * p.ldo * out_sz;
        if (p.out2) q.out2 = (char*)p.out2 + r0 * p.ldo * out_sz;
        if (p.aux) q.aux = (const char*)p.aux + r0 * p.ldaux * aux_sz;
        int grid = pl.halves;
        if (grid > num_cus()) grid = num_cus();
        int total = grid;
        LnJob job = {};
        if constexpr (EPI == EPI_RESID) {
            // idle CUs beside the tail: they normalise leading rows of the output (complete since the full-tile launch).
            // One CU streams about 26 GB/s whatever is in flight, so the 88 idle CUs of the N = 768 tail get through about
            // half of the 21760 complete rows while the tail runs (scripts/bench_lnfuse.py: with every row the launch
            // grows by 12 us, with half by 5); the LayerNorm launch that follows takes the rest.
            static double env_share = -1.0;
            if (env_share < 0.0) {
                const char* e = getenv("MMAE_LN_TAIL_SHARE");
                env_share = e ? atof(e) : 0.5;
            }
            const int idle = num_cus() - grid;
            if (ln && ln->rows > 0 && idle >= 32 && g_nt_phase == 0 && env_share > 0.0 && (ln->dim == 768 || ln->dim == 1024)) {
                job = *ln;
                long rows = (long)(env_share * (double)r0);
                rows = rows < r0 ? rows : r0;
                job.rows = (int)(rows / 8 * 8);
                if (job.rows > 0) {
                    total = num_cus();
                    if (ln_rows_done) *ln_rows_done = job.rows;
                }
            }
        }
        if (pl.tail_rows == 64)
            hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, 64>), dim3(total), dim3(512), PP_LDS, s, q, (q.M + 63) / 64, pl.tiles_n, grid, job);
        else
            hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, 128>), dim3(total), dim3(512), PP_LDS, s, q, (q.M + 127) / 128, pl.tiles_n, grid, job);
        MMAE_CHECK_LAUNCH("gemm_nt_pp(half tiles)");
    }
    return MMAE_OK;
}


// ------------------------------------------------------------------------------------------
// 128x256x64 persistent kernel, four waves (each 128 rows x 64 columns = 128 accumulator registers, the wave tile of the
// 256x256 kernel), TWO workgroups per CU.  The 256x256 kernel keeps one workgroup per CU, so while its eight waves store a
// tile the matrix cores of the CU idle (a quarter to a half of the tile time for the epilogues that move a second large
// matrix).  Here the two workgroups of a CU are independent: one stores while the other multiplies, and the per-SIMD
// pairing "one wave reads LDS / loads, the other runs its MFMA cluster" comes from the hardware arbiter instead of a
// barrier protocol.  Price: 48 KB of operands per k-tile for a 128x256 tile against 64 KB for 256x256 (1.5x the L2 -> LDS
// bytes per FLOP), and a main loop that runs at about 0.7 of the 256x256 kernel's rate (K = 2304, bias-only epilogue: 126 us
// against 91 us); starting the second workgroup of each CU late, or giving the two unequal issue priorities during their
// MFMA clusters, changed nothing.  So it pays where the tile time is mostly epilogue: short reductions under the GELU
// epilogues (K = 256).
// LDS per workgroup (<= 80 KiB so two fit): a ring of three 8-KiB activation slots (64 rows x 128 B) and a ring of three
// 16-KiB weight slots (4 waves x 32 rows); k-tile T uses activation slots 2T, 2T+1 (rows 0-63, 64-127) and weight slots
// 2T, 2T+1 (columns 0-31, 32-63 of every wave's 64) modulo 3.  A k-tile is four phases as in the 256x256 kernel:
//   phase 0: read A-lo(T), W-lo(T)                          -> quadrant (A-lo, W-lo)
//   phase 1: read W-hi(T);  refill: A-hi(T+1), W-hi(T+1)    -> (A-lo, W-hi)        (slots read in phase 0)
//   phase 2: read A-hi(T);  refill: W-lo(T+2)               -> (A-hi, W-hi)        (slot read in phase 1)
//   phase 3:                refill: A-lo(T+2)               -> (A-hi, W-lo)        (slot read in phase 2)
// one s_barrier per phase: a slot is refilled in the phase after the one that read it, and read in a phase after the
// counted vmcnt of every wave says its copy has landed (12 copy instructions per thread and k-tile: 2 per activation
// slot, 4 per weight slot; in-order completion).  The ring runs on across tile boundaries.
// ------------------------------------------------------------------------------------------
#define P2_A_BYTES 8192
#define P2_W_BYTES 16384
#define P2_LDS (3 * P2_A_BYTES + 3 * P2_W_BYTES + 4 * 256)  // + per-wave bias slice (64 floats)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_p2_kernel(GemmNT p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char p2smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;
    const int fr = lane & 15, fq = lane >> 4, swz = fr >> 1;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int me = xcd_remap(blockIdx.x, G);
    const int nk = p.K >> 6;
    if (me >= ntiles) return;
    const int my_tiles = (ntiles - 1 - me) / G + 1;
    const int total_kt = my_tiles * nk;
    char* const a_ring = p2smem;
    char* const w_ring = p2smem + 3 * P2_A_BYTES;
    float* bias_lds = (float*)(p2smem + 3 * P2_A_BYTES + 3 * P2_W_BYTES) + wave * 64;
    if (!p.bias) bias_lds[lane] = 0.f;

    auto tile_origin = [&](int t, int& m0, int& n0) {  // groups of 8 row-tiles x all column-tiles, rows fastest
        const int per_group = 8 * tiles_n;
        const int grp = t / per_group, within = t - grp * per_group;
        int rows = tiles_m - grp * 8; rows = rows < 8 ? rows : 8;
        m0 = (grp * 8 + within % rows) * 128;
        n0 = (within / rows) * 256;
    };

    // ---- copy cursors: 0 = A-lo, 1 = A-hi (2 instructions per thread), 2 = W-lo, 3 = W-hi (4 per thread) ----
    const bf16* asrc[2][2];
    const bf16* wsrc[2][4];
    int c_kt[4], c_tile[4], c_slot[4];
    auto set_cursor = [&](auto cur_c, int t) {
        constexpr int c = decltype(cur_c)::value;
        int m0, n0;
        tile_origin(t, m0, n0);
        int ln = lane;
        asm volatile("" : "+v"(ln));  // recomputed at every tile change instead of living in registers across the main loop
        if constexpr (c < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (j * 4 + wave) * 8 + (ln >> 3);
                int g = m0 + c * 64 + r;
                g = g < p.M ? g : p.M - 1;
                asrc[c][j] = p.A + (long)g * p.lda + (((ln & 7) ^ ((r >> 1) & 7)) << 3);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (j * 4 + wave) * 8 + (ln >> 3);
                int g = n0 + (r >> 5) * 64 + (c - 2) * 32 + (r & 31);
                g = g < p.N ? g : p.N - 1;
                wsrc[c - 2][j] = p.W + (long)g * p.ldw + (((ln & 7) ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1))) << 3);
            }
        }
    };
    auto issue = [&](auto cur_c) {  // copy the cursor's next k-tile into its ring slot, then advance the cursor
        constexpr int c = decltype(cur_c)::value;
        if constexpr (c < 2) {
            char* dst = a_ring + c_slot[c] * P2_A_BYTES + wave * 1024;
            glds16(asrc[c][0], dst);
            glds16(asrc[c][1], dst + 4 * 1024);
        } else {
            char* dst = w_ring + c_slot[c] * P2_W_BYTES + wave * 1024;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(wsrc[c - 2][j], dst + j * 4 * 1024);
        }
        c_slot[c] = c_slot[c] == 0 ? 2 : c_slot[c] - 1;  // + 2 modulo 3
        if (c_tile[c] < ntiles) {
            if constexpr (c < 2) { asrc[c][0] += 64; asrc[c][1] += 64; }
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) wsrc[c - 2][j] += 64;
            }
            if (++c_kt[c] == nk) {
                c_kt[c] = 0;
                c_tile[c] += G;
                if (c_tile[c] < ntiles) {
                    set_cursor(cur_c, c_tile[c]);
                } else {  // end of this workgroup's stream: the run-ahead copies read the zero page (never consumed)
                    if constexpr (c < 2) { asrc[c][0] = asrc[c][1] = (const bf16*)g_zero_page; }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) wsrc[c - 2][j] = (const bf16*)g_zero_page;
                    }
                }
            }
        }
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
#pragma unroll
    for (int c = 0; c < 4; ++c) { c_kt[c] = 0; c_tile[c] = me; }
    c_slot[0] = 0; c_slot[1] = 1; c_slot[2] = 0; c_slot[3] = 1;
    set_cursor(C0{}, me); set_cursor(C1{}, me); set_cursor(C2{}, me); set_cursor(C3{}, me);

    // ---- prologue, in the order the main loop issues: W-lo(0) A-lo(0) | A-hi(0) W-hi(0) | W-lo(1) | A-lo(1) ----
    issue(C2{}); issue(C0{}); issue(C1{}); issue(C3{}); issue(C2{}); issue(C0{});
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // everything but the 12 newest instructions: W-lo(0), A-lo(0) landed
    __builtin_amdgcn_s_barrier();

    f32x4 acc[2][4][4];  // [A half][weight fragment][activation fragment]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 af[4][2], wlo[2][2], whi[2][2];
    const int a_off = fr * 128;                                             // + mi * 2048
    const int w_off = (wn * 32 + 8 * (fr >> 2) + (fr & 3)) * 128;           // + t * 512
    const int ch0 = ((fq ^ swz) << 4), ch1 = (((4 + fq) ^ swz) << 4);
    auto read_a = [&](const char* slot) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(slot + a_off + mi * 2048 + ch0);
            af[mi][1] = *(const bf16x8*)(slot + a_off + mi * 2048 + ch1);
        }
    };
    auto read_w = [&](const char* slot, bf16x8 (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            w[t][0] = *(const bf16x8*)(slot + w_off + t * 512 + ch0);
            w[t][1] = *(const bf16x8*)(slot + w_off + t * 512 + ch1);
        }
    };
    auto mfma_quad = [&](f32x4 (&c)[4][4], int nbase, const bf16x8 (&w)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    c[nbase + t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][ks], af[mi][ks], c[nbase + t][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    int tile = me, kt = 0;
    int ra = 0, rw = 0;  // ring slots of A-lo(T) / W-lo(T); the -hi slots are the next ones
    for (int T = 0; T < total_kt; ++T) {
        const int ra1 = ra == 2 ? 0 : ra + 1, rw1 = rw == 2 ? 0 : rw + 1;
        // ---- phase 0 ----
        read_a(a_ring + ra * P2_A_BYTES);
        read_w(w_ring + rw * P2_W_BYTES, wlo);
        if (kt == 0 && p.bias) {  // this tile's bias slice for the wave (256 B)
            int m0, n0;
            tile_origin(tile, m0, n0);
            int n = n0 + wn * 64 + lane; n = n < p.N ? n : p.N - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                             (__attribute__((address_space(3))) void*)bias_lds, 4, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad(acc[0], 0, wlo);
        // W-hi(T) (and the older A-hi(T)) have landed once at most the copies issued after them are in flight:
        // W-lo(T+1) (4), A-lo(T+1) (2), and possibly this phase's bias copy
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 1 ----
        issue(C1{});  // A-hi(T+1) -> the slot A-lo(T) was read from
        issue(C3{});  // W-hi(T+1) -> the slot W-lo(T) was read from
        read_w(w_ring + rw1 * P2_W_BYTES, whi);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad(acc[0], 2, whi);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2 ----
        issue(C2{});  // W-lo(T+2) -> the slot W-hi(T) was read from
        read_a(a_ring + ra1 * P2_A_BYTES);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad(acc[1], 2, whi);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3 ----
        issue(C0{});  // A-lo(T+2) -> the slot A-hi(T) was read from
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad(acc[1], 0, wlo);
        ra = ra == 0 ? 2 : ra - 1;
        rw = rw == 0 ? 2 : rw - 1;
        if (++kt == nk) {
            int m0, n0;
            tile_origin(tile, m0, n0);
            float bv[16];  // the wave's bias slice (zeros without a bias: written once at the start)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 b0 = *(const f32x4*)(bias_lds + 32 * u + 8 * fq), b1 = *(const f32x4*)(bias_lds + 32 * u + 8 * fq + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[8 * u + e] = b0[e]; bv[8 * u + 4 + e] = b1[e]; }
            }
            const bool full = m0 + 128 <= p.M && n0 + 256 <= p.N;  // workgroup-uniform
            if constexpr (EPI == EPI_DGELU) {
                bf16x8 hs[2][4][2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        int m = m0 + h2 * 64 + mi * 16 + fr;
                        m = m < p.M ? m : p.M - 1;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            int n = n0 + wn * 64 + 32 * u + 8 * fq;
                            n = n < p.N ? n : p.N - 8;
                            hs[h2][mi][u] = *(const bf16x8*)((const bf16*)p.aux + (long)m * p.ldaux + n);
                        }
                    }
                if (full) {
                    store_wave_tile<EPI, true, true>(p, acc[0], m0, n0 + wn * 64, fr, fq, bv, hs[0]);
                    store_wave_tile<EPI, true, true>(p, acc[1], m0 + 64, n0 + wn * 64, fr, fq, bv, hs[1]);
                } else {
                    store_wave_tile<EPI, false, true>(p, acc[0], m0, n0 + wn * 64, fr, fq, bv, hs[0]);
                    store_wave_tile<EPI, false, true>(p, acc[1], m0 + 64, n0 + wn * 64, fr, fq, bv, hs[1]);
                }
            } else if (full) {
                store_wave_tile<EPI, true, true>(p, acc[0], m0, n0 + wn * 64, fr, fq, bv);
                store_wave_tile<EPI, true, true>(p, acc[1], m0 + 64, n0 + wn * 64, fr, fq, bv);
            } else {
                store_wave_tile<EPI, false, true>(p, acc[0], m0, n0 + wn * 64, fr, fq, bv);
                store_wave_tile<EPI, false, true>(p, acc[1], m0 + 64, n0 + wn * 64, fr, fq, bv);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            kt = 0;
            tile += G;
        }
        // A-lo(T+1), W-lo(T+1) have landed once at most this k-tile's own 12 copy instructions are in flight (after an
        // epilogue its stores are newer than those copies: the wait then also covers older stores, never fewer copies)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead copies past the end of the stream
}

template <int EPI>
static int launch_nt_p2(const GemmNT& p, hipStream_t s) {
    static bool reserved = false;
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_p2_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        if (e != hipSuccess) {
            mmae_set_error("gemm_nt: cannot reserve %d bytes of LDS: %s", P2_LDS, hipGetErrorString(e));
            return MMAE_ELAUNCH;
        }
        reserved = true;
    }
    const int tm = (p.M + 127) / 128, tn = (p.N + 255) / 256;
    int grid = tm * tn;
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    hipLaunchKernelGGL((gemm_nt_p2_kernel<EPI>), dim3(grid), dim3(256), P2_LDS, s, p, tm, tn);
    MMAE_CHECK_LAUNCH("gemm_nt_p2");
    return MMAE_OK;
}

static thread_local int g_tn_phase = 0;  // 0 = split GEMM and slab reduction, 1 = GEMM only, 2 = reduction only
extern "C" void mmae_gemm_tn_set_phase(int phase) { g_tn_phase = phase; }
static thread_local int g_last_nt = 0, g_last_tn = 0;  // kernel chosen by this thread's last call (same codes as *_set_variant)
extern "C" int mmae_gemm_nt_last_kernel(void) { return g_last_nt; }
extern "C" int mmae_gemm_tn_last_kernel(void) { return g_last_tn; }

static thread_local int g_nt_variant = 0;  // 0 = choose by size, 1 = 128x128 grid kernel, 4 = persistent 128x128, 5 = 256x256 ping-pong
extern "C" void mmae_gemm_nt_set_variant(int v) { g_nt_variant = v; }
extern "C" void mmae_gemm_nt_set_half_tiles(int on) { g_pp_halves = on; }
extern "C" void mmae_gemm_nt_set_phase(int phase) { g_nt_phase = phase; }
extern "C" int mmae_gemm_nt_last_full_rows(void) { return g_nt_full_rows; }
extern "C" int mmae_gemm_nt_last_tail_rows(void) { return g_nt_tail_rows; }

template <int EPI>
static int launch_nt(const GemmNT& p, hipStream_t s, const LnJob* ln = nullptr, int* ln_rows_done = nullptr) {
    if (ln_rows_done) *ln_rows_done = 0;
    // Measured on MI355X at the shapes of the ViT-B step.  The 256x256 ping-pong kernel runs a k-tile in about 60 % of the
    // time the 128x128 kernels need per FLOP, but a workgroup owns a whole CU, so it pays only where the tile count
    // fills the rounds of CUs, counting a tail of half tiles at its cost (N = 2304 / 3072 at M = 25344: 93 %, the decoder
    // shapes at M = 50176: 77 %; N = 768 at M = 25344: 58 % with full tiles, where it loses, 66 % with half tiles).  Among the
    // 128x128 kernels the persistent ring wins for many tiles of a short K loop (N >= 2048), the plain grid elsewhere.
    const long tiles128 = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    g_nt_full_rows = p.M;
    if (g_nt_variant == 5 && p.K % 128 == 0) { g_last_nt = 5; return launch_nt_pp<EPI>(p, s, ln, ln_rows_done); }
    if constexpr (EPI != EPI_TOKENS && EPI != EPI_UNPATCH) {
        if (g_nt_variant == 7 && p.K % 64 == 0 && p.K >= 128) { g_last_nt = 7; return launch_nt_p2<EPI>(p, s); }
    }
    if constexpr (EPI == EPI_DGELU) {
        // Short reductions under the GELU' epilogue (the decoders' fc1 backward, K = 256): the tile time is mostly epilogue,
        // which the two workgroups per CU of the 128x256 kernel overlap.  Measured with operands that come from HBM, as
        // inside the step (scripts/bench_gemm.py cold), M = 50176, N = 1024: 74.5 us against 77.4 (256x256) and 76.2
        // (128x128).  The forward (GELU, two outputs) is the other way round since the 256x256 kernel runs its two
        // epilogues side by side: 65.7 against 72.3 us.  Long reductions stay with the 256x256 kernel (its main loop is
        // the faster one: K = 768, 140 against 166 us).
        const long tiles_p2 = (long)((p.M + 127) / 128) * ((p.N + 255) / 256);
        if (g_nt_variant == 0 && p.K % 64 == 0 && p.K >= 128 && p.K <= 256 && p.N >= 512 && tiles_p2 >= 2 * num_cus()) {
            g_last_nt = 7;
            return launch_nt_p2<EPI>(p, s);
        }
    }
    if (g_nt_variant == 0 && p.K % 128 == 0 && EPI != EPI_TOKENS && EPI != EPI_UNPATCH) {
        const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        const double fill = (double)tiles256 / (pp_plan(p.M, p.N, p.K).rounds * num_cus());
        static double env_fill = -1.0;
        if (env_fill < 0.0) {
            const char* e = getenv("MMAE_PP_FILL");
            env_fill = e ? atof(e) : 0.0;
        }
        // with the two groups' epilogues side by side the 256x256 kernel wins down to a third of the CUs' rounds filled
        // (M = 25344: N = 768, K = 768, fill 0.66: 57 against 64 us with the residual epilogue, 38 against 42 us without;
        // N = 256, K = 768, fill 0.39: 15 against 17 us)
        const double min_fill = env_fill > 0.0 ? env_fill : 0.35;
        if (fill >= min_fill) { g_last_nt = 5; return launch_nt_pp<EPI>(p, s, ln, ln_rows_done); }
    }
    if (g_nt_variant == 4 || (g_nt_variant == 0 && p.N >= 2048 && tiles128 >= 4 * 2 * num_cus())) {
        g_last_nt = 4;
        return launch_nt_persist<EPI, 2, 2>(p, s);
    }
    g_last_nt = 1;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(tm * tn), dim3(256), 0, s, p);
    MMAE_CHECK_LAUNCH("gemm_nt");
    return MMAE_OK;
}

static int check_nt(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const void* out) {
    MMAE_REQUIRE(A && W && out, MMAE_EINVAL, "gemm_nt: null pointer");
    MMAE_REQUIRE(M > 0 && N > 0 && K > 0, MMAE_EINVAL, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    MMAE_REQUIRE(K % 8 == 0, MMAE_ESHAPE, "gemm_nt: K=%d must be a multiple of 8", K);
    MMAE_REQUIRE(N % 8 == 0, MMAE_ESHAPE, "gemm_nt: N=%d must be a multiple of 8", N);
    MMAE_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, MMAE_ESHAPE,
                 "gemm_nt: row strides (lda=%ld, ldw=%ld) must be >= K and multiples of 8", lda, ldw);
    MMAE_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), MMAE_EINVAL, "gemm_nt: operands must be 16-byte aligned");
    return MMAE_OK;
}

extern "C" int mmae_gemm_nt(int epi, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                            const float* bias, void* out, long ldo, void* out2, const void* aux, long ldaux,
                            hipStream_t stream) {
    int rc = check_nt(A, lda, W, ldw, M, N, K, out);
    if (rc) return rc;
    GemmNT p = {};
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.out = out; p.ldo = ldo; p.out2 = out2; p.aux = aux; p.ldaux = ldaux;
    MMAE_REQUIRE(ldo >= N && ldo % 8 == 0, MMAE_ESHAPE, "gemm_nt: ldo=%ld must be >= N and a multiple of 8", ldo);
    MMAE_REQUIRE((uintptr_t)out % 16 == 0 && (!aux || ((uintptr_t)aux % 16 == 0 && ldaux % 8 == 0)) && (!bias || (uintptr_t)bias % 16 == 0),
                 MMAE_EINVAL, "gemm_nt: out / aux / bias must be 16-byte aligned (aux row stride a multiple of 8)");
    switch (epi) {
        case EPI_BF16: return launch_nt<EPI_BF16>(p, stream);
        case EPI_GELU:
            MMAE_REQUIRE(out2 && bias, MMAE_EINVAL, "gemm_nt(gelu): needs bias and a second output");
            return launch_nt<EPI_GELU>(p, stream);
        case EPI_RESID:
            MMAE_REQUIRE(aux && ldaux >= N, MMAE_EINVAL, "gemm_nt(resid): needs the residual input");
            return launch_nt<EPI_RESID>(p, stream);
        case EPI_DGELU:
            MMAE_REQUIRE(aux && ldaux >= N, MMAE_EINVAL, "gemm_nt(dgelu): needs the saved pre-activation");
            return launch_nt<EPI_DGELU>(p, stream);
        case EPI_F32: return launch_nt<EPI_F32>(p, stream);
        default: break;
    }
    mmae_set_error("gemm_nt: unknown epilogue %d", epi);
    return MMAE_EINVAL;
}

// out(f32) = aux + A W^T + bias, then ln(bf16) = LayerNorm(out) with its row statistics: the residual product of a block and the
// LayerNorm that follows it (multimae/multimae_utils.py:283-284 of the reference: x = x + attn(...); ... norm2(x), and the
// next block's norm1 after fc2).  Same kernels as mmae_gemm_nt(epi 2) + mmae_layernorm_fwd, bitwise the same results; where the
// product ends in a tail launch that leaves CUs idle, those CUs normalise the rows that are already complete meanwhile.
extern "C" int mmae_gemm_nt_resid_ln(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                                     float* out, const float* aux, long ldaux, const float* gamma, const float* beta, float eps,
                                     void* ln_bf16, float* mean, float* rstd, hipStream_t stream) {
    int rc = check_nt(A, lda, W, ldw, M, N, K, out);
    if (rc) return rc;
    MMAE_REQUIRE(aux && ldaux >= N && gamma && beta && ln_bf16 && mean && rstd, MMAE_EINVAL, "gemm_nt_resid_ln: null pointer");
    MMAE_REQUIRE(N % 4 == 0 && N <= 1024, MMAE_ESHAPE, "gemm_nt_resid_ln: N=%d must be a multiple of 4 and <= 1024 (one LayerNorm row)", N);
    MMAE_REQUIRE((uintptr_t)out % 16 == 0 && (uintptr_t)aux % 16 == 0 && ldaux % 8 == 0 && (!bias || (uintptr_t)bias % 16 == 0),
                 MMAE_EINVAL, "gemm_nt_resid_ln: out / aux / bias must be 16-byte aligned (aux row stride a multiple of 8)");
    GemmNT p = {};
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.out = out; p.ldo = N; p.aux = aux; p.ldaux = ldaux;
    const LnJob job = {out, gamma, beta, (bf16*)ln_bf16, mean, rstd, M, N, eps};
    int done = 0;
    rc = launch_nt<EPI_RESID>(p, stream, &job, &done);
    if (rc) return rc;
    if (done >= M) return MMAE_OK;
    return mmae_ln_fwd_launch(out + (long)done * N, gamma, beta, (bf16*)ln_bf16 + (long)done * N, mean + done, rstd + done, M - done, N,
                              eps, stream);
}

extern "C" int mmae_patch_embed_gemm(const void* A, const void* W, int M, int N, int K, const int* add_idx,
                                     const float* add_tab, int seg_in, int seg_out, float* out, long ldo,
                                     hipStream_t stream) {
    int rc = check_nt(A, K, W, K, M, N, K, out);
    if (rc) return rc;
    MMAE_REQUIRE(add_idx && add_tab && seg_in > 0 && seg_out >= seg_in, MMAE_EINVAL, "patch_embed_gemm: bad row map");
    GemmNT p = {};
    p.A = (const bf16*)A; p.lda = K; p.W = (const bf16*)W; p.ldw = K; p.M = M; p.N = N; p.K = K;
    p.out = out; p.ldo = ldo; p.seg_in = seg_in; p.seg_out = seg_out; p.add_idx = add_idx; p.add_tab = add_tab;
    return launch_nt<EPI_TOKENS>(p, stream);
}

extern "C" int mmae_unpatchify_gemm(const void* A, long lda, const void* W, int M, int N, int K, const float* bias,
                                    void* out_image, int out_is_f32, int C, int P, int NH, int NW, hipStream_t stream) {
    int rc = check_nt(A, lda, W, K, M, N, K, out_image);
    if (rc) return rc;
    MMAE_REQUIRE(N == C * P * P && P % 4 == 0 && M % (NH * NW) == 0, MMAE_ESHAPE,
                 "unpatchify_gemm: N=%d must equal C*P*P (C=%d, P=%d a multiple of 4) and M=%d a multiple of NH*NW", N, C, P, M);
    GemmNT p = {};
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = K; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.out = out_image; p.C = C; p.P = P; p.NH = NH; p.NW = NW; p.out_f32 = out_is_f32;
    return launch_nt<EPI_UNPATCH>(p, stream);
}

// ------------------------------------------------------------------------------------------
// gemm_tn: D[N,K] = sum_m X[m,N] * Y[m,K]; both operands are m-major (their rows are the
// reduction index), so fragments are fetched with the transposing LDS read.
// ------------------------------------------------------------------------------------------
#define TN_ROWS 64  // granularity of an M-split (the 256x256 kernel stages 64 rows at a time, the 128x128 kernel 32)

struct GemmTN {
    const bf16* X; long ldx;  // [M,N]
    const bf16* Y; long ldy;  // [M,K]
    int M, N, K;
    int rows_per_split;       // multiple of TN_ROWS
    float* out;               // [splits][N][K] slabs (or the final [N,K] when splits == 1)
    long slab;                // elements per slab
    float* colsum;            // optional: [splits][N] column sums of X (bias gradient), same slab scheme
};

__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// Work item (tile, M-split) of a workgroup of the 1-D grid of 8 * ceil(tiles * splits / 8) workgroups.  Workgroups are
// dealt to the 8 XCDs round-robin, each XCD with its own L2; the items, ordered split-major with the k-tile fastest, are cut
// into 8 contiguous runs, one per XCD, so the workgroups resident on an XCD work on the same rows of X and Y and on
// neighbouring column panels: each panel then leaves HBM about once instead of once per workgroup that uses it
// (measured on dW fc1: 698 MB fetched per launch before, against 195 MB of operands).
__device__ __forceinline__ bool tn_work_item(int tiles, int splits, int& tile, int& split) {
    const int run = gridDim.x >> 3;
    const int item = (blockIdx.x & 7) * run + (blockIdx.x >> 3);
    split = item / tiles;
    tile = item - split * tiles;
    return item < tiles * splits;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTN p, int tiles, int splits) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;  // wave tile: 64 n x 64 k
    const int tiles_k = (p.K + 127) / 128;
    int wg, split;
    if (!tn_work_item(tiles, splits, wg, split)) return;
    const int n0 = (wg / tiles_k) * 128, k0 = (wg % tiles_k) * 128;
    const int m_begin = split * p.rows_per_split;
    int m_end = m_begin + p.rows_per_split; m_end = m_end < p.M ? m_end : p.M;

    // Operand ring: four slots of 32 reduction rows, X image [32][128] bf16 (256-byte rows, 16-column blocks swizzled by
    // tn_swz) followed by the Y image, filled by LDS-DMA three slots ahead.  Wave-instruction j of a slot fills rows
    // (j*4 + wave)*4 .. +3, 16 lanes per row; tn_swz(r + 16) == tn_swz(r), so both instructions use the same columns.
    const int r_st = wave * 4 + (lane >> 4);
    const bf16* xsrc; const bf16* ysrc;
    {
        const int c16 = lane & 15;
        const int col = (((c16 >> 1) ^ tn_swz(r_st)) << 4) + (c16 & 1) * 8;
        int xc = n0 + col; xc = xc + 8 <= p.N ? xc : p.N - 8;
        int yc = k0 + col; yc = yc + 8 <= p.K ? yc : p.K - 8;
        xsrc = p.X + (long)(m_begin + r_st) * p.ldx + xc;
        ysrc = p.Y + (long)(m_begin + r_st) * p.ldy + yc;
    }
    const int nh = (m_end - m_begin + 31) / 32;
    int issued = 0;
    auto issue = [&]() {  // the next 32 rows into slot issued & 3; rows past the split read zeros
        char* sx = smem + (issued & 3) * TILE_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = issued * 32 + j * 16;
            const bool ok = m_begin + row + r_st < m_end;
            // the selected addresses are made opaque so that each copy stays one instruction for all lanes (a branch per
            // source would issue two where a wave straddles the end of the split, and the counted waits below assume four
            // per thread and slot)
            const void* xs = ok ? (const void*)(xsrc + (long)row * p.ldx) : (const void*)g_zero_page;
            const void* ys = ok ? (const void*)(ysrc + (long)row * p.ldy) : (const void*)g_zero_page;
            asm volatile("" : "+v"(xs), "+v"(ys));
            glds16(xs, sx + j * 4096);
            glds16(ys, sx + 8192 + j * 4096);
        }
        ++issued;
    };
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (issued < nh) issue();

    f32x4 acc[4][4];  // [ki][ni] : D'[k = ki*16 + fq*4 + r][n = ni*16 + fr]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // column sums of X ride along as one more MFMA per column fragment with an all-ones A operand
    // (every row of that product is sum_m X[m][n]); only the wk == 0 waves of the k-tile-0 workgroups do it
    const bool do_colsum = p.colsum != nullptr && k0 == 0 && wk == 0;
    f32x4 csum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) csum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    // Transposing reads of rows 8g + q and 8g + q + 4 (tn_swz of both is q | (g & 1) << 2).  They go through inline asm:
    // for the builtin the compiler orders the read after every LDS-DMA in flight (s_waitcnt vmcnt(0) in front of the
    // reads), which turns the run-ahead into a synchronous copy.  The explicit lgkmcnt wait retires them.
    const int rswz = q | ((g & 1) << 2);
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + (8 * g + q) * 256 + pp * 8;
    auto frag = [&](unsigned lb, auto off_c, int nb) -> bf16x8 {
        constexpr int off = decltype(off_c)::value;
        const unsigned a = lb + ((nb ^ rswz) << 5);
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(off + 1024));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    for (int hs = 0; hs < nh; ++hs) {
        // slot hs has landed once all but the two newer slots' loads (four per thread and slot) are back
        if (hs + 2 < nh) wait_vmcnt<8>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading slot hs - 1
        if (issued < nh) issue();      // refill slot hs - 1
        unsigned lb = lbase + (hs & 3) * TILE_BYTES;
        asm volatile("" : "+v"(lb));
        bf16x8 yf[4], xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xf[t] = frag(lb, std::integral_constant<int, 0>{}, wn * 4 + t);
            yf[t] = frag(lb, std::integral_constant<int, 8192>{}, wk * 4 + t);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[ki], xf[ni], acc[ki][ni], 0, 0, 0);
        if (do_colsum) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) csum[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[ni], csum[ni], 0, 0, 0);
        }
    }

    float* out = p.out + (long)split * p.slab;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + i16;
        if (n >= p.N) continue;
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            const int k = k0 + wk * 64 + ki * 16 + g * 4;
            if (k >= p.K) continue;
            *(f32x4*)(out + (long)n * p.K + k) = acc[ki][ni];
        }
        if (do_colsum && g == 0) p.colsum[(long)split * p.N + n] = csum[ni][0];
    }
}

// ------------------------------------------------------------------------------------------
// 256 (n) x 256 (k) weight-gradient kernel with the structure of gemm_nt_pp_kernel: eight waves (2 along n x 4 along k,
// 128 x 64 outputs each), 64 reduction rows per stage, four 16-KiB slots per stage filled by LDS-DMA seven slots ahead,
// the two wave groups one barrier apart.  Slot images are [64 m][128 columns] (256-byte rows, 16-column blocks swizzled
// by tn_swz) read with the transposing LDS read: slot 0 / 3 = X columns {wn*128 + (0|64) + 0..63}, slot 1 / 2 = Y columns
// {wk*64 + (0|32) + 0..31}.  One workgroup = one (tile, M-split) pair; splits are chosen so that tiles x splits fills
// the CUs once.  The bias gradient (column sums of X) is accumulated on the vector ALU from the X fragments already in
// registers, each wave of a k-tile-0 workgroup taking two of its eight column fragments.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void gemm_tn_pp_kernel(GemmTN p, int tiles_k, int tiles, int splits) {
    extern __shared__ __attribute__((aligned(16))) char tnsmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    int tile, split;
    if (!tn_work_item(tiles, splits, tile, split)) return;
    const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;
    const int m_begin = split * p.rows_per_split;
    int m_end = m_begin + p.rows_per_split; m_end = m_end < p.M ? m_end : p.M;
    const int nst = (m_end - m_begin + 127) / 128 * 2;  // stages of 64 rows, in pairs (one per LDS buffer)

    // ---- staging: wave-instruction j of this wave fills slot rows (j*8 + wave)*4 .. +3, 16 lanes per 256-byte row.
    // N and K are multiples of 256 here, so no column needs clamping: slot 3 is slot 0 shifted by 64 columns, slot 2 is
    // slot 1 shifted by 32, and the second instruction is the first shifted by 32 rows.
    // tn_swz(r + 32) == tn_swz(r), so both instructions hold the same image columns: one pointer per operand.
    const int r_st = wave * 4 + (lane >> 4);
    const bf16* xsrc; const bf16* ysrc;
    {
        const int c16 = lane & 15;
        const int col = (((c16 >> 1) ^ tn_swz(r_st)) << 4) + (c16 & 1) * 8;  // column of the slot image held at this lane's position
        xsrc = p.X + (long)(m_begin + r_st) * p.ldx + n0 + (col >> 6) * 128 + (col & 63);
        ysrc = p.Y + (long)(m_begin + r_st) * p.ldy + k0 + (col >> 5) * 64 + (col & 31);
    }
    int s_st[4] = {0, 0, 0, 0};  // next stage index per slot
    auto stage_slot = [&](auto slot_c) {
        constexpr int s = decltype(slot_c)::value;
        char* dst = tnsmem + (s_st[s] & 1) * PP_KT_BYTES + s * PP_H_BYTES + wave * 1024;
        constexpr bool is_x = (s == 0 || s == 3);
        const long ld = is_x ? p.ldx : p.ldy;
        const long off = (long)s_st[s] * 64 * ld + (s == 3 ? 64 : (s == 2 ? 32 : 0));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = m_begin + s_st[s] * 64 + j * 32 + r_st < m_end;   // rows past the split read zeros
            const bf16* g = (is_x ? xsrc : ysrc) + off + j * 32 * ld;
            glds16(ok ? (const void*)g : (const void*)g_zero_page, dst + j * 8 * 1024);
        }
        ++s_st[s];
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;

    stage_slot(S0{}); stage_slot(S1{}); stage_slot(S2{}); stage_slot(S3{});
    stage_slot(S0{}); stage_slot(S1{}); stage_slot(S2{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 acc[2][4][4];  // [X half][ki][ni] : D'[k = ki*16 + g*4 + r][n = ni*16 + i16]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    bf16x8 xf[4][2], ylo[2][2], yhi[2][2];
    // transposing reads: rows r = ms*32 + 8g + q + 4h; tn_swz(r) = q | (g & 1) << 2 for all of them
    const int rbase = (8 * g + q) * 256 + pp * 8, rswz = q | ((g & 1) << 2);
    // The transposing reads go through inline asm: for the builtin the compiler orders the read after every LDS-DMA in
    // flight (s_waitcnt vmcnt(0) in front of each group of reads), which would turn the seven-slot run-ahead into a
    // synchronous copy.  The reads are retired by the explicit lgkmcnt waits before the MFMA clusters.
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) char*)tnsmem) + rbase;
    // lb: lbase + buffer offset, made opaque once per phase so that the (loop-invariant) read addresses are recomputed
    // there instead of being hoisted out of the loop into two dozen extra registers
    auto frag = [&](unsigned lb, auto off_c, int nb) -> bf16x8 {
        constexpr int off = decltype(off_c)::value;
        const unsigned a = lb + ((nb ^ rswz) << 5);
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(off + 1024));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // Bias gradient: the column sums of the X tile, taken on the vector ALU from the fragments already in registers.  The
    // tiles_k workgroups that share an X tile share the work: fragment f = 2 wk + t (of the eight 16-column fragments a
    // wave row holds; all four wk waves hold the same ones) belongs to the workgroup with k-tile index f % min(tiles_k, 8).
    // Left to the k-tile-0 workgroups alone it made them (a third of the grid at K = 768) 11 % slower than the rest, and
    // the launch as slow as them.
    const int cs_share = tiles_k < 8 ? tiles_k : 8, cs_kt = tile % tiles_k;
    bool cs_mine[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) cs_mine[t] = p.colsum != nullptr && (wk * 2 + t) % cs_share == cs_kt;
    const bool do_colsum = cs_mine[0] || cs_mine[1];
    float cs[2] = {0.f, 0.f};  // column sums of X fragments ni = (wk & 1) * 2 + {0, 1} of X half (wk >> 1)
    auto read_x = [&](unsigned lb, auto slot_c) {
        constexpr int so = decltype(slot_c)::value * PP_H_BYTES;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            xf[ni][0] = frag(lb, std::integral_constant<int, so>{}, wn * 4 + ni);
            xf[ni][1] = frag(lb, std::integral_constant<int, so + 8192>{}, wn * 4 + ni);
        }
    };
    auto colsum_x = [&](int half) {
        if (do_colsum && (wk >> 1) == half) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (cs_mine[t]) {
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms) cs[t] = add_bf16x8(cs[t], (wk & 1) ? xf[2 + t][ms] : xf[t][ms]);
                }
        }
    };
    auto read_y = [&](unsigned lb, auto slot_c, bf16x8 (&y)[2][2]) {
        constexpr int so = decltype(slot_c)::value * PP_H_BYTES;
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
            y[ki][0] = frag(lb, std::integral_constant<int, so>{}, wk * 2 + ki);
            y[ki][1] = frag(lb, std::integral_constant<int, so + 8192>{}, wk * 2 + ki);
        }
    };
    auto mfma_quad = [&](f32x4 (&c)[4][4], int kbase, const bf16x8 (&y)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ki = 0; ki < 2; ++ki)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    c[kbase + ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[ki][ms], xf[ni][ms], c[kbase + ki][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    if (wn == 1) __builtin_amdgcn_s_barrier();  // second wave group runs one barrier behind

    for (int st = 0; st < nst; st += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            unsigned lb = lbase + b * PP_KT_BYTES;
            // ---- phase 0: X-lo (slot 0) and Y-lo (slot 1) ----
            asm volatile("" : "+v"(lb));
            read_x(lb, S0{});
            read_y(lb, S1{}, ylo);
            stage_slot(S3{});
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the slot-0 reads are done before the barrier
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[0], 0, ylo);
            colsum_x(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1: Y-hi (slot 2) ----
            asm volatile("" : "+v"(lb));
            read_y(lb, S2{}, yhi);
            stage_slot(S0{});
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[0], 2, yhi);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2: X-hi (slot 3) ----
            asm volatile("" : "+v"(lb));
            read_x(lb, S3{});
            stage_slot(S1{});
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[1], 2, yhi);
            colsum_x(1);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3 ----
            stage_slot(S2{});
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(acc[1], 0, ylo);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wn == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // The wave's 128 x 64 block leaves through the (now idle) operand ring, 64 rows x 256 bytes at a time, so that a store
    // instruction covers four whole 256-byte rows: in the accumulator layout the sixteen lanes i16 hold sixteen different
    // rows, 64 separate segments per instruction, about a quarter of the store rate of a CU.  Chunk c (16 bytes) of row r
    // sits at c ^ (r & 15) in the wave's 16-KiB piece: no bank conflicts in either direction; DS operations of a wave
    // execute in order, and the piece is private to the wave (the loads into the ring were waited for above, by every
    // wave of its own, and the other group no longer reads the ring once it is past its last cluster's reads:
    // the barrier below).
    __builtin_amdgcn_s_barrier();
    float* out = p.out + (long)split * p.slab + (long)(n0 + wn * 128) * p.K + k0 + wk * 64;
    char* piece = tnsmem + wave * 16384;
    const int rr = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int ki = 0; ki < 4; ++ki)
                *(f32x4*)(piece + (ni * 16 + i16) * 256 + (((ki * 4 + g) ^ i16) << 4)) = acc[h][ki][ni];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = 4 * j + rr;
            const f32x4 v = *(const f32x4*)(piece + r * 256 + ((cc ^ (r & 15)) << 4));
            *(f32x4*)(out + (long)(h * 64 + r) * p.K + 4 * cc) = v;
        }
    }
    if (do_colsum) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = cs[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n0 + wn * 128 + (wk >> 1) * 64 + ((wk & 1) * 2 + t) * 16 + i16;
            if (cs_mine[t] && g == 0 && n < p.N) p.colsum[(long)split * p.N + n] = v;
        }
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s][i]; the same for the column-sum slabs (scalar tail loop)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, long slab, int nsplit, float* __restrict__ out,
                                   long n4, int accumulate, const float* __restrict__ cslabs, int n_cols,
                                   float* __restrict__ cout, int c_accumulate) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    for (long i = tid; i < n4; i += nth) {
        f32x4 s = accumulate ? ((const f32x4*)out)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        // unrolled so that eight slab loads are in flight per thread (a small output with many splits is otherwise one
        // dependent round trip per split); the additions keep their order
#pragma unroll 8
        for (int k = 0; k < nsplit; ++k) s += ((const f32x4*)(slabs + k * slab))[i];
        ((f32x4*)out)[i] = s;
    }
    if (cout) {
        for (long i = tid; i < n_cols; i += nth) {
            float s = c_accumulate ? cout[i] : 0.f;
#pragma unroll 8
            for (int k = 0; k < nsplit; ++k) s += cslabs[(long)k * n_cols + i];
            cout[i] = s;
        }
    }
}

static thread_local int g_tn_variant = 0;  // 0 = by problem size, 1 = 128x128 kernel, 2 = 256x256 ping-pong kernel
extern "C" void mmae_gemm_tn_set_variant(int v) { g_tn_variant = v; }

static bool tn_use_pp(int M, int N, int K) {
    if (g_tn_variant == 1 || N % 256 != 0 || K % 256 != 0) return false;
    if (g_tn_variant == 2) return true;
    // worth a whole CU per workgroup when there are enough 256x256 tiles to share few splits and a long reduction
    const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
    return tiles >= 9 && M >= 8192;
}

static int tn_splits(int M, int N, int K) {
    if (tn_use_pp(M, N, K)) {
        const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
        int s = num_cus() / tiles;  // one resident round of one workgroup per CU
        const int max_s = (M + 1023) / 1024;
        if (s > max_s) s = max_s;
        return s < 1 ? 1 : s;
    }
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    // One resident round: 256 CUs x 2 workgroups of this kernel fit at once; a grid a little over that would run a
    // second, nearly empty round (576 workgroups took as long as 1024), so split M only as far as tiles * s <= 512.
    int s = 512 / tiles;
    const int max_s = (M + 16 * TN_ROWS - 1) / (16 * TN_ROWS);  // at least 1024 rows per split: slab traffic stays small
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

extern "C" size_t mmae_gemm_tn_workspace_bytes(int M, int N, int K) {
    return (size_t)tn_splits(M, N, K) * ((size_t)N * (size_t)K + (size_t)N) * sizeof(float);
}

extern "C" int mmae_gemm_tn(const void* X, long ldx, const void* Y, long ldy, int M, int N, int K, float* out,
                            int accumulate, float* colsum_out, int colsum_accumulate, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(X && Y && out, MMAE_EINVAL, "gemm_tn: null pointer");
    MMAE_REQUIRE(M > 0 && N >= 8 && K >= 8, MMAE_EINVAL, "gemm_tn: bad problem M=%d N=%d K=%d", M, N, K);
    MMAE_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= N && ldy >= K, MMAE_ESHAPE,
                 "gemm_tn: N, K and row strides must be multiples of 8 (N=%d K=%d ldx=%ld ldy=%ld)", N, K, ldx, ldy);
    MMAE_REQUIRE(((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0) && ((uintptr_t)out % 16 == 0), MMAE_EINVAL,
                 "gemm_tn: pointers must be 16-byte aligned");
    const int s = tn_splits(M, N, K);
    const bool direct = (s == 1 && !accumulate && !(colsum_out && colsum_accumulate));
    const size_t need = direct ? 0 : mmae_gemm_tn_workspace_bytes(M, N, K);
    MMAE_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), MMAE_EWORKSPACE,
                 "gemm_tn: workspace %zu bytes < required %zu", workspace_bytes, need);
    GemmTN p = {};
    p.X = (const bf16*)X; p.ldx = ldx; p.Y = (const bf16*)Y; p.ldy = ldy; p.M = M; p.N = N; p.K = K;
    const bool pp = tn_use_pp(M, N, K);
    g_last_tn = pp ? 2 : 1;
    const int gran = pp ? 2 * TN_ROWS : TN_ROWS;
    int rps = (M + s - 1) / s; rps = (rps + gran - 1) / gran * gran;
    p.rows_per_split = rps;
    p.slab = (long)N * K;
    float* tmp = direct ? out : (float*)workspace;
    float* ctmp = colsum_out ? (direct ? colsum_out : (float*)workspace + (size_t)s * N * K) : nullptr;
    p.out = tmp;
    p.colsum = ctmp;
    if (g_tn_phase == 2) {
        // reduction only: the slabs are already in the workspace
    } else if (pp) {
        static bool reserved = false;
        if (!reserved) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_KT_BYTES);
            MMAE_REQUIRE(e == hipSuccess, MMAE_ELAUNCH, "gemm_tn: cannot reserve LDS: %s", hipGetErrorString(e));
            reserved = true;
        }
        const int tk = (K + 255) / 256, tiles = ((N + 255) / 256) * tk;
        hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3((tiles * s + 7) / 8 * 8), dim3(512), 2 * PP_KT_BYTES, stream, p, tk, tiles, s);
        MMAE_CHECK_LAUNCH("gemm_tn_pp");
    } else {
        const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3((tiles * s + 7) / 8 * 8), dim3(256), 0, stream, p, tiles, s);
        MMAE_CHECK_LAUNCH("gemm_tn");
    }
    if (!direct && g_tn_phase != 1) {
        const long n4 = (long)N * K / 4;
        int blocks = (int)((n4 + 255) / 256); blocks = blocks > 2048 ? 2048 : blocks;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, stream, tmp, p.slab, s, out, n4, accumulate,
                           ctmp, N, colsum_out, colsum_accumulate);
        MMAE_CHECK_LAUNCH("slab_reduce");
    }
    return MMAE_OK;
}

// ------------------------------------------------------------------------------------------
// column sums of a bf16 [M,N] matrix (bias gradients): two deterministic passes.
// ------------------------------------------------------------------------------------------
#define CS_CHUNKS 64
__global__ void colsum_partial_kernel(const bf16* __restrict__ X, long ldx, int M, int N, float* __restrict__ part) {
    // block: 32 column-groups of 8 columns (256 columns) x 8 row lanes
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per;
    int r1 = r0 + rows_per; r1 = r1 < M ? r1 : M;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        for (int r = r0 + rl; r < r1; r += 8) {
            const bf16x8 v = *(const bf16x8*)(X + (long)r * ldx + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = s[e];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][c];
        part[(long)blockIdx.y * N + blockIdx.x * 256 + c] = t;
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int chunks, int N, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float t = accumulate ? out[c] : 0.f;
#pragma unroll 8
    for (int k = 0; k < chunks; ++k) t += part[(long)k * N + c];
    out[c] = t;
}

extern "C" size_t mmae_colsum_workspace_bytes(int M, int N) { (void)M; return (size_t)CS_CHUNKS * N * sizeof(float); }

extern "C" int mmae_colsum_bf16(const void* X, long ldx, int M, int N, float* out, int accumulate, void* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
    MMAE_REQUIRE(X && out && workspace, MMAE_EINVAL, "colsum: null pointer");
    MMAE_REQUIRE(N % 8 == 0 && ldx % 8 == 0 && M > 0, MMAE_ESHAPE, "colsum: N=%d and ldx=%ld must be multiples of 8", N, ldx);
    MMAE_REQUIRE(workspace_bytes >= mmae_colsum_workspace_bytes(M, N), MMAE_EWORKSPACE, "colsum: workspace too small");
    int chunks = CS_CHUNKS; if (chunks > (M + 63) / 64) chunks = (M + 63) / 64;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, stream, (const bf16*)X, ldx, M, N,
                       (float*)workspace);
    MMAE_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const float*)workspace, chunks, N,
                       out, accumulate);
    MMAE_CHECK_LAUNCH("colsum_final");
    return MMAE_OK;
}
