// Store-rate probe: 256 workgroups x 8 waves write 256x256 bf16 tiles of a 25344 x 3072 matrix (each wave a 128-row x
// 128-byte block, the epilogue of the 256x256 GEMM kernel) with different lane -> address patterns.
//   0: lane (fr = l & 15, fq = l >> 4) -> row 16*mi + fr, bytes 64*u + 16*fq        (MFMA accumulator layout)
//   1: lane l -> row 8*j + l / 8, bytes 16 * (l % 8)                                (8 rows x 128 B per instruction)
//   2: lane l -> row 16*mi + l / 4, bytes 64*u + 16 * (l % 4)                       (16 rows x 64 B, quads contiguous)
//   3: as 0 with 8-byte stores (two per 16 bytes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int PAT>
__global__ __launch_bounds__(512, 1) void store_kernel(char* out, long ld_bytes, int tiles_m, int tiles_n, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    const int ntiles = tiles_m * tiles_n;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int rep = 0; rep < reps; ++rep)
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const int m0 = (t % tiles_m) * 256, n0 = (t / tiles_m) * 256;
            char* base = out + (long)(m0 + wm * 128) * ld_bytes + (long)(n0 + wn * 64) * 2;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (PAT == 0 || PAT == 3) {
                    const int mi = i >> 1, u = i & 1, fr = lane & 15, fq = lane >> 4;
                    char* p = base + (long)(mi * 16 + fr) * ld_bytes + 64 * u + 16 * fq;
                    if (PAT == 0) *(f32x4*)p = v;
                    else { *(f32x2*)p = (f32x2){v[0], v[1]}; *(f32x2*)(p + 8) = (f32x2){v[2], v[3]}; }
                } else if (PAT == 1) {
                    char* p = base + (long)(i * 8 + (lane >> 3)) * ld_bytes + 16 * (lane & 7);
                    *(f32x4*)p = v;
                } else {
                    const int mi = i >> 1, u = i & 1;
                    char* p = base + (long)(mi * 16 + (lane >> 2)) * ld_bytes + 64 * u + 16 * (lane & 3);
                    *(f32x4*)p = v;
                }
            }
        }
}

template <int PAT>
static void run(char* buf, const char* name, int grid = 256) {
    const int tm = 99, tn = 12;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    store_kernel<PAT><<<grid, 512>>>(buf, 3072 * 2, tm, tn, 1);
    hipEventRecord(e0);
    const int reps = 4;
    store_kernel<PAT><<<grid, 512>>>(buf, 3072 * 2, tm, tn, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)reps * tm * tn * 256 * 256 * 2;
    printf("pattern %d (%s), %d workgroups: %.1f us per pass, %.2f TB/s, %.1f GB/s per CU, %.2f us per 128-KiB tile\n", PAT, name, grid, ms * 1e3 / reps,
           bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / grid, ms * 1e3 / reps / (tm * tn / (double)grid));
}

int main() {
    char* buf;
    if (hipMalloc(&buf, 25344L * 3072 * 2) != hipSuccess) return 1;
    run<0>(buf, "accumulator layout, 16 B");
    run<1>(buf, "8 rows x 128 B");
    run<2>(buf, "16 rows x 64 B, quads contiguous");
    run<3>(buf, "accumulator layout, 8 B");
    for (int g : {8, 16, 32, 64, 128, 192}) run<0>(buf, "accumulator layout, 16 B", g);
    for (int g : {8, 32, 128}) run<1>(buf, "8 rows x 128 B", g);
    for (int g : {8, 32, 128}) run<2>(buf, "16 rows x 64 B, quads contiguous", g);
    for (int g : {8, 32}) run<3>(buf, "accumulator layout, 8 B", g);
    hipFree(buf);
    return 0;
}
