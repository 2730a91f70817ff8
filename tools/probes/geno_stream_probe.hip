// How fast can 256 workgroups stream a sample-major uint8 genotype matrix the way the large-M GEMM needs it - every
// workgroup 128 rows x a narrow SNP slice per step - and which slice shape / group mapping does HBM like?
//   rows x K bytes, workgroup (mt, g) as in l1_gemm_i8.hip (block b on XCD b % 8, g = xcd + 8 * (idx / n_mt)),
//   per step 128 rows x LINE bytes (LINE = 128, 256, 512), 16 bytes per lane, DEPTH steps in flight;
//   mapping 0: groups interleaved (step s of group g at (g + s G) LINE)     <- what the GEMM does with LINE = 128
//   mapping 1: groups contiguous  (group g owns [g K/G, (g+1) K/G), walks it in LINE steps)
//   hipcc --offload-arch=gfx950 -O3 geno_stream_probe.hip -o geno_stream_probe && ./geno_stream_probe [rows] [K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LINE, int MAP, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const uint8_t* __restrict__ X, int64_t pitch, int n_mt, int G, int K,
                                                      uint32_t* __restrict__ sink) {
    constexpr int LPR = LINE / 16;            // lanes per row
    constexpr int RPP = 512 / LPR;            // rows per pass
    constexpr int NP = 128 / RPP;             // passes (loads per thread) per step
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mt = idx % n_mt, g = xcd + 8 * (idx / n_mt);
    const int nsteps = K / (G * LINE);
    const uint8_t* base[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) base[p] = X + (int64_t)(mt * 128 + p * RPP + t / LPR) * pitch + (t % LPR) * 16;
    u32x4 acc = u32x4{0};
    for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
        u32x4 v[DEPTH][NP];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            int s = s0 + d;
            if (s >= nsteps) s = nsteps - 1;
            const int64_t off = MAP == 0 ? (int64_t)(g + s * G) * LINE : (int64_t)g * (K / G) + (int64_t)s * LINE;
#pragma unroll
            for (int p = 0; p < NP; ++p) v[d][p] = *reinterpret_cast<const u32x4*>(base[p] + off);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc = acc ^ v[d][p];
    }
    sink[(size_t)blockIdx.x * 512 + t] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int LINE, int MAP, int DEPTH>
static void run(const uint8_t* X, int64_t pitch, int rows, int K, uint32_t* sink) {
    const int n_mt = rows / 128, G = 256 / n_mt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<LINE, MAP, DEPTH>), dim3(n_mt * G), dim3(512), 0, 0, X, pitch, n_mt, G, K, sink);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((stream_kernel<LINE, MAP, DEPTH>), dim3(n_mt * G), dim3(512), 0, 0, X, pitch, n_mt, G, K, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)rows * (K / (G * LINE)) * G * LINE;
    printf("{\"probe\": \"geno_stream\", \"rows\": %d, \"K\": %d, \"line_bytes\": %d, \"mapping\": \"%s\", \"steps_in_flight\": %d, "
           "\"us\": %.1f, \"gbs\": %.0f}\n", rows, K, LINE, MAP == 0 ? "interleaved" : "contiguous", DEPTH, ms * 1e3 / iters,
           bytes / (ms * 1e-3 / iters) * 1e-9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 4096;
    const int K = argc > 2 ? atoi(argv[2]) : 98304;      // a multiple of 8 groups x 512 bytes
    const int64_t pitch = 100000;
    uint8_t* X; uint32_t* sink;
    CK(hipMalloc(&X, (size_t)rows * pitch)); CK(hipMalloc(&sink, 256 * 512 * 4));
    CK(hipMemset(X, 1, (size_t)rows * pitch));
    run<128, 0, 4>(X, pitch, rows, K, sink);
    run<128, 0, 8>(X, pitch, rows, K, sink);
    run<128, 1, 4>(X, pitch, rows, K, sink);
    run<128, 1, 8>(X, pitch, rows, K, sink);
    run<256, 0, 4>(X, pitch, rows, K, sink);
    run<256, 1, 4>(X, pitch, rows, K, sink);
    run<512, 0, 2>(X, pitch, rows, K, sink);
    run<512, 0, 4>(X, pitch, rows, K, sink);
    run<512, 1, 2>(X, pitch, rows, K, sink);
    run<512, 1, 4>(X, pitch, rows, K, sink);
    return 0;
}
