// What does the matrix pipe sustain, and at which clock?  Settles "clock-down vs barrier bubble" for l1_gemm.hip with a
// number (VERDICT r02, weak #2b) and picks the number format of the large-M layer-1 GEMM by measurement:
//   bf16 32x32x16, f16 32x32x16, i8 32x32x32 - back-to-back MFMAs from registers on every SIMD, operands either
//   zero, genotype-like (A in {0,1,2}, B random) or fully random; one or two waves per SIMD.
// Each wave stamps s_memtime (shader clock ticks) and s_memrealtime (constant 100 MHz) around its loop, so
//   effective clock = d(memtime) / d(memrealtime) * 100 MHz          (no profiler in the way)
//   rate            = MFMAs * flops / wall (hipEvent)
// Also checks (a) the i8 operand layout used by the kernel (lane (j, hi) holds 16 consecutive k of row / column j),
// (b) whether f16 MFMA honours subnormal inputs (a u8 genotype as the fp16 bit pattern 0x00xx = x * 2^-24).
//   hipcc --offload-arch=gfx950 -O3 mfma_clock_probe.hip -o mfma_clock_probe && ./mfma_clock_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0 bf16, 1 f16, 2 i8.  A: 16 fragments (4 k-steps x 4 row tiles), B: 4 fragments x 2 sets; 4 accumulators.
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(const u32x4* __restrict__ abuf, const u32x4* __restrict__ bbuf,
                                                    int iters, float* __restrict__ sink, uint64_t* __restrict__ stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 a[4][4], b[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) a[k][t] = abuf[((blockIdx.x * 8 + wave) % 61 * 16 + k * 4 + t) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) b[s][k] = bbuf[((blockIdx.x * 8 + wave) % 53 * 8 + s * 4 + k) * 64 + lane];
    f32x16 accf[4];
    i32x16 acci[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { accf[t] = f32x16{0}; acci[t] = i32x16{0}; }
    __syncthreads();
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (MODE == 0)
                        accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[k][t]),
                                                                          __builtin_bit_cast(bf16x8, b[s][k]), accf[t], 0, 0, 0);
                    else if (MODE == 1)
                        accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[k][t]),
                                                                         __builtin_bit_cast(f16x8, b[s][k]), accf[t], 0, 0, 0);
                    else
                        acci[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a[k][t]),
                                                                        __builtin_bit_cast(i32x4, b[s][k]), acci[t], 0, 0, 0);
                }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float z = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) z += accf[t][r] + (float)acci[t][r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = z;
    if (lane == 0) {
        stamps[((size_t)blockIdx.x * 8 + wave) * 2] = c1 - c0;
        stamps[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

// ---- correctness: one wave, D = A (32 x KK) x B (KK x 32), row-major host arrays --------------------------------
__global__ void check_i8(const int8_t* A, const int8_t* B, int* D) {   // A[32][32], B[32(k)][32(j)]
    const int lane = threadIdx.x, j = lane & 31, hi = lane >> 5;
    union { int8_t b[16]; i32x4 v; } fa, fb;
    for (int e = 0; e < 16; ++e) { fa.b[e] = A[j * 32 + hi * 16 + e]; fb.b[e] = B[(hi * 16 + e) * 32 + j]; }
    i32x16 acc = i32x16{0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.v, fb.v, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + j] = acc[r];
}
__global__ void check_f16_subnormal(const uint8_t* A, const float* B, float scaleB, float* D) {   // A[32][16] u8, B[16][32]
    const int lane = threadIdx.x, j = lane & 31, hi = lane >> 5;
    union { uint16_t h[8]; f16x8 v; } fa;
    f16x8 fb;
    for (int e = 0; e < 8; ++e) { fa.h[e] = A[j * 16 + hi * 8 + e]; fb[e] = (_Float16)(B[(hi * 8 + e) * 32 + j] * scaleB); }
    f32x16 acc = f32x16{0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.v, fb, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + j] = acc[r];
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static uint16_t f2h(float f) {   // round-to-nearest-even fp32 -> fp16, normal range only (probe data is O(1))
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u; int e = (int)((u >> 23) & 255) - 127 + 15; uint32_t m = u & 0x7FFFFFu;
    if (e <= 0) return (uint16_t)s;
    uint32_t h = (uint32_t)(e << 10) | (m >> 13);
    if ((m & 0x1FFFu) > 0x1000u || ((m & 0x1FFFu) == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(s | h);
}

template <int MODE>
static void run_rate(const char* name, int data, int threads, int iters, int launches) {
    // data: 0 zeros, 1 genotype-like A / random B, 2 random / random
    const size_t na = 61 * 16 * 64, nb = 53 * 8 * 64;
    std::vector<u32x4> ha(na), hb(nb);
    auto elemA = [&](void) -> uint32_t {   // one 16-bit (bf16/f16) or 8-bit (i8) element
        if (data == 0) return 0;
        int g = (rnd() % 10 < 6) ? 0 : (rnd() % 3 ? 1 : 2);
        if (data == 2) g = rnd() % 120;
        if (MODE == 2) return (uint32_t)g;
        if (MODE == 1) return f2h((float)g);
        float f = (float)g; uint32_t u; memcpy(&u, &f, 4); return u >> 16;
    };
    auto elemB = [&](void) -> uint32_t {
        if (data == 0) return 0;
        if (MODE == 2) return rnd() & 255u;
        float f = ((int)(rnd() % 20001) - 10000) * 1e-4f;
        if (MODE == 1) return f2h(f * 64.f);
        uint32_t u; memcpy(&u, &f, 4); return u >> 16;
    };
    for (auto& v : ha) for (int d = 0; d < 4; ++d) v[d] = MODE == 2 ? (elemA() | elemA() << 8 | elemA() << 16 | elemA() << 24) : (elemA() | elemA() << 16);
    for (auto& v : hb) for (int d = 0; d < 4; ++d) v[d] = MODE == 2 ? (elemB() | elemB() << 8 | elemB() << 16 | elemB() << 24) : (elemB() | elemB() << 16);
    u32x4 *da, *db; float* sink; uint64_t* st;
    const int grid = 256;
    CK(hipMalloc(&da, na * 16)); CK(hipMalloc(&db, nb * 16)); CK(hipMalloc(&sink, (size_t)grid * 512 * 4)); CK(hipMalloc(&st, (size_t)grid * 8 * 16));
    CK(hipMemcpy(da, ha.data(), na * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), nb * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 0, 0, da, db, iters, sink, st);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 0, 0, da, db, iters, sink, st);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> hs((size_t)grid * 8 * 2);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    const int waves = threads / 64;
    double cyc = 0, rt = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) { cyc += (double)hs[(b * 8 + w) * 2]; rt += (double)hs[(b * 8 + w) * 2 + 1]; }
    const double n_mfma = (double)grid * waves * iters * 32.0 * launches;
    const double flops = n_mfma * 32.0 * 32.0 * (MODE == 2 ? 32.0 : 16.0) * 2.0;
    const double ghz = cyc / rt * 0.1;
    // Mean over WAVES of (a wave's own lifetime in shader cycles / its MFMAs), divided by the waves per SIMD.  This is the
    // SIMD's issue interval only when the waves of a SIMD share it evenly - it reads 32.0 (= 8 passes x 4 cycles, what
    // 2.5 PF / 1024 SIMDs / 2.4 GHz implies) with ONE wave per SIMD.  With two waves per SIMD the issue arbiter serves the
    // older wave first: it runs at 32 cycles per MFMA for its whole life, the younger one gets the gaps and then the SIMD
    // to itself, so the two lifetimes are T and 2 T, their mean is 48 cycles per MFMA and this column reads 24 - at an
    // unchanged rate (the `tops` column, which is what the roofline uses).  It is a wave-lifetime statistic, not a pipe rate.
    const double cyc_per_mfma_simd = (cyc / (grid * waves)) / (iters * 32.0) / (waves > 4 ? 2.0 : 1.0);
    printf("{\"probe\": \"mfma_rate\", \"type\": \"%s\", \"data\": \"%s\", \"waves_per_simd\": %d, \"launch_ms\": %.3f, \"launches\": %d, "
           "\"tops\": %.1f, \"frac_of_2500\": %.3f, \"clock_ghz\": %.3f, \"mean_wave_lifetime_cycles_per_mfma_div_waves_per_simd\": %.2f}\n",
           name, data == 0 ? "zeros" : data == 1 ? "genotype A, random B" : "random", waves / 4, ms / launches, launches,
           flops / (ms * 1e-3) * 1e-12, flops / (ms * 1e-3) * 1e-12 / 2500.0, ghz, cyc_per_mfma_simd);
    fflush(stdout);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(sink); (void)hipFree(st);
}

int main(int argc, char** argv) {
    // ---- layout / semantics checks
    {
        std::vector<int8_t> A(32 * 32), B(32 * 32); std::vector<int> D(32 * 32), R(32 * 32);
        for (auto& v : A) v = (int8_t)(rnd() % 3);
        for (auto& v : B) v = (int8_t)((int)(rnd() % 255) - 127);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
        int8_t *dA, *dB; int* dD;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_i8, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
        printf("{\"probe\": \"i8_layout\", \"mismatches\": %d}\n", bad);
    }
    {
        std::vector<uint8_t> A(32 * 16); std::vector<float> B(16 * 32), D(1024);
        for (auto& v : A) v = (uint8_t)(rnd() % 3);
        A[0] = 255; A[17] = 200;
        for (auto& v : B) v = ((int)(rnd() % 2001) - 1000) * 1e-3f;
        uint8_t* dA; float *dB, *dD;
        CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
        const float sB = 4096.f;
        hipLaunchKernelGGL(check_f16_subnormal, dim3(1), dim3(64), 0, 0, dA, dB, sB, dD);
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        double maxrel = 0, maxref = 0; int zeros = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 16; ++k) { _Float16 hb = (_Float16)(B[k * 32 + j] * sB); s += (double)A[i * 16 + k] * (double)(float)hb; }
            const double got = (double)D[i * 32 + j] * 16777216.0;   // x * 2^-24 as a subnormal
            if (D[i * 32 + j] == 0.f && s != 0) ++zeros;
            if (fabs(s) > maxref) maxref = fabs(s);
            if (fabs(got - s) > maxrel) maxrel = fabs(got - s);
        }
        printf("{\"probe\": \"f16_subnormal_A\", \"flushed_results\": %d, \"max_abs_err\": %.3g, \"max_ref\": %.3g}\n", zeros, maxrel, maxref);
    }
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;   // 4000 x 32 MFMAs x 32 cycles = 4.1 M cycles = 1.7 ms per launch
    for (int data = 0; data < 3; ++data) {
        run_rate<0>("bf16_32x32x16", data, 512, iters, 10);
        run_rate<1>("f16_32x32x16", data, 512, iters, 10);
        run_rate<2>("i8_32x32x32", data, 512, iters, 10);
    }
    run_rate<0>("bf16_32x32x16", 1, 256, iters, 10);
    run_rate<2>("i8_32x32x32", 1, 256, iters, 10);
    // short launches from idle (the shape of one predict sweep): 50 us each
    run_rate<0>("bf16_32x32x16", 1, 512, 60, 1);
    run_rate<2>("i8_32x32x32", 1, 512, 60, 1);
    run_rate<0>("bf16_32x32x16", 1, 512, 60, 20);
    run_rate<2>("i8_32x32x32", 1, 512, 60, 20);
    return 0;
}
