// what does the split-K hand-over of the large-M int8 GEMM (l1_gemm_i8.hip) cost in each of the
// forms that could replace "fp32 slabs + l1_gemm_reduce_kernel", with the GEMM's own geometry and nothing else running?
//
//   geometry   rows x 256 units, 128-row tiles (n_mt of them), G = 256 / n_mt SNP groups per tile, one workgroup of 512
//              threads per (tile, group), block b on XCD b % 8 with g = xcd + 8 (idx / n_mt) - l1_gemm_i8_kernel's mapping.
//              A wave owns 32 units x 128 rows as four 32x32 accumulator tiles of two digit planes (lane = unit, 16 rows
//              per lane and tile: rowmap of common.h).
//   slab       what ships: every workgroup writes its 128 x 256 fp32 tile (hi 256 + lo) delta with 16-byte stores through
//              a wave-private LDS image; a second launch adds the G slabs in a fixed order (+ shift + b1, ELU).
//   atomic32   every accumulator register goes out as one no-return agent-scope int32 atomicAdd per digit plane into
//              acc[plane][rows][256] (exact and order-independent: genotypes <= 3, |digit| <= 128); the last workgroup of a
//              tile to arrive (ticket counter) turns the sums into a1 and zeroes them for the next call.
//   atomic64   the same with one 64-bit atomicAdd of hi 256 + lo per output (half the requests).
//   ticket     slabs as in `slab`, written with plain stores + agent release; the tile's last arriver adds the G slabs itself
//              (the guide's in-launch split-K ending) - no second launch, 4 MB read by one workgroup per tile at G = 32.
//
//   hipcc --offload-arch=gfx950 -O3 splitk_reduce_probe.hip -o splitk_reduce_probe && ./splitk_reduce_probe [rows]
// Prints one JSON line per (rows, variant): microseconds per call (mean of 20 back-to-back calls) and a checksum that must be
// the same for every variant of a shape.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }

// the accumulator a real GEMM would hold: a cheap exact function of (plane, row, unit, group)
__device__ __forceinline__ int fake_acc(int p, int row, int unit, int g) {
    return ((row * 131 + unit * 17 + g * 7 + p * 3) % 2001) - 1000;
}

__device__ __forceinline__ void tile_of(int& mt, int& g, int n_mt, int G) {
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
}

// ---- slab: workgroup writes its fp32 tile; MODE 1 = also take a ticket and let the last arriver reduce
template <int MODE>
__global__ __launch_bounds__(512) void slab_kernel(float* __restrict__ partial, int G, int n_mt, const float* __restrict__ delta,
                                                   unsigned* __restrict__ cnt, const float* __restrict__ cb, float* __restrict__ a1) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;
    int mt, g;
    tile_of(mt, g, n_mt, G);
    const int Mp = n_mt * 128;
    float* const ep = reinterpret_cast<float*>(smem) + w * (128 * 32);
    float* const pout = partial + ((int64_t)g * Mp + mt * 128) * 256 + w * 32;
    const float dl = delta[w * 32 + jl];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + rowmap(r, hi);
            ep[row * 32 + jl] = (256.f * (float)fake_acc(0, mt * 128 + row, w * 32 + jl, g) +
                                 (float)fake_acc(1, mt * 128 + row, w * 32 + jl, g)) * dl;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i * 8 + lane / 8, c4 = (lane % 8) * 4;
        *reinterpret_cast<f32x4*>(pout + (int64_t)row * 256 + c4) = *reinterpret_cast<const f32x4*>(ep + row * 32 + c4);
    }
    if (MODE == 0) return;
    // in-launch ending (cdna_hip_programming.md, split-K recipe): drain, barrier, ONE agent release, ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem);            // the one LDS array, reused
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned k = __hip_atomic_fetch_add(cnt + mt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (k == (unsigned)(G - 1)) ? 1u : 0u;
        if (*flag) { cnt[mt] = 0; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    }
    __syncthreads();
    if (!*flag) return;
    // the last arriver adds the G slabs of its tile in group order: 128 x 256 outputs, 16 f32x4 per thread
    for (int i = 0; i < 16; ++i) {
        const int64_t o4 = ((int64_t)mt * 128 * 256) + ((int64_t)i * 512 + t) * 4;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int gg = 0; gg < G; ++gg) s = s + *reinterpret_cast<const f32x4*>(partial + (int64_t)gg * Mp * 256 + o4);
        const int h = (int)(o4 & 255);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = elu_f(s[e] + cb[h + e]);
        *reinterpret_cast<f32x4*>(a1 + o4) = o;
    }
}

__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ partial, int G, int64_t MH, const float* __restrict__ cb,
                                                     float* __restrict__ a1) {
    __shared__ f32x4 red[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i4 = ((int64_t)blockIdx.x * 64 + o) * 4;
    const int gq = (G + 3) / 4, g0 = q * gq, g1 = g0 + gq < G ? g0 + gq : G;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g = g0; g < g1; ++g) s = s + *reinterpret_cast<const f32x4*>(partial + (int64_t)g * MH + i4);
    red[q][o] = s;
    __syncthreads();
    if (q == 0) {
        f32x4 z = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        const int h = (int)(i4 & 255);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = elu_f(z[e] + cb[h + e]);
        *reinterpret_cast<f32x4*>(a1 + i4) = r;
    }
}

// ---- atomics: W = 32: two int32 adds per output; W = 64: one int64 add of hi 256 + lo
template <int W>
__global__ __launch_bounds__(512) void atomic_kernel(int32_t* __restrict__ acc32, long long* __restrict__ acc64, int G, int n_mt,
                                                     const float* __restrict__ delta, unsigned* __restrict__ cnt,
                                                     const float* __restrict__ cb, float* __restrict__ a1) {
    __shared__ unsigned flag;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;
    int mt, g;
    tile_of(mt, g, n_mt, G);
    const int64_t Mp = n_mt * 128;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt * 128 + tm * 32 + rowmap(r, hi), unit = w * 32 + jl;
            const int a0 = fake_acc(0, row, unit, g), a1v = fake_acc(1, row, unit, g);
            if (W == 32) {
                __hip_atomic_fetch_add(acc32 + (int64_t)row * 256 + unit, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(acc32 + (Mp + row) * 256 + unit, a1v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(acc64 + (int64_t)row * 256 + unit, (long long)a0 * 256 + a1v, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        const unsigned k = __hip_atomic_fetch_add(cnt + mt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = (k == (unsigned)(G - 1)) ? 1u : 0u;
        if (flag) cnt[mt] = 0;
    }
    __syncthreads();
    if (!flag) return;
    // last arriver of the tile: sums -> a1, and zero them for the next call (atomic exchange: the sums live at the memory side)
    for (int i = 0; i < 64; ++i) {
        const int64_t o = ((int64_t)mt * 128 * 256) + (int64_t)i * 512 + t;
        float z;
        if (W == 32) {
            const int s0 = __hip_atomic_exchange(acc32 + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int s1 = __hip_atomic_exchange(acc32 + Mp * 256 + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z = (256.f * (float)s0 + (float)s1);
        } else {
            z = (float)__hip_atomic_exchange(acc64 + o, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int h = (int)(o & 255);
        a1[o] = elu_f(z * delta[h] + cb[h]);
    }
}

static double checksum(const float* d_a1, int64_t n) {
    float* h = (float*)malloc(n * 4);
    CK(hipMemcpy(h, d_a1, n * 4, hipMemcpyDeviceToHost));
    double s = 0;
    for (int64_t i = 0; i < n; ++i) s += (double)h[i] * (1 + (i % 7));
    free(h);
    return s;
}

template <typename F>
static float time_us(F&& call, int iters = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) call();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) call();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int shapes_default[] = {1000, 4096, 16384};
    for (int si = 0; si < 3; ++si) {
        const int rows = argc > 1 ? atoi(argv[1]) : shapes_default[si];
        const int n_mt = (rows + 127) / 128;
        int G = 256 / n_mt; if (G < 1) G = 1; if (G >= 8) G &= ~7;
        const int64_t Mp = (int64_t)n_mt * 128, MH = Mp * 256;
        float *partial, *delta, *cb, *a1; int32_t* acc32; long long* acc64; unsigned* cnt;
        CK(hipMalloc(&partial, (size_t)G * MH * 4)); CK(hipMalloc(&delta, 1024)); CK(hipMalloc(&cb, 1024)); CK(hipMalloc(&a1, MH * 4));
        CK(hipMalloc(&acc32, 2 * MH * 4)); CK(hipMalloc(&acc64, MH * 8)); CK(hipMalloc(&cnt, n_mt * 4));
        CK(hipMemset(acc32, 0, 2 * MH * 4)); CK(hipMemset(acc64, 0, MH * 8)); CK(hipMemset(cnt, 0, n_mt * 4));
        float hd[256], hc[256];
        for (int i = 0; i < 256; ++i) { hd[i] = 1.f / (float)(1 << (10 + i % 3)); hc[i] = 0.01f * (i % 11) - 0.05f; }
        CK(hipMemcpy(delta, hd, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(cb, hc, 1024, hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute((const void*)slab_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        CK(hipFuncSetAttribute((const void*)slab_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        const dim3 grid(n_mt * G);
        auto slab = [&] {
            hipLaunchKernelGGL(slab_kernel<0>, grid, dim3(512), 131072, 0, partial, G, n_mt, delta, cnt, cb, a1);
            hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)(MH / 256)), dim3(256), 0, 0, partial, G, MH, cb, a1);
        };
        auto slab_only = [&] { hipLaunchKernelGGL(slab_kernel<0>, grid, dim3(512), 131072, 0, partial, G, n_mt, delta, cnt, cb, a1); };
        auto reduce_only = [&] { hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)(MH / 256)), dim3(256), 0, 0, partial, G, MH, cb, a1); };
        auto ticket = [&] { hipLaunchKernelGGL(slab_kernel<1>, grid, dim3(512), 131072, 0, partial, G, n_mt, delta, cnt, cb, a1); };
        auto at32 = [&] { hipLaunchKernelGGL(atomic_kernel<32>, grid, dim3(512), 0, 0, acc32, acc64, G, n_mt, delta, cnt, cb, a1); };
        auto at64 = [&] { hipLaunchKernelGGL(atomic_kernel<64>, grid, dim3(512), 0, 0, acc32, acc64, G, n_mt, delta, cnt, cb, a1); };
        struct { const char* name; float us; double sum; } res[6];
        int nr = 0;
#define RUN(nm, fn) { CK(hipMemset(a1, 0, MH * 4)); float us = time_us(fn); res[nr++] = {nm, us, checksum(a1, MH)}; }
        RUN("slab_write+reduce_launch", slab)
        RUN("slab_write_only", slab_only)
        RUN("reduce_launch_only", reduce_only)
        RUN("slab_write+last_arriver_reduces", ticket)
        RUN("atomic_int32_x2+last_arriver", at32)
        RUN("atomic_int64+last_arriver", at64)
        for (int i = 0; i < nr; ++i)
            printf("{\"rows\": %d, \"row_tiles\": %d, \"groups\": %d, \"variant\": \"%s\", \"us\": %.2f, \"checksum\": %.6e}\n", rows, n_mt, G,
                   res[i].name, res[i].us, res[i].sum);
        fflush(stdout);
        hipFree(partial); hipFree(delta); hipFree(cb); hipFree(a1); hipFree(acc32); hipFree(acc64); hipFree(cnt);
        if (argc > 1) break;
    }
    return 0;
}
