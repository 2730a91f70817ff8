// Which XCDs does a CU-masked stream run on?  (hipExtStreamCreateWithCUMask; bit b of the mask = "CU b" in the
// runtime's numbering.)  Each workgroup records its XCC id (s_getreg_b32 HW_REG_XCC_ID) and spins a little so that the
// grid spreads over every CU the stream may use.   hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void probe(int* xcc, int* cu, long long spin) {
    if (threadIdx.x == 0) {
        unsigned x, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        xcc[blockIdx.x] = x & 0xF;
        cu[blockIdx.x] = hwid;
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
    }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
    const int n = 2048;
    int *dx, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, s, dx, dc, 2000LL);   // 20 us per block at 100 MHz
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<int> hx(n), hc(n);
    hipMemcpy(hx.data(), dx, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0};
    for (int i = 0; i < n; ++i) hist[hx[i] & 15]++;
    std::vector<int> seen;
    for (int i = 0; i < n; ++i) { int key = (hx[i] << 16) | (hc[i] & 0xFFFFF0); bool f = false; for (int v : seen) f |= v == key; if (!f) seen.push_back(key); }
    printf("%-28s %7.3f ms, %3zu distinct (xcc, hw_id>>4):", name, ms, seen.size());
    for (int i = 0; i < 8; ++i) printf(" %4d", hist[i]);
    printf("\n");
    hipFree(dx); hipFree(dc);
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = 8;                                   // 256 bits
    run("no mask", {});
    std::vector<uint32_t> m(words, 0);
    for (int b = 0; b < 256; ++b) if (b % 8 == 7) m[b / 32] |= 1u << (b % 32);
    run("bits b % 8 == 7", m);
    std::fill(m.begin(), m.end(), 0);
    for (int b = 0; b < 256; ++b) if (b % 8 != 7) m[b / 32] |= 1u << (b % 32);
    run("bits b % 8 != 7", m);
    std::fill(m.begin(), m.end(), 0);
    for (int b = 224; b < 256; ++b) m[b / 32] |= 1u << (b % 32);
    run("bits 224..255", m);
    std::fill(m.begin(), m.end(), 0);
    for (int b = 0; b < 32; ++b) m[b / 32] |= 1u << (b % 32);
    run("bits 0..31", m);
    std::fill(m.begin(), m.end(), 0);
    for (int b = 0; b < 8; ++b) m[b / 32] |= 1u << (b % 32);
    run("bits 0..7", m);
    std::vector<uint32_t> one(1, 0xFF);
    run("1 word, 0xFF", one);
    std::vector<uint32_t> w16(16, 0);
    w16[0] = 0xF;
    run("16 words, bits 0..3", w16);
    return 0;
}
