// Layer-1 backward + Adam of minibatch t CHAINED with the layer-1 forward of minibatch t + 1; widths that pad to 512, 256,
// 128 or 64 (round 4: NHT = 16, 8, 4, 2 unit tiles; with fewer than 8 a workgroup owns 2 or 4 k-tiles at a time so that all
// eight of its waves stream weights whatever the width, with 16 every wave streams two unit tiles per k-tile)
// (reference: one `model.fit` step after another, /root/reference/locator/locator.py:367-376; layer 1 is
// BatchNormalization + Dense(width, elu), :318-320).
//
// The backward kernel of l1_kernels.hip holds every updated weight W1' in registers once per step; the next step's
// forward contracts the same W1' with the next minibatch's xhat.  Run as two kernels the forward re-reads all of W1
// (102 MB at 100k SNPs: 24 us of a 192 us step).  Here the weight tile is used for the next forward while it is
// still in registers, so a step streams W1 / m / v exactly once (read + write) and the separate forward launch
// disappears for every step but the first of an epoch.
//
// What makes that legal: xhat_{t+1}[b][k] = gamma'_k * (x - mu_k) * rstd_k + beta'_k needs the gamma / beta that step t
// itself updates, and their gradients reduce over ALL units of SNP k.  So one workgroup owns whole k-tiles (g, g + G,
// g + 2G, ...: at any moment the G workgroups stream one contiguous window of W1 / m / v); wave w (of 8) streams unit
// tile w of each, per k-tile:
//     Gn[h][k]     = sum_b dZ[b][h] xn[b][k]        ONE fp32 MFMA chain (A = dZ from LDS, B = xn = (x - mean) * rstd)
//     dW^T[h][k]   = gamma_k Gn[h][k] + beta_k sum_b dZ[b][h]                                -> Adam on W, m, v, 16-byte stores
//     (sum_b dxhat xn, sum_b dxhat)[k] = (sum_h W[h][k] Gn[h][k], sum_h W[h][k] sum_b dZ[b][h])   this wave's 32 units -> LDS
//     --- one LDS-only barrier per k-tile (the weight prefetch stays in flight across it) ---
//     every wave adds the 8 partial sums in wave order, applies Adam to gamma_k / beta_k (wave 0 stores them and the
//     next step's [scale|shift|mean|rstd]), builds xhat_{t+1} for the tile, and contracts it with its W' tile:
//     z_{t+1}[b][h] += sum_k xhat_{t+1}[b][k] W'[k][h]   (A = xhat_{t+1}, lane = row; B = W' through a 4 KB per-wave
//     LDS transpose: the accumulator layout has lane = SNP, the B operand needs lane = unit)
// and the workgroup leaves partial[g][32][256] for l1_reduce_kernel, exactly like l1_fwd_partial_kernel.
// Adam per weight as in l1_bwd_adam_kernel (v_rcp / v_sqrt).  dW1 and the gamma / beta gradient are associated differently
// from that kernel (which forms sum_b dZ xhat and dxhat = dZ W^T with two MFMA chains): same sums, fp32 round-off differs;
// against the fp64 oracle this association is the closer one (tests/chain_vs_oracle.py).
// Trailing workgroups of the launch run the step's hidden-layer / head Adam tail (stack_tail.h).
//
// Narrower layers (round 4).  With NHT < 8 unit tiles a k-tile keeps only NHT waves busy, and the kernel lives on bytes in
// flight per compute unit, so a workgroup owns KTW = 8 / NHT CONSECUTIVE k-tiles at a time (a
// "super-tile"): wave w streams unit tile w % NHT of k-tile slot w / NHT.  Everything per k-tile - the small operands in
// LDS, the cross-wave gamma / beta reduction (now over the NHT waves of a slot), the gamma / beta Adam, the next step's
// scale / shift - exists once per slot; the 7 loader roles per slot are dealt round-robin over the 8 waves (1, 2 or 4 roles
// per wave: more untracked small loads per wave, all OLDER than the 12 streaming loads the one hand-counted wait counts,
// so the count stays 12).  A wave's forward accumulator covers its slot's k-tiles only, so the launch leaves KTW partial
// groups per workgroup (partial[g * KTW + slot]) for l1_reduce_kernel.  The last super-tile may be short: its missing
// slots run the same loads on dummy addresses, contribute zeros and store nothing.
//
// Width 512 (NHT = 16).  Sixteen waves would leave each 128 registers - less than the two weight / moment register sets, the
// MFMA accumulators and the forward accumulators need - so the workgroup stays at eight waves and wave w streams unit tiles
// w and w + 8 of every k-tile one after the other ("sub-steps"): the register sets keep alternating (tile w in A while
// w + 8 lands in B, w + 8 in B while the next k-tile's w lands in A), every sub-step is the same "12 loads | wait for 12"
// pattern, the wave adds its two (dgamma | dbeta) partials before they go to LDS, keeps two transposed W' tiles and two
// forward accumulators, and everything per k-tile (barrier, gamma / beta Adam, small operands) happens after the second
// sub-step.  dZ alone is 64 KB of LDS at this width, the kernel's 147 KB still fit one workgroup per compute unit.
#include <type_traits>
#include "common.h"
#include "stack_tail.h"

#define KT 32

// timing ablations (make ablate_chain A=<bits>): results are WRONG with any bit set; they only answer "what does this
// part of the loop cost".  1 = no barrier, 2 = no per-tile small operands, 4 = no next-forward part, 16 = no Adam
// arithmetic, 32 = no weight stores
#ifndef LOC_CHAIN_ABLATE
#define LOC_CHAIN_ABLATE 0
#endif

__device__ __forceinline__ void ch_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void ch_adam_fast(float& w, float& m, float& v, float g, float alpha) {
    m = m + (g - m) * ADAM_C1;
    v = v + (g * g - v) * ADAM_C2;
    w = w - (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + ADAM_EPS);
}

// Vector-memory loads the compiler does not track, with hand-counted s_waitcnt (as in l1_gemm.hip).  Left to itself the
// compiler waits for the 12 KB prefetch it has just issued before the first use of the CURRENT register set (its
// per-register wait counts are merged conservatively around the loop), so nothing overlaps.  The "memory" clobbers keep
// these loads, the compiler's own stores and the waits in program order, which is what the counts below rely on.
// Every untracked load is issued UNCONDITIONALLY (a dummy address where there is nothing to fetch): the compiler believes
// an asm output is valid at once, so a load under a branch would meet "not loaded" in a phi, and a copy inserted for that
// phi would read the register before the data lands.  tests/test_chain_asm.py checks the generated code for exactly that.
// 16 bytes at (wave-uniform base) + (32-bit byte offset of the lane) + OFF: one offset register serves W1, m and v
template <int OFF, bool NT>
__device__ __forceinline__ void ch_gload16(f32x4& v, const float* base, uint32_t voff) {
    if (NT) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(v) : "v"(voff), "s"(base), "n"(OFF) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
// The two wait states behind the store are part of it: on gfx940+ a vector-memory store of more than 8 bytes still reads its
// data registers for two wait states after it issues, and a vector-ALU write to them in that window corrupts what is
// stored.  The compiler's hazard recognizer pads its OWN stores; it cannot see into an asm statement, and once the data
// registers are dead after the asm it is free to reuse them at once - round 4's register allocation of the second step did
// (`v_pk_add_f32 v[24:25]` right behind `global_store_dwordx4 v142, v[22:25]`: wrong Adam moments in memory, right weights).
template <int OFF, bool NT>
__device__ __forceinline__ void ch_gstore16(const f32x4& v, float* base, uint32_t voff) {
    if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}
// 4 bytes at a per-lane 64-bit address
__device__ __forceinline__ void ch_gload4(uint32_t& v, const void* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
// Wait until at most N vector-memory operations are outstanding; the operands tie the registers the wait protects, so
// no use of them can be scheduled above it.  COUNTING RULE (gfx9 has one counter for loads and stores): loads retire in
// order among loads and stores among stores, but a store may retire before an older load.  So "at most N outstanding"
// proves that a load has landed only if N is the number of LOADS issued after it -- stores never count.
// (a -DLOC_CHAIN_DEBUG_DRAIN build - `make debug_drain`, the parity-debug twin library - turns every hand count into
// vmcnt(0); tests/test_gpu_chain.py compares the two builds bit for bit)
#ifdef LOC_CHAIN_DEBUG_DRAIN
#define CH_VMCNT(N) 0
#else
#define CH_VMCNT(N) (N)
#endif
template <int N>
__device__ __forceinline__ void ch_wait_unit(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], uint32_t& s0, uint32_t& s1,
                                             uint32_t& s2) {
    asm volatile("s_waitcnt vmcnt(%15)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(s0), "+v"(s1), "+v"(s2)
                 : "n"(CH_VMCNT(N))
                 : "memory");
}

constexpr int CH_TP = 33;
constexpr int CH_SM = 896;   // floats of one k-tile's small operands in LDS (see `sm` in the kernel)
constexpr int ch_ktw(int nht) { return nht >= 8 ? 1 : 8 / nht; }     // k-tiles (slots) a workgroup owns at a time
constexpr int ch_upw(int nht) { return nht > 8 ? nht / 8 : 1; }      // unit tiles a wave streams per k-tile (sub-steps)
// rb = 32-row blocks per minibatch (1: --batch_size <= 32; 2: 33..64, width 256 only): dZ, the row lists and the two genotype
// tiles of a k-tile's small operands grow with it
constexpr int ch_smf(int rb) { return CH_SM + 512 * (rb - 1); }       // floats of one k-tile's small operands
constexpr size_t ch_lds_floats(int nht, int rb = 1) {
    return 32 * rb * (nht * 32 + 1) + 64 * rb + 8 * ch_upw(nht) * 32 * CH_TP + 2 * 8 * 64 + 8 * 64 + 2 * ch_ktw(nht) * ch_smf(rb) +
           ((nht != 8 || rb > 1) ? nht * 32 : 0);
}

// The hand-counted wait: at most N vector-memory operations outstanding; the operands tie every register an untracked load
// of this wave may still be writing (the current unit's three register sets and the small-operand words of its loader
// roles), so no use of them can be scheduled above it.
template <int N>
__device__ __forceinline__ void ch_wait_unit(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], uint32_t (&ld)[1][3]) {
    ch_wait_unit<N>(a, b, c, ld[0][0], ld[0][1], ld[0][2]);
}
template <int N>
__device__ __forceinline__ void ch_wait_unit(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], uint32_t (&ld)[2][3]) {
    asm volatile("s_waitcnt vmcnt(%18)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(ld[0][0]), "+v"(ld[0][1]), "+v"(ld[0][2]),
                   "+v"(ld[1][0]), "+v"(ld[1][1]), "+v"(ld[1][2])
                 : "n"(CH_VMCNT(N))
                 : "memory");
}
template <int N>
__device__ __forceinline__ void ch_wait_unit(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], uint32_t (&ld)[4][3]) {
    asm volatile("s_waitcnt vmcnt(%24)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(ld[0][0]), "+v"(ld[0][1]), "+v"(ld[0][2]),
                   "+v"(ld[1][0]), "+v"(ld[1][1]), "+v"(ld[1][2]), "+v"(ld[2][0]), "+v"(ld[2][1]), "+v"(ld[2][2]),
                   "+v"(ld[3][0]), "+v"(ld[3][1]), "+v"(ld[3][2])
                 : "n"(CH_VMCNT(N))
                 : "memory");
}

template <int NTM, int NHT, int RB>
__global__ __launch_bounds__(512) void l1_bwd_adam_chain_kernel(
    const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows, int n_b,
    const int32_t* __restrict__ rows_next, int n_b_next, int K, int Kp, float* bn4,
    const float* __restrict__ next_stats, const float* __restrict__ dz1, float* __restrict__ w1s,
    float* __restrict__ m1s, float* __restrict__ v1s, float* __restrict__ gamma, float* __restrict__ beta,
    float* __restrict__ m_gamma, float* __restrict__ v_gamma, float* __restrict__ m_beta, float* __restrict__ v_beta,
    float* __restrict__ b1, float* __restrict__ m_b1, float* __restrict__ v_b1, const float* __restrict__ alpha_tab,
    int alpha_tab_len, const float* __restrict__ lr, const int* __restrict__ t_base, int t_off,
    float* __restrict__ partial_out, int n_tail, loc_dw_tail_args ta) {
    constexpr int Hp = NHT * 32, PZ = Hp + 1, TP = CH_TP;
    constexpr int KTW = ch_ktw(NHT);             // k-tiles (slots) a workgroup owns at a time
    constexpr int UPW = ch_upw(NHT);             // unit tiles a wave streams per k-tile, one after the other
    constexpr int WPS = NHT / UPW;               // waves per slot
    constexpr int NR = 32 * RB;                  // rows the workgroup is built for (RB row blocks of 32)
    constexpr int NK = 4 * RB + 3;               // loader roles per k-tile: 2 RB + 2 RB genotype pieces, bn4, gamma / beta, next stats
    constexpr int NROLE = NK * KTW;              // loader roles per super-tile
    constexpr int RPW = (NROLE + 7) / 8;         // ... per wave
    constexpr int SMF = ch_smf(RB);              // floats of one k-tile's small operands
    constexpr int FO = 512 * (RB - 1);           // shift of their float section behind the (RB times larger) genotype tiles
    constexpr bool DZS_REG = NHT == 8 && RB == 1;   // dZ column sums in registers (else re-read from LDS)
    static_assert(NHT == 16 || NHT == 8 || NHT == 4 || NHT == 2, "unit tiles per k-tile");
    static_assert(RB == 1 || (RB == 2 && NHT == 8), "row blocks: two at width 256 only");
    // Trailing workgroups (n_tail of them): the step's other Adam tail -- hidden-layer dW / db, heads, batch loss
    // (stack_tail.h).  It depends on nothing this kernel writes.  Workgroups are dispatched in index order, so these start
    // when the first layer-1 workgroups retire: 3125 k-tiles over 256 workgroups leave most compute units idle during the
    // last iteration of the others, which is where this work goes instead of into a launch of its own.
    const int G = (int)gridDim.x - n_tail;
    if ((int)blockIdx.x >= G) {
        loc_gb_tail none = {};
        // (two row blocks: a short last minibatch uses one, so the tail counts its blocks at run time)
        stack_dw_all_body<NHT, RB == 1 ? 1 : 0>((int)blockIdx.x - G, ta, none);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzl = smem;                                          // [NR][PZ]   dZ of this step
    int* rows_l = reinterpret_cast<int*>(dzl + NR * PZ);        // [NR]
    int* rown_l = rows_l + NR;                                  // [NR]       rows of the next minibatch
    float* Tt = reinterpret_cast<float*>(rown_l + NR);          // [8][UPW][32][TP] per-wave W' tiles, [SNP][unit]
    float* red = Tt + 8 * UPW * 32 * TP;                            // [2][8][64]  per-wave (dgamma | dbeta) partials, by k-tile parity
    float* ssl = red + 2 * 8 * 64;                              // [8][64]     per-wave (scale' | shift') of the next step
    // [2][KTW][CH_SM] the small operands of a k-tile (slot), fetched two super-tiles ahead by the loader roles (one to three
    // load instructions each instead of 26 per wave): bytes 0..1023 genotype tile of this minibatch [32 rows][32 SNPs],
    // 1024..2047 the same for the next minibatch, then floats [scale|shift|mean|rstd][32], (gamma|beta), their Adam
    // m, v [64] each, next [mean|var][32]
    float* sm = ssl + 8 * 64;
    float* dzs_l = sm + 2 * KTW * SMF;                          // [Hp] column sums of dZ (unless DZS_REG: see dzs below)

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    const int ut0 = w % WPS, kq = w / WPS;       // this wave's (first) unit tile, and its k-tile slot inside the super-tile
    const int nkt = Kp / KT, S = (nkt + KTW - 1) / KTW;
    // k-tile of slot `slot` of super-tile T, or -1 (no such super-tile, or past the end of a short last one)
    auto ktile = [&](int T, int slot) { return (T >= 0 && T < S && T * KTW + slot < nkt) ? T * KTW + slot : -1; };
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    const bool chain = rows_next != nullptr;

    // byte offset of this lane's first 16 bytes of unit (kt, w) in each of W1S / m / v  (Kp * 1024 < 2^32: checked by the launcher)
    auto unit_off = [&](int kt, int ut) { return (uint32_t)(((uint32_t)kt * NHT + ut) * 4096u + lane * 16u); };
    auto load_unit = [&](int kt, int ut, f32x4 (&wq)[4], f32x4 (&mq)[4], f32x4 (&vq)[4]) {
        // 12 loads, ALWAYS (the wait counts depend on it).  kt < 0: nothing left to fetch -- the same 12 instructions with
        // ONE address for the whole wave (one 16-byte request each instead of 1 KB), a different line per wave so that
        // the 2048 waves do not queue on one channel; results unused
        const uint32_t o = kt >= 0 ? unit_off(kt, ut) : (uint32_t)((((uint32_t)blockIdx.x * 8 + w) & 2047u) * 64u);
        ch_gload16<0, (NTM & 2) != 0>(wq[0], w1s, o);    ch_gload16<0, (NTM & 1) != 0>(mq[0], m1s, o);    ch_gload16<0, (NTM & 1) != 0>(vq[0], v1s, o);
        ch_gload16<1024, (NTM & 2) != 0>(wq[1], w1s, o); ch_gload16<1024, (NTM & 1) != 0>(mq[1], m1s, o); ch_gload16<1024, (NTM & 1) != 0>(vq[1], v1s, o);
        ch_gload16<2048, (NTM & 2) != 0>(wq[2], w1s, o); ch_gload16<2048, (NTM & 1) != 0>(mq[2], m1s, o); ch_gload16<2048, (NTM & 1) != 0>(vq[2], v1s, o);
        ch_gload16<3072, (NTM & 2) != 0>(wq[3], w1s, o); ch_gload16<3072, (NTM & 1) != 0>(mq[3], m1s, o); ch_gload16<3072, (NTM & 1) != 0>(vq[3], v1s, o);
    };
    // The first unit is requested before anything else; the prologue ends with vmcnt(0).
    f32x4 wA[4], mA[4], vA[4], wB[4], mB[4], vB[4];
    load_unit(ktile((int)blockIdx.x, kq), ut0, wA, mA, vA);

    if constexpr (RB == 1) {
        for (int i = t; i < 32 * Hp; i += 512) dzl[(i / Hp) * PZ + (i % Hp)] = dz1[i];
    } else {
        // the stack kernel carries (and zeroes beyond n_b) the rows of the 32-row blocks IN USE only: a short last minibatch
        // leaves the other block's dZ rows stale
        const int n_used = 32 * ((n_b + 31) / 32);
        for (int i = t; i < NR * Hp; i += 512) dzl[(i / Hp) * PZ + (i % Hp)] = (i / Hp) < n_used ? dz1[i] : 0.f;
    }
    if (t < NR) {
        rows_l[t] = t < n_b ? rows[t] : 0;
        rown_l[t] = (chain && t < n_b_next) ? rows_next[t] : 0;
    }
    __syncthreads();

    // dzsum[h] = sum_b dZ[b][h] for this lane's 16 units h = w*32 + rowmap(r, hi): column sums through `red` (free until
    // the loop; the barriers below separate the two uses), then registers for the whole kernel
    if (t < Hp) {
        float sum = 0.f;
#pragma unroll 8
        for (int b = 0; b < NR; ++b) sum += dzl[b * PZ + t];
        red[t] = sum;
    }
    __syncthreads();
    // (every width but 256 re-reads them from LDS at each use instead: two unit tiles per wave would need 32 registers, and the
    // narrow widths' extra loader-role words already spill without the 16)
    float dzs_r[16];
    if constexpr (DZS_REG) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dzs_r[r] = red[ut0 * 32 + rowmap(r, hi)];
    } else {
        if (t < Hp) dzs_l[t] = red[t];
    }

    // gamma (lanes 0..31) or beta (lanes 32..63) of SNP jl of the tile: one Adam per lane.  beta / m_beta / v_beta sit Kp
    // floats behind gamma / m_gamma / v_gamma (loc_param_layout; checked by the launcher): one wave-uniform base each
    const int gbo = hi * Kp;
    (void)beta; (void)m_beta; (void)v_beta;
    // loader roles (wave-uniform): role r = w + 8 rr (rr < RPW) serves slot r / 7 with kind r % 7: kinds 0, 1 the genotype
    // tile (8 bytes per lane), 2, 3 the next minibatch's, 4 bn4, 5 gamma / beta + moments, 6 the next minibatch's batch
    // statistics.  fetch() issues the global loads of super-tile T, stage() puts them into sm[buf] an iteration later (under
    // load a small load takes about as long as a big one).
    uint32_t ld[RPW][3];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) ld[rr][0] = ld[rr][1] = ld[rr][2] = 0u;
    auto fetch = [&](int T) {
        // three 4-byte loads per role and lane, ALWAYS; the role (and "nothing to fetch") only chooses the addresses
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int role = w + 8 * rr, slot = role / NK, kind = role - NK * slot;
            const int kt = role < NROLE ? ktile(T, slot) : -1;
            const int ktc = kt >= 0 ? kt : 0;
            const int k = ktc * KT + jl;
            const void *a0 = alpha_tab + w, *a1 = a0, *a2 = a0;        // every "nothing to fetch" case: one read-only word per wave
            if (kt >= 0) {
                if (kind < 4 * RB) {
                    const int pc = (kind & (2 * RB - 1)) * 64 + lane;  // 8-byte piece of the [NR rows][32 bytes] tile
                    const int row = kind < 2 * RB ? rows_l[pc >> 2] : rown_l[pc >> 2];   // (no next minibatch: row 0, unused)
                    const uint8_t* a = X + (int64_t)row * pitch + (int64_t)ktc * KT + 8 * (pc & 3);
                    a0 = a; a1 = a + 4; a2 = a;
                } else if (kind == 4 * RB) {
                    a0 = bn4 + (int64_t)hi * Kp + k; a1 = bn4 + (int64_t)(2 + hi) * Kp + k; a2 = a0;
                } else if (kind == 4 * RB + 1) {
                    a0 = gamma + gbo + k; a1 = m_gamma + gbo + k; a2 = v_gamma + gbo + k;
                } else if (kind == 4 * RB + 2 && chain) {
                    a0 = next_stats + (int64_t)hi * Kp + k; a1 = a0; a2 = a0;
                }
            }
            ch_gload4(ld[rr][0], a0); ch_gload4(ld[rr][1], a1); ch_gload4(ld[rr][2], a2);
        }
    };
    auto stage = [&](int buf, int T) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int role = w + 8 * rr, slot = role / NK, kind = role - NK * slot;
            if (role >= NROLE || ktile(T, slot) < 0) continue;
            const uint32_t ld0 = ld[rr][0], ld1 = ld[rr][1], ld2 = ld[rr][2];
            float* b = sm + (buf * KTW + slot) * SMF;
            float* bf = b + FO;                       // the float section (offsets as for one row block)
            if (kind < 2 * RB) {
                // this minibatch's tile goes in TRANSPOSED, [SNP][32 bytes], the byte of row b at position
                // 16 * ((b >> 2) & 1) + (b & 3) + 4 * (b >> 3): lane (SNP jl, half hi) then reads its 16 rows rowmap(r, hi),
                // r = 0..15, as one 16-byte word
                // (RB row blocks: [SNP][RB][32 bytes])
                uint8_t* xb = reinterpret_cast<uint8_t*>(b);
                const int pc = (kind & (2 * RB - 1)) * 64 + lane, row = pc >> 2, snp0 = 8 * (pc & 3), r5 = RB == 1 ? row : (row & 31);
                const int pos = (RB == 1 ? 0 : 32 * (row >> 5)) + 16 * ((r5 >> 2) & 1) + (r5 & 3) + 4 * (r5 >> 3);
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[(snp0 + e) * NR + pos] = (uint8_t)(((e < 4 ? ld0 : ld1) >> (8 * (e & 3))) & 255u);
            } else if (kind < 4 * RB) {
                uint2 v; v.x = ld0; v.y = ld1;
                *reinterpret_cast<uint2*>(b + 256 * RB + ((kind & (2 * RB - 1)) * 64 + lane) * 2) = v;
            } else if (kind == 4 * RB) {
                bf[512 + lane] = bitsf(ld0);
                bf[576 + lane] = bitsf(ld1);
            } else if (kind == 4 * RB + 1) {
                bf[640 + lane] = bitsf(ld0); bf[704 + lane] = bitsf(ld1); bf[768 + lane] = bitsf(ld2);
            } else {
                bf[832 + lane] = bitsf(ld0);
            }
        }
    };
    {
        const int T0 = blockIdx.x;
        fetch(T0);
        ch_wait_unit<0>(wA, mA, vA, ld);
        stage(0, T0);
        fetch(T0 + G);                                          // staged during the first iteration
        ch_wait_unit<0>(wA, mA, vA, ld);
    }
    __syncthreads();

    // bias of layer 1: db1[h] = sum_b dZ[b][h]   (workgroup 0, the waves of slot 0: wave <-> unit tile)
    if (blockIdx.x == 0 && kq == 0) {
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int ut = ut0 + WPS * u;
            float s = 0.f;
            if constexpr (RB == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += dzl[rowmap(r, hi) * PZ + ut * 32 + jl];
                s += __shfl_xor(s, 32);
            } else {
                s = dzs_l[ut * 32 + jl];               // the column sums over all NR rows (in LDS since the prologue)
            }
            if (hi == 0) {
                const int h = ut * 32 + jl;
                float wv = b1[h], mv = m_b1[h], vv = v_b1[h];
                adam_update(wv, mv, vv, s, alpha);
                b1[h] = wv; m_b1[h] = mv; v_b1[h] = vv;
            }
        }
    }

    float* Tw0 = Tt + w * UPW * 32 * TP;
    float* ssw = ssl + w * 64;
    bool row_next_ok[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) row_next_ok[rb] = chain && 32 * rb + jl < n_b_next;

    f32x16 facc[UPW][RB];
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) facc[u][rb] = f32x16{0};
    float pg_acc = 0.f, pb_acc = 0.f;     // this wave's (dgamma | dbeta) partial over the sub-steps of a k-tile

    // Vector-memory operations of one sub-step, in program order (the hand-counted wait depends on it):
    //   12 prefetch loads | wait | 12 stores of this unit | last sub-step of the k-tile: 3 small loads of tile + 2 per role |
    //   barrier | 0..5 small stores
    // SUB = which of the wave's UPW unit tiles of k-tile (T, slot kq) this is; the unit prefetched meanwhile is sub-step
    // (SUB + 1) % UPW of super-tile T_pref (= T unless this is the last sub-step)
    auto step = [&](auto sub_c, int T, int T_pref, int T_next, int T_next2, int par, f32x4 (&wq)[4], f32x4 (&mq)[4],
                    f32x4 (&vq)[4], f32x4 (&wn)[4], f32x4 (&mn)[4], f32x4 (&vn)[4]) {
        constexpr int SUB = decltype(sub_c)::value;
        constexpr bool first = SUB == 0, last = SUB == UPW - 1;
        const int ut = ut0 + WPS * SUB;
        float* Tw = Tw0 + SUB * 32 * TP;
        float dzs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dzs[r] = DZS_REG ? dzs_r[r] : dzs_l[ut * 32 + rowmap(r, hi)];
        const int ktv = ktile(T, kq);
        // false only for the missing slots of a short last super-tile (wave-uniform); a workgroup that owns whole k-tiles
        // (UPW > 1 implies KTW == 1) has none
        const bool valid = UPW > 1 || ktv >= 0;
        const int kt = valid ? ktv : 0;
        const int k = kt * KT + jl;
        // this tile's small operands from LDS
        const float* smc = sm + (par * KTW + kq) * SMF;
        const float* smf = smc + FO;             // the float section (offsets as for one row block)
        const uint8_t* xt = reinterpret_cast<const uint8_t*>(smc);
        u32x4 xp[RB];                            // 16 rows of SNP jl per row block, packed
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xp[rb] = *reinterpret_cast<const u32x4*>(xt + jl * NR + rb * 32 + hi * 16);
        auto xv = [&](int rb, int r) { return (float)((xp[rb][r >> 2] >> (8 * (r & 3))) & 255u); };
        const float mu = smf[576 + jl], rs = smf[608 + jl];
        const float gam = smf[640 + jl], bet = smf[672 + jl];          // gamma_k, beta_k before this step's update
        float pv = smf[640 + lane], pm = smf[704 + lane], pvv = smf[768 + lane];
        float nmu = smf[832 + jl], nvar = smf[864 + jl];
        u32x4 xr[RB];                            // next minibatch: row 32 rb + jl, SNPs 16 hi .. 16 hi + 15
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xr[rb] = *reinterpret_cast<const u32x4*>(smc + 256 * RB + (rb * 32 + jl) * 8 + 4 * hi);
        // ONE wait per iteration, right after the next unit's 12 loads (always 12: dummies on the last tile): "at most 12
        // outstanding" proves every older load landed -- this unit (requested an iteration ago) and the next tile's small
        // operands ld0..2 (requested at the end of the previous iteration).
        load_unit(ktile(T_pref, kq), ut0 + WPS * ((SUB + 1) % UPW), wn, mn, vn);
        ch_wait_unit<12>(wq, mq, vq, ld);

        // ONE fp32 MFMA chain per unit:  Gn[h][k] = sum_b dZ[b][h] xn[b][k]  (D[i = unit][j = SNP], contraction over the
        // batch rows b = rowmap(s, hi); xn = (x - mean) * rstd).  Everything else follows from it without forming dxhat
        // (rows beyond the minibatch need no mask: their dZ rows are zero and the bytes staged for them are finite):
        //   dW^T[h][k]          = sum_b dZ[b][h] (gamma_k xn[b][k] + beta_k)        = gamma_k Gn[h][k] + beta_k dzsum[h]
        //   sum_b dxhat[b][k] xn[b][k] = sum_b sum_h dZ[b][h] W[h][k] xn[b][k]      = sum_h W[h][k] Gn[h][k]
        //   sum_b dxhat[b][k]          = sum_h W[h][k] dzsum[h]                      (this wave's 32 units; old W)
        f32x16 g = {0};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                g = mfma32(dzl[(32 * rb + rowmap(s, hi)) * PZ + ut * 32 + jl], (xv(rb, s) - mu) * rs, g);
        {
            float pg = 0.f, pb = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pg = fmaf(wq[r >> 2][r & 3], g[r], pg);
                pb = fmaf(wq[r >> 2][r & 3], dzs[r], pb);
            }
            pg += __shfl_xor(pg, 32);
            pb += __shfl_xor(pb, 32);
            if (!first) { pg += pg_acc; pb += pb_acc; }
            if (last) red[(par * 8 + w) * 64 + lane] = valid ? (hi ? pb : pg) : 0.f;
            else { pg_acc = pg; pb_acc = pb; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = fmaf(gam, g[r], bet * dzs[r]);
        // Adam on the weight tile, stores, and the tile's transpose for the next forward
        const uint32_t so = unit_off(kt, ut);
        auto adam4 = [&](int q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float wv = wq[q][c], mv = mq[q][c], vv = vq[q][c];
#if LOC_CHAIN_ABLATE & 16
                wv += g[q * 4 + c];
#else
                ch_adam_fast(wv, mv, vv, g[q * 4 + c], alpha);
#endif
                wq[q][c] = wv; mq[q][c] = mv; vq[q][c] = vv;
            }
        };
        // 12 stores, always
#if LOC_CHAIN_ABLATE & 32
        adam4(0); adam4(1); adam4(2); adam4(3);
#else
        // (stores never enter the load count - see the counting rule - so a missing slot may simply skip them)
        if (!valid) { adam4(0); adam4(1); adam4(2); adam4(3); } else {
        adam4(0); ch_gstore16<0, (NTM & 4) != 0>(wq[0], w1s, so);    ch_gstore16<0, (NTM & 8) != 0>(mq[0], m1s, so);    ch_gstore16<0, (NTM & 8) != 0>(vq[0], v1s, so);
        adam4(1); ch_gstore16<1024, (NTM & 4) != 0>(wq[1], w1s, so); ch_gstore16<1024, (NTM & 8) != 0>(mq[1], m1s, so); ch_gstore16<1024, (NTM & 8) != 0>(vq[1], v1s, so);
        adam4(2); ch_gstore16<2048, (NTM & 4) != 0>(wq[2], w1s, so); ch_gstore16<2048, (NTM & 8) != 0>(mq[2], m1s, so); ch_gstore16<2048, (NTM & 8) != 0>(vq[2], v1s, so);
        adam4(3); ch_gstore16<3072, (NTM & 4) != 0>(wq[3], w1s, so); ch_gstore16<3072, (NTM & 8) != 0>(mq[3], m1s, so); ch_gstore16<3072, (NTM & 8) != 0>(vq[3], v1s, so);
        }
#endif
        if (chain && valid && !(LOC_CHAIN_ABLATE & 4)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Tw[jl * TP + rowmap(r, hi)] = wq[r >> 2][r & 3];   // T[SNP][unit]
        }
        if (!last) return;            // the rest once per k-tile, after the wave's last unit tile
        // The next tile's small operands (landed: see the wait above) go to the other LDS buffer, then the request for
        // the tile after next.
        if (!(LOC_CHAIN_ABLATE & 2)) stage(par ^ 1, T_next);
        fetch(T_next2);
#if !(LOC_CHAIN_ABLATE & 1)
        // every wave's (dgamma | dbeta) partial of this k-tile and the next tile's small operands are in LDS; all reads
        // of this tile's small operands are above this line, so the buffer is free for tile + 2 after it
        ch_lds_barrier();
#endif

        float dsum = red[(par * 8 + kq * WPS) * 64 + lane];       // the waves of this slot, in wave order
#pragma unroll
        for (int w2 = 1; w2 < WPS; ++w2) dsum += red[(par * 8 + kq * WPS + w2) * 64 + lane];
        adam_update(pv, pm, pvv, dsum, alpha);
        const bool live = valid && k < K;
        if (ut0 == 0 && live) { gamma[gbo + k] = pv; m_gamma[gbo + k] = pm; v_gamma[gbo + k] = pvv; }
        if (chain && valid && !(LOC_CHAIN_ABLATE & 4)) {
            const float other = __shfl_xor(pv, 32);
            const float gam = hi ? other : pv, bet = hi ? pv : other;
            float rstd = 1.0f / sqrtf(nvar + BN_EPS);
            float scn = gam * rstd;
            float shn = bet - nmu * scn;
            if (!live) { scn = 0.f; shn = 0.f; nmu = 0.f; rstd = 0.f; }
            if (ut0 == 0) {
                bn4[(hi ? Kp : 0) + k] = hi ? shn : scn;
                bn4[(int64_t)(2 + hi) * Kp + k] = hi ? rstd : nmu;
            }
            ssw[lane] = hi ? shn : scn;
            __builtin_amdgcn_wave_barrier();
            // xhat of the next minibatch for SNPs 16*hi + s of the tile (lane = row jl), times W'[SNP][unit = jl]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(ssw + 16 * hi + 4 * j);
                const f32x4 h4 = *reinterpret_cast<const f32x4*>(ssw + 32 + 16 * hi + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float xb = (float)((xr[rb][j] >> (8 * e)) & 255u);
                        a[rb] = row_next_ok[rb] ? fmaf(xb, s4[e], h4[e]) : 0.f;   // rows beyond the next minibatch: staged from row 0, unused
                    }
#pragma unroll
                    for (int u = 0; u < UPW; ++u) {
                        const float wt = Tw0[(u * 32 + 16 * hi + 4 * j + e) * TP + jl];
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) facc[u][rb] = mfma32(a[rb], wt, facc[u][rb]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // workgroup g owns k-tiles g, g + G, g + 2G, ...: at any moment the G workgroups stream one contiguous G * 32 KB
    // window of W1 / m / v, which spreads over every HBM channel
    {
        using sub0 = std::integral_constant<int, 0>;
        using sub1 = std::integral_constant<int, 1>;
        for (int T = blockIdx.x; T < S; T += 2 * G) {
            auto nx = [&](int j) { return T + j * G < S ? T + j * G : -1; };
            if constexpr (UPW == 1) {
                step(sub0{}, T, nx(1), nx(1), nx(2), 0, wA, mA, vA, wB, mB, vB);
                if (T + G < S) step(sub0{}, T + G, nx(2), nx(2), nx(3), 1, wB, mB, vB, wA, mA, vA);
            } else {
                step(sub0{}, T, T, nx(1), nx(2), 0, wA, mA, vA, wB, mB, vB);
                step(sub1{}, T, nx(1), nx(1), nx(2), 0, wB, mB, vB, wA, mA, vA);
                if (T + G < S) {
                    step(sub0{}, T + G, T + G, nx(2), nx(3), 1, wA, mA, vA, wB, mB, vB);
                    step(sub1{}, T + G, nx(2), nx(2), nx(3), 1, wB, mB, vB, wA, mA, vA);
                }
            }
        }
    }
    // the last iteration's dummy requests are still in flight: nothing below may reuse their registers before they land
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (chain) {
        // D[i = row b][j = unit]: lane holds unit w*32 + jl, rows rowmap(r, hi) -- the layout l1_reduce_kernel sums
        // (one partial group per k-tile slot of this workgroup: group g * KTW + slot)
        float* pout = partial_out + ((int64_t)blockIdx.x * KTW + kq) * NR * Hp;
#pragma unroll
        for (int u = 0; u < UPW; ++u)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pout[(32 * rb + rowmap(r, hi)) * Hp + (ut0 + WPS * u) * 32 + jl] = facc[u][rb][r];
    }
}

extern "C" int loc_l1_chain_supported(int Hp) { return Hp == 512 || Hp == 256 || Hp == 128 || Hp == 64; }
// partial groups the chained kernel leaves per workgroup (k-tile slots a workgroup owns at a time)
extern "C" int loc_l1_chain_groups_per_workgroup(int Hp) { return loc_l1_chain_supported(Hp) ? ch_ktw(Hp / 32) : 0; }

int l1_chain_launch(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                          const int32_t* rows_next, int n_b_next, const loc_dims* d, float* bn4,
                                          const float* bn_next_stats, const float* dz1, float* w1s, float* m1s,
                                          float* v1s, float* gamma, float* beta, float* m_gamma, float* v_gamma,
                                          float* m_beta, float* v_beta, float* b1, float* m_b1, float* v_b1,
                                          const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                                          int t_off, int grid, float* partial, int64_t partial_floats,
                                          const loc_tuning* tune, const loc_dw_tail_args* tail, int rb, void* stream) {
    // rb = 32-row blocks the kernel is built for (1, or 2 for --batch_size 33..64 at width 256): fixed for a fit, because
    // the partial sums it leaves are [group][32 rb][Hp]
    if (rb < 1 || rb > 2 || (rb == 2 && d->Hp != 256)) {
        loc_set_error("loc_l1_backward_adam_chain: %d row blocks at width %d (two need width 256)", rb, d->Hp);
        return -1;
    }
    const int max_rows = LOC_ROWS * rb;
    if (!loc_l1_chain_supported(d->Hp)) { loc_set_error("loc_l1_backward_adam_chain: width must pad to 64, 128, 256 or 512 (got %d)", d->Hp); return -1; }
    if (n_b < 1 || n_b > max_rows) { loc_set_error("loc_l1_backward_adam_chain: n_b=%d out of 1..%d", n_b, max_rows); return -1; }
    if (rows_next && (n_b_next < 1 || n_b_next > max_rows || !bn_next_stats)) {
        loc_set_error("loc_l1_backward_adam_chain: the next minibatch needs 1..%d rows (got %d) and its batch statistics",
                      max_rows, n_b_next);
        return -1;
    }
    if ((x_pitch % 16) != 0 || ((uintptr_t)X % 16) != 0) {
        loc_set_error("loc_l1_backward_adam_chain: genotype rows must be 16-byte aligned (pitch %lld)", (long long)x_pitch);
        return -1;
    }
    if (beta != gamma + d->Kp || m_beta != m_gamma + d->Kp || v_beta != v_gamma + d->Kp) {
        loc_set_error("loc_l1_backward_adam_chain: beta / m_beta / v_beta must sit Kp floats behind gamma / m_gamma / v_gamma "
                      "(loc_param_layout)");
        return -1;
    }
    if ((int64_t)d->Kp * 1024 >= ((int64_t)1 << 32)) {
        loc_set_error("loc_l1_backward_adam_chain: more than 4M SNPs exceed the kernel's 32-bit byte offsets");
        return -1;
    }
    const int nht = d->Hp / 32, ktw = ch_ktw(nht);
    const int nkt = d->Kp / KT, n_super = (nkt + ktw - 1) / ktw;
    if (grid < 1) grid = 1;
    if (grid > n_super) grid = n_super;
    if (rows_next && (int64_t)grid * ktw * max_rows * d->Hp > partial_floats) {
        loc_set_error("loc_l1_backward_adam_chain: partial buffer too small for %d workgroups x %d groups", grid, ktw);
        return -1;
    }
    const size_t lds = ch_lds_floats(nht, rb) * sizeof(float);
    // the step's hidden-layer / head Adam tail as trailing workgroups: (L - 1) * 64 weight tiles + the head block
    loc_dw_tail_args ta = {};
    int n_tail = 0;
    if (tail) {
        if (tail->n_b != n_b || tail->L < 2) { loc_set_error("loc_l1_backward_adam_chain: inconsistent tail arguments"); return -1; }
        ta = *tail;
        n_tail = (tail->L - 1) * nht * nht + 1;
    }
    const int ntm = !tune || tune->l1b_nt_mask == 0 ? 13 : (tune->l1b_nt_mask < 0 ? 0 : tune->l1b_nt_mask);
#define LAUNCH_CHAIN_NR(M, N, R)                                                                                   \
    {                                                                                                              \
        LOC_ENSURE_LDS((l1_bwd_adam_chain_kernel<M, N, R>), lds);                                                  \
        hipLaunchKernelGGL((l1_bwd_adam_chain_kernel<M, N, R>), dim3(grid + n_tail), dim3(512), lds, (hipStream_t)stream, X, \
                           x_pitch, rows, n_b, rows_next, n_b_next, d->K, d->Kp, bn4, bn_next_stats, dz1, w1s, m1s, v1s, \
                           gamma, beta, m_gamma, v_gamma, m_beta, v_beta, b1, m_b1, v_b1, alpha_tab, alpha_tab_len, lr, \
                           t_base, t_off, partial, n_tail, ta);                                                    \
    }
#define LAUNCH_CHAIN_N(M, N) LAUNCH_CHAIN_NR(M, N, 1)
#define LAUNCH_CHAIN(M)                                                                                            \
    {                                                                                                              \
        if (nht == 16) LAUNCH_CHAIN_N(M, 16) else if (nht == 8) LAUNCH_CHAIN_N(M, 8)                               \
        else if (nht == 4) LAUNCH_CHAIN_N(M, 4) else LAUNCH_CHAIN_N(M, 2)                                          \
    }
    if (rb == 2) {                        // two row blocks: width 256, default cache policy
        LAUNCH_CHAIN_NR(13, 8, 2)
        LOC_CHECK_LAUNCH();
        return 0;
    }
    if (nht != 8) {                       // the cache-policy measurement switches exist for the default width only
        LAUNCH_CHAIN(13)
        LOC_CHECK_LAUNCH();
        return 0;
    }
    switch (ntm) {
        case 0: LAUNCH_CHAIN_N(0, 8) break;
        case 9: LAUNCH_CHAIN_N(9, 8) break;
        case 15: LAUNCH_CHAIN_N(15, 8) break;
        default: LAUNCH_CHAIN_N(13, 8) break;
    }
#undef LAUNCH_CHAIN_NR
#undef LAUNCH_CHAIN_N
#undef LAUNCH_CHAIN
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_l1_backward_adam_chain(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                          const int32_t* rows_next, int n_b_next, const loc_dims* d, float* bn4,
                                          const float* bn_next_stats, const float* dz1, float* w1s, float* m1s,
                                          float* v1s, float* gamma, float* beta, float* m_gamma, float* v_gamma,
                                          float* m_beta, float* v_beta, float* b1, float* m_b1, float* v_b1,
                                          const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                                          int t_off, int grid, float* partial, int64_t partial_floats,
                                          const loc_tuning* tune, void* stream) {
    return l1_chain_launch(X, x_pitch, rows, n_b, rows_next, n_b_next, d, bn4, bn_next_stats, dz1, w1s, m1s, v1s, gamma, beta,
                           m_gamma, v_gamma, m_beta, v_beta, b1, m_b1, v_b1, alpha_tab, alpha_tab_len, lr, t_base, t_off,
                           grid, partial, partial_floats, tune, nullptr, 1, stream);
}
