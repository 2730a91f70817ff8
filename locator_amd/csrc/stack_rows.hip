// Hidden Dense(256, elu) layers + Dense(2) x 2 for MANY rows in inference mode (reference: model.predict on thousands of
// rows - the batched --jacknife replicates, predictions over large sample sets, /root/reference/locator/locator.py:414,
// :441, :683-747; layers :319-325).
//
// The row-parallel kernel of stack_fused.hip carries 2 rows per workgroup through the layers on the vector ALU and
// re-streams all (L - 1) x 256 KB of weights from L2 for every pair of rows: right for a 32-row training step (16
// workgroups, latency-bound) and for the 90-row validation sweep, but for 4096 rows it moves 4.8 GB through the L2 -> CU
// ports and takes longer than the first-layer GEMM it follows (tools/predict_timeline.py, round 4: 235 us against 170 us at
// 4096 rows x 100,000 SNPs; 800 against 640 at 16,384).  Here a workgroup takes 32 rows (one MFMA M tile) through every layer
// on v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation - the same arithmetic as the vector-ALU kernel up to the
// order of the sums - and streams the weights once per 32 rows (16 x less L2 traffic); activations never leave the LDS.
//   wave w (of 8) owns output units [32 w, 32 w + 32) of every layer:  D[i = row][j = unit] += A[i][k] B[k][j], two k per
//   MFMA: k = s + 128 hi for lane half hi, s = 0..127, so a lane's A values are CONSECUTIVE floats of its row in the LDS
//   tile (one ds_read_b128 per four MFMAs) and its B values are W[k][32 w + jl].
//   The weights reach that layout through a per-wave LDS stage (round 4, late): fetched as 16-byte-per-lane requests (1 KB
//   contiguous runs, 4 per chunk of 16 steps) three chunks ahead, across layer boundaries (they do not depend on the
//   activations), parked in the stage, read back one value per MFMA.  Fetched straight into the MFMA layout (one 4-byte load
//   per MFMA and lane) the kernel made 1,024 vector-memory instructions per layer and compute unit and its MFMA loop followed
//   their NUMBER - 18-26k cycles per layer where the MFMAs need 16.4k, 19.5k with a quarter of the loads (timing build
//   `make sr_stamps XDEF=-DSR_ABLATE=1`), unchanged by where the lines come from, by the prefetch depth, by a second
//   accumulator chain or by earlier LDS reads (tools/probes/sr_stamps.py).  Staged: 18-22k.
// Optional input stage as in stack_fused.hip: the many-row GEMM's SNP-group partial sums are added up here (+ shift + b1,
// ELU), same association as l1_gemm_reduce_kernel.
#include "common.h"
#include <type_traits>

#define SR_HP 256
#define SR_ROWS 32
#define SR_PITCH 260     /* floats per activation row in LDS: 16-lane ds_read_b128 groups start 4 banks apart */

// -DSR_STAMPS=<block>: a measurement build (tools/probes/sr_stamps.py) in which every wave of one workgroup leaves the cycle
// counter at the phase boundaries of every layer
#ifdef SR_STAMPS
__device__ unsigned long long sr_stamp_buf[8 * 64];
#define SR_STAMP(i)                                                                                       \
    if ((int)blockIdx.x == SR_STAMPS && (threadIdx.x & 63) == 0) sr_stamp_buf[(threadIdx.x >> 6) * 64 + (i)] = __builtin_readcyclecounter();
extern "C" int loc_debug_sr_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sr_stamp_buf), sizeof(unsigned long long) * 8 * 64);
}
#else
#define SR_STAMP(i)
#endif

__device__ __forceinline__ void sr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512) void stack_rows_eval_kernel(
    const float* __restrict__ a1, const float* __restrict__ rd_partial, int rd_G, int64_t rd_MH,
    const float* __restrict__ rd_cvec8, const float* __restrict__ rd_b1, const float* __restrict__ Wh,
    const float* __restrict__ bh, const float* __restrict__ wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ bb, int L, int n_b, const int32_t* __restrict__ rows, const float* __restrict__ Y,
    float* __restrict__ yhat, float* __restrict__ dist) {
    constexpr int Hp = SR_HP, P = SR_PITCH;
    // two activation tiles + the head partials (68.6 KB) + two weight-chunk stages per wave (67.6 KB)
    extern __shared__ __attribute__((aligned(16))) float sr_smem[];
    float (*act)[SR_ROWS * P] = reinterpret_cast<float (*)[SR_ROWS * P]>(sr_smem);
    float (*hp)[SR_ROWS][2] = reinterpret_cast<float (*)[SR_ROWS][2]>(sr_smem + 2 * SR_ROWS * P);
    float* wst = sr_smem + 2 * SR_ROWS * P + 8 * SR_ROWS * 2;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * SR_ROWS;
    const int64_t HH = (int64_t)Hp * Hp;

    // weight stream: chunk c of a layer = steps s = 16 c .. 16 c + 15 for both lane halves: the 32 rows W[16 c + r + 128 h][32 w ..
    // 32 w + 31] (r < 16, h < 2), 128 bytes each.  One 4-byte load per MFMA and lane - the layout the MFMA wants - makes 1,024
    // vector-memory instructions per layer and compute unit, and the MFMA loop was bound by their number (see the header).  So a
    // chunk travels as FOUR 16-byte loads per lane (lane = 8 rows x 8 column quads: 1 KB contiguous runs), is parked in a
    // per-wave LDS stage [32 rows][32] and read back one value per MFMA: LDS operations of one wave execute in order, so the
    // wave needs no barrier between its own stage writes and reads, and nobody else touches its stage.
    constexpr int WS = 32 * 32 + 32;                     // floats per stage; rows 16..31 (lane half 1) shifted by 32 floats = 32 banks
    float* stg = wst + w * 2 * WS;
    const int grow = lane >> 3, gcol = 4 * (lane & 7);   // this lane's row (of 8 per request) and column quad
    f32x4 g[4];
    auto gload = [&](const float* __restrict__ W, int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ri = 8 * q + grow;                 // 0..31: half ri >> 4, step ri & 15
#if defined(SR_ABLATE) && SR_ABLATE == 1
            g[q] = *reinterpret_cast<const f32x4*>(W + (int64_t)(128 * (ri >> 4)) * Hp + 32 * w + gcol);   // timing only (wrong results)
#else
            g[q] = *reinterpret_cast<const f32x4*>(W + (int64_t)(16 * c + (ri & 15) + 128 * (ri >> 4)) * Hp + 32 * w + gcol);
#endif
        }
    };
    auto sput = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ri = 8 * q + grow;
            *reinterpret_cast<f32x4*>(stg + buf * WS + ri * 32 + (ri >> 4) * 32 + gcol) = g[q];
        }
    };
    auto bget = [&](int buf, float (&bv)[16]) {
        const float* p = stg + buf * WS + (hi * 16) * 32 + hi * 32 + jl;
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = p[e * 32];
    };
    float bA[16], bB[16];
    gload(Wh, 0); sput(0);
    gload(Wh, 1); sput(1);
    gload(Wh, 2);

    // ---- input: rows of this workgroup -> act[0]
    if (rd_partial != nullptr) {
        const int gq = (rd_G + 3) / 4;
        for (int i = t; i < SR_ROWS * Hp; i += 512) {
            const int r = i / Hp, n = i % Hp;
            float v = 0.f;
            if (r0 + r < n_b) {
                const float* src = rd_partial + (int64_t)(r0 + r) * Hp + n;
                float sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g0 = q * gq, g1 = g0 + gq < rd_G ? g0 + gq : rd_G;
                    float z = 0.f;
                    for (int g = g0; g < g1; ++g) z += src[(int64_t)g * rd_MH];
                    sq[q] = z;
                }
                float c = rd_cvec8[n];
#pragma unroll
                for (int sl = 1; sl < 8; ++sl) c += rd_cvec8[sl * Hp + n];
                v = elu_f((((sq[0] + sq[1]) + sq[2]) + sq[3]) + (c + rd_b1[n]));
            }
            act[0][r * P + n] = v;
        }
    } else {
        for (int i = t; i < SR_ROWS * Hp; i += 512) {
            const int r = i / Hp, n = i % Hp;
            act[0][r * P + n] = r0 + r < n_b ? a1[(int64_t)(r0 + r) * Hp + n] : 0.f;
        }
    }
    __syncthreads();

    // ---- layers 2..L
    int cur = 0;
    bget(0, bA);
    for (int l = 2; l <= L; ++l) {
        const float* Wc = Wh + (int64_t)(l - 2) * HH;
        const float* Wn = l < L ? Wc + HH : Wh;                // after the last layer: a dummy prefetch, never used
        const float bias = bh[(int64_t)(l - 2) * Hp + 32 * w + jl];
        const float* arow = act[cur] + jl * P + 128 * hi;      // this lane's row, its half of the k range
        SR_STAMP(4 * (l - 2) + 0)
        f32x16 acc = {0};
        // chunk c (even) sits in bA and came from stage 0, chunk c + 1 goes stage 1 -> bB; `g` holds chunk c + 2 on entry.  After
        // the last chunk of a layer the stream continues with the next layer's first chunks (they do not depend on activations).
#pragma unroll 1
        for (int c = 0; c < 8; c += 2) {
            {
                bget(1, bB);
                f32x4 a4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = *reinterpret_cast<const f32x4*>(arow + 16 * c + 4 * q);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc = mfma32(a4[e >> 2][e & 3], bA[e], acc);
                sput(0);                                       // chunk c + 2 (stage 0 was read into bA an iteration ago)
                if (c + 3 < 8) gload(Wc, c + 3); else gload(Wn, c + 3 - 8);
            }
            {
                bget(0, bA);                                   // chunk c + 2
                f32x4 a4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = *reinterpret_cast<const f32x4*>(arow + 16 * (c + 1) + 4 * q);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc = mfma32(a4[e >> 2][e & 3], bB[e], acc);
                sput(1);                                       // chunk c + 3
                if (c + 4 < 8) gload(Wc, c + 4); else gload(Wn, c + 4 - 8);
            }
        }
        SR_STAMP(4 * (l - 2) + 1)
        // bias + ELU -> the other activation buffer: lane holds unit 32 w + jl of rows rowmap(r, hi)
        float* out = act[cur ^ 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) out[rowmap(r, hi) * P + 32 * w + jl] = elu_f(acc[r] + bias);
        cur ^= 1;
        SR_STAMP(4 * (l - 2) + 2)
        sr_lds_barrier();
        SR_STAMP(4 * (l - 2) + 3)
    }
    asm volatile("" ::"v"(bA[0]), "v"(bB[0]), "v"(g[0]));

    // ---- Dense(2), Dense(2), distance (locator.py:324-325, :314-315): per row, fixed summation order
    {
        const float* af = act[cur];
        const int u = 32 * w + jl;
        const float w0 = wa[2 * u], w1 = wa[2 * u + 1];
        // lane (unit u, half hi) covers rows 16 hi .. 16 hi + 15: sum over the 32 units of this wave by lane shuffles
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            const int row = 16 * hi + rr;
            const float a = af[row * P + u];
            float p0 = a * w0, p1 = a * w1;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { p0 += __shfl_xor(p0, o); p1 += __shfl_xor(p1, o); }
            if (jl == 0) { hp[w][row][0] = p0; hp[w][row][1] = p1; }
        }
        sr_lds_barrier();
        if (t < SR_ROWS) {
            const int b = r0 + t;
            if (b < n_b) {
                float y10 = ba[0], y11 = ba[1];
                for (int w2 = 0; w2 < 8; ++w2) { y10 += hp[w2][t][0]; y11 += hp[w2][t][1]; }
                const float y20 = y10 * wb[0] + y11 * wb[2] + bb[0];
                const float y21 = y10 * wb[1] + y11 * wb[3] + bb[1];
                yhat[2 * (int64_t)b] = y20;
                yhat[2 * (int64_t)b + 1] = y21;
                if (dist != nullptr && Y != nullptr) {
                    const int64_t yr = rows[b];
                    const float e0 = y20 - Y[2 * yr], e1 = y21 - Y[2 * yr + 1];
                    dist[b] = sqrtf(fmaxf(e0 * e0 + e1 * e1, 0.f));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 16 rows per workgroup on v_mfma_f32_16x16x4_f32 (round 5).  At 3072..8191 rows per chunk the 32-row form has 96..255
// workgroups for 256 compute units and every one of them walks its nine dependent layers for ~60 us whatever the row count
// (117 us at 4096 rows where the MFMAs of a FULL chip would need 38).  Halving the row tile doubles the workgroups and halves
// each one's MFMA time per layer (128 MFMAs of 32 cycles per wave and layer instead of 128 of 64); the weight stream per
// workgroup is the same, i.e. twice the L2 -> CU traffic per row (2.3 MB per 16 rows: 0.6 GB at 4096 rows, far from a limit).
//   wave w (of 8) owns units [32 w, 32 w + 32) as two 16-unit tiles t:  D[i = row][j = unit] += A[i][k] B[k][j], four k per MFMA:
//   lane = (i16 = lane & 15, kq = lane >> 4), k = 64 kq + s for step s = 0..63, so that a lane's A values are CONSECUTIVE floats
//   of its row (one ds_read_b128 per four steps = eight MFMAs) and its B values are W[64 kq + s][32 w + 16 t + i16], read back
//   from the same per-wave LDS stage as the 32-row form: chunk c = steps 8 c .. 8 c + 7 of the four k quarters = 32 rows of W x
//   32 units = four 16-byte requests per lane, three chunks ahead across layer boundaries.  Stage row ri = 8 kq + r at
//   ri 32 + 16 (ri >> 3) floats: the four k quarters of a read fall on different bank halves.
//   D register r of a lane is row 4 kq + r, unit 16 t + i16.
//   CS = steps per weight chunk: 4 is what ships (71 KB of LDS: TWO workgroups per compute unit, so that one's epilogue, barrier
//   and stage traffic sit under the other's MFMAs once there are more workgroups than compute units); 8 (104 KB, one
//   workgroup per compute unit) was the first form and measured 3-15 % slower at every row count
//   (profiles/r05_stack_rows_bench_2percu.jsonl: 4096 rows 66.5 against 63.6 us, 12,288 rows 186 against 165).
#define SR16_ROWS 16
template <int CS>
__global__ __launch_bounds__(512, CS == 4 ? 2 : 1) void stack_rows16_eval_kernel(
    const float* __restrict__ a1, const float* __restrict__ rd_partial, int rd_G, int64_t rd_MH,
    const float* __restrict__ rd_cvec8, const float* __restrict__ rd_b1, const float* __restrict__ Wh,
    const float* __restrict__ bh, const float* __restrict__ wa, const float* __restrict__ ba, const float* __restrict__ wb,
    const float* __restrict__ bb, int L, int n_b, const int32_t* __restrict__ rows, const float* __restrict__ Y,
    float* __restrict__ yhat, float* __restrict__ dist) {
    constexpr int Hp = SR_HP, P = SR_PITCH, R = SR16_ROWS;
    extern __shared__ __attribute__((aligned(16))) float sr_smem[];
    float (*act)[R * P] = reinterpret_cast<float (*)[R * P]>(sr_smem);
    float (*hp)[R][2] = reinterpret_cast<float (*)[R][2]>(sr_smem + 2 * R * P);
    float* wst = sr_smem + 2 * R * P + 8 * R * 2;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i16 = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * R;
    const int64_t HH = (int64_t)Hp * Hp;

    static_assert(CS == 8 || CS == 4, "steps per chunk");
    constexpr int NQ = CS / 2;                           // 16-byte requests per lane and chunk (4 CS rows of W x 128 bytes per wave)
    constexpr int NCH = 64 / CS;                         // chunks per layer
    constexpr int WS = 4 * CS * 32 + 64;                 // floats per stage (4 CS rows of 32 + 16 floats of shift per k quarter)
    float* stg = wst + w * 2 * WS;
    const int grow = lane >> 3, gcol = 4 * (lane & 7);   // this lane's row (of 8 per request) and column quad
    // Register ring of RD chunks between the request and the stage write: a chunk is requested RD sub-iterations (of CS steps =
    // 2 CS MFMAs of 32 cycles) before it is parked in the stage, i.e. RD + 2 chunks before its MFMAs.  The first 16-row form
    // kept ONE chunk in registers (the 32-row form's schedule, where a sub-iteration is 1,024 cycles), which with four-step
    // chunks leaves 256 cycles for an L2 round trip.  Measured: RD = 4 against 1 changes nothing at <= 4096 rows (65.6 against
    // 65.3 us: the waves' parked half - profiles/r05_stack_rows16_pmc_4096.json, 0.53 - is not the weight stream) and gives
    // 107 against 111 us at 8192 rows, two workgroups per compute unit; kept.
    constexpr int RD = 4;
    static_assert(NCH % RD == 0 && RD % 2 == 0, "the loop body is RD sub-iterations; stage parity = sub-iteration parity");
    f32x4 g[RD][NQ];
    auto gload = [&](const float* __restrict__ W, int c, f32x4 (&gv)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                   // stage row ri = 8 q + grow = CS x (k quarter) + (step in the chunk)
            const int ri = 8 * q + grow;
            gv[q] = *reinterpret_cast<const f32x4*>(W + (int64_t)(64 * (ri / CS) + CS * c + ri % CS) * Hp + 32 * w + gcol);
        }
    };
    auto sput = [&](int buf, const f32x4 (&gv)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ri = 8 * q + grow;
            *reinterpret_cast<f32x4*>(stg + buf * WS + ri * 32 + 16 * (ri / CS) + gcol) = gv[q];
        }
    };
    auto bget = [&](int buf, float (&bv)[2 * CS]) {      // bv[2 e + t]: step e of the chunk, unit tile t
        const float* p = stg + buf * WS + (CS * kq) * 32 + 16 * kq + i16;
#pragma unroll
        for (int e = 0; e < CS; ++e) {
            bv[2 * e] = p[e * 32];
            bv[2 * e + 1] = p[e * 32 + 16];
        }
    };
    float bA[2 * CS], bB[2 * CS];
    // chunks 0 and 1 of the first layer into the stages, chunks 2 .. RD + 1 into the ring (slot j holds chunk j + 2)
    gload(Wh, 0, g[0]); gload(Wh, 1, g[1]);
    sput(0, g[0]); sput(1, g[1]);
#pragma unroll
    for (int j = 0; j < RD; ++j) gload(Wh, j + 2, g[j]);

    // ---- input: rows of this workgroup -> act[0]
    if (rd_partial != nullptr) {
        const int gq = (rd_G + 3) / 4;
        for (int i = t; i < R * Hp; i += 512) {
            const int r = i / Hp, n = i % Hp;
            float v = 0.f;
            if (r0 + r < n_b) {
                const float* src = rd_partial + (int64_t)(r0 + r) * Hp + n;
                float sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g0 = q * gq, g1 = g0 + gq < rd_G ? g0 + gq : rd_G;
                    float z = 0.f;
                    for (int gg = g0; gg < g1; ++gg) z += src[(int64_t)gg * rd_MH];
                    sq[q] = z;
                }
                float c = rd_cvec8[n];
#pragma unroll
                for (int sl = 1; sl < 8; ++sl) c += rd_cvec8[sl * Hp + n];
                v = elu_f((((sq[0] + sq[1]) + sq[2]) + sq[3]) + (c + rd_b1[n]));
            }
            act[0][r * P + n] = v;
        }
    } else {
        for (int i = t; i < R * Hp / 4; i += 512) {      // 16-byte loads: 64 per row
            const int r = i / (Hp / 4), n4 = (i % (Hp / 4)) * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + r < n_b) v = *reinterpret_cast<const f32x4*>(a1 + (int64_t)(r0 + r) * Hp + n4);
            *reinterpret_cast<f32x4*>(&act[0][r * P + n4]) = v;
        }
    }
    __syncthreads();

    // ---- layers 2..L
    int cur = 0;
    bget(0, bA);
    for (int l = 2; l <= L; ++l) {
        const float* Wc = Wh + (int64_t)(l - 2) * HH;
        const float* Wn = l < L ? Wc + HH : Wh;                // after the last layer: a dummy prefetch, never used
        const float bias0 = bh[(int64_t)(l - 2) * Hp + 32 * w + i16], bias1 = bh[(int64_t)(l - 2) * Hp + 32 * w + 16 + i16];
        const float* arow = act[cur] + i16 * P + 64 * kq;      // this lane's row, its quarter of the k range
        SR_STAMP(4 * (l - 2) + 0)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // sub-iteration j of a loop trip handles chunk c + j: its B values are in bA (j even) / bB (j odd); it fetches chunk
        // c + j + 1 from the other stage, issues its 2 CS MFMAs, parks ring slot j (chunk c + j + 2, requested RD sub-iterations
        // ago) in its own stage and requests chunk c + j + 2 + RD into the slot.  Past the last chunk of a layer the stream
        // continues with the next layer's first chunks (they do not depend on the activations).
        // (stamps, `make sr_stamps` + tools/probes/sr_stamps.py at 4096 rows: MFMA loop 9.6-13.3k cycles per layer and wave where
        // the pipe needs 8.2k per SIMD, epilogue 1.4-2.6k, barrier wait 0.1-2.8k; reading the A values one sub-iteration ahead
        // as well changed none of it and was taken out again)
        auto sub = [&](int c, auto JC) {
            constexpr int j = decltype(JC)::value, par = j & 1;
            float (&bcur)[2 * CS] = par ? bB : bA;
            float (&bnxt)[2 * CS] = par ? bA : bB;
            bget(1 - par, bnxt);
            f32x4 a4[CS / 4];
#pragma unroll
            for (int q = 0; q < CS / 4; ++q) a4[q] = *reinterpret_cast<const f32x4*>(arow + CS * (c + j) + 4 * q);
#pragma unroll
            for (int e = 0; e < CS; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e >> 2][e & 3], bcur[2 * e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e >> 2][e & 3], bcur[2 * e + 1], acc1, 0, 0, 0);
            }
            sput(par, g[j]);
            const int tc = c + j + 2 + RD;
            if (tc < NCH) gload(Wc, tc, g[j]); else gload(Wn, tc - NCH, g[j]);
        };
#pragma unroll 1
        for (int c = 0; c < NCH; c += RD) {
            sub(c, std::integral_constant<int, 0>{});
            sub(c, std::integral_constant<int, 1>{});
            sub(c, std::integral_constant<int, 2>{});
            sub(c, std::integral_constant<int, 3>{});
        }
        SR_STAMP(4 * (l - 2) + 1)
        // bias + ELU -> the other activation buffer: register r of a lane is row 4 kq + r, unit 32 w + 16 t + i16
        float* out = act[cur ^ 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out[(4 * kq + r) * P + 32 * w + i16] = elu_f(acc0[r] + bias0);
            out[(4 * kq + r) * P + 32 * w + 16 + i16] = elu_f(acc1[r] + bias1);
        }
        cur ^= 1;
        SR_STAMP(4 * (l - 2) + 2)
        sr_lds_barrier();
        SR_STAMP(4 * (l - 2) + 3)
    }
    asm volatile("" ::"v"(bA[0]), "v"(bB[0]), "v"(g[0][0]), "v"(g[1][0]), "v"(g[2][0]), "v"(g[3][0]));

    // ---- Dense(2), Dense(2), distance (locator.py:324-325, :314-315): per row, fixed summation order
    {
        const float* af = act[cur];
        const int jl = lane & 31, hi = lane >> 5, u = 32 * w + jl;
        const float w0 = wa[2 * u], w1 = wa[2 * u + 1];
        // lane (unit u, half hi) covers rows 8 hi .. 8 hi + 7: sum over the 32 units of this wave by lane shuffles
#pragma unroll 4
        for (int rr = 0; rr < 8; ++rr) {
            const int row = 8 * hi + rr;
            const float a = af[row * P + u];
            float p0 = a * w0, p1 = a * w1;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { p0 += __shfl_xor(p0, o); p1 += __shfl_xor(p1, o); }
            if (jl == 0) { hp[w][row][0] = p0; hp[w][row][1] = p1; }
        }
        sr_lds_barrier();
        if (t < R) {
            const int b = r0 + t;
            if (b < n_b) {
                float y10 = ba[0], y11 = ba[1];
                for (int w2 = 0; w2 < 8; ++w2) { y10 += hp[w2][t][0]; y11 += hp[w2][t][1]; }
                const float y20 = y10 * wb[0] + y11 * wb[2] + bb[0];
                const float y21 = y10 * wb[1] + y11 * wb[3] + bb[1];
                yhat[2 * (int64_t)b] = y20;
                yhat[2 * (int64_t)b + 1] = y21;
                if (dist != nullptr && Y != nullptr) {
                    const int64_t yr = rows[b];
                    const float e0 = y20 - Y[2 * yr], e1 = y21 - Y[2 * yr + 1];
                    dist[b] = sqrtf(fmaxf(e0 * e0 + e1 * e1, 0.f));
                }
            }
        }
    }
}


// Which form takes how many rows: a workgroup's nine dependent layers take their time whatever the row count, so what counts
// is the time of a ROUND of workgroups (one - or for the 16-row tiles two - per compute unit) and how many rows it covers.
// Measured on 256 compute units (tools/stack_rows_bench.py, profiles/r05_stack_rows_bench*.jsonl):
//   vector ALU (stack_fused.hip), 2 rows per workgroup   26-29 us per round (512 rows)      -> up to 2 x CUs rows
//   vector ALU, 4 rows per workgroup                     30-33 us per round (1024 rows)     -> up to 4 x CUs rows
//   vector ALU, 8 rows per workgroup                     45-48 us per round (2048 rows)     -> up to 8 x CUs rows
//   16-row tiles, two per compute unit                   64 us up to one workgroup per compute unit (4096 rows), then + 50-55
//                                                        us per further 4096 rows (110 us at 8192, 165 at 12,288, 216 at 16,384)
//   32-row tiles, one per compute unit                   103-106 us per round (8192 rows)
// Above 8 x CUs rows the launcher takes the matrix-pipe form with the smaller modelled time: 16-row tiles up to 4096 rows and
// at 8193..12,288, 32-row tiles at 4097..8192 and 12,289..16,384 (a predict chunk is at most 16,384 rows).
#define SR16_FIRST_US 12
#define SR16_ROUND_US 52
#define SR32_ROUND_US 105
int sr_compute_units() {
    static int ncu[LOC_MAX_DEVICES] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 ? 0 : dev % LOC_MAX_DEVICES;
    if (ncu[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        ncu[dev] = v;
    }
    return ncu[dev];
}
static bool sr_takes_16_row_tiles(int n_b) {
    const int cu = sr_compute_units();
    const int r16 = ((n_b + 15) / 16 + cu - 1) / cu, r32 = ((n_b + 31) / 32 + cu - 1) / cu;
    return SR16_FIRST_US + r16 * SR16_ROUND_US < r32 * SR32_ROUND_US;
}
extern "C" int loc_stack_rows_min_rows(void) { return 8 * sr_compute_units() + 1; }
extern "C" int loc_stack_rows_supported(int Hp, int L) { return Hp == SR_HP && L >= 2; }

#ifdef SR_STAMPS
extern "C" int loc_debug_sr_occupancy(int lds_bytes) {
    int nb = -1;
    LOC_ENSURE_LDS(stack_rows_eval_kernel, (size_t)lds_bytes);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stack_rows_eval_kernel, 512, (size_t)lds_bytes) != hipSuccess) return -2;
    return nb;
}
#endif

// tile_rows: 32 or 16 = that form; 0 = by row count
int sr_eval_launch(const float* a1, const float* rd_partial, int rd_G, int64_t rd_MH, const float* rd_cvec8, const float* rd_b1,
                   const float* Wh, const float* bh, const float* wa, const float* ba, const float* wb, const float* bb, int L,
                   int n_b, const int32_t* rows, const float* Y, float* yhat, float* dist, int tile_rows, void* stream) {
    if (tile_rows == 16 || (tile_rows == 0 && sr_takes_16_row_tiles(n_b))) {
        // 16-row tiles, four-step weight chunks: 71 KB of LDS, two workgroups per compute unit
        constexpr size_t lds16 = (2 * SR16_ROWS * SR_PITCH + 8 * SR16_ROWS * 2 + 8 * 2 * (4 * 4 * 32 + 64)) * sizeof(float);
        LOC_ENSURE_LDS(stack_rows16_eval_kernel<4>, lds16);
        hipLaunchKernelGGL(stack_rows16_eval_kernel<4>, dim3((n_b + SR16_ROWS - 1) / SR16_ROWS), dim3(512), lds16, (hipStream_t)stream,
                           a1, rd_partial, rd_G, rd_MH, rd_cvec8, rd_b1, Wh, bh, wa, ba, wb, bb, L, n_b, rows, Y, yhat, dist);
        LOC_CHECK_LAUNCH();
        return 0;
    }
    constexpr size_t lds = (2 * SR_ROWS * SR_PITCH + 8 * SR_ROWS * 2 + 8 * 2 * (32 * 32 + 32)) * sizeof(float);
    LOC_ENSURE_LDS(stack_rows_eval_kernel, lds);
    hipLaunchKernelGGL(stack_rows_eval_kernel, dim3((n_b + SR_ROWS - 1) / SR_ROWS), dim3(512), lds, (hipStream_t)stream, a1,
                       rd_partial, rd_G, rd_MH, rd_cvec8, rd_b1, Wh, bh, wa, ba, wb, bb, L, n_b, rows, Y, yhat, dist);
    LOC_CHECK_LAUNCH();
    return 0;
}
