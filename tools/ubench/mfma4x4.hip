// Microbenchmark / semantics check of v_mfma_f32_4x4x1_16B_f32 on gfx950 (for conv layers with few output channels):
//   (1) operand / result layout and the A-block broadcast (cbsz = 4, abid = k): D_b[i][j] += A_abid[i] * B_b[j] for all 16 blocks;
//   (2) exactness: a chain of K such instructions equals fmaf(b[k], a[k], acc) with k ascending, bit for bit;
//   (3) issue rate of independent / dependent chains against v_mfma_f32_16x16x4_f32.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o mfma4x4 mfma4x4.hip && ./mfma4x4
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave: 64 "pixels" (B operand, lane = block*4 + j), 4 "channels" (A operand, broadcast from block abid), K = 32 steps.
// wpack: lane (blk, i) holds w[k = blk + 16*half][i] for half = 0 (reg 0), 1 (reg 1)
__global__ void k_sem(const float* __restrict__ x /*[64][32]*/, const float* __restrict__ w /*[32][4]*/, const float* __restrict__ bias /*[4]*/,
                      float* __restrict__ out /*[64][4]*/) {
    const int lane = threadIdx.x;
    const int blk = lane >> 2, i = lane & 3;
    float wreg[2] = {w[(blk)*4 + i], w[(blk + 16) * 4 + i]};
    f32x4 acc = {bias[0], bias[1], bias[2], bias[3]};
#define STEP(k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[(k) >> 4], x[lane * 32 + (k)], acc, 4, (k) & 15, 0);
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
    STEP(16) STEP(17) STEP(18) STEP(19) STEP(20) STEP(21) STEP(22) STEP(23) STEP(24) STEP(25) STEP(26) STEP(27) STEP(28) STEP(29) STEP(30) STEP(31)
#undef STEP
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = acc[r];
}

template <int AB>
__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, AB, 0); }
#define REP16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float w0) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    float x = 1.0f + tid * 1e-9f, res = 0.f;
    if (MODE == 0) {          // 4x4x1, 3 independent chains (the conv1 pattern: one B operand, three channel groups)
        f32x4 a0 = {0, 1, 2, 3}, a1 = a0, a2 = a0;
        for (int it = 0; it < iters; it++) {
#define F3(u) a0 = mm<u>(w0, x, a0); a1 = mm<(u + 1) & 15>(w0, x, a1); a2 = mm<(u + 2) & 15>(w0, x, a2);
            REP16(F3)
#undef F3
        }
        res = a0[0] + a1[1] + a2[2];
    } else if (MODE == 1) {   // 4x4x1, 6 independent chains (two pixel groups x three channel groups)
        f32x4 a[6];
        for (int j = 0; j < 6; j++) a[j] = f32x4{(float)j, 1, 2, 3};
        for (int it = 0; it < iters; it++) {
#define F6(u) a[0] = mm<u>(w0, x, a[0]); a[1] = mm<(u + 1) & 15>(w0, x, a[1]); a[2] = mm<(u + 2) & 15>(w0, x, a[2]); \
              a[3] = mm<(u + 3) & 15>(w0, x, a[3]); a[4] = mm<(u + 4) & 15>(w0, x, a[4]); a[5] = mm<(u + 5) & 15>(w0, x, a[5]);
            F6(0) F6(1) F6(2) F6(3) F6(4) F6(5) F6(6) F6(7)
#undef F6
        }
        for (int j = 0; j < 6; j++) res += a[j][0];
    } else if (MODE == 2) {   // 16x16x4, 2 independent chains (today's conv1)
        f32x4 a0 = {0, 1, 2, 3}, a1 = a0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w0, a1, 0, 0, 0);
            }
        }
        res = a0[0] + a1[1];
    } else {                  // 4x4x1, ONE dependent chain
        f32x4 a0 = {0, 1, 2, 3};
        for (int it = 0; it < iters; it++) {
#define F1(u) a0 = mm<u>(w0, x, a0);
            REP16(F1)
#undef F1
        }
        res = a0[0];
    }
    out[tid] = res;
}

int main() {
    float hx[64 * 32], hw[32 * 4], hb[4], ho[64 * 4];
    srand(1);
    for (float& v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (float& v : hw) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (float& v : hb) v = (float)rand() / RAND_MAX;
    float *dx, *dw, *db, *dout;
    hipMalloc(&dx, sizeof hx); hipMalloc(&dw, sizeof hw); hipMalloc(&db, sizeof hb); hipMalloc(&dout, 1 << 24);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(dx, dw, db, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < 64; p++)
        for (int c = 0; c < 4; c++) {
            float acc = hb[c];
            for (int k = 0; k < 32; k++) acc = fmaf(hx[p * 32 + k], hw[k * 4 + c], acc);
            if (memcmp(&acc, &ho[p * 4 + c], 4) != 0) { if (bad < 5) printf("mismatch pixel %d channel %d: %.9g vs %.9g\n", p, c, ho[p * 4 + c], acc); bad++; }
        }
    printf("semantics: D[channel r][pixel lane] with A broadcast from block abid, k-ordered fmaf chain: %d mismatches of 256\n", bad);
    const int iters = 2000, blocks = 256 * 8;
    const char* names[4] = {"4x4x1 x3 chains", "4x4x1 x6 chains", "16x16x4 x2 chains", "4x4x1 x1 chain"};
    const double flop_per_it[4] = {48.0 * 512, 48.0 * 512, 16.0 * 2048, 16.0 * 512};
    for (int m = 0; m < 4; m++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) k_rate<0><<<blocks, 256>>>(dout, iters, 0.5f);
            else if (m == 1) k_rate<1><<<blocks, 256>>>(dout, iters, 0.5f);
            else if (m == 2) k_rate<2><<<blocks, 256>>>(dout, iters, 0.5f);
            else k_rate<3><<<blocks, 256>>>(dout, iters, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop_per_it[m] * iters * blocks * 4 / (ms * 1e-3) / 1e12;     // 4 waves per block
        printf("%-20s %.3f ms  %.1f TFLOP/s\n", names[m], ms, tf);
    }
    return bad != 0;
}
