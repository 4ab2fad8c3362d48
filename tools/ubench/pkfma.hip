// Microbenchmark: f32 FMA throughput of the VALU (v_pk_fma_f32, v_fma_f32) vs the f32 matrix core on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o pkfma pkfma.hip && ./pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float w0, float w1) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    float res = 0.f;
    if (MODE == 0) {          // packed FMA, 8 independent chains per lane, scalar weights
        f32x2 acc[8];
        for (int i = 0; i < 8; i++) acc[i] = f32x2{(float)tid, (float)i};
        f32x2 x = {1.0f + tid * 1e-9f, 1.0f}, w = {w0, w1};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_elementwise_fma(x, w, acc[i]);
        }
        for (int i = 0; i < 8; i++) res += acc[i][0] + acc[i][1];
    } else if (MODE == 1) {   // scalar FMA
        float acc[16];
        for (int i = 0; i < 16; i++) acc[i] = (float)(tid + i);
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(x, w0, acc[i]);
        }
        for (int i = 0; i < 16; i++) res += acc[i];
    } else if (MODE == 2) {   // MFMA 32x32x2 f32, 2 independent chains
        f32x16 a0, a1;
        for (int i = 0; i < 16; i++) { a0[i] = (float)i; a1[i] = (float)(i + tid); }
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w1, a1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; i++) res += a0[i] + a1[i];
    } else if (MODE == 4) {   // MFMA 32x32x2 f32, 1 chain
        f32x16 a0;
        for (int i = 0; i < 16; i++) a0[i] = (float)(i + tid);
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16; u++) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w0, a0, 0, 0, 0);
        }
        for (int i = 0; i < 16; i++) res += a0[i];
    } else if (MODE == 5) {   // MFMA 32x32x2 f32, 4 chains
        f32x16 a[4];
        for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) a[j][i] = (float)(i + tid + j);
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w0, a[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) res += a[j][i];
    } else if (MODE == 6) {   // MFMA 16x16x4 f32, 1 chain
        f32x4 a0 = {(float)tid, 1.f, 2.f, 3.f};
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 32; u++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w0, a0, 0, 0, 0);
        }
        res += a0[0] + a0[1] + a0[2] + a0[3];
    } else if (MODE == 7) {   // MFMA 16x16x4 f32, 2 chains
        f32x4 a0 = {(float)tid, 1.f, 2.f, 3.f}, a1 = {1.f, (float)tid, 2.f, 3.f};
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16; u++) { a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w0, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w1, a1, 0, 0, 0); }
        }
        res += a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
    } else {                  // MFMA 16x16x4 f32, 4 chains
        f32x4 a[4];
        for (int i = 0; i < 4; i++) a[i] = f32x4{(float)i, (float)tid, 1.f, 2.f};
        float x = 1.0f + tid * 1e-9f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w0, a[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) res += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    }
    out[tid] = res;
}

template <int MODE>
double run(float* d, int blocks, int iters, double flop_per_thread_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10, 1.0001f, 0.9999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return flop_per_thread_iter * iters * blocks * 256.0 / (ms * 1e-3) / 1e12;
}

int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    const int blocks = 256 * 8, iters = 4000;
    printf("v_pk_fma_f32   : %.1f TFLOP/s\n", run<0>(d, blocks, iters, 8 * 8 * 2 * 2.0));
    printf("v_fma_f32      : %.1f TFLOP/s\n", run<1>(d, blocks, iters, 4 * 16 * 2.0));
    printf("mfma 32x32x2   : %.1f TFLOP/s\n", run<2>(d, blocks, iters, 16 * 4096.0 / 64.0));
    printf("mfma 16x16x4   : %.1f TFLOP/s\n", run<3>(d, blocks, iters, 32 * 2048.0 / 64.0));
    printf("mfma 32x32x2 x1 chain : %.1f TFLOP/s\n", run<4>(d, blocks, iters, 16 * 4096.0 / 64.0));
    printf("mfma 32x32x2 x4 chains: %.1f TFLOP/s\n", run<5>(d, blocks, iters, 16 * 4096.0 / 64.0));
    printf("mfma 16x16x4 x1 chain : %.1f TFLOP/s\n", run<6>(d, blocks, iters, 32 * 2048.0 / 64.0));
    printf("mfma 16x16x4 x2 chains: %.1f TFLOP/s\n", run<7>(d, blocks, iters, 32 * 2048.0 / 64.0));
    for (int wg = 1; wg <= 2; wg++) {   // occupancy as in the PNet kernel: 1 or 2 waves per SIMD
        printf("  %d wave(s)/SIMD: 32x32x2 x1 %.1f  x2 %.1f   16x16x4 x1 %.1f  x2 %.1f  x4 %.1f\n", wg, run<4>(d, 256 * wg, iters * 4, 16 * 4096.0 / 64.0),
               run<2>(d, 256 * wg, iters * 4, 16 * 4096.0 / 64.0), run<6>(d, 256 * wg, iters * 4, 32 * 2048.0 / 64.0),
               run<7>(d, 256 * wg, iters * 4, 32 * 2048.0 / 64.0), run<3>(d, 256 * wg, iters * 4, 32 * 2048.0 / 64.0));
    }
    return 0;
}
