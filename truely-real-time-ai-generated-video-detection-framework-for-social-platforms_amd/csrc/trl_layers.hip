// trl_layers.hip -- generic NHWC layer kernels for gfx950 (MI355X).
//
// conv_igemm: implicit-GEMM convolution on the f32 matrix cores.
//   GEMM view   M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin, k = (ky*KW+kx)*Cin + c.
//   MFMA        v_mfma_f32_32x32x2_f32: one instruction = two chained fmaf per accumulator
//               (k, k+1), bit-identical to the scalar chain acc = fmaf(x[k], w[k], acc) with k
//               ascending (MI355X_MICROARCH "FP32-input MFMA").  The K loop below walks k in
//               ascending order and the accumulator starts at the bias, so the result equals
//               the oracle's conv2d() bit for bit whatever the tile shape.
//   layout      output channel on the MFMA column (= lane), pixels on the rows: per-channel
//               epilogue constants are lane constants and a store instruction writes 128-byte
//               channel runs of NHWC.
//   staging     A (pixels x k) gathered from NHWC into LDS as [k][m]; B (k x cout) as [k][n];
//               ds_read_b32 operand reads are conflict-free (32 consecutive floats per half-wave).
//               Next chunk's global loads are issued before the MFMAs of the current chunk.
#include "trl_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

namespace {

template <int BM, int BN, int WM, int WN, int BK, bool VEC>
__global__ __launch_bounds__(256) void conv_igemm(ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KG = BK / 4;                                   // float4 groups along k per chunk
    constexpr int ASLOTS = BM * KG, APT = (ASLOTS + 255) / 256;
    constexpr int BSLOTS = BK * (BN / 4), BPT = (BSLOTS + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float As[BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= trl_live_rows(a)) return;     // device-sized batch: nothing of this tile exists (block-uniform)

    // ---- per-thread A row ------------------------------------------------------------------
    const int row = tid % BM;
    const int m = m0 + row;
    const bool mvalid = m < a.M;
    const int mm = mvalid ? m : 0;
    const int ohw = a.OH * a.OW;
    const int nimg = mm / ohw;
    const int rem = mm - nimg * ohw;
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
    const float* xbase = a.x + (size_t)nimg * a.H * a.W * a.ldx + a.xoff;

    float4 areg[APT];
    float4 breg[BPT];
    const int kpad = (a.K + 15) & ~15;   // rows present in the zero-padded weight matrix

    // Per-slot im2col cursor for the vector path: (channel, kx, ky) of the slot's current k, advanced by BK per
    // chunk with a carry loop instead of two integer divisions per load (f32 MFMA shares the FP32 pipe with the
    // VALU on gfx950, so every VALU instruction in the K loop is paid in matrix throughput).
    int cur_c[APT], cur_kx[APT], cur_ky[APT];
#pragma unroll
    for (int i = 0; i < APT; i++) {
        const int slot = tid + i * 256;
        const int k = 4 * (slot / BM);
        const int tap = k / a.Cin;
        cur_c[i] = k - tap * a.Cin;
        cur_ky[i] = tap / a.KW;
        cur_kx[i] = tap - cur_ky[i] * a.KW;
    }
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < APT; i++) {
            const int slot = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < ASLOTS && mvalid) {
                const int k = k0 + 4 * (slot / BM);
                if (VEC) {
                    if (k < a.K) {
                        const int iy = iy0 + cur_ky[i], ix = ix0 + cur_kx[i];
                        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                            v = *reinterpret_cast<const float4*>(xbase + ((size_t)iy * a.W + ix) * a.ldx + cur_c[i]);
                    }
                } else {
                    float t[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int kk = k + j;
                        t[j] = 0.f;
                        if (kk < a.K) {
                            const int tap = kk / a.Cin, c = kk - tap * a.Cin;
                            const int ky = tap / a.KW, kx = tap - ky * a.KW;
                            const int iy = iy0 + ky, ix = ix0 + kx;
                            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                                t[j] = xbase[((size_t)iy * a.W + ix) * a.ldx + c];
                        }
                    }
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            areg[i] = v;
            if (VEC) {   // advance the cursor to the next chunk
                cur_c[i] += BK;
                while (cur_c[i] >= a.Cin) {
                    cur_c[i] -= a.Cin;
                    if (++cur_kx[i] == a.KW) { cur_kx[i] = 0; ++cur_ky[i]; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int slot = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < BSLOTS) {
                const int kk = slot / (BN / 4), n = n0 + 4 * (slot % (BN / 4));
                if (n < a.ldw && k0 + kk < kpad) v = *reinterpret_cast<const float4*>(a.w + (size_t)(k0 + kk) * a.ldw + n);
            }
            breg[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < APT; i++) {
            const int slot = tid + i * 256;
            if (slot < ASLOTS) {
                const int g = slot / BM;
                As[(4 * g + 0) * BM + row] = areg[i].x;
                As[(4 * g + 1) * BM + row] = areg[i].y;
                As[(4 * g + 2) * BM + row] = areg[i].z;
                As[(4 * g + 3) * BM + row] = areg[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int slot = tid + i * 256;
            if (slot < BSLOTS) {
                const int kk = slot / (BN / 4), n4 = slot % (BN / 4);
                *reinterpret_cast<float4*>(&Bs[kk * BN + 4 * n4]) = breg[i];
            }
        }
    };

    // ---- accumulators start at the bias (chain head) ---------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 32 + r;
        const float b = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[tm][tn][i] = b;
    }

    const int nchunks = (a.K + BK - 1) / BK;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ch++) {
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; tm++) av[tm] = As[(2 * s + h) * BM + (wm * TM + tm) * 32 + r];
#pragma unroll
            for (int tn = 0; tn < TN; tn++) bv[tn] = Bs[(2 * s + h) * BN + (wn * TN + tn) * 32 + r];
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 32 + r;
        if (n >= a.Cout) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sf = a.scale ? a.shift[n] : 0.f;
        const float sl = a.act == TRL_ACT_PRELU ? a.slope[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int mr = m0 + (wm * TM + tm) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (mr >= a.M) continue;
                float v = acc[tm][tn][i];
                if (a.scale) v = __builtin_fmaf(v, sc, sf);
                if (a.res) {
                    v = v * a.res_scale;
                    v = v + a.res[(size_t)mr * a.ldres + n];
                }
                if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (a.act == TRL_ACT_PRELU) v = v > 0.f ? v : sl * v;
                a.y[(size_t)mr * a.ldy + a.yoff + n] = v;
            }
        }
    }
}

// ---- split-K(4) variant --------------------------------------------------------------------------------
// Tiny maps with long reductions (OH*OW <= 9, K >= 512: FaceNet's 3x3 / 1x1-spatial tail, the R/O-Net dense
// layers) have few output tiles and a serial chain of K/2 dependent 64-cycle MFMAs each.  The oracle defines
// their accumulation as four chains over consecutive quarters of k combined as (c0+c1)+(c2+c3); here wave w of
// a workgroup owns quarter w of one 32x64 output tile: staging is wave-private (no workgroup barrier in the K
// loop), the four SIMDs of the CU work on the same tile, and the chain per wave is four times shorter.
__global__ __launch_bounds__(256) void conv_splitk4(ConvArgs a) {
    constexpr int BM = 32, BN = 64, BK = 32;
    __shared__ __attribute__((aligned(16))) float As[4][BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[4][BK * BN];     // later: the 4 x (32x64) partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= trl_live_rows(a)) return;     // device-sized batch: nothing of this tile exists (block-uniform)
    const int segK = a.K >> 2, ks = wave * segK, ke = ks + segK;

    const int m = m0 + r;
    const bool mvalid = m < a.M;
    const int mm = mvalid ? m : 0;
    const int ohw = a.OH * a.OW;
    const int nimg = mm / ohw;
    const int rem = mm - nimg * ohw;
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
    const float* xbase = a.x + (size_t)nimg * a.H * a.W * a.ldx + a.xoff;

    // A: 32 rows x 8 float4 groups per chunk = 4 slots per lane (same row, groups h, h+2, h+4, h+6)
    int cur_c[4], cur_kx[4], cur_ky[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int k = ks + 4 * (h + 2 * i);
        const int tap = k / a.Cin;
        cur_c[i] = k - tap * a.Cin;
        cur_ky[i] = tap / a.KW;
        cur_kx[i] = tap - cur_ky[i] * a.KW;
    }
    float4 areg[4], breg[8];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = k0 + 4 * (h + 2 * i);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mvalid && k < ke) {
                const int iy = iy0 + cur_ky[i], ix = ix0 + cur_kx[i];
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v = *reinterpret_cast<const float4*>(xbase + ((size_t)iy * a.W + ix) * a.ldx + cur_c[i]);
            }
            areg[i] = v;
            cur_c[i] += BK;
            while (cur_c[i] >= a.Cin) {
                cur_c[i] -= a.Cin;
                if (++cur_kx[i] == a.KW) { cur_kx[i] = 0; ++cur_ky[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int slot = lane + 64 * i;
            const int kk = slot >> 4, n = n0 + 4 * (slot & 15);
            breg[i] = (k0 + kk < ke && n < a.ldw) ? *reinterpret_cast<const float4*>(a.w + (size_t)(k0 + kk) * a.ldw + n)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float* Aw = As[wave];
    float* Bw = Bs[wave];
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int g = h + 2 * i;
            Aw[(4 * g + 0) * BM + r] = areg[i].x; Aw[(4 * g + 1) * BM + r] = areg[i].y;
            Aw[(4 * g + 2) * BM + r] = areg[i].z; Aw[(4 * g + 3) * BM + r] = areg[i].w;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int slot = lane + 64 * i;
            *reinterpret_cast<float4*>(&Bw[(slot >> 4) * BN + 4 * (slot & 15)]) = breg[i];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int n = n0 + tn * 32 + r;
        const float b = (wave == 0 && a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;   // chain 0 starts at the bias
#pragma unroll
        for (int i = 0; i < 16; i++) acc[tn][i] = b;
    }
    load_chunk(ks);
    for (int k0 = ks; k0 < ke; k0 += BK) {
        store_chunk();                                   // wave-private staging: LDS ops of one wave stay in order
        if (k0 + BK < ke) load_chunk(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            const float av = Aw[(2 * s + h) * BM + r];
            const float b0 = Bw[(2 * s + h) * BN + r], b1 = Bw[(2 * s + h) * BN + 32 + r];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = &Bs[0][0];                              // [wave][tn][reg][lane]
#pragma unroll
    for (int tn = 0; tn < 2; tn++)
#pragma unroll
        for (int i = 0; i < 16; i++) red[((wave * 2 + tn) * 16 + i) * 64 + lane] = acc[tn][i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int e = tid + 256 * j;
        const int tn = e >> 10, reg = (e >> 6) & 15, ln = e & 63;
        const int mr = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
        const int n = n0 + tn * 32 + (ln & 31);
        if (mr >= a.M || n >= a.Cout) continue;
        float v = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
        if (a.scale) v = __builtin_fmaf(v, a.scale[n], a.shift[n]);
        if (a.res) {
            v = v * a.res_scale;
            v = v + a.res[(size_t)mr * a.ldres + n];
        }
        if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (a.act == TRL_ACT_PRELU) v = v > 0.f ? v : a.slope[n] * v;
        a.y[(size_t)mr * a.ldy + a.yoff + n] = v;
    }
}

// conv_splitk4_tap: conv_splitk4 with whole-tap chunks (Cin % BKC == 0 and (K/4) % BKC == 0): the im2col cursor of a
// wave's quarter is a scalar, gather addresses are a per-lane 32-bit offset plus that scalar (see conv_tap).
template <int BKC, bool PAD>
__global__ __launch_bounds__(256) void conv_splitk4_tap(ConvArgs a) {
    constexpr int BM = 32, BN = 64;
    constexpr int AS = BKC / 8, BSL = BKC / 4;                        // float4 slots per lane: A (rows x k-groups), B
    __shared__ __attribute__((aligned(16))) float As[4][BKC * BM];
    __shared__ __attribute__((aligned(16))) float Bs[4][32 * BN];     // staging uses BKC rows; later the 4 x (32x64) partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= trl_live_rows(a)) return;     // device-sized batch: nothing of this tile exists (block-uniform)
    const int segK = a.K >> 2, ks = wave * segK;

    const int m = m0 + r;
    const int mm = m < a.M ? m : 0;
    const int ohw = a.OH * a.OW;
    const int nimg = mm / ohw;
    const int rem = mm - nimg * ohw;
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
    const int aoff = ((nimg * a.H + iy0) * a.W + ix0) * a.ldx + a.xoff + 4 * h;
    int bn = n0 + 4 * (lane & 15);
    bn = bn < a.ldw ? bn : 0;
    const int boff = (lane >> 4) * a.ldw + bn;

    // scalar cursor of the wave's quarter
    const int tap0 = ks / a.Cin;
    int c0 = ks - tap0 * a.Cin, ky = tap0 / a.KW, kx = tap0 - ky * a.KW, k0 = ks;
    f32x4s areg[AS], breg[BSL];
    auto load_chunk = [&]() __attribute__((always_inline)) {
        const int soff = (ky * a.W + kx) * a.ldx + c0;
        bool inside = true;
        if (PAD) inside = (unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W;
#pragma unroll
        for (int i = 0; i < AS; i++) {
            f32x4s v = {0.f, 0.f, 0.f, 0.f};
            if (inside) v = *reinterpret_cast<const f32x4s*>(a.x + (aoff + soff + 8 * i));
            areg[i] = v;
        }
        const float* wrow = a.w + (size_t)k0 * a.ldw;
#pragma unroll
        for (int i = 0; i < BSL; i++) breg[i] = *reinterpret_cast<const f32x4s*>(wrow + (boff + 4 * i * a.ldw));
        k0 += BKC; c0 += BKC;
        if (c0 >= a.Cin) { c0 = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
    };
    float* Aw = As[wave];
    float* Bw = Bs[wave];
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AS; i++) {
            const int g = h + 2 * i;
            Aw[(4 * g + 0) * BM + r] = areg[i][0]; Aw[(4 * g + 1) * BM + r] = areg[i][1];
            Aw[(4 * g + 2) * BM + r] = areg[i][2]; Aw[(4 * g + 3) * BM + r] = areg[i][3];
        }
#pragma unroll
        for (int i = 0; i < BSL; i++) *reinterpret_cast<f32x4s*>(&Bw[((lane >> 4) + 4 * i) * BN + 4 * (lane & 15)]) = breg[i];
    };

    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int n = n0 + tn * 32 + r;
        const float b = (wave == 0 && a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;   // chain 0 starts at the bias
#pragma unroll
        for (int i = 0; i < 16; i++) acc[tn][i] = b;
    }
    const int nchunks = segK / BKC;
    load_chunk();
    for (int ch = 0; ch < nchunks; ch++) {
        store_chunk();                                   // wave-private staging: LDS ops of one wave stay in order
        if (ch + 1 < nchunks) load_chunk();
#pragma unroll
        for (int s = 0; s < BKC / 2; s++) {
            const float av = Aw[(2 * s + h) * BM + r];
            const float b0 = Bw[(2 * s + h) * BN + r], b1 = Bw[(2 * s + h) * BN + 32 + r];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = &Bs[0][0];                              // [wave][tn][reg][lane]
#pragma unroll
    for (int tn = 0; tn < 2; tn++)
#pragma unroll
        for (int i = 0; i < 16; i++) red[((wave * 2 + tn) * 16 + i) * 64 + lane] = acc[tn][i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int e = tid + 256 * j;
        const int tn = e >> 10, reg = (e >> 6) & 15, ln = e & 63;
        const int mr = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
        const int n = n0 + tn * 32 + (ln & 31);
        if (mr >= a.M || n >= a.Cout) continue;
        float v = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
        if (a.scale) v = __builtin_fmaf(v, a.scale[n], a.shift[n]);
        if (a.res) {
            v = v * a.res_scale;
            v = v + a.res[(size_t)mr * a.ldres + n];
        }
        if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (a.act == TRL_ACT_PRELU) v = v > 0.f ? v : a.slope[n] * v;
        a.y[(size_t)mr * a.ldy + a.yoff + n] = v;
    }
}

// conv_tap: the same implicit GEMM for layers whose K chunks never straddle a filter tap (Cin % BK == 0): the
// im2col cursor (ky, kx, c0) of a chunk is then UNIFORM and lives on the scalar unit, a thread's gather address is
// one 32-bit per-thread offset plus that scalar, rows past M re-read row 0 (their results are never stored) and the
// weight matrix is addressed the same way.  The K loop of conv_igemm spends ~160 VALU instructions per chunk on
// per-slot cursors and bounds tests; f32 MFMA and VALU share the FP32 pipe on gfx950, so that was ~30 % of the
// loop.  Here it is ~20.  Same k order, same bias-seeded chain: bit-identical results.
// PAD = the layer has spatial padding (taps falling outside the image load nothing and contribute zeros).
template <int BM, int BN, int WM, int WN, int BK, bool PAD>
__global__ __launch_bounds__(256) void conv_tap(ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KG = BK / 4;                                   // float4 groups along k per chunk
    constexpr int ASLOTS = BM * KG, APT = (ASLOTS + 255) / 256;
    constexpr int BSLOTS = BK * (BN / 4), BPT = (BSLOTS + 255) / 256;
    constexpr int LDA = BK + 1;                                  // A tile [m][k], odd row stride: operand reads (lane = row) and the staging writes are conflict-free
    static_assert(WM * WN == 4 && BK % 4 == 0 && BK % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= trl_live_rows(a)) return;     // device-sized batch: nothing of this tile exists (block-uniform)

    // ---- A slots: slot = tid + 256 i -> (k-group = slot % KG, row = slot / KG).  The KG lanes of a row read the BK contiguous
    // channels one pixel contributes to a chunk, so a wave's load touches 64 / KG pixels' lines instead of 64 (row-per-lane
    // gathers spend the vector L1's request rate, not its bandwidth: round 2).  Element offset of
    // (image, iy0, ix0, channel 4 g) from a.x; rows past M re-read row 0 (never stored).
    const int ohw = a.OH * a.OW;
    int aoff[APT], adst[APT], iy0v[APT], ix0v[APT];
#pragma unroll
    for (int i = 0; i < APT; i++) {
        const int slot = tid + 256 * i;
        const int g = slot % KG, row = slot / KG;
        const int m = m0 + row;
        const int mm = (m < a.M && row < BM) ? m : 0;
        const int nimg = mm / ohw;
        const int rem = mm - nimg * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        iy0v[i] = oy * a.sh - a.ph; ix0v[i] = ox * a.sw - a.pw;
        aoff[i] = ((nimg * a.H + iy0v[i]) * a.W + ix0v[i]) * a.ldx + a.xoff + 4 * g;      // < 2^31: checked by the launcher
        adst[i] = row * LDA + 4 * g;
    }
    // ---- per-thread B slot: (k row kk0, column n) ------------------------------------------------------------
    const int bkk = tid / (BN / 4), bn4 = tid % (BN / 4);
    int bn = n0 + 4 * bn4;
    bn = bn < a.ldw ? bn : 0;                                   // columns past the matrix re-read column 0 (never stored)
    const int boff = bkk * a.ldw + bn;
    constexpr int BKSTEP = 256 / (BN / 4);                       // k-row stride between a thread's B slots

    float4 areg[APT];
    float4 breg[BPT];
    int ky = 0, kx = 0, c0 = 0;                                 // scalar cursor of the NEXT chunk to load
    int k0 = 0;
    auto load_chunk = [&]() {
        const int soff = (ky * a.W + kx) * a.ldx + c0;         // scalar
#pragma unroll
        for (int i = 0; i < APT; i++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = ASLOTS % 256 == 0 || tid + i * 256 < ASLOTS;
            if (PAD) ok = ok && (unsigned)(iy0v[i] + ky) < (unsigned)a.H && (unsigned)(ix0v[i] + kx) < (unsigned)a.W;
            if (ok) v = *reinterpret_cast<const float4*>(a.x + (aoff[i] + soff));
            areg[i] = v;
        }
        const float* wrow = a.w + (size_t)k0 * a.ldw;           // scalar
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BSLOTS % 256 == 0 || tid + i * 256 < BSLOTS) v = *reinterpret_cast<const float4*>(wrow + (boff + BKSTEP * i * a.ldw));
            breg[i] = v;
        }
        // advance: next BK channels of the tap, else next tap
        k0 += BK; c0 += BK;
        if (c0 >= a.Cin) { c0 = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < APT; i++) {
            if (ASLOTS % 256 == 0 || tid + i * 256 < ASLOTS) {
                As[adst[i] + 0] = areg[i].x; As[adst[i] + 1] = areg[i].y;
                As[adst[i] + 2] = areg[i].z; As[adst[i] + 3] = areg[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            if (BSLOTS % 256 == 0 || tid + i * 256 < BSLOTS)
                *reinterpret_cast<float4*>(&Bs[(bkk + BKSTEP * i) * BN + 4 * bn4]) = breg[i];
        }
    };

    // ---- accumulators start at the bias (chain head) ---------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 32 + r;
        const float b = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[tm][tn][i] = b;
    }

    const int nchunks = a.K / BK;
    load_chunk();
    for (int ch = 0; ch < nchunks; ch++) {
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk();
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; tm++) av[tm] = As[((wm * TM + tm) * 32 + r) * LDA + 2 * s + h];
#pragma unroll
            for (int tn = 0; tn < TN; tn++) bv[tn] = Bs[(2 * s + h) * BN + (wn * TN + tn) * 32 + r];
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue (as conv_igemm) --------------------------------------------------------------------------------
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 32 + r;
        if (n >= a.Cout) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sf = a.scale ? a.shift[n] : 0.f;
        const float sl = a.act == TRL_ACT_PRELU ? a.slope[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int mr = m0 + (wm * TM + tm) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (mr >= a.M) continue;
                float v = acc[tm][tn][i];
                if (a.scale) v = __builtin_fmaf(v, sc, sf);
                if (a.res) {
                    v = v * a.res_scale;
                    v = v + a.res[(size_t)mr * a.ldres + n];
                }
                if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (a.act == TRL_ACT_PRELU) v = v > 0.f ? v : sl * v;
                a.y[(size_t)mr * a.ldy + a.yoff + n] = v;
            }
        }
    }
}

// conv_tap48: whole-tap implicit GEMM for Cout = 48 (R-Net conv2, 31 k candidates x 81 pixels per step): a 128 x 48 tile on
// v_mfma_f32_16x16x4_f32 -- three 16-column tiles instead of two 32-column ones, so no MFMA cycle is spent on the 16 padding
// columns a 64-wide tile would carry (25 % of that layer).  Same LDS layout ([k][m], [k][n]) and loader as conv_tap; each of the
// 4 waves owns 32 rows = 2 x 3 accumulator tiles; k = 4s + (lane >> 4) ascending, bias-seeded: the oracle's chain.
typedef float f32x4t __attribute__((ext_vector_type(4)));
template <int BK, bool PAD>
__global__ __launch_bounds__(256) void conv_tap48(ConvArgs a) {
    constexpr int BM = 128, BN = 48;
    constexpr int KG = BK / 4;
    constexpr int ASLOTS = BM * KG, APT = (ASLOTS + 255) / 256;
    constexpr int BSLOTS = BK * (BN / 4), BPT = (BSLOTS + 255) / 256;
    constexpr int GSTEP = 256 / BM;
    static_assert(BK % 4 == 0, "k-steps of 4");
    __shared__ __attribute__((aligned(16))) float As[BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    if (m0 >= trl_live_rows(a)) return;     // device-sized batch: nothing of this tile exists (block-uniform)

    const int row = tid % BM, g0 = tid / BM;
    const int m = m0 + row;
    const int mm = m < a.M ? m : 0;
    const int ohw = a.OH * a.OW;
    const int nimg = mm / ohw;
    const int rem = mm - nimg * ohw;
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
    const int aoff = ((nimg * a.H + iy0) * a.W + ix0) * a.ldx + a.xoff + 4 * g0;
    // B slots: slot = tid + 256 i -> (k row, column group of 4); 12 groups per row
    int boff[BPT], bdst[BPT];
#pragma unroll
    for (int i = 0; i < BPT; i++) {
        const int slot = tid + 256 * i, kk = slot / (BN / 4), n4 = slot - kk * (BN / 4);
        boff[i] = kk * a.ldw + 4 * n4;
        bdst[i] = kk * BN + 4 * n4;
    }

    float4 areg[APT];
    float4 breg[BPT];
    int ky = 0, kx = 0, c0 = 0, k0 = 0;
    auto load_chunk = [&]() {
        const int soff = (ky * a.W + kx) * a.ldx + c0;
        bool inside = true;
        if (PAD) inside = (unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W;
#pragma unroll
        for (int i = 0; i < APT; i++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((ASLOTS % 256 == 0 || tid + i * 256 < ASLOTS) && inside)
                v = *reinterpret_cast<const float4*>(a.x + (aoff + soff + 4 * GSTEP * i));
            areg[i] = v;
        }
        const float* wrow = a.w + (size_t)k0 * a.ldw;
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BSLOTS % 256 == 0 || tid + i * 256 < BSLOTS) v = *reinterpret_cast<const float4*>(wrow + boff[i]);
            breg[i] = v;
        }
        k0 += BK; c0 += BK;
        if (c0 >= a.Cin) { c0 = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < APT; i++) {
            if (ASLOTS % 256 == 0 || tid + i * 256 < ASLOTS) {
                const int g = g0 + GSTEP * i;
                As[(4 * g + 0) * BM + row] = areg[i].x;
                As[(4 * g + 1) * BM + row] = areg[i].y;
                As[(4 * g + 2) * BM + row] = areg[i].z;
                As[(4 * g + 3) * BM + row] = areg[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            if (BSLOTS % 256 == 0 || tid + i * 256 < BSLOTS) *reinterpret_cast<float4*>(&Bs[bdst[i]]) = breg[i];
        }
    };

    f32x4t acc[2][3];
#pragma unroll
    for (int tn = 0; tn < 3; tn++) {
        const float b = a.bias != nullptr ? a.bias[tn * 16 + l15] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) acc[tm][tn] = f32x4t{b, b, b, b};
    }

    const int nchunks = a.K / BK;
    load_chunk();
    for (int ch = 0; ch < nchunks; ch++) {
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk();
#pragma unroll
        for (int s = 0; s < BK / 4; s++) {
            float av[2], bv[3];
#pragma unroll
            for (int tm = 0; tm < 2; tm++) av[tm] = As[(4 * s + kq) * BM + wave * 32 + tm * 16 + l15];
#pragma unroll
            for (int tn = 0; tn < 3; tn++) bv[tn] = Bs[(4 * s + kq) * BN + tn * 16 + l15];
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 3; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int tn = 0; tn < 3; tn++) {
        const int n = tn * 16 + l15;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sf = a.scale ? a.shift[n] : 0.f;
        const float sl = a.act == TRL_ACT_PRELU ? a.slope[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int mr = m0 + wave * 32 + tm * 16 + kq * 4 + q;
                if (mr >= a.M) continue;
                float v = acc[tm][tn][q];
                if (a.scale) v = __builtin_fmaf(v, sc, sf);
                if (a.res) {
                    v = v * a.res_scale;
                    v = v + a.res[(size_t)mr * a.ldres + n];
                }
                if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (a.act == TRL_ACT_PRELU) v = v > 0.f ? v : sl * v;
                a.y[(size_t)mr * a.ldy + a.yoff + n] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK>
int launch_tap(const ConvArgs& a, dim3 grid, hipStream_t s) {
    if (a.ph || a.pw) conv_tap<BM, BN, WM, WN, BK, true><<<grid, 256, 0, s>>>(a);
    else conv_tap<BM, BN, WM, WN, BK, false><<<grid, 256, 0, s>>>(a);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

template <int BM, int BN, int WM, int WN, int BK>
int launch_cfg(const ConvArgs& a, bool vec, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    // whole-tap chunks (conv_tap) whenever the channel count allows it and 32-bit element offsets suffice
    static const bool tap_off = trl_tune_set("TRL_NO_TAP");
    const long long x_elems = (long long)a.N * a.H * a.W * a.ldx + a.xoff;
    if (vec && !tap_off && a.K == a.KH * a.KW * a.Cin && x_elems < 0x7fffffffll && (long long)a.K * a.ldw < 0x7fffffffll) {
        if (BK >= 64 && a.Cin % 64 == 0) return launch_tap<BM, BN, WM, WN, 64>(a, grid, s);
        static const bool bk16 = trl_tune_set("TRL_CONV_BK16");   // tuning aid: 16-channel chunks (half the LDS tile) where 32 divide Cin
        if (a.Cin % 32 == 0 && !bk16) return launch_tap<BM, BN, WM, WN, 32>(a, grid, s);
        if (BM == 128 && BN == 64 && a.Cin % 28 == 0) return launch_tap<BM, BN, WM, WN, 28>(a, grid, s);
        if (a.Cin % 16 == 0) return launch_tap<BM, BN, WM, WN, 16>(a, grid, s);
    }
    if (vec) conv_igemm<BM, BN, WM, WN, BK, true><<<grid, 256, 0, s>>>(a);
    else conv_igemm<BM, BN, WM, WN, 16, false><<<grid, 256, 0, s>>>(a);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

// ---- max pool -------------------------------------------------------------------------------
template <int V>   // V channels per thread (4 when every stride / offset is a multiple of 4 floats)
__global__ void maxpool_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx, int xoff,
                               int k, int st, float* __restrict__ y, int ldy, int yoff, int OH, int OW,
                               const int32_t* __restrict__ n_dev, int n_base) {
    const int CV = C / V;
    if (n_dev) {   // device-sized batch: the first clamp(*n_dev - n_base, 0, N) items exist
        int t = *n_dev - n_base;
        N = t < 0 ? 0 : (t > N ? N : t);
    }
    const size_t total = (size_t)N * OH * OW * CV;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CV) * V;
        size_t pix = idx / CV;
        const int ox = (int)(pix % OW); pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        float best[V];
#pragma unroll
        for (int j = 0; j < V; j++) best[j] = -INFINITY;
        for (int ky = 0; ky < k; ky++) {
            const int iy = oy * st + ky;
            if (iy >= H) break;
            for (int kx = 0; kx < k; kx++) {
                const int ix = ox * st + kx;
                if (ix >= W) break;
                const float* p = x + (((size_t)n * H + iy) * W + ix) * ldx + xoff + c;
                if (V == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    best[0] = v.x > best[0] ? v.x : best[0]; best[1] = v.y > best[1] ? v.y : best[1];
                    best[2 % V] = v.z > best[2 % V] ? v.z : best[2 % V]; best[3 % V] = v.w > best[3 % V] ? v.w : best[3 % V];
                } else {
                    best[0] = p[0] > best[0] ? p[0] : best[0];
                }
            }
        }
        float* q = y + (((size_t)n * OH + oy) * OW + ox) * ldy + yoff + c;
        if (V == 4) *reinterpret_cast<float4*>(q) = make_float4(best[0], best[1 % V], best[2 % V], best[3 % V]);
        else q[0] = best[0];
    }
}

// AdaptiveAvgPool2d(1): sequential row-major sum / count (oracle order)
__global__ void gap_kernel(const float* __restrict__ x, int N, int HW, int C, float* __restrict__ y) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int n = idx / C, c = idx - n * C;
    float s = 0.f;
    for (int p = 0; p < HW; p++) s = s + x[((size_t)n * HW + p) * C + c];
    y[idx] = s / (float)HW;
}

// F.normalize(p=2, dim=1, eps=1e-12) on [n][512]; one wave per row.
__global__ __launch_bounds__(64) void l2norm512_kernel(const float* __restrict__ x, const uint8_t* __restrict__ valid,
                                                       int n, float* __restrict__ y) {
    const int row = blockIdx.x, lane = threadIdx.x;
    if (row >= n) return;
    const float* v = x + (size_t)row * 512;
    float* o = y + (size_t)row * 512;
    if (valid && !valid[row]) {
        for (int i = 0; i < 8; i++) o[lane + 64 * i] = 0.f;
        return;
    }
    float nrm = sqrtf(trl_wave_dot512(v, v, lane));
    if (nrm < 1e-12f) nrm = 1e-12f;
    for (int i = 0; i < 8; i++) o[lane + 64 * i] = v[lane + 64 * i] / nrm;
}

// Carry of the drift state machine between consecutive windows of one clip (trl_drift_update): what model.py's loop variables
// `previous_embedding`, `consecutive_count` and `ai_detected_frames` hold between two sampled frames (model.py:60-75).
struct DriftState {
    float prev[512];                 // embedding of the last frame that had one (model.py:75)
    int32_t has_prev, run, hits, pad;
};
static_assert(sizeof(DriftState) == TRL_DRIFT_STATE_BYTES, "drift state layout");

// server/model.py:60-61 -- cosine similarity of every embedded frame with the previous embedded frame.
// One wave per sampled frame; the three 512-long dots use the oracle's fixed lane order (trl_wave_dot512).
// `st` (optional): the window continues a clip -- a frame with no embedded predecessor inside the window compares with st->prev.
__global__ __launch_bounds__(256) void drift_sims_kernel(const float* __restrict__ emb, const uint8_t* __restrict__ valid, int n,
                                                         float* __restrict__ sims, const DriftState* __restrict__ st) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float sim = 2.0f;                       // 2.0 = "no comparison happened"
    if (valid[i]) {
        int prev = i - 1;
        while (prev >= 0 && !valid[prev]) prev--;   // frames without a face neither compare nor replace `previous` (model.py:48-75)
        const float* pv = prev >= 0 ? emb + (size_t)prev * 512 : ((st && st->has_prev) ? st->prev : nullptr);
        if (pv) {
            const float* cur = emb + (size_t)i * 512;
            const float d = trl_wave_dot512(cur, pv, lane);
            const float na = sqrtf(trl_wave_dot512(cur, cur, lane)), nb = sqrtf(trl_wave_dot512(pv, pv, lane));
            sim = d / (na * nb);
        }
    }
    if (lane == 0) sims[i] = sim;
}

// server/model.py:62-66,70,86-95 -- the run-length state machine is a scan over the similarities: staged in
// LDS in chunks, walked by one lane, then the score in double like the reference's Python floats.
// `st` (optional): counters start from the carried state, which then takes the window's final counters and last embedding.
__global__ __launch_bounds__(256) void drift_scan_kernel(const float* __restrict__ sims, int n, long long frame_count, int fps,
                                                         uint8_t* __restrict__ flags, int32_t* __restrict__ result,
                                                         DriftState* __restrict__ st, const float* __restrict__ emb,
                                                         const uint8_t* __restrict__ valid) {
    __shared__ float sh[4096];
    __shared__ uint8_t fl[4096];
    __shared__ int last_valid;
    const float thr_sim = 0.99f;   // model.py:16
    const int thr_frames = 15;     // model.py:17
    int run = st ? st->run : 0, hits = st ? st->hits : 0;
    if (threadIdx.x == 0) last_valid = -1;
    for (int base = 0; base < n; base += 4096) {
        const int m = (n - base) < 4096 ? (n - base) : 4096;
        for (int t = threadIdx.x; t < m; t += blockDim.x) sh[t] = sims[base + t];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 0; t < m; t++) {
                const float sim = sh[t];
                uint8_t f = 0;
                if (sim <= 1.5f) {          // a comparison happened
                    run = (sim < thr_sim) ? run + 1 : 0;
                    if (run > thr_frames) { hits++; f = 1; }
                }
                fl[t] = f;
            }
        }
        __syncthreads();
        if (flags) for (int t = threadIdx.x; t < m; t += blockDim.x) flags[base + t] = fl[t];
        if (st) for (int t = threadIdx.x; t < m; t += blockDim.x) if (valid[base + t]) atomicMax(&last_valid, base + t);
        __syncthreads();
    }
    if (st) {                               // (the similarities were computed by the previous kernel: st->prev is free to change)
        __syncthreads();
        const int lv = last_valid;
        if (lv >= 0) for (int t = threadIdx.x; t < 512; t += blockDim.x) st->prev[t] = emb[(size_t)lv * 512 + t];
    }
    if (threadIdx.x == 0) {
        if (st) { st->run = run; st->hits = hits; if (last_valid >= 0) st->has_prev = 1; }
        int score = 0;
        long long total = 0;
        if (frame_count > 0 && fps > 0) {
            int step = (int)((double)fps / 7.0);
            if (step < 1) step = 1;
            total = (frame_count + step - 1) / step;
            if (total > 0) {
                double pct = ((double)hits / (double)total) * 100.0;
                double conf = pct * ((double)run / (double)thr_frames);
                if (conf > 100.0) conf = 100.0;
                const double w = (frame_count > (long long)fps * 30) ? 0.5 : 0.3;
                double ws = pct + conf * w;
                if (ws > 100.0) ws = 100.0;
                score = (int)ws;
                if (score < 0) score = 0;
                if (score > 100) score = 100;
            }
        }
        result[0] = score; result[1] = run; result[2] = hits; result[3] = (int32_t)total;
    }
}

}  // namespace

int trl_launch_conv(const ConvArgs& a, hipStream_t s) {
    if (a.M <= 0) return TRL_OK;
    if (trl_fn_eligible(a)) return trl_launch_fn_group(&a, 1, s);   // small maps (FaceNet's 7x7 / 3x3 / 1x1 stages): trl_fnconv.hip
    const bool vec = (a.Cin % 4 == 0) && (a.ldx % 4 == 0) && (a.xoff % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
    // The oracle's four-chain rule for tiny maps with long reductions (oracle/trl_oracle.c conv2d)
    if (a.OH * a.OW <= 9 && a.K >= 512 && (a.K & 15) == 0) {
        if (!vec) { trl_set_error("split-K layer needs Cin %% 4 == 0 and 16-byte aligned input"); return TRL_ERR_INVALID; }
        dim3 grid((a.M + 31) / 32, (a.Cout + 63) / 64);
        static const bool tap_off = trl_tune_set("TRL_NO_TAP");
        const int segK = a.K >> 2;
        const bool small = (long long)a.N * a.H * a.W * a.ldx + a.xoff < 0x7fffffffll && (long long)a.K * a.ldw < 0x7fffffffll;
        const bool pad = a.ph || a.pw;
        if (!tap_off && small && a.K == a.KH * a.KW * a.Cin && a.Cin % 32 == 0 && segK % 32 == 0) {
            if (pad) conv_splitk4_tap<32, true><<<grid, 256, 0, s>>>(a); else conv_splitk4_tap<32, false><<<grid, 256, 0, s>>>(a);
        } else if (!tap_off && small && a.K == a.KH * a.KW * a.Cin && a.Cin % 16 == 0 && segK % 16 == 0) {
            if (pad) conv_splitk4_tap<16, true><<<grid, 256, 0, s>>>(a); else conv_splitk4_tap<16, false><<<grid, 256, 0, s>>>(a);
        } else {
            conv_splitk4<<<grid, 256, 0, s>>>(a);
        }
        TRL_LAUNCH_CHECK();
        return TRL_OK;
    }
    // Deep K chunks (BK = 64) when K is long: a chunk's MFMAs (BK/2 x 64 cycles per wave tile) must cover
    // the global-load round trip of the next chunk, the only latency hiding a lone workgroup per CU has.
    const bool deep = a.K >= 192;
    {   // Cout == 48 with whole-tap chunks: the 128 x 48 tile (no padded MFMA columns)
        static const bool tap_off = trl_tune_set("TRL_NO_TAP");
        const bool small = (long long)a.N * a.H * a.W * a.ldx + a.xoff < 0x7fffffffll && (long long)a.K * a.ldw < 0x7fffffffll;
        if (vec && !tap_off && small && a.Cout == 48 && a.ldw >= 48 && a.M >= 16384 && a.K == a.KH * a.KW * a.Cin && (a.Cin % 28 == 0 || a.Cin % 32 == 0)) {
            dim3 grid((a.M + 127) / 128, 1);
            const bool pad = a.ph || a.pw;
            if (a.Cin % 32 == 0) { if (pad) conv_tap48<32, true><<<grid, 256, 0, s>>>(a); else conv_tap48<32, false><<<grid, 256, 0, s>>>(a); }
            else { if (pad) conv_tap48<28, true><<<grid, 256, 0, s>>>(a); else conv_tap48<28, false><<<grid, 256, 0, s>>>(a); }
            TRL_LAUNCH_CHECK();
            return TRL_OK;
        }
    }
    if (a.Cout <= 32) return deep ? launch_cfg<128, 32, 4, 1, 64>(a, vec, s) : launch_cfg<128, 32, 4, 1, 16>(a, vec, s);
    static const int bigm = trl_tune_int("TRL_CONV_BIGM", 128);   // tuning aid: row tile of the large-M layers
    if (a.M >= 16384 && bigm == 128) return deep ? launch_cfg<128, 64, 2, 2, 32>(a, vec, s) : launch_cfg<128, 64, 2, 2, 16>(a, vec, s);
    if (a.M >= 1024) return deep ? launch_cfg<64, 64, 2, 2, 64>(a, vec, s) : launch_cfg<64, 64, 2, 2, 16>(a, vec, s);
    return deep ? launch_cfg<32, 128, 1, 4, 64>(a, vec, s) : launch_cfg<32, 128, 1, 4, 16>(a, vec, s);
}

int trl_launch_maxpool(const float* x, int N, int H, int W, int C, int ldx, int xoff, int k, int st, int ceil_mode,
                       float* y, int ldy, int yoff, int OH, int OW, hipStream_t s, const int32_t* n_dev, int n_base) {
    (void)ceil_mode;
    const bool v4 = (C % 4 == 0) && (ldx % 4 == 0) && (xoff % 4 == 0) && (ldy % 4 == 0) && (yoff % 4 == 0) &&
                    (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0);
    const size_t total = (size_t)N * OH * OW * (v4 ? C / 4 : C);
    if (total == 0) return TRL_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    if (v4) maxpool_kernel<4><<<(unsigned)blocks, 256, 0, s>>>(x, N, H, W, C, ldx, xoff, k, st, y, ldy, yoff, OH, OW, n_dev, n_base);
    else maxpool_kernel<1><<<(unsigned)blocks, 256, 0, s>>>(x, N, H, W, C, ldx, xoff, k, st, y, ldy, yoff, OH, OW, n_dev, n_base);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_launch_gap(const float* x, int N, int HW, int C, float* y, hipStream_t s) {
    if (N * C == 0) return TRL_OK;
    gap_kernel<<<(N * C + 255) / 256, 256, 0, s>>>(x, N, HW, C, y);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_launch_l2norm512(const float* x, const uint8_t* valid, int n, float* y, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    l2norm512_kernel<<<n, 64, 0, s>>>(x, valid, n, y);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_launch_drift(const float* emb, const uint8_t* valid, int n, long long frame_count, int fps, float* sims,
                     uint8_t* flags, int32_t* result, hipStream_t s, void* state) {
    DriftState* st = (DriftState*)state;
    if (n > 0) {
        drift_sims_kernel<<<(n + 3) / 4, 256, 0, s>>>(emb, valid, n, sims, st);
        TRL_LAUNCH_CHECK();
    }
    drift_scan_kernel<<<1, 256, 0, s>>>(sims, n, frame_count, fps, flags, result, st, emb, valid);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
