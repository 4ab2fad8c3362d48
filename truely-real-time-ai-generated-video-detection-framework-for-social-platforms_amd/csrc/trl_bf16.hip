// trl_bf16.hip -- optional reduced-precision embedder: trl_config.embed_precision = 1 (bf16, BASELINE configs[2] "bf16 MFMA")
// or 2 (fp16, BASELINE configs[4] "fp16"); every kernel is a template on the 16-bit format P.
//
// InceptionResnetV1 with bf16 activations + bf16 weights on v_mfma_f32_32x32x16_bf16 (f32 accumulate), folded BN /
// residual / ReLU epilogue in f32, one rounding to bf16 per stored activation.  NOT the parity path: the default
// (embed_precision = 0) stays the bit-exact f32 pipeline; this mode trades ~1e-2 relative embedding error (measured
// and bounded in tests/test_gpu_api.py) for a ~3x faster embedder.  The detector is never run in reduced precision:
// boxes, crops and the valid mask are identical to the f32 path.
//
// conv_bf16: whole-tap implicit GEMM (see conv_tap in trl_layers.hip): K chunks of BK channels of one filter tap,
// scalar im2col cursor, A = [pixel][k] and B = [cout][k] staged k-contiguous in LDS so that an MFMA operand is one
// ds_read_b128 (8 bf16); rows padded by 16 bytes: the 32 rows x 2 k-halves of an operand read hit distinct banks.
#include "trl_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {            // round to nearest even (inputs are finite)
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// P = 1: bf16 (8 exponent bits: no range concerns), P = 2: IEEE fp16 (3 more mantissa bits; FaceNet's activations are O(1..100),
// far inside +-65504, and a value that did overflow would surface as inf -> NaN embedding, never as a silent wrong number)
template <int P> __device__ __forceinline__ float lp2f(uint16_t h) {
    if (P == 1) return bf2f(h);
    return (float)__builtin_bit_cast(_Float16, h);
}
template <int P> __device__ __forceinline__ uint16_t f2lp(float f) {
    if (P == 1) return f2bf(f);
    return __builtin_bit_cast(uint16_t, (_Float16)f);       // v_cvt_f16_f32: round to nearest even
}

template <int BM, int BN, int BK, bool PAD, int P>
__global__ __launch_bounds__(256) void conv_bf16(ConvArgs a) {
    constexpr int LDS_K = BK + 8;                                // bf16 elements per staged row (16-byte pad)
    constexpr int KG = BK / 8;                                   // 16-byte groups along k per chunk
    constexpr int ASLOTS = BM * KG, APT = (ASLOTS + 255) / 256;
    constexpr int BSLOTS = BN * KG, BPT = (BSLOTS + 255) / 256;
    constexpr int TM = BM / 64, TN = BN / 64;                    // 2 x 2 waves, 32x32 MFMA tiles per wave
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BK % 16 == 0 && 256 % KG == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) uint16_t As[BM * LDS_K];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BN * LDS_K];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const uint16_t* __restrict__ xg = reinterpret_cast<const uint16_t*>(a.x);
    const uint16_t* __restrict__ rg = reinterpret_cast<const uint16_t*>(a.res);
    uint16_t* __restrict__ yg = reinterpret_cast<uint16_t*>(a.y);

    // per-thread A slots: k-group g = tid % KG, rows tid / KG + (256/KG) * i
    const int ag = tid % KG;
    int aoff[APT];
    bool ain[APT];
    int iy0[APT], ix0[APT];
#pragma unroll
    for (int i = 0; i < APT; i++) {
        const int row = tid / KG + (256 / KG) * i;
        const int m = m0 + row;
        const int mm = (m < a.M && row < BM) ? m : 0;
        const int ohw = a.OH * a.OW;
        const int nimg = mm / ohw;
        const int rem = mm - nimg * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        iy0[i] = oy * a.sh - a.ph; ix0[i] = ox * a.sw - a.pw;
        aoff[i] = ((nimg * a.H + iy0[i]) * a.W + ix0[i]) * a.ldx + a.xoff + 8 * ag;
        ain[i] = row < BM;
    }
    const int bg = tid % KG;
    int boff[BPT];
#pragma unroll
    for (int i = 0; i < BPT; i++) boff[i] = (n0 + tid / KG + (256 / KG) * i) * a.ldwt + 8 * bg;   // wt rows are padded to 64 couts

    u32x4 areg[APT], breg[BPT];
    int ky = 0, kx = 0, c0 = 0, k0 = 0;
    auto load_chunk = [&]() __attribute__((always_inline)) {
        const int soff = (ky * a.W + kx) * a.ldx + c0;
#pragma unroll
        for (int i = 0; i < APT; i++) {
            u32x4 v = {0u, 0u, 0u, 0u};
            bool ok = ain[i];
            if (PAD) ok = ok && (unsigned)(iy0[i] + ky) < (unsigned)a.H && (unsigned)(ix0[i] + kx) < (unsigned)a.W;
            if (ok) v = *reinterpret_cast<const u32x4*>(xg + (aoff[i] + soff));
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (BSLOTS % 256 == 0 || tid + 256 * i < BSLOTS) v = *reinterpret_cast<const u32x4*>(a.wt + (boff[i] + k0));
            breg[i] = v;
        }
        k0 += BK; c0 += BK;
        if (c0 >= a.Cin) { c0 = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < APT; i++) {
            const int row = tid / KG + (256 / KG) * i;
            if (row < BM) *reinterpret_cast<u32x4*>(&As[row * LDS_K + 8 * ag]) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            const int row = tid / KG + (256 / KG) * i;
            if (row < BN) *reinterpret_cast<u32x4*>(&Bs[row * LDS_K + 8 * bg]) = breg[i];
        }
    };

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
        for (int kk = 0; kk < BK / 16; kk++) {
            u32x4 av[TM], bv[TN];
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
                av[tm] = *reinterpret_cast<const u32x4*>(&As[((wm * TM + tm) * 32 + r) * LDS_K + 16 * kk + 8 * h]);
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
                bv[tn] = *reinterpret_cast<const u32x4*>(&Bs[((wn * TN + tn) * 32 + r) * LDS_K + 16 * kk + 8 * h]);
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++) {
                    if (P == 1) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[tm]), __builtin_bit_cast(bf16x8, bv[tn]), acc[tm][tn], 0, 0, 0);
                    else acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[tm]), __builtin_bit_cast(f16x8, bv[tn]), acc[tm][tn], 0, 0, 0);
                }
        }
        __syncthreads();
    }

#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 32 + r;
        if (n >= a.Cout) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sf = a.scale ? a.shift[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int mr = m0 + (wm * TM + tm) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (mr >= a.M) continue;
                float v = acc[tm][tn][i];
                if (a.scale) v = __builtin_fmaf(v, sc, sf);
                if (a.res) v = v * a.res_scale + lp2f<P>(rg[(size_t)mr * a.ldres + n]);
                if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
                yg[(size_t)mr * a.ldy + a.yoff + n] = f2lp<P>(v);
            }
        }
    }
}

// f32 NHWC -> 16-bit (same shape, dense)
template <int P>
__global__ void k_to_bf16(const float* __restrict__ x, size_t n, uint16_t* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = f2lp<P>(x[i]);
}

// [K][ld] f32 (HWIO rows) -> bf16 transposed [Cp][Kp], zero padded
template <int P>
__global__ void k_weight_bf16_t(const float* __restrict__ w, int K, int ld, int Cout, uint16_t* __restrict__ wt, int Kp, int Cp) {
    const size_t total = (size_t)Cp * Kp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kp), k = (int)(i - (size_t)n * Kp);
        wt[i] = (n < Cout && k < K) ? f2lp<P>(w[(size_t)k * ld + n]) : (uint16_t)0;
    }
}

// max pool k x k / stride st (no padding, floor mode) on 16-bit NHWC views, 8 channels per thread
template <int P>
__global__ void k_maxpool_bf16(const uint16_t* __restrict__ x, int N, int H, int W, int C, int ldx, int xoff, int k, int st,
                               uint16_t* __restrict__ y, int ldy, int yoff, int OH, int OW) {
    const int C8 = C / 8;
    const size_t total = (size_t)N * OH * OW * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        size_t p = i / C8;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        float best[8];
#pragma unroll
        for (int j = 0; j < 8; j++) best[j] = -__builtin_inff();
        for (int dy = 0; dy < k; dy++)
            for (int dx = 0; dx < k; dx++) {
                const int iy = oy * st + dy, ix = ox * st + dx;
                if (iy >= H || ix >= W) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + ((size_t)(n * H + iy) * W + ix) * ldx + xoff + 8 * c8);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float lo = lp2f<P>((uint16_t)(v[j] & 0xFFFFu)), hi = lp2f<P>((uint16_t)(v[j] >> 16));
                    best[2 * j] = lo > best[2 * j] ? lo : best[2 * j];
                    best[2 * j + 1] = hi > best[2 * j + 1] ? hi : best[2 * j + 1];
                }
            }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (unsigned)f2lp<P>(best[2 * j]) | ((unsigned)f2lp<P>(best[2 * j + 1]) << 16);   // exact: the values ARE 16-bit
        *reinterpret_cast<u32x4*>(y + ((size_t)(n * OH + oy) * OW + ox) * ldy + yoff + 8 * c8) = o;
    }
}

// global average pool of a bf16 map -> f32 [N][C] (sum in pixel order, then / HW: the f32 kernel's expression)
template <int P>
__global__ void k_gap_bf16(const uint16_t* __restrict__ x, int N, int HW, int C, float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f;
    for (int p = 0; p < HW; p++) s += lp2f<P>(x[((size_t)n * HW + p) * C + c]);
    y[i] = s / (float)HW;
}

template <int BM, int BN, int BK>
int launch_bf16(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    const bool pad = a.ph || a.pw;
    if (a.lowp == 2) {
        if (pad) conv_bf16<BM, BN, BK, true, 2><<<grid, 256, 0, s>>>(a); else conv_bf16<BM, BN, BK, false, 2><<<grid, 256, 0, s>>>(a);
    } else {
        if (pad) conv_bf16<BM, BN, BK, true, 1><<<grid, 256, 0, s>>>(a); else conv_bf16<BM, BN, BK, false, 1><<<grid, 256, 0, s>>>(a);
    }
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

}  // namespace

int trl_launch_conv_bf16(const ConvArgs& a, hipStream_t s) {
    if (a.M <= 0) return TRL_OK;
    if (!a.wt || a.K != a.KH * a.KW * a.Cin || (a.Cin % 16) || (a.ldx % 8) || (a.xoff % 8) || (((uintptr_t)a.x) & 15) ||
        (long long)a.N * a.H * a.W * a.ldx + a.xoff >= 0x7fffffffll) {
        trl_set_error("bf16 conv: unsupported layer shape (Cin=%d ldx=%d xoff=%d K=%d)", a.Cin, a.ldx, a.xoff, a.K);
        return TRL_ERR_INVALID;
    }
    const bool k32 = a.Cin % 32 == 0;
    if (a.M >= 8192) return k32 ? launch_bf16<128, 64, 32>(a, s) : launch_bf16<128, 64, 16>(a, s);
    return k32 ? launch_bf16<64, 64, 32>(a, s) : launch_bf16<64, 64, 16>(a, s);
}

int trl_launch_to_bf16(const float* x, size_t n, uint16_t* y, hipStream_t s, int fmt) {
    if (n == 0) return TRL_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (fmt == 2) k_to_bf16<2><<<(unsigned)blocks, 256, 0, s>>>(x, n, y); else k_to_bf16<1><<<(unsigned)blocks, 256, 0, s>>>(x, n, y);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_make_weight_bf16(DevW* w, hipStream_t s, int fmt) {
    if (w->pt) return TRL_OK;
    const int Kp = (w->K + 31) & ~31, Cp = (w->Cout + 63) & ~63;
    TRL_HIP(hipMalloc((void**)&w->pt, (size_t)Kp * Cp * sizeof(uint16_t) + 64));
    w->ldt = Kp;
    size_t blocks = ((size_t)Kp * Cp + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (fmt == 2) k_weight_bf16_t<2><<<(unsigned)blocks, 256, 0, s>>>(w->p, w->K, w->ld, w->Cout, w->pt, Kp, Cp);
    else k_weight_bf16_t<1><<<(unsigned)blocks, 256, 0, s>>>(w->p, w->K, w->ld, w->Cout, w->pt, Kp, Cp);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_launch_maxpool_bf16(const uint16_t* x, int N, int H, int W, int C, int ldx, int xoff, int k, int st, uint16_t* y, int ldy,
                            int yoff, int OH, int OW, hipStream_t s, int fmt) {
    if ((C % 8) || (ldx % 8) || (xoff % 8) || (ldy % 8) || (yoff % 8)) { trl_set_error("bf16 max-pool needs channel counts in multiples of 8"); return TRL_ERR_INVALID; }
    const size_t total = (size_t)N * OH * OW * (C / 8);
    if (total == 0) return TRL_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    if (fmt == 2) k_maxpool_bf16<2><<<(unsigned)blocks, 256, 0, s>>>(x, N, H, W, C, ldx, xoff, k, st, y, ldy, yoff, OH, OW);
    else k_maxpool_bf16<1><<<(unsigned)blocks, 256, 0, s>>>(x, N, H, W, C, ldx, xoff, k, st, y, ldy, yoff, OH, OW);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_launch_gap_bf16(const uint16_t* x, int N, int HW, int C, float* y, hipStream_t s, int fmt) {
    if (N * C == 0) return TRL_OK;
    if (fmt == 2) k_gap_bf16<2><<<(N * C + 255) / 256, 256, 0, s>>>(x, N, HW, C, y); else k_gap_bf16<1><<<(N * C + 255) / 256, 256, 0, s>>>(x, N, HW, C, y);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
