// trl_common.h -- shared declarations of libtruely_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/truely_hip.h"

// ---- error plumbing --------------------------------------------------------------------------
void trl_set_error(const char* fmt, ...);
#define TRL_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            trl_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return TRL_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
#define TRL_CHECK(expr)                       \
    do {                                      \
        int s__ = (expr);                     \
        if (s__ != TRL_OK) return s__;        \
    } while (0)
#define TRL_LAUNCH_CHECK() TRL_HIP(hipGetLastError())

// ---- tuning switches ---------------------------------------------------------------------------
// Experiment / ablation switches (TRL_* environment variables: forced kernel variants, timing-only ablations that produce WRONG
// detections by construction, tile-size overrides) exist only in a tuning build (`make TUNING=1`); the shipped library reads no
// environment variable at all.  What the tests need from the shipped library is behind trl_debug_option().
#ifdef TRL_TUNING
#include <stdlib.h>
static inline int trl_tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool trl_tune_set(const char* name) { return getenv(name) != nullptr; }
static inline const char* trl_tune_str(const char* name) { return getenv(name); }
#else
#define trl_tune_int(name, dflt) (dflt)
#define trl_tune_set(name) (false)
#define trl_tune_str(name) ((const char*)nullptr)
#endif
extern int g_trl_no_fnconv;   // trl_debug_option("no_fnconv"): test hook, process-wide -- FaceNet's small maps through the generic conv kernels

// ---- device tensors ---------------------------------------------------------------------------
struct DevW {          // a weight matrix on the device, [Kpad][ld] row-major, zero padded
    float* p = nullptr;
    int K = 0, Cout = 0, Kpad = 0, ld = 0;
    uint16_t* pt = nullptr;   // optional bf16 copy, TRANSPOSED [Cout rounded to 64][ldt] (k contiguous: the MFMA B operand)
    int ldt = 0;              // K rounded up to 32
};
struct DevV {          // a per-channel vector, padded to a multiple of 128 floats
    float* p = nullptr;
    int n = 0;
};

// NHWC activation view
struct Act {
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;   // c = channels of this view
    int ld = 0;                        // floats between consecutive pixels (>= c)
    int coff = 0;                      // channel offset of the view inside the pixel
    bool bf = false;                   // elements are bf16 (2 bytes), p is then a uint16_t* in disguise (embed_precision = 1)
    size_t pixels() const { return (size_t)n * h * w; }
};

enum { TRL_ACT_NONE = 0, TRL_ACT_RELU = 1, TRL_ACT_PRELU = 2 };

struct ConvArgs {
    const float* x; int N, H, W, Cin, ldx, xoff;
    const float* w; int ldw, K;
    const float* bias;            // accumulator init (or null -> 0)
    const float* scale; const float* shift;   // folded BN (or null)
    const float* slope;           // PReLU (act == PRELU)
    const float* res; int ldres; float res_scale;   // residual: v = v*res_scale + res
    float* y; int ldy, yoff;
    int KH, KW, sh, sw, ph, pw, Cout, OH, OW, act;
    int M;                        // N*OH*OW
    // Device-sized batches (R-/O-Net candidate lists): when m_dev is set the launch covers a CAPACITY of N items and only the
    // first clamp(*m_dev - m_base, 0, N) exist; workgroups whose rows all lie past them exit at once (rows past the count
    // inside a live workgroup compute on in-bounds scratch and are never read).  No host round trip to size the grid.
    const int32_t* m_dev = nullptr; int m_base = 0; int m_per = 1;
    int ysplit = 1 << 30, yskip = 0;   // output column n >= ysplit is stored yskip channels further right (trl_fnconv.hip only)
    int lowp = 0;                 // 1 / 2: x, y, res are bf16 / fp16 and the weights come from wt (conv_bf16, FaceNet only)
    const uint16_t* wt = nullptr; int ldwt = 0;
};

// bump allocator over one device allocation
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    void reset() { off = 0; }
    void* alloc(size_t bytes) {
        size_t a = (off + 255) & ~(size_t)255;
        if (a + bytes > cap) return nullptr;
        off = a + bytes;
        return base + a;
    }
};

#ifdef __HIPCC__
// max without the compiler's operand canonicalisation: with IEEE mode on, hipcc quiets possible signalling NaNs in front of a
// two-operand v_max_f32 whose inputs it cannot prove canonical (values out of an MFMA or a load: one extra v_max_f32 x, x, x per
// operand).  The data here is finite, and in kernels that share the FP32 pipe with f32 MFMAs every VALU instruction costs
// matrix throughput.
__device__ __forceinline__ float vmax_nc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3_nc(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin_nc(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin3_nc(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmed3_nc(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// PReLU for ANY slope in two VALU instructions.  prelu(v) = v > 0 ? v : s v is max(v, s v) for s <= 1 (negative slopes
// included: then s v >= 0 >= v for v <= 0 and s v < 0 < v for v > 0) and min(v, s v) for s > 1; med3(v, s v, c) with the
// per-channel constant c = +inf / -inf selects between the two.  Exact: the result is one of the two operands, the same
// product the reference expression rounds (only the sign of a zero result can differ, which no later step observes).
__device__ __forceinline__ float trl_prelu_sel(float slope) { return slope > 1.f ? -__builtin_inff() : __builtin_inff(); }
__device__ __forceinline__ float trl_prelu_med3(float v, float slope, float sel) { return vmed3_nc(v, slope * v, sel); }
// max over a pool window of prelu(v_i), from the window's max m and min n only (PReLU after the pool, any slope sign):
//   s >= 0: prelu is monotone, the max is prelu(m);
//   s <  0: prelu(v) = max(v, s v), so max_i prelu(v_i) = max(max_i v_i, max_i s v_i) = max(m, s n)  (s n is the exact product
//           of the element that attains the min: float multiplication by a constant is monotone).
// Both are med3(m, s x, sel) with x = (s < 0 ? n : m).
__device__ __forceinline__ float trl_prelu_pooled(float m, float n, float slope, float sel) {
    return vmed3_nc(m, slope * (slope < 0.f ? n : m), sel);
}
__device__ __forceinline__ int trl_live_rows(const ConvArgs& a) {   // rows of the GEMM that exist (uniform: scalar loads)
    if (!a.m_dev) return a.M;
    int t = *a.m_dev - a.m_base;
    t = t < 0 ? 0 : (t > a.N ? a.N : t);
    return t * a.m_per;
}
#endif

// ---- kernels (launch wrappers) ----------------------------------------------------------------
int trl_launch_conv(const ConvArgs& a, hipStream_t s);
bool trl_fn_split4_rule(const ConvArgs& a);                                // the oracle's four-chain rule applies to the layer
bool trl_fn_eligible(const ConvArgs& a);                                   // trl_fnconv.hip: small-map conv family
int trl_launch_fn_group(const ConvArgs* convs, int nz, hipStream_t s);     // 1..3 independent convs in one launch
int trl_launch_maxpool(const float* x, int N, int H, int W, int C, int ldx, int xoff, int k, int st,
                       int ceil_mode, float* y, int ldy, int yoff, int OH, int OW, hipStream_t s,
                       const int32_t* n_dev = nullptr, int n_base = 0);
int trl_launch_gap(const float* x, int N, int HW, int C, float* y, hipStream_t s);
// reduced-precision embedder (trl_bf16.hip)
int trl_launch_conv_bf16(const ConvArgs& a, hipStream_t s);
int trl_launch_to_bf16(const float* x, size_t n, uint16_t* y, hipStream_t s, int fmt = 1);     // fmt: 1 = bf16, 2 = fp16
int trl_make_weight_bf16(DevW* w, hipStream_t s, int fmt = 1);
int trl_launch_maxpool_bf16(const uint16_t* x, int N, int H, int W, int C, int ldx, int xoff, int k, int st, uint16_t* y, int ldy,
                            int yoff, int OH, int OW, hipStream_t s, int fmt = 1);
int trl_launch_gap_bf16(const uint16_t* x, int N, int HW, int C, float* y, hipStream_t s, int fmt = 1);
int trl_launch_l2norm512(const float* x, const uint8_t* valid, int n, float* y, hipStream_t s);
int trl_launch_drift(const float* emb, const uint8_t* valid, int n, long long frame_count, int fps,
                     float* sims, uint8_t* flags, int32_t* result, hipStream_t s, void* state = nullptr);

static inline int trl_pool_out(int L, int k, int s, int ceil_mode) {
    int o;
    if (ceil_mode) {
        o = (L - k + s - 1) / s + 1;
        if ((o - 1) * s >= L) o--;
    } else {
        o = (L - k) / s + 1;
    }
    return o;
}

// ---- device math shared with the oracle (oracle/trl_oracle.c: orc_expf, softmax2_p1, orc_dot512)
#ifdef __HIPCC__
// Same operation sequence as orc_expf; with -ffp-contract=off every op rounds once, as on the CPU.
__device__ __forceinline__ float trl_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int e = (int)n;
    float s = __uint_as_float((unsigned)(e + 127) << 23);
    return y * s;
}
__device__ __forceinline__ float trl_softmax2_p1(float a0, float a1) {
    float m = a0 > a1 ? a0 : a1;
    float e0 = trl_expf(a0 - m), e1 = trl_expf(a1 - m);
    return e1 / (e0 + e1);
}
// 512-long dot by ONE wave: lane j accumulates elements j+64*i (i ascending, fmaf), then an xor
// butterfly (32,16,...,1).  Every lane returns the same value.  Mirrors orc_dot512.
__device__ __forceinline__ float trl_wave_dot512(const float* a, const float* b, int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s = __builtin_fmaf(a[lane + 64 * i], b[lane + 64 * i], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    return s;
}
#endif
