// trl_fnconv.hip -- implicit-GEMM convolutions for FaceNet's SMALL maps (7x7, 3x3, 1x1 at the reference's 80x80 crops:
// repeat_1/2/3, mixed_6a/7a, last_linear; server/model.py:59) on v_mfma_f32_16x16x4_f32.
//
// Why a second conv family next to trl_layers.hip: at 256 faces these layers have M = 12544 / 2304 / 256 output
// rows.  The 32x32x2-based kernels tile them into grids that do not match the chip (288 workgroups on 256 CUs = two
// rounds for 12 % more work; 48 workgroups for M = 256) and prefetch ONE K chunk ahead, so a chunk's ~1k MFMA cycles
// barely cover a cache-miss round trip: the rocprofv3 timeline showed 13-24 us launches for 2-6 us of matrix work.
// Here: 16-row MFMA tiles (finer M/N granularity, same 64 FLOP/clk/SIMD peak), a host-side chooser that picks the
// workgroup tile so the grid fits the 256 CUs in whole rounds, K chunks prefetched TWO ahead through registers,
// zero taps of fully padded windows skipped, and up to three independent convs per launch (blockIdx.z).
//
// Arithmetic contract (oracle/trl_oracle.c conv2d): one bias-seeded fmaf chain per output, k = (ky*KW+kx)*Cin + c
// ascending -- v_mfma_f32_16x16x4_f32 is that chain, 4 k per instruction; layers with OH*OW <= 9 and K >= 512 use
// FOUR chains over consecutive quarters of k combined as (c0 + c1) + (c2 + c3) (SPLIT4: one wave per quarter).
// A tap that lies in the padding for every output row contributes fmaf(0, w, acc) == acc and is skipped.
#include "trl_common.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include <utility>

typedef float f32x4m __attribute__((ext_vector_type(4)));

struct FnGroup {
    ConvArgs a[3];
    int gx[3];     // M-tiles of conv z (workgroups past them exit)
    int gy[3];     // N-tiles of conv z
    unsigned long long* dbg;   // diagnostic stamps (trl_debug_fn_arm), null in production
    int skip;                  // timing-only ablation (TRL_FN_SKIP): 1 = no MFMA, 2 = no global loads, 4 = no LDS staging; 0 in production
};

namespace {

constexpr int FBK = 32;                                   // K chunk: 32 channels of one filter tap
constexpr int FDEPTH = 2;                                 // chunks in flight: these layers wait on the Infinity Cache / HBM round
                                                          // trip of the previous layer's activations, not on bandwidth

template <int... I, typename F>
__device__ __forceinline__ void fn_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
constexpr int fn_ld(int b) { return b + ((b % 32 == 0) ? 16 : 0); }   // B tile [k][n]: leading dim == 16 (mod 32): conflict-free operand reads
constexpr int LDK = FBK + 2;                              // A tile [m][k], k contiguous: row stride 34 -> (2 m + k) mod 32 banks, conflict-free
// (A permuted-k layout read with ds_read_b64 was tried: the four scattered ds_write_b32 per staged float4 cost more than the
// narrower reads saved -- K loop +10..25 %.)
typedef float f32x2m __attribute__((ext_vector_type(2)));

// uniform: the tap (ky, kx) lies in the zero padding for EVERY output pixel (1x1 maps with padded 1x3 / 3x1 filters)
__device__ __forceinline__ bool fn_dead_tap(const ConvArgs& a, int ky, int kx) {
    return a.H == 1 && a.W == 1 && (ky != a.ph || kx != a.pw);
}

// Per-column epilogue constants live in registers; the residual values of a lane's outputs are all requested before the first
// store (stores may alias the residual buffer as far as the compiler knows: per-element load/store pairs would serialise on
// memory latency -- the first version of this kernel spent 80 % of its time there).
struct FnCol { float sc, sf, sl; };
__device__ __forceinline__ FnCol fn_col(const ConvArgs& a, int n) {
    FnCol c;
    const bool ok = n < a.Cout;
    c.sc = (a.scale && ok) ? a.scale[n] : 1.f;
    c.sf = (a.scale && ok) ? a.shift[n] : 0.f;
    c.sl = (a.act == TRL_ACT_PRELU && ok) ? a.slope[n] : 0.f;
    return c;
}
__device__ __forceinline__ float fn_finish(const ConvArgs& a, const FnCol& c, float v, float r) {
    if (a.scale) v = __builtin_fmaf(v, c.sc, c.sf);
    if (a.res) {
        v = v * a.res_scale;
        v = v + r;
    }
    if (a.act == TRL_ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (a.act == TRL_ACT_PRELU) v = v > 0.f ? v : c.sl * v;
    return v;
}

// ---- single chain: the four waves tile BM x BN as WM x WN ----------------------------------------------------------------
// DBG (both kernels): the diagnostic instantiation honours g.dbg (per-wave cycle stamps) and g.skip (timing-only ablations); the
// production instantiation reads neither.
template <int BM, int BN, int WM, int WN, bool DBG>
__global__ __launch_bounds__(256) void fn_conv(FnGroup g) {
    unsigned long long* const dbgp = DBG ? g.dbg : nullptr;   // compile-time null / 0 in production
    const int skipm = DBG ? g.skip : 0;
    const int z = blockIdx.z;
    if ((int)blockIdx.x >= g.gx[z] || (int)blockIdx.y >= g.gy[z]) return;
    // one batch of scalar loads for the conv's arguments (indexing the kernarg by a runtime z field by field cost ~15
    // dependent scalar-load round trips, 4k cycles, before the first global load was even issued)
    const ConvArgs a = g.a[z];
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    if (dbgp) st0 = __builtin_readcyclecounter();
    static_assert(WM * WN == 4 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "tile shape");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LDB = fn_ld(BN);
    constexpr int ASL = BM * (FBK / 4), APT = (ASL + 255) / 256;       // float4 slots of a chunk, per thread
    constexpr int BSL = FBK * (BN / 4), BPT = (BSL + 255) / 256;
    constexpr int ASZ = (BM * LDK + 3) & ~3, BSZ = FBK * LDB;
    __shared__ __attribute__((aligned(16))) float Asm[2 * ASZ];      // two staging buffers: one barrier per chunk
    __shared__ __attribute__((aligned(16))) float Bsm[2 * BSZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // A slots: slot = tid + 256 i -> (k-group = slot % 8, row = slot / 8): eight lanes read the 128 contiguous bytes one row
    // contributes to a chunk, so a wave's load touches 16 cache lines, not 64 (a row-per-lane gather spent half of every wave's
    // life waiting on the vector L1: SQ_WAIT_ANY 52 % in the first version).  Rows past M re-read row 0 (never stored).
    int aoff[APT], adst[APT];
    bool ain[APT][1];
    int iy0v[APT], ix0v[APT];
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < APT; i++) {
        const int slot = tid + 256 * i;
        const int gk = slot & 7, row = slot >> 3;
        const int m = m0 + row, mm = m < a.M ? m : 0;
        const int nimg = mm / ohw, rem = mm - nimg * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        iy0v[i] = oy * a.sh - a.ph; ix0v[i] = ox * a.sw - a.pw;
        aoff[i] = ((nimg * a.H + iy0v[i]) * a.W + ix0v[i]) * a.ldx + a.xoff + 4 * gk;
        adst[i] = row * LDK + 4 * gk;
        ain[i][0] = slot < ASL;
    }
    int boff[BPT], bdst[BPT];
#pragma unroll
    for (int i = 0; i < BPT; i++) {
        const int slot = tid + 256 * i;
        const int kk = slot / (BN / 4), n4 = slot - kk * (BN / 4);
        int bn = n0 + 4 * n4;
        bn = bn < a.ldw ? bn : 0;                          // columns past the matrix re-read column 0 (never stored)
        boff[i] = kk * a.ldw + bn;
        bdst[i] = kk * LDB + 4 * n4;
    }
    const bool pad = a.ph || a.pw;

    f32x4m ar[FDEPTH][APT], br[FDEPTH][BPT];   // ext_vector arrays stay in registers (HIP float4 structs captured by a lambda do not)
    int ky = 0, kx = 0, c0 = 0, k0 = 0;                    // scalar cursor of the next chunk to load
    auto load = [&](auto ST) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value;
        const int soff = (ky * a.W + kx) * a.ldx + c0;
#pragma unroll
        for (int i = 0; i < APT; i++) {
            f32x4m v = {0.f, 0.f, 0.f, 0.f};
            bool ok = ain[i][0];
            if (pad) ok = ok && (unsigned)(iy0v[i] + ky) < (unsigned)a.H && (unsigned)(ix0v[i] + kx) < (unsigned)a.W;
            if (ok && !(skipm & 2)) v = *reinterpret_cast<const f32x4m*>(a.x + (aoff[i] + soff));
            ar[st][i] = v;
        }
        const float* wrow = a.w + (size_t)k0 * a.ldw;
#pragma unroll
        for (int i = 0; i < BPT; i++) {
            f32x4m v = {0.f, 0.f, 0.f, 0.f};
            if ((BSL % 256 == 0 || tid + 256 * i < BSL) && !(skipm & 2)) v = *reinterpret_cast<const f32x4m*>(wrow + boff[i]);
            br[st][i] = v;
        }
        k0 += FBK; c0 += FBK;
        if (c0 >= a.Cin) { c0 = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
    };
    auto stage = [&](auto ST, int buf) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value;
        float* As = Asm + buf * ASZ;
        float* Bs = Bsm + buf * BSZ;
#pragma unroll
        for (int i = 0; i < APT; i++) {
            if (ASL % 256 == 0 || ain[i][0]) {      // two 8-byte stores (row stride 136 B keeps 8-byte alignment)
                *reinterpret_cast<f32x2m*>(&As[adst[i]]) = f32x2m{ar[st][i][0], ar[st][i][1]};
                *reinterpret_cast<f32x2m*>(&As[adst[i] + 2]) = f32x2m{ar[st][i][2], ar[st][i][3]};
            }
        }
#pragma unroll
        for (int i = 0; i < BPT; i++)
            if (BSL % 256 == 0 || tid + 256 * i < BSL) *reinterpret_cast<f32x4m*>(&Bs[bdst[i]]) = br[st][i];
    };

    f32x4m acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 16 + l15;
        const float b = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;     // the chain starts at the bias
#pragma unroll
        for (int tm = 0; tm < TM; tm++) acc[tm][tn] = f32x4m{b, b, b, b};
    }
    // All operands of a chunk are requested before its first MFMA (hipcc sinks each LDS read next to its use, which exposes
    // the LDS latency once per k-step: 2-3x the MFMA time of these short steps).
    float av[FBK / 4][TM], bv[FBK / 4][TN];
    auto fetch = [&](int buf) __attribute__((always_inline)) {
        const float* As = Asm + buf * ASZ;
        const float* Bs = Bsm + buf * BSZ;
#pragma unroll
        for (int s = 0; s < FBK / 4; s++) {
#pragma unroll
            for (int tm = 0; tm < TM; tm++) av[s][tm] = As[((wm * TM + tm) * 16 + l15) * LDK + 4 * s + kq];
#pragma unroll
            for (int tn = 0; tn < TN; tn++) bv[s][tn] = Bs[(4 * s + kq) * LDB + (wn * TN + tn) * 16 + l15];
        }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < FBK / 4; s++)
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][tm], bv[s][tn], acc[tm][tn], 0, 0, 0);
    };

    // chunks are loaded FDEPTH ahead: while chunk c is multiplied, c+1 .. c+FDEPTH-1 sit in registers or are in flight
    const int nchunks = a.K / FBK;
    constexpr auto SEQ = std::make_integer_sequence<int, FDEPTH>{};
    fn_static_for(SEQ, [&](auto J) __attribute__((always_inline)) { if (decltype(J)::value < nchunks) load(J); });
    // epilogue operands are requested NOW, behind the first chunks: per-column constants and the residual values of this
    // lane's outputs arrive while the K loop runs (fetched after it, their round trip was the whole epilogue)
    const float* __restrict__ rp = a.res;
    FnCol colc[TN];
    float rv[TM][TN][4];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 16 + l15;
        colc[tn] = fn_col(a, n);
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int mr = m0 + (wm * TM + tm) * 16 + kq * 4 + q;
                rv[tm][tn][q] = (rp && mr < a.M && n < a.Cout) ? rp[(size_t)mr * a.ldres + n] : 0.f;
            }
    }
    if (dbgp) st1 = __builtin_readcyclecounter();
    // chunk c lives in LDS buffer c & 1.  Per chunk: read its operands, drop chunk c+1 into the other buffer (free since the
    // barrier that ended chunk c-1), issue the global loads of chunk c+FDEPTH+... , multiply, ONE barrier.
    stage(std::integral_constant<int, 0>{}, 0);
    if (FDEPTH < nchunks) load(std::integral_constant<int, 0>{});
    __syncthreads();
    if (dbgp) st2 = __builtin_readcyclecounter();
    for (int ch = 0; ch < nchunks; ch += FDEPTH) {
        fn_static_for(SEQ, [&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            constexpr int jn = (j + 1) % FDEPTH;               // register stage holding chunk ch + j + 1
            const int c = ch + j;
            if (c < nchunks) {
                fetch(c & 1);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < nchunks) {
                    if (!(skipm & 4)) stage(std::integral_constant<int, jn>{}, (c + 1) & 1);
                    if (c + 1 + FDEPTH < nchunks) load(std::integral_constant<int, jn>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(skipm & 1)) multiply();
                __syncthreads();
            }
        });
    }

    if (dbgp) st3 = __builtin_readcyclecounter();
    float* __restrict__ yp = a.y;
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + (wn * TN + tn) * 16 + l15;
        if (n >= a.Cout) continue;
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int mr = m0 + (wm * TM + tm) * 16 + kq * 4 + q;
                if (mr < a.M) yp[(size_t)mr * a.ldy + a.yoff + n + (n >= a.ysplit ? a.yskip : 0)] = fn_finish(a, colc[tn], acc[tm][tn][q], rv[tm][tn][q]);
            }
    }
    if (dbgp && lane == 0) {
        unsigned long long* d = dbgp + (size_t)(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = __builtin_readcyclecounter(); d[5] = wall_clock64();
    }
}

// ---- four chains: wave w owns quarter w of k for the whole BM x BN tile; wave-private staging, no barrier in the K loop --
template <int BM, int BN, bool DBG>
__global__ __launch_bounds__(256) void fn_conv_split4(FnGroup g) {
    unsigned long long* const dbgp = DBG ? g.dbg : nullptr;   // compile-time null / 0 in production
    const int skipm = DBG ? g.skip : 0;
    const int z = blockIdx.z;
    if ((int)blockIdx.x >= g.gx[z] || (int)blockIdx.y >= g.gy[z]) return;
    const ConvArgs a = g.a[z];                                // one batch of scalar loads (see fn_conv)
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    if (dbgp) st0 = __builtin_readcyclecounter();
    constexpr int TM = BM / 16, TN = BN / 16;
    constexpr int LDB = fn_ld(BN);
    constexpr int APL = BM * (FBK / 4) / 64, BPL = FBK * (BN / 4) / 64;     // float4 slots per lane
    static_assert(BM * (FBK / 4) % 64 == 0 && FBK * (BN / 4) % 64 == 0, "slots per lane");
    constexpr int STG = (BM * LDK + FBK * LDB + 3) & ~3;
    constexpr int RED = BM * BN;                             // floats of one wave's partial tile
    constexpr int SM = 4 * STG > 4 * RED ? 4 * STG : 4 * RED;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // SM floats, dynamic: the 48x64 tile needs 67 KB (> the 64 KB static limit)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    float* As = sm + wave * STG;
    float* Bs = As + ((BM * LDK + 3) & ~3);
    // Wave w owns quarter w of k.  On a 1x1 map with a padded filter only the centre tap sees data: every other k multiplies a
    // zero (fmaf(0, w, acc) == acc), so the quarter is clipped to the centre tap's channels [lo, hi) -- the launcher checks
    // that the clipped ranges start and end on chunk boundaries inside the tap.
    const int segK = a.K >> 2;
    int ks = wave * segK, ke = ks + segK;
    if (a.H == 1 && a.W == 1 && (a.ph || a.pw)) {
        const int lo = (a.ph * a.KW + a.pw) * a.Cin, hi = lo + a.Cin;
        ks = ks > lo ? ks : lo;
        ke = ke < hi ? ke : hi;
        if (ke < ks) ke = ks;
    }

    int aoff[APL], adst[APL], iy0v[APL], ix0v[APL];
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < APL; i++) {
        const int slot = lane + 64 * i;
        const int gk = slot & 7, row = slot >> 3;              // eight lanes per row: see fn_conv
        const int m = m0 + row, mm = m < a.M ? m : 0;
        const int nimg = mm / ohw, rem = mm - nimg * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        iy0v[i] = oy * a.sh - a.ph; ix0v[i] = ox * a.sw - a.pw;
        aoff[i] = ((nimg * a.H + iy0v[i]) * a.W + ix0v[i]) * a.ldx + a.xoff + 4 * gk;
        adst[i] = row * LDK + 4 * gk;
    }
    int boff[BPL], bdst[BPL];
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int slot = lane + 64 * i;
        const int kk = slot / (BN / 4), n4 = slot - kk * (BN / 4);
        int bn = n0 + 4 * n4;
        bn = bn < a.ldw ? bn : 0;
        boff[i] = kk * a.ldw + bn;
        bdst[i] = kk * LDB + 4 * n4;
    }
    const bool pad = a.ph || a.pw;

    // scalar cursor of the wave's quarter
    const int tap0 = ks / a.Cin;
    int c0 = ks - tap0 * a.Cin, ky = tap0 / a.KW, kx = tap0 - ky * a.KW, k0 = ks;
    f32x4m ar[FDEPTH][APL], br[FDEPTH][BPL];
    bool live[FDEPTH];                                       // chunk in the stage is not an all-padding tap (indexed at compile time)
    auto load = [&](auto ST) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value;
        const bool lv = !(pad && fn_dead_tap(a, ky, kx));
        live[st] = lv;
        if (lv) {
            const int soff = (ky * a.W + kx) * a.ldx + c0;
#pragma unroll
            for (int i = 0; i < APL; i++) {
                f32x4m v = {0.f, 0.f, 0.f, 0.f};
                bool ok = true;
                if (pad) ok = (unsigned)(iy0v[i] + ky) < (unsigned)a.H && (unsigned)(ix0v[i] + kx) < (unsigned)a.W;
                if (ok && !(skipm & 2)) v = *reinterpret_cast<const f32x4m*>(a.x + (aoff[i] + soff));
                ar[st][i] = v;
            }
            const float* wrow = a.w + (size_t)k0 * a.ldw;
#pragma unroll
            for (int i = 0; i < BPL; i++) br[st][i] = (skipm & 2) ? f32x4m{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4m*>(wrow + boff[i]);
        }
        k0 += FBK; c0 += FBK;
        if (c0 >= a.Cin) { c0 = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
    };
    auto stage = [&](auto ST) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value;
#pragma unroll
        for (int i = 0; i < APL; i++) {
            *reinterpret_cast<f32x2m*>(&As[adst[i]]) = f32x2m{ar[st][i][0], ar[st][i][1]};
            *reinterpret_cast<f32x2m*>(&As[adst[i] + 2]) = f32x2m{ar[st][i][2], ar[st][i][3]};
        }
#pragma unroll
        for (int i = 0; i < BPL; i++) *reinterpret_cast<f32x4m*>(&Bs[bdst[i]]) = br[st][i];
    };

    f32x4m acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int n = n0 + tn * 16 + l15;
        const float b = (wave == 0 && a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;   // chain 0 starts at the bias
#pragma unroll
        for (int tm = 0; tm < TM; tm++) acc[tm][tn] = f32x4m{b, b, b, b};
    }
    auto compute = [&]() __attribute__((always_inline)) {      // every operand of the chunk first, then the MFMAs (see fn_conv)
        float av[FBK / 4][TM], bv[FBK / 4][TN];
#pragma unroll
        for (int s = 0; s < FBK / 4; s++) {
#pragma unroll
            for (int tm = 0; tm < TM; tm++) av[s][tm] = As[(tm * 16 + l15) * LDK + 4 * s + kq];
#pragma unroll
            for (int tn = 0; tn < TN; tn++) bv[s][tn] = Bs[(4 * s + kq) * LDB + tn * 16 + l15];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (skipm & 1) {      // timing ablation: keep the operand reads alive without the matrix pipe
            float t = 0.f;
#pragma unroll
            for (int s = 0; s < FBK / 4; s++) {
#pragma unroll
                for (int tm = 0; tm < TM; tm++) t += av[s][tm];
#pragma unroll
                for (int tn = 0; tn < TN; tn++) t += bv[s][tn];
            }
            acc[0][0][0] += t;
            return;
        }
#pragma unroll
        for (int s = 0; s < FBK / 4; s++)
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][tm], bv[s][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nchunks = (ke - ks) / FBK;
    constexpr auto SEQ = std::make_integer_sequence<int, FDEPTH>{};
    fn_static_for(SEQ, [&](auto J) __attribute__((always_inline)) { if (decltype(J)::value < nchunks) load(J); });
    // epilogue operands requested behind the first chunks (see fn_conv): thread (wave = q, lane) finishes element q of every block
    const float* __restrict__ rp = a.res;
    FnCol cc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) cc[tn] = fn_col(a, n0 + tn * 16 + l15);
    float rv[TM * TN];
#pragma unroll
    for (int blk = 0; blk < TM * TN; blk++) {
        const int tm = blk / TN, tn = blk - tm * TN;
        const int mr = m0 + tm * 16 + kq * 4 + wave;
        const int n = n0 + tn * 16 + l15;
        rv[blk] = (rp && mr < a.M && n < a.Cout) ? rp[(size_t)mr * a.ldres + n] : 0.f;
    }
    if (dbgp) st1 = __builtin_readcyclecounter();
    for (int ch = 0; ch < nchunks; ch += FDEPTH) {
        // wave-private staging: the LDS operations of one wave stay in order, no workgroup barrier needed
        fn_static_for(SEQ, [&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if (ch + j < nchunks) {
                const bool lj = live[j];
                if (lj && !(skipm & 4)) stage(J);
                if (dbgp && ch + j == 0) { __builtin_amdgcn_s_waitcnt(0); st2 = __builtin_readcyclecounter(); }
                if (ch + j + FDEPTH < nchunks) load(J);
                if (lj) compute();
            }
        });
    }
    if (dbgp) st3 = __builtin_readcyclecounter();
    __syncthreads();
    static_assert(SM * sizeof(float) <= 160 * 1024, "LDS per workgroup");
    float* red = sm;                                         // [wave][tm][tn][q][lane]
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[wave * RED + ((tm * TN + tn) * 4 + q) * 64 + lane] = acc[tm][tn][q];
    __syncthreads();
    // thread (wave = q, lane) owns element q of every 16x16 block: e = tid + 256 blk
    float* __restrict__ yp = a.y;
#pragma unroll
    for (int blk = 0; blk < TM * TN; blk++) {
        const int tm = blk / TN, tn = blk - tm * TN;
        const int e = tid + 256 * blk;
        const int mr = m0 + tm * 16 + kq * 4 + wave;
        const int n = n0 + tn * 16 + l15;
        if (mr >= a.M || n >= a.Cout) continue;
        const float v = (red[e] + red[RED + e]) + (red[2 * RED + e] + red[3 * RED + e]);
        yp[(size_t)mr * a.ldy + a.yoff + n + (n >= a.ysplit ? a.yskip : 0)] = fn_finish(a, cc[tn], v, rv[blk]);
    }
    if (dbgp && lane == 0) {
        unsigned long long* d = dbgp + (size_t)(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = __builtin_readcyclecounter(); d[5] = wall_clock64();
    }
}

struct Tile { int bm, bn; };

template <int BM, int BN, int WM, int WN>
int launch1(const FnGroup& g, int nz, hipStream_t s) {
    int gx = 0, gy = 0;
    for (int z = 0; z < nz; z++) { gx = g.gx[z] > gx ? g.gx[z] : gx; gy = g.gy[z] > gy ? g.gy[z] : gy; }
    if (g.dbg || g.skip) fn_conv<BM, BN, WM, WN, true><<<dim3(gx, gy, nz), 256, 0, s>>>(g);
    else fn_conv<BM, BN, WM, WN, false><<<dim3(gx, gy, nz), 256, 0, s>>>(g);
    return TRL_OK;
}
template <int BM, int BN>
int launch4(const FnGroup& g, int nz, hipStream_t s) {
    int gx = 0, gy = 0;
    for (int z = 0; z < nz; z++) { gx = g.gx[z] > gx ? g.gx[z] : gx; gy = g.gy[z] > gy ? g.gy[z] : gy; }
    constexpr int STG = (BM * LDK + FBK * fn_ld(BN) + 3) & ~3, RED = BM * BN;
    constexpr size_t bytes = sizeof(float) * (size_t)(4 * STG > 4 * RED ? 4 * STG : 4 * RED);
    // The dynamic-LDS limit is a per-DEVICE function attribute (the 48x64 tile needs 67 KB, over the 64 KB default): set on every
    // launch -- a process may hold contexts on several devices, and worker threads launch concurrently (a host-side call, no
    // device work; a process-wide "already set" flag was wrong on both counts).
    if (g.dbg || g.skip) {
        TRL_HIP(hipFuncSetAttribute((const void*)fn_conv_split4<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        fn_conv_split4<BM, BN, true><<<dim3(gx, gy, nz), 256, bytes, s>>>(g);
    } else {
        TRL_HIP(hipFuncSetAttribute((const void*)fn_conv_split4<BM, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        fn_conv_split4<BM, BN, false><<<dim3(gx, gy, nz), 256, bytes, s>>>(g);
    }
    return TRL_OK;
}

// diagnostic stamps: the k-th launch after trl_debug_fn_arm(k) records per-wave cycle stamps
unsigned long long* g_dbg_buf = nullptr;
int g_dbg_arm = -1, g_dbg_waves = 0;
constexpr int DBG_WAVES = 1 << 16;

bool fn_split4(const ConvArgs& a) { return a.OH * a.OW <= 9 && a.K >= 512 && (a.K & 15) == 0; }

}  // namespace

bool trl_fn_split4_rule(const ConvArgs& a) { return fn_split4(a); }

// Does this family take the layer?  Whole-tap chunks of 32 channels, float4-aligned input, 32-bit element offsets, small M.
bool trl_fn_eligible(const ConvArgs& a) {
    const bool off = g_trl_no_fnconv != 0 || trl_tune_set("TRL_NO_FNCONV");
    if (off || a.lowp || a.m_dev) return false;
    if (a.M <= 0 || a.M > 16384 || a.Cin % 32 != 0 || a.K != a.KH * a.KW * a.Cin) return false;
    if ((a.ldx & 3) || (a.xoff & 3) || (((uintptr_t)a.x) & 15) || (a.ldw & 3)) return false;
    if ((long long)a.N * a.H * a.W * a.ldx + a.xoff >= 0x7fffffffll || (long long)a.K * a.ldw >= 0x7fffffffll) return false;
    if (fn_split4(a)) {
        const int seg = a.K >> 2;
        if (a.H == 1 && a.W == 1 && (a.ph || a.pw)) {         // quarters clipped to the centre tap must align to chunks
            const int lo = (a.ph * a.KW + a.pw) * a.Cin, hi = lo + a.Cin;
            for (int q = 0; q < 4; q++) {
                int s0 = q * seg > lo ? q * seg : lo, s1 = (q + 1) * seg < hi ? (q + 1) * seg : hi;
                if (s1 > s0 && ((s0 - lo) % FBK != 0 || (s1 - s0) % FBK != 0)) return false;
            }
        } else if (seg % FBK != 0) return false;
    }
    return true;
}

// Workgroup tile by MEASUREMENT (tools/tune_fn_tiles.sh: every tile forced in turn under rocprofv3, per-launch times compared at the
// 256-face geometry M = 12544 / 2304 / 256): these launches live on occupancy, not on operand reuse -- 32x32 tiles (many small
// workgroups hide each other's load latency) beat the "whole rounds of 256 CUs" model of the first version by 13 % overall; the
// exceptions are the four-chain layers on 3x3 maps, whose wave-private staging favours 48-row tiles for N <= 256.
static Tile pick_tile(int M, int N, bool split4, int nz) {
    if (!split4) return (M <= 4096 && N >= 512) ? Tile{64, 64} : Tile{32, 32};
    if (nz > 1 || M <= 512) return Tile{16, 32};
    if (M <= 4096) return N >= 384 ? Tile{32, 32} : (N >= 192 ? Tile{48, 64} : Tile{48, 32});
    return Tile{32, 32};
}

// Up to three convs of the SAME class (all split-4 or all single-chain) in one launch; they must not depend on each other.
int trl_launch_fn_group(const ConvArgs* convs, int nz, hipStream_t s) {
    if (nz < 1 || nz > 3) { trl_set_error("fn group size"); return TRL_ERR_INVALID; }
    const bool sp = fn_split4(convs[0]);
    int M = 0, N = 0;
    for (int z = 0; z < nz; z++) {
        if (!trl_fn_eligible(convs[z]) || fn_split4(convs[z]) != sp) { trl_set_error("fn group: ineligible or mixed convs"); return TRL_ERR_INVALID; }
        M = convs[z].M > M ? convs[z].M : M;
        N += convs[z].Cout;                                  // the group shares the chip: choose the tile for the combined width
    }
    FnGroup g;
    // instantiated tiles: single chain {32,64,128}x{32,64,96,128} subset below; four chains {16,32,48}x{32,64}
    Tile t;
    // tuning aid: TRL_FN_FORCE1 / TRL_FN_FORCE4 = "BMxBN" forces the tile of every single-chain / four-chain launch
    static const char* f1 = trl_tune_str("TRL_FN_FORCE1");
    static const char* f4 = trl_tune_str("TRL_FN_FORCE4");
    const char* force = sp ? f4 : f1;
    int fbm = 0, fbn = 0;
    if (force && sscanf(force, "%dx%d", &fbm, &fbn) == 2) t = Tile{fbm, fbn};
    else t = pick_tile(M, nz == 1 ? convs[0].Cout : N, sp, nz);
    static const int skip = trl_tune_int("TRL_FN_SKIP", 0);
    g.skip = skip;
    g.dbg = nullptr;
    if (g_dbg_arm >= 0 && g_dbg_arm-- == 0) g.dbg = g_dbg_buf;
    for (int z = 0; z < 3; z++) {
        g.a[z] = convs[z < nz ? z : 0];
        g.gx[z] = z < nz ? (convs[z].M + t.bm - 1) / t.bm : 0;
        g.gy[z] = z < nz ? (convs[z].Cout + t.bn - 1) / t.bn : 0;
    }
    if (g.dbg) {
        int gx = 0, gy = 0;
        for (int z = 0; z < nz; z++) { gx = g.gx[z] > gx ? g.gx[z] : gx; gy = g.gy[z] > gy ? g.gy[z] : gy; }
        g_dbg_waves = gx * gy * nz * 4;
        if (g_dbg_waves > DBG_WAVES) { g.dbg = nullptr; g_dbg_waves = 0; }
        else fprintf(stderr, "[fn stamps] tile %dx%d split4=%d grid %dx%dx%d M=%d N=%d K=%d Cin=%d\n", t.bm, t.bn, (int)sp, gx, gy, nz, convs[0].M, convs[0].Cout, convs[0].K, convs[0].Cin);
    }
#define FN1(BM, BN, WM, WN) if (t.bm == BM && t.bn == BN) { TRL_CHECK((launch1<BM, BN, WM, WN>(g, nz, s))); TRL_LAUNCH_CHECK(); return TRL_OK; }
#define FN4(BM, BN) if (t.bm == BM && t.bn == BN) { TRL_CHECK((launch4<BM, BN>(g, nz, s))); TRL_LAUNCH_CHECK(); return TRL_OK; }
    if (sp) {
        FN4(16, 32) FN4(16, 64) FN4(32, 32) FN4(32, 64) FN4(48, 32) FN4(48, 64)
    } else {
        FN1(32, 32, 2, 2) FN1(32, 64, 2, 2) FN1(64, 32, 2, 2) FN1(64, 64, 2, 2) FN1(64, 96, 2, 2) FN1(128, 32, 4, 1) FN1(128, 64, 4, 1)
        FN1(32, 128, 1, 4) FN1(16, 64, 1, 4)
    }
#undef FN1
#undef FN4
    trl_set_error("fn tile %dx%d not instantiated", t.bm, t.bn);
    return TRL_ERR_STATE;
}

// Diagnostic: arm per-wave cycle stamps for the k-th small-map conv launch from now; read them back after a synchronise.
// Rows of 8 u64 per wave: entry, loads issued, first chunk staged, K loop done, stores issued, wall clock (100 MHz).
extern "C" int trl_debug_fn_arm(int k) {
    if (!g_dbg_buf && hipMalloc((void**)&g_dbg_buf, (size_t)DBG_WAVES * 64) != hipSuccess) return TRL_ERR_HIP;
    (void)hipMemset(g_dbg_buf, 0, (size_t)DBG_WAVES * 64);
    g_dbg_arm = k; g_dbg_waves = 0;
    return TRL_OK;
}
extern "C" int trl_debug_fn_read(unsigned long long* h_rows, int max_waves, int* n_waves) {
    if (!g_dbg_buf || !h_rows || !n_waves) return TRL_ERR_INVALID;
    TRL_HIP(hipDeviceSynchronize());
    const int n = g_dbg_waves < max_waves ? g_dbg_waves : max_waves;
    if (n > 0) TRL_HIP(hipMemcpy(h_rows, g_dbg_buf, (size_t)n * 64, hipMemcpyDeviceToHost));
    *n_waves = n;
    return TRL_OK;
}
