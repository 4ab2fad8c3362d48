// trl_front.hip -- fused front end of R-Net / O-Net for gfx950 (detect_face stages 2 and 3):
//   crop [y-1:ey, x-1:ex] of the u8 frame -> imresample to SxS -> (v-127.5)*0.0078125
//   -> conv1 3x3 (3->28 / 3->32) + PReLU -> MaxPool(3, 2, ceil_mode)            (one workgroup per candidate;
//      PReLU moves behind the pool when the slopes are monotone, see MODE)
// The conv1 activation (54 KB / 271 KB per candidate) never leaves LDS: only the pooled map is written,
// which removes ~7.6 GB of HBM traffic per 256-frame batch.  conv1 runs on v_mfma_f32_16x16x4_f32 with the
// weights in registers, k ascending from the bias: bit-identical to the oracle's conv2d chain.
#include "trl_ctx.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// a / b for b's correctly rounded reciprocal r: the correctly rounded quotient on the verified domain (see the reduce step below)
__device__ __forceinline__ float rdiv(float a, float b, float r) {
    const float q0 = a * r;
    const float e = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e, r, q0);
}

typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at dword alignment

__device__ __forceinline__ unsigned valid_bytes(int rel, int nbytes) {   // 0xFF for bytes b of the dword with rel+b < nbytes
    int hi = nbytes - rel; hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
    return hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
}

// PReLU is applied once per POOLED value, never to the conv1 map (the reference order is PReLU, then pool; the result is the same
// bits -- trl_common.h trl_prelu_pooled):
// MODE: 0 = some slope is negative: the pool keeps the window's min next to its max, max_i prelu(v_i) = med3(m, s (s < 0 ? n : m), +-inf);
//       1 = all slopes >= 0: PReLU is monotone and commutes with max: med3(m, s m, +-inf) (any slope size);
//       2 = all slopes in [0, 1]: prelu(m) == max(m, s m).
// The kernel is VALU-bound (crop unpacking, pooling), and VALU shares the FP32 pipe with the f32 MFMAs.
// DBG: the timing-only phase ablations (TRL_FRONT_SKIP, tools/front_ablation.sh); compiled out of the production instantiation.
#ifndef TRL_FRONT_MINW
#define TRL_FRONT_MINW 1      // tuning aid: minimum waves per SIMD the kernel is compiled for (register budget)
#endif
template <int S, int C1, int R, int MODE, bool DBG>
__global__ __launch_bounds__(256, TRL_FRONT_MINW) void k_mtcnn_front(const uint8_t* __restrict__ frames, int nframes, int H, int W,
                                                     const int4* __restrict__ cbox, const int32_t* __restrict__ d_total, int t0,
                                                     const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ s1,
                                                     float* __restrict__ out, int dbg_skip) {
    constexpr int CW = S - 2;                        // conv1 output side
    constexpr int P = (CW - 3 + 1) / 2 + 1;          // MaxPool(3,2,ceil) output side (11 / 23)
    constexpr int SR = 2 * R + 1;                    // conv1 rows per strip
    constexpr int IN_N = S * S * 3;
    constexpr int CLD = 36;                          // channel stride of the conv1 strip: float4 pooling reads, and the
                                                     // epilogue's rows 4*kq+q land 16 banks apart (conflict free)
    constexpr int MROWS = (SR * CW + 15) & ~15;      // strip rows padded to whole M-tiles: unguarded epilogue stores
    // + zero tail: the zero-weight k = 27 pad of the last conv row reads in_s[IN_N + 3 x] for x < CW.  It must hold ZEROS, not
    // just any finite value: stale LDS can contain NaN bit patterns and NaN * 0 would poison that conv output.
    constexpr int IN_PAD = (3 * S + 15) & ~15;
    __shared__ __attribute__((aligned(16))) float in_s[IN_N + IN_PAD];
    __shared__ __attribute__((aligned(16))) float c1_s[MROWS * CLD];
    // per-wave column-sum strip (aliases c1_s, which is dead during the crop)
    constexpr int COLCAP = ((MROWS * CLD - 16) / 4 < 2048 ? (MROWS * CLD - 16) / 4 : 2048) & ~3;
    static_assert(COLCAP >= 512, "column strip too small");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int t = t0 + blockIdx.x;
    if (!DBG) dbg_skip = 0;                          // a compile-time 0 in production
    // The candidate record and the device-side total are requested together: ONE memory round trip between the launch and the
    // crop's pixel loads (the crop is bound by dependent round trips).  Records past the total are unwritten memory inside the
    // list's allocation; they are read and discarded.
    const int4 cb0 = cbox[2 * (size_t)t], cb1 = cbox[2 * (size_t)t + 1];
    const int total = *d_total;
    asm volatile("" ::"s"(cb0.x), "s"(cb1.x), "s"(total));   // both scalar loads are issued before the branch below (hipcc would sink the record's load behind it)
    if (t >= total) return;             // the launch is sized by a capacity; candidates past the device-side total do not exist

    // ---- crop + area resample + normalise -> in_s [S][S][3] ----------------------------------------------
    // Each WAVE owns output rows oy = wave, wave+4, ...: (1) column sums of the bin's source rows, lanes along
    // the row (coalesced aligned dwords re-aligned with v_alignbyte, four byte-columns per lane) into a
    // per-wave LDS strip, (2) horizontal bins from the strip.  Integer sums: exact in any order.  No block barrier.
    {
        const int f = __builtin_amdgcn_readfirstlane(cb0.x), y0 = __builtin_amdgcn_readfirstlane(cb0.y), x0 = __builtin_amdgcn_readfirstlane(cb0.z);
        const int ih = __builtin_amdgcn_readfirstlane(cb0.w), iw = __builtin_amdgcn_readfirstlane(cb1.x);
        const uint32_t* base32 = reinterpret_cast<const uint32_t*>(frames);
        const long long last_dw = ((long long)nframes * H * W * 3 - 1) >> 2;   // last dword holding frame bytes
        const long long fbyte0 = (long long)f * H * W * 3;
        unsigned* colbuf = reinterpret_cast<unsigned*>(c1_s) + wave * COLCAP;
        if (dbg_skip & 1) {
            for (int p = tid; p < IN_N; p += 256) in_s[p] = 0.f;
        } else if (ih <= 3 * S && iw <= 3 * S) {
            // small boxes: bins of at most 4x4 pixels, i.e. <= 12 bytes per source row = one dword-aligned 16-byte load.
            // One thread per output pixel, three pixels per pass: all row loads are issued before any is consumed (one memory
            // round trip per pass instead of one per tap); bytes -> channel sums with v_alignbyte + v_dot4.
            // The path is VALU-bound (VALU shares the FP32 pipe with conv1's MFMAs), so: offsets are 32-bit from a scalar,
            // dword-aligned frame base; only the rows a bin of THIS box can have are fetched (ROWS = 2..4, uniform); the
            // end-of-buffer clamp runs only for a box at the very end of the last frame; the bin mean uses the exhaustively
            // verified reciprocal division (bins <= 4x4: far inside its domain) instead of two IEEE divisions per channel.
            const int fb3 = (int)(fbyte0 & 3);
            const char* fptr = reinterpret_cast<const char*>(frames) + (fbyte0 - fb3);         // dword aligned, scalar
            const int row_bytes = W * 3;
            const int khmax_c = (ih + S - 1) / S + 1;                                        // bins are ceil(ih/S) or one more rows tall
            // furthest byte any lane can touch: last row of the box, last bin's aligned 16-byte fetch
            const long long far = fbyte0 + ((long long)(y0 + ih - 1) * W + x0 + iw) * 3 + 16;
            const bool safe = far <= (long long)nframes * H * W * 3;
            auto small_pass = [&](auto ROWS_T, auto SAFE_T) {
                constexpr int ROWS = decltype(ROWS_T)::value;
                constexpr bool SAFE = decltype(SAFE_T)::value;
                for (int pb = tid; pb < S * S; pb += 3 * 256) {
                    unsigned ww[3][ROWS][4], shv[3][ROWS];
                    int khv[3], kwv[3];
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const int pp = pb + 256 * u, p = pp < S * S ? pp : 0;
                        const int oy = p / S, ox = p - oy * S;
                        const int ys = (oy * ih) / S, ye = ((oy + 1) * ih + S - 1) / S;
                        const int xs = (ox * iw) / S, xe = ((ox + 1) * iw + S - 1) / S;
                        khv[u] = ye - ys; kwv[u] = xe - xs;
                        const unsigned lo0 = (unsigned)((y0 + ys) * row_bytes + (x0 + xs) * 3 + fb3);
#pragma unroll
                        for (int r = 0; r < ROWS; r++) {
                            const unsigned lo = lo0 + (unsigned)((r < khv[u] ? r : khv[u] - 1) * row_bytes);     // dead rows re-read the bin's last row
                            shv[u][r] = lo & 3u;
                            if (SAFE) {
                                const u32x4_a4 v4 = *reinterpret_cast<const u32x4_a4*>(fptr + (lo & ~3u));
                                ww[u][r][0] = v4[0]; ww[u][r][1] = v4[1]; ww[u][r][2] = v4[2]; ww[u][r][3] = v4[3];
                            } else {
                                const long long dw = ((fbyte0 - fb3) + (long long)(lo & ~3u)) >> 2;
#pragma unroll
                                for (int j = 0; j < 4; j++) ww[u][r][j] = base32[dw + j <= last_dw ? dw + j : last_dw];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const int nbytes = kwv[u] * 3;
                        const unsigned vm0 = valid_bytes(0, nbytes), vm1 = valid_bytes(4, nbytes), vm2 = valid_bytes(8, nbytes);
                        unsigned a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                        for (int r = 0; r < ROWS; r++) {
                            const bool live = r < khv[u];
                            const unsigned sh = shv[u][r];
                            const unsigned d0 = __builtin_amdgcn_alignbyte(ww[u][r][1], ww[u][r][0], sh) & (live ? vm0 : 0u);
                            const unsigned d1 = __builtin_amdgcn_alignbyte(ww[u][r][2], ww[u][r][1], sh) & (live ? vm1 : 0u);
                            const unsigned d2 = __builtin_amdgcn_alignbyte(ww[u][r][3], ww[u][r][2], sh) & (live ? vm2 : 0u);
                            a0 = __builtin_amdgcn_udot4(d0, 0x01000001u, a0, false); a1 = __builtin_amdgcn_udot4(d0, 0x00000100u, a1, false);
                            a2 = __builtin_amdgcn_udot4(d0, 0x00010000u, a2, false);
                            a0 = __builtin_amdgcn_udot4(d1, 0x00010000u, a0, false); a1 = __builtin_amdgcn_udot4(d1, 0x01000001u, a1, false);
                            a2 = __builtin_amdgcn_udot4(d1, 0x00000100u, a2, false);
                            a0 = __builtin_amdgcn_udot4(d2, 0x00000100u, a0, false); a1 = __builtin_amdgcn_udot4(d2, 0x00010000u, a1, false);
                            a2 = __builtin_amdgcn_udot4(d2, 0x01000001u, a2, false);
                        }
                        const int pp = pb + 256 * u;
                        if (pp < S * S) {
                            // a / kh / kw through the correctly rounded reciprocals of kh, kw in 1..4 (rdiv: equal to the two IEEE
                            // divisions for every byte sum, oracle/trl_oracle.c orc_selftest_recip_div)
                            const int kh = khv[u], kw = kwv[u];
                            const float fkh = (float)kh, fkw = (float)kw;
                            const float rkh = kh == 3 ? (1.0f / 3.0f) : (kh == 1 ? 1.0f : (kh == 2 ? 0.5f : 0.25f));
                            const float rkw = kw == 3 ? (1.0f / 3.0f) : (kw == 1 ? 1.0f : (kw == 2 ? 0.5f : 0.25f));
                            in_s[3 * pp + 0] = (rdiv(rdiv((float)a0, fkh, rkh), fkw, rkw) - 127.5f) * 0.0078125f;
                            in_s[3 * pp + 1] = (rdiv(rdiv((float)a1, fkh, rkh), fkw, rkw) - 127.5f) * 0.0078125f;
                            in_s[3 * pp + 2] = (rdiv(rdiv((float)a2, fkh, rkh), fkw, rkw) - 127.5f) * 0.0078125f;
                        }
                    }
                }
            };
            if (!safe) small_pass(std::integral_constant<int, 4>{}, std::false_type{});
            else if (khmax_c <= 2) small_pass(std::integral_constant<int, 2>{}, std::true_type{});
            else if (khmax_c == 3) small_pass(std::integral_constant<int, 3>{}, std::true_type{});
            else small_pass(std::integral_constant<int, 4>{}, std::true_type{});
        } else {
            // big boxes: a wave owns output rows oy = wave (mod 4) and works on TWO of them (oy, oy+4) in lock step, so
            // 32 independent dword loads (2 bins x 4 source rows x 4 chunks) are in flight per round trip -- the crop is
            // bound by dependent round trips, not by bytes.  A chunk is 63 payload dwords: lane i re-aligns dword i
            // with dword i+1 taken from lane i+1 by a DPP wave shift (no second load); lane 63 only feeds lane 62.
            // Addressing is a scalar row base plus a clamped 32-bit lane offset, the re-alignment shift is a scalar per
            // row, bytes accumulate in packed 16-bit halves (<= 256 rows x 255 per flush).
            static_assert((S / 4) % 2 == 0, "rows per wave must pair up");
            constexpr int CAP2 = (COLCAP / 2) & ~3;
            unsigned* colA = colbuf;
            unsigned* colB = colbuf + CAP2;
            const long long row_pitch = (long long)W * 3;
            const bool same_phase = (row_pitch & 3) == 0;
            // adaptive-pool bins [floor(i a), ceil((i+1) a)), a = iw / S, are ceil(a) or ceil(a) + 1 columns wide
            const int kwA0 = (iw + S - 1) / S;
            const float rkw0 = 1.0f / (float)kwA0, rkw1 = 1.0f / (float)(kwA0 + 1);
            const bool fastdiv = ih <= 94 * S && iw <= 94 * S;      // every bin <= 96 x 96: the verified domain of rdiv
            for (int oyA = wave; oyA < S; oyA += 8) {
                const int oyB = oyA + 4;
                const int ysA = (oyA * ih) / S, yeA = ((oyA + 1) * ih + S - 1) / S, khA = yeA - ysA;
                const int ysB = (oyB * ih) / S, yeB = ((oyB + 1) * ih + S - 1) / S, khB = yeB - ysB;
                const int khm = khA > khB ? khA : khB;
                int oxa = 0;
                while (oxa < S) {
                    // segment of output columns whose source span fits half the strip
                    const int xsa = (oxa * iw) / S;
                    int oxb = oxa + 1;
                    if ((iw - xsa) * 3 <= CAP2 - 4) oxb = S;      // the rest of the row fits (every box up to ~300 px): no search
                    else while (oxb < S && (((oxb + 1) * iw + S - 1) / S - xsa) * 3 <= CAP2 - 4) oxb++;
                    const int xeb = (oxb * iw + S - 1) / S;
                    const int seg_bytes = (xeb - xsa) * 3;
                    const long long o_segA = fbyte0 + ((long long)(y0 + ysA) * W + x0 + xsa) * 3;
                    const long long o_segB = fbyte0 + ((long long)(y0 + ysB) * W + x0 + xsa) * 3;
                    // Column sums of the segment's source bytes over the bin's rows, one 32-bit accumulator per byte column and
                    // bin (v_dot4 with a one-hot byte weight: acc += byte j; a row only the other bin has gets weight 0).  This
                    // path is bound by its VALU instructions, so (a) only the chunks the segment reaches are processed (one
                    // straight-line instantiation per chunk count: guards inside the unrolled body made hipcc wait for the loads
                    // chunk by chunk), (b) when the row pitch is a multiple of 4 every row of the box has the SAME byte phase, so
                    // the dwords are summed as loaded (aligned columns, 64 payload dwords per chunk) and the phase is applied once,
                    // as an index shift at the flush -- 4 VALU per dword instead of 8; other pitches re-align every dword first
                    // (lane i takes dword i+1 from lane i+1 by a DPP wave shift; lane 63 only feeds lane 62: 63 payload dwords).
                    const int sh0 = same_phase ? (int)(o_segA & 3) : 0;              // byte phase of the segment start
                    const int cstride = same_phase ? 256 : 252;                      // payload bytes per chunk
                    const int span = seg_bytes + sh0;                                // aligned-relative bytes to cover
                    for (int c0 = 0; c0 < span; c0 += 4 * cstride) {
                        const int rem = span - c0;
                        const int nch = 1 + (rem > cstride ? 1 : 0) + (rem > 2 * cstride ? 1 : 0) + (rem > 3 * cstride ? 1 : 0);   // uniform
                        unsigned sum[2][4][4];
#pragma unroll
                        for (int z = 0; z < 2; z++)
#pragma unroll
                            for (int c = 0; c < 4; c++) { sum[z][c][0] = 0; sum[z][c][1] = 0; sum[z][c][2] = 0; sum[z][c][3] = 0; }
                        // One step = 4 source rows of both bins x NCH chunks, all loads in flight together.  (More rows per step for
                        // narrow segments -- 16 x 1 chunk, 8 x 2 -- was measured and is slower: dead rows still cost their loads.)
                        // SAFE: no byte any lane of this pass can touch lies past the frame buffer (true except for boxes at the very
                        // end of the last frame), so a load is a scalar row base + the lane's constant offset: no clamp, no VALU.
                        auto step = [&](auto NCH_T, auto PHASE_T, auto SAFE_T, int yy) {
                            constexpr int NCH = decltype(NCH_T)::value;
                            constexpr int RS = 4;
                            constexpr bool PH = decltype(PHASE_T)::value, SAFE = decltype(SAFE_T)::value;
                            unsigned lo[2][RS][NCH];
                            unsigned shr[2][RS];
#pragma unroll
                            for (int z = 0; z < 2; z++)
#pragma unroll
                                for (int r = 0; r < RS; r++) {
                                    const int kh_z = z ? khB : khA;
                                    const int rr = (yy + r < kh_z) ? yy + r : 0;                          // dead rows re-read row 0 (weight 0)
                                    const long long o = (z ? o_segB : o_segA) + (long long)rr * row_pitch - sh0 + c0;   // scalar
                                    shr[z][r] = (unsigned)(o & 3);
                                    const char* rowp = reinterpret_cast<const char*>(frames) + (PH ? o : (o & ~3ll));
                                    if (SAFE) {
#pragma unroll
                                        for (int c = 0; c < NCH; c++)
                                            lo[z][r][c] = *reinterpret_cast<const uint32_t*>(rowp + (PH ? 256 : 252) * c + 4 * lane);
                                    } else {
                                        const long long room = last_dw - (o >> 2);                       // >= 0: byte o is a frame byte
                                        const unsigned lim = room > 0x3fffffll ? 0xfffffcu : (unsigned)room * 4u;
#pragma unroll
                                        for (int c = 0; c < NCH; c++) {
                                            const unsigned ob = (PH ? 256u : 252u) * c + 4u * lane;
                                            lo[z][r][c] = *reinterpret_cast<const uint32_t*>(rowp + (ob < lim ? ob : lim));
                                        }
                                    }
                                }
#pragma unroll
                            for (int z = 0; z < 2; z++)
#pragma unroll
                                for (int r = 0; r < RS; r++) {
                                    const unsigned w0 = (yy + r < (z ? khB : khA)) ? 1u : 0u;        // scalar: 0 for a row this bin lacks
#pragma unroll
                                    for (int c = 0; c < NCH; c++) {
                                        unsigned v = lo[z][r][c];
                                        if (!PH) {
                                            const unsigned h = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xF, 0xF, false);   // wave_shl:1
                                            v = __builtin_amdgcn_alignbyte(h, v, shr[z][r]);
                                        }
                                        sum[z][c][0] = __builtin_amdgcn_udot4(v, w0, sum[z][c][0], false);
                                        sum[z][c][1] = __builtin_amdgcn_udot4(v, w0 << 8, sum[z][c][1], false);
                                        sum[z][c][2] = __builtin_amdgcn_udot4(v, w0 << 16, sum[z][c][2], false);
                                        sum[z][c][3] = __builtin_amdgcn_udot4(v, w0 << 24, sum[z][c][3], false);
                                    }
                                }
                        };
                        auto rows = [&](auto NCH_T, auto SAFE_T) {
                            constexpr int RS = 4;
                            if (same_phase) { for (int yy = 0; yy < khm; yy += RS) step(NCH_T, std::true_type{}, SAFE_T, yy); }
                            else { for (int yy = 0; yy < khm; yy += RS) step(NCH_T, std::false_type{}, SAFE_T, yy); }
                        };
                        // furthest byte of the pass: bin B's last row (oyB > oyA and bin edges are monotone, so yeB >= yeA), last chunk, lane 63
                        const long long o_far = o_segB + (long long)(khB - 1) * row_pitch - sh0 + c0 + (long long)(nch - 1) * cstride + 255;
                        if ((o_far >> 2) > last_dw) rows(std::integral_constant<int, 4>{}, std::false_type{});   // clamped loads
                        else if (nch == 1) rows(std::integral_constant<int, 1>{}, std::true_type{});
                        else if (nch == 2) rows(std::integral_constant<int, 2>{}, std::true_type{});
                        else if (nch == 3) rows(std::integral_constant<int, 3>{}, std::true_type{});
                        else rows(std::integral_constant<int, 4>{}, std::true_type{});
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if (c < nch) {
                                const int b = c0 + cstride * c + 4 * lane - sh0;     // segment byte of this lane's byte 0
                                if (same_phase || lane < 63) {
#pragma unroll
                                    for (int j = 0; j < 4; j++) {
                                        if (b + j >= 0 && b + j < seg_bytes) { colA[b + j] = sum[0][c][j]; colB[b + j] = sum[1][c][j]; }
                                    }
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    // bin means: the exhaustively verified reciprocal division (two fmas per division, pyr_div in trl_pnet.hip's terms;
                    // oracle/trl_oracle.c orc_selftest_recip_div: equal to a / kh / kw for every bin <= 96 x 96 and every byte sum)
                    const float fkhA = (float)khA, fkhB = (float)khB;
                    const float rkhA = 1.0f / fkhA, rkhB = 1.0f / fkhB;
                    for (int idx = lane; idx < (oxb - oxa) * 3; idx += 64) {
                        const int ox = oxa + idx / 3, c = idx % 3;
                        const int xs = (ox * iw) / S, xe = ((ox + 1) * iw + S - 1) / S;
                        unsigned accA = 0, accB = 0;
                        for (int xx = xs; xx < xe; xx++) { accA += colA[(xx - xsa) * 3 + c]; accB += colB[(xx - xsa) * 3 + c]; }
                        const int kw = xe - xs;
                        const float fkw = (float)kw;
                        float qA, qB;
                        if (fastdiv && (kw == kwA0 || kw == kwA0 + 1)) {
                            const float rkw = kw == kwA0 ? rkw0 : rkw1;
                            qA = rdiv(rdiv((float)accA, fkhA, rkhA), fkw, rkw);
                            qB = rdiv(rdiv((float)accB, fkhB, rkhB), fkw, rkw);
                        } else {
                            qA = (float)accA / fkhA / fkw;
                            qB = (float)accB / fkhB / fkw;
                        }
                        in_s[(oyA * S + ox) * 3 + c] = (qA - 127.5f) * 0.0078125f;
                        in_s[(oyB * S + ox) * 3 + c] = (qB - 127.5f) * 0.0078125f;
                    }
                    __builtin_amdgcn_wave_barrier();
                    oxa = oxb;
                }
            }
        }
        for (int p = IN_N + tid; p < IN_N + IN_PAD; p += 256) in_s[p] = 0.f;   // read by the zero-weight k = 27 pad
    }
    // ---- conv1 weights: B operands of both 16-channel N-tiles, in registers ----------------------------------
    float B0[7], B1[7];
    int koff[7];
#pragma unroll
    for (int s = 0; s < 7; s++) {
        const int k = 4 * s + kq;
        B0[s] = w1[k * 32 + l15];
        B1[s] = w1[k * 32 + 16 + l15];
        koff[s] = k + (S * 3 - 9) * (k / 9);
    }
    const float bias0 = b1[l15], bias1 = b1[16 + l15];
    __syncthreads();

    float* dst = out + (size_t)blockIdx.x * P * P * C1;
    float* const erow = c1_s + (kq * 4) * CLD + l15;             // epilogue lane base: row 4*kq (+q), channel l15
    if (dbg_skip & 2) return;
    constexpr int C4 = C1 / 4;
    // the pooled value's PReLU (slope classes: see MODE) and the store, shared by both strip schemes
    auto finish_pool = [&](float4 best, float4 low, int c4, int pr_abs, int px) __attribute__((always_inline)) {
        const float4 sl = *reinterpret_cast<const float4*>(s1 + 4 * c4);
        if (MODE == 2) {
            best.x = vmax_nc(best.x, sl.x * best.x); best.y = vmax_nc(best.y, sl.y * best.y);
            best.z = vmax_nc(best.z, sl.z * best.z); best.w = vmax_nc(best.w, sl.w * best.w);
        } else if (MODE == 1) {
            best.x = trl_prelu_med3(best.x, sl.x, trl_prelu_sel(sl.x)); best.y = trl_prelu_med3(best.y, sl.y, trl_prelu_sel(sl.y));
            best.z = trl_prelu_med3(best.z, sl.z, trl_prelu_sel(sl.z)); best.w = trl_prelu_med3(best.w, sl.w, trl_prelu_sel(sl.w));
        } else {
            best.x = trl_prelu_pooled(best.x, low.x, sl.x, trl_prelu_sel(sl.x)); best.y = trl_prelu_pooled(best.y, low.y, sl.y, trl_prelu_sel(sl.y));
            best.z = trl_prelu_pooled(best.z, low.z, sl.z, trl_prelu_sel(sl.z)); best.w = trl_prelu_pooled(best.w, low.w, sl.w, trl_prelu_sel(sl.w));
        }
        *reinterpret_cast<float4*>(dst + ((size_t)pr_abs * P + px) * C1 + 4 * c4) = best;
    };
    if constexpr (R == 1 && C1 == 32) {
        // ONE pooled row per strip (the O-Net front: three workgroups per CU) WITHOUT recomputing the conv row two consecutive pool
        // windows share: conv row r lives in slot r % 3 of a three-row ring, a strip computes the two rows its window adds (three for
        // the first strip) instead of three -- a third of the conv work of the plain strip scheme -- and every M-tile lies inside
        // one conv row (3 tiles of 16 pixels, the row padded to 48), so the 6 tiles x 2 channel halves split evenly: each wave runs
        // three independent chains of ONE half (its B operand: 7 registers), 21 matrix instructions per strip where the plain scheme's
        // busiest wave issued 42.
        constexpr int RW = 48;                                   // padded conv row (3 M-tiles)
        static_assert(3 * RW * CLD <= MROWS * CLD, "ring fits the strip buffer");
        const int half = wave & 1, tsel = wave >> 1;             // this wave: channel half, tiles tsel, tsel + 2, ...
        float Bh[7];
#pragma unroll
        for (int s = 0; s < 7; s++) Bh[s] = half ? B1[s] : B0[s];
        const float biash = half ? bias1 : bias0;
        for (int p0 = 0; p0 < P; p0++) {
            const int r_first = p0 == 0 ? 0 : 2 * p0 + 1;
            const int r_last = 2 * p0 + 2 < CW ? 2 * p0 + 2 : CW - 1;
            const int ntl = (r_last - r_first + 1) * 3;          // tiles of this strip: tile t -> conv row r_first + t / 3, pixels 16 (t % 3) ..
            for (int t0 = tsel; t0 < ntl; t0 += 6) {             // up to three tiles of this wave at a time: t0, t0 + 2, t0 + 4
                f32x4 acc[3];
                float xa[3][7];
                int srow[3], x0[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int t = t0 + 2 * u < ntl ? t0 + 2 * u : t0;          // (a missing tile repeats the first: computed, not stored)
                    const int row = r_first + t / 3;
                    x0[u] = 16 * (t - 3 * (t / 3));
                    srow[u] = row - 3 * (row / 3);
                    int x = x0[u] + l15; x = x < CW ? x : CW - 1;
                    const int base = (row * S + x) * 3;
#pragma unroll
                    for (int s = 0; s < 7; s++) xa[u][s] = in_s[base + koff[s]];
                    acc[u] = f32x4{biash, biash, biash, biash};
                }
#pragma unroll
                for (int s = 0; s < 7; s++)
#pragma unroll
                    for (int u = 0; u < 3; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][s], Bh[s], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    if (t0 + 2 * u < ntl) {
                        float* e = c1_s + (srow[u] * RW + x0[u] + 4 * kq) * CLD + 16 * half + l15;     // pixels past CW land in the row's padding
#pragma unroll
                        for (int q = 0; q < 4; q++) e[q * CLD] = acc[u][q];
                    }
                }
            }
            __syncthreads();
            for (int idx = tid; idx < P * C4; idx += 256) {
                const int c4 = idx % C4, px = idx / C4;
                const int rA = 2 * p0, rB = rA + 1, rC = rA + 2;            // the window's conv rows (rC may lie below the map)
                const float* sA = c1_s + ((rA - 3 * (rA / 3)) * RW + 2 * px) * CLD + 4 * c4;
                const float* sB = c1_s + ((rB - 3 * (rB / 3)) * RW + 2 * px) * CLD + 4 * c4;
                const float* sC = c1_s + ((rC - 3 * (rC / 3)) * RW + 2 * px) * CLD + 4 * c4;
                float4 best = *reinterpret_cast<const float4*>(sA), low = best;
                const bool x2 = 2 * px + 2 < CW;
                const int nrw = rC < CW ? 3 : (rB < CW ? 2 : 1);
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    if (dy >= nrw) break;
                    const float* sr = dy == 0 ? sA : (dy == 1 ? sB : sC);
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        if (dy == 0 && dx == 0) continue;
                        if (dx == 2 && !x2) continue;
                        const float4 v = *reinterpret_cast<const float4*>(sr + dx * CLD);
                        best.x = vmax_nc(best.x, v.x); best.y = vmax_nc(best.y, v.y); best.z = vmax_nc(best.z, v.z); best.w = vmax_nc(best.w, v.w);
                        if (MODE == 0) { low.x = vmin_nc(low.x, v.x); low.y = vmin_nc(low.y, v.y); low.z = vmin_nc(low.z, v.z); low.w = vmin_nc(low.w, v.w); }
                    }
                }
                finish_pool(best, low, c4, p0, px);
            }
            __syncthreads();
        }
        return;
    }
    for (int p0 = 0; p0 < P; p0 += R) {
        const int rows0 = 2 * p0;
        const int nrows = (CW - rows0) < SR ? (CW - rows0) : SR;
        const int M = nrows * CW;
        const int ntiles = (M + 15) >> 4;
        // two M-tiles x two N-tiles = four independent accumulator chains per wave (the M-tile index is scalar)
        for (int mt = wave; mt < ntiles; mt += 8) {
            const bool hasB = mt + 4 < ntiles;
            int mA = mt * 16 + l15; mA = mA < M ? mA : M - 1;
            const int yA = mA / CW, xA = mA - yA * CW;
            const int baseA = ((rows0 + yA) * S + xA) * 3;
            f32x4 a0 = {bias0, bias0, bias0, bias0}, a1 = {bias1, bias1, bias1, bias1}, c0 = a0, c1 = a1;
            if (hasB) {
                int mB = (mt + 4) * 16 + l15; mB = mB < M ? mB : M - 1;
                const int yB = mB / CW, xB = mB - yB * CW;
                const int baseB = ((rows0 + yB) * S + xB) * 3;
                float xa[7], xb[7];
#pragma unroll
                for (int s = 0; s < 7; s++) { xa[s] = in_s[baseA + koff[s]]; xb[s] = in_s[baseB + koff[s]]; }
#pragma unroll
                for (int s = 0; s < 7; s++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B0[s], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B1[s], a1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[s], B0[s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[s], B1[s], c1, 0, 0, 0);
                }
            } else {
                float xa[7];
#pragma unroll
                for (int s = 0; s < 7; s++) xa[s] = in_s[baseA + koff[s]];
#pragma unroll
                for (int s = 0; s < 7; s++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B0[s], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B1[s], a1, 0, 0, 0);
                }
            }
            // rows past M and channels past C1 fall into the strip's padding: nothing reads them
            float* ea = erow + mt * (16 * CLD);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ea[q * CLD] = a0[q];
                ea[q * CLD + 16] = a1[q];
            }
            if (hasB) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    ea[(64 + q) * CLD] = c0[q];
                    ea[(64 + q) * CLD + 16] = c1[q];
                }
            }
        }
        __syncthreads();
        // ---- MaxPool(3, 2, ceil_mode) of the strip -> global NHWC [P][P][C1], four channels per thread ------------
        const int prow = (P - p0) < R ? (P - p0) : R;
        for (int idx = tid; idx < prow * P * C4; idx += 256) {
            const int c4 = idx % C4;
            const int px = (idx / C4) % P;
            const int pr = idx / (C4 * P);
            const float* src = c1_s + ((2 * pr) * CW + 2 * px) * CLD + 4 * c4;
            float4 best = *reinterpret_cast<const float4*>(src);      // (dy, dx) = (0, 0) always exists
            float4 low = best;                                        // MODE 0 only: the window's min
            const bool x2 = 2 * px + 2 < CW;                          // ceil mode: the last window is clipped
            if (x2 && 2 * pr + 2 < nrows) {                           // whole 3x3 window: four three-operand max (min) per channel
                float4 v[8];
#pragma unroll
                for (int j = 1; j < 9; j++) v[j - 1] = *reinterpret_cast<const float4*>(src + ((j / 3) * CW + (j % 3)) * CLD);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    best.x = vmax3_nc(best.x, v[j].x, v[j + 1].x); best.y = vmax3_nc(best.y, v[j].y, v[j + 1].y);
                    best.z = vmax3_nc(best.z, v[j].z, v[j + 1].z); best.w = vmax3_nc(best.w, v[j].w, v[j + 1].w);
                    if (MODE == 0) {
                        low.x = vmin3_nc(low.x, v[j].x, v[j + 1].x); low.y = vmin3_nc(low.y, v[j].y, v[j + 1].y);
                        low.z = vmin3_nc(low.z, v[j].z, v[j + 1].z); low.w = vmin3_nc(low.w, v[j].w, v[j + 1].w);
                    }
                }
            } else {
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    if (2 * pr + dy >= nrows) break;
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        if (dy == 0 && dx == 0) continue;
                        if (dx == 2 && !x2) continue;
                        const float4 v = *reinterpret_cast<const float4*>(src + (dy * CW + dx) * CLD);
                        best.x = vmax_nc(best.x, v.x); best.y = vmax_nc(best.y, v.y); best.z = vmax_nc(best.z, v.z); best.w = vmax_nc(best.w, v.w);
                        if (MODE == 0) { low.x = vmin_nc(low.x, v.x); low.y = vmin_nc(low.y, v.y); low.z = vmin_nc(low.z, v.z); low.w = vmin_nc(low.w, v.w); }
                    }
                }
            }
            finish_pool(best, low, c4, p0 + pr, px);
        }
        __syncthreads();
    }
}

// conv1 PReLU slope class of a net (see MODE above)
int slope_mode(const DevV* sl, int n) {
    std::vector<float> h(n);
    if (hipMemcpy(h.data(), sl->p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    int mode = 2;
    for (float v : h) {
        if (!(v >= 0.f)) return 0;
        if (!(v <= 1.f)) mode = 1;
    }
    return mode;
}

}  // namespace

// timing-only ablation (TRL_FRONT_SKIP: 1/2 = R-Net crop / conv+pool, 4/8 = O-Net); 0 in production
static int front_dbg() { static const int v = trl_tune_int("TRL_FRONT_SKIP", 0); return v; }

// R-Net front: pooled [nc][11][11][28]
int trl_launch_rnet_front(trl_ctx* c, const uint8_t* d_frames, int H, int W, const int32_t* d_total, int t0, int nc,
                          float* d_pool, hipStream_t s) {
    if (nc <= 0) return TRL_OK;
    const DevW* w = trl_w(c, "rnet.conv1.w");
    const DevV *b = trl_v(c, "rnet.conv1.b"), *sl = trl_v(c, "rnet.prelu1");
    if (!w || !b || !sl || w->ld != 32 || w->K != 27) { trl_set_error("rnet.conv1 weights"); return TRL_ERR_WEIGHTS; }
    if (c->rnet_front_mode < 0) c->rnet_front_mode = slope_mode(sl, 28);
    static const int rr = trl_tune_int("TRL_RNET_R", 4);     // tuning aid: pooled rows per conv1 strip
#define TRL_RF(RR, MODE, DBG) k_mtcnn_front<24, 28, RR, MODE, DBG><<<nc, 256, 0, s>>>(d_frames, c->cb.n, H, W, reinterpret_cast<const int4*>(c->cb.cbox), \
                                                                       d_total, t0, w->p, b->p, sl->p, d_pool, front_dbg() & 3)
    if (front_dbg() & 3) { if (c->rnet_front_mode == 2) TRL_RF(4, 2, true); else if (c->rnet_front_mode == 1) TRL_RF(4, 1, true); else TRL_RF(4, 0, true); }
    else if (c->rnet_front_mode == 2) { if (rr == 2) TRL_RF(2, 2, false); else if (rr == 3) TRL_RF(3, 2, false); else if (rr == 6) TRL_RF(6, 2, false); else TRL_RF(4, 2, false); }
    else if (c->rnet_front_mode == 1) TRL_RF(4, 1, false); else TRL_RF(4, 0, false);
#undef TRL_RF
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
// O-Net front: pooled [nc][23][23][32]
int trl_launch_onet_front(trl_ctx* c, const uint8_t* d_frames, int H, int W, const int32_t* d_total, int t0, int nc,
                          float* d_pool, hipStream_t s) {
    if (nc <= 0) return TRL_OK;
    const DevW* w = trl_w(c, "onet.conv1.w");
    const DevV *b = trl_v(c, "onet.conv1.b"), *sl = trl_v(c, "onet.prelu1");
    if (!w || !b || !sl || w->ld != 32 || w->K != 27) { trl_set_error("onet.conv1 weights"); return TRL_ERR_WEIGHTS; }
    if (c->onet_front_mode < 0) c->onet_front_mode = slope_mode(sl, 32);
    static const int orr = trl_tune_int("TRL_ONET_R", 1);   // measured: one pooled row per strip = 49 KB of LDS = three resident
                                                                                   // workgroups per CU: 1.18 vs 1.27 ms (R = 3, two per CU) for the O-Net front
#define TRL_OF(RR, MODE, DBG) k_mtcnn_front<48, 32, RR, MODE, DBG><<<nc, 256, 0, s>>>(d_frames, c->cb.n, H, W, reinterpret_cast<const int4*>(c->cb.cbox), \
                                                                       d_total, t0, w->p, b->p, sl->p, d_pool, (front_dbg() >> 2) & 3)
    if ((front_dbg() >> 2) & 3) { if (c->onet_front_mode == 2) TRL_OF(1, 2, true); else if (c->onet_front_mode == 1) TRL_OF(1, 1, true); else TRL_OF(1, 0, true); }
    else if (c->onet_front_mode == 2) { if (orr == 3) TRL_OF(3, 2, false); else if (orr == 2) TRL_OF(2, 2, false); else if (orr == 4) TRL_OF(4, 2, false); else TRL_OF(1, 2, false); }
    else if (c->onet_front_mode == 1) TRL_OF(1, 1, false); else TRL_OF(1, 0, false);
#undef TRL_OF
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
