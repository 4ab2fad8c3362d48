// trl_pnet.hip -- MTCNN stage 1 (PNet over the image pyramid) for gfx950, the dominant kernel of the
// hot path (83 % of the conv FLOPs at 720p; server/model.py:47 -> detect_face stage 1).
//
// Per batch of frames: one k_pyramid launch per (level, Infinity-Cache-sized frame chunk), then ONE fused launch.
//
//  k_pyramid      u8 BGR frames -> every pyramid level, imresample (F.interpolate mode="area") +
//                 (x-127.5)*0.0078125, stored as three floats {b,g,r} per pixel (12 B, streamed past the caches).  One
//                 lane (fine levels) or lane group (coarse levels) per output pixel, dword-aligned 16/12-byte
//                 loads + v_dot4 byte sums (integer-exact, order independent), bin edges by exact multiply-high
//                 division, bin mean by the exhaustively verified reciprocal division (pyr_div).
//
//  k_pnet_fused   ONE persistent launch over all (frame, level, 16x16-cell tile) work items.  Per tile,
//                 entirely in LDS / registers:
//                   42x42x3 input tile -> conv1 3x3 (3->10) + 2x2 ceil max-pool + PReLU (the pool is an
//                   elementwise max over accumulators, or two DPP steps) -> conv2 3x3 (10->16) + PReLU
//                   -> conv3 3x3 (16->32) + PReLU -> 1x1 heads (32->2+4) -> softmax -> thr0 ->
//                   generateBoundingBox record appended to the (frame, level) candidate list.
//                 All four layers run on the f32 matrix cores -- the layers with few output channels (conv1: 10, heads: 6)
//                 on v_mfma_f32_4x4x1_16B_f32 with the weight block broadcast (no padding of N to 16), conv2 on
//                 v_mfma_f32_16x16x4_f32, conv3 on v_mfma_f32_32x32x2_f32 -- k ascending, accumulator seeded with the
//                 bias: the same fmaf chain as the oracle, so maps and candidates are bit-identical.
//                 conv1 / conv2 / head weights live in registers for the whole launch (6+23+4 VGPRs per lane),
//                 conv3's in LDS; activations never leave the CU: HBM traffic is the pyramid read only.  f32 MFMA and VALU share the FP32 pipe, so the loops carry almost no VALU: the
//                 tile decode is scalar (multiply-high by host magic numbers), LDS addresses are lane bases +
//                 compile-time offsets, pooling precedes PReLU when the slopes allow, edge logic only on edge tiles.
//                 blockIdx -> tile mapping keeps an XCD on a contiguous run of tiles (halo rows of
//                 neighbouring tiles hit the same L2).
#include "trl_ctx.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at dword alignment
struct __attribute__((packed, aligned(4))) u32x3_a4 { unsigned x, y, z; };

namespace {

constexpr int TS = 16;            // output cells per tile side
constexpr int IN_T = 2 * TS + 10; // 42 input pixels
constexpr int P1_T = TS + 4;      // 20 pooled cells
constexpr int C2_T = TS + 2;      // 18 conv2 cells
constexpr int C2_LD = 17;         // padded channel stride of the conv2 tile (bank spread for conv3 reads)
constexpr int REGION_A = C2_T * C2_T * C2_LD;        // 5508 floats: input tile, later conv2 output
constexpr int REGION_B = P1_T * P1_T * 10 + 224;     // 4000 floats of pooled conv1 + the reach of conv2's zero-weight k padding (k = 90, 91 of the last cells: finite, zeroed once)
static_assert(3 * 7 * 256 <= REGION_A, "input tile (+ the 28 overhang pixels of the 7x256 copy) fits region A");
static_assert(P1_T * P1_T * 10 <= REGION_B, "pooled tile fits region B");
// Horizontal carry between consecutive tiles of a tile row (a workgroup takes RUNS of consecutive tiles): the right-most 4 pooled
// columns and 2 conv2 columns of tile (ty, tx) ARE the left-most ones of tile (ty, tx + 1) -- the halo that 16x16-cell tiles
// otherwise compute twice (conv1 x1.56, conv2 x1.27 of the useful work).  They are kept in LDS across the tile boundary, and a tile
// that follows its left neighbour computes only 16 new pooled columns (80 instead of 100 conv1 M-tiles) and 16 new conv2 columns
// (18 instead of 21 M-tiles).  Same values: every cell is the same fmaf chain over the same pixels whichever tile computes it.
constexpr int CARRY_P = P1_T * 4 * 10;               // 800 floats: pooled rows x 4 columns x 10 channels
constexpr int CARRY_C = C2_T * 2 * C2_LD;            // 612 floats: conv2 rows x 2 columns x 17 (padded channels)
// Vertical carry (round 3): the tiles of a level are walked in BANDS of BAND tile rows, column by column (top tile, the tiles below
// it, then the next column), so a lower tile follows its upper neighbour in the same workgroup: ITS first 4 pooled rows and
// 2 conv2 rows are the upper tile's last ones.  A tile with both carries computes 16 x 16 new pooled cells (one super unit per
// wave and nothing else) and 16 x 16 new conv2 cells (16 M-tiles, four per wave).  The horizontal strips need one slot per band row:
// three rows (two of three tiles carry vertically) fill the 80 KB a workgroup may use at two workgroups per CU; two rows were 1.5 %
// slower, four do not fit.
constexpr int BAND = 3;
constexpr int VCARRY_P = 4 * P1_T * 10;              // 800 floats: 4 pooled rows x 20 columns x 10 channels (contiguous in the tile)
constexpr int VCARRY_C = 2 * C2_T * C2_LD;           // 612 floats: 2 conv2 rows x 18 columns x 17
constexpr int DYN_LDS = (144 * 32 + BAND * (CARRY_P + CARRY_C) + VCARRY_P + VCARRY_C) * 4;   // conv3 weights + the carry strips (dynamic: static LDS is capped at 64 KB)
static_assert((144 * 32) % 4 == 0 && CARRY_P % 4 == 0 && CARRY_C % 4 == 0 && VCARRY_P % 4 == 0, "strips stay 16-byte aligned");

// One pyramid pixel = three floats (a fourth padding float would be 25 % of the pyramid's write + read traffic).
struct PyrPx { float b, g, r; };
typedef float f32x3_nt __attribute__((ext_vector_type(3), aligned(4)));

struct PLevel {
    int h, w, oh, ow;        // level size, PNet map size
    int tiles_x, tile0;      // tiles per row, first tile index of the level inside a frame
    int tiles_y;             // tile rows
    unsigned bmagic;         // ceil(2^32 / (BAND * tiles_x)): band of a tile index
    int pix0;                // pixel offset of the level inside a frame's pyramid
    int pix_pad;             // h*w rounded up to 64 (pyramid slots of the level)
    int gshift, work0;       // pyramid kernel: log2(lanes per pixel), first thread of the level inside a frame
    int ytab0, xtab0;        // offsets of the level's row / column bin-edge tables
    int mode, nd, grshift;   // pyramid kernel path, re-aligned dwords per row (mode 0), log2 groups per row (mode 1)
    unsigned wmagic;         // ceil(2^32 / w): pixel / w by __umulhi
    unsigned txmagic;        // ceil(2^32 / tiles_x): tile decode stays on the scalar unit
    int khA, kwA, khmax, kwmax;   // adaptive-pool bins of the level are khA or khA+1 rows (kwA / kwA+1 columns)
    float rkh[2], rkw[2];    // RN(1/khA), RN(1/(khA+1)), same for kw: reciprocal division (see pyr_div)
    int fastdiv;             // bin sizes small enough for the exhaustively verified reciprocal division
    unsigned vmA[4], vmB[4]; // mode 0: valid-byte masks of the 4 re-aligned dwords of a row for bins kwA / kwA+1 wide
    unsigned hmagic;         // ceil(2^32 / h); with wmagic: bin edges by multiply-high when 'arith' (no table load in the
    int arith;               // dependent-latency chain of a pixel): requires H*h*h < 2^32 and W*w*w < 2^32
    float scale;
    int cap, rec0;           // candidate records of the level: slots per frame, first slot inside a frame's block (LvLayout)
};
struct PnetArgs {
    const PyrPx* pyr; long long pyr_stride;    // pixels per frame
    int n_frames, L, tiles_per_frame, H, W;
    unsigned tpf_magic;                        // ceil(2^32 / tiles_per_frame)
    long long work_per_frame;                  // pyramid kernel threads per frame
    PLevel lv[16];
    const float *w1, *w2, *w3, *wh;            // [Kpad][32] zero padded
    const float *b1, *b2, *b3, *bh, *s1, *s2, *s3;
    float thr; int rec_stride;                 // record slots per frame (LvLayout::S)
    float dthr;                                // logit-difference prefilter: no cell with logit1 - logit0 < dthr can reach thr (-inf: off)
    int dbg_skip;                              // timing-only ablation mask (TRL_PNET_SKIP); read by the DBG instantiation only
    int32_t* lvl_cnt; Cand* lvl_rec; int32_t* flags;
    int32_t* xcd_next;                         // per-XCD dynamic tile cursor (8 counters, zeroed before the launch)
    int run;                                   // consecutive tiles a workgroup takes per cursor fetch (>= 1)
    unsigned long long* clk;                   // [0] = earliest workgroup start, [1] = latest workgroup end (device wall clock); DBG: [2..] phase clocks
    int prof;                                  // DBG instantiation: accumulate the per-phase wave clocks
};

// ---- pyramid -------------------------------------------------------------------------------------
// One LANE GROUP of G lanes per output pixel (G = 1, 4, 16 or 64 by level: coarse levels average
// thousands of source bytes per pixel, so their bins are split across lanes and summed with xor
// shuffles -- integer sums, exact in any order).  Bytes are fetched as aligned dwords; the three
// channel sums of a dword are three v_dot4_u32_u8 against 0/1 byte masks selected by the dword's
// phase (byte offset mod 3) inside the BGR span.
__device__ __forceinline__ void dword_sums(unsigned v, int rel, int nbytes, unsigned& s0, unsigned& s1, unsigned& s2) {
    // rel = byte offset of this dword relative to the first byte of the span (-3 .. nbytes-1)
    const int lo = rel < 0 ? -rel : 0;
    const int hi = (nbytes - rel) < 4 ? (nbytes - rel) : 4;
    const unsigned vm = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
    v &= vm;
    const int phase = (rel + 3) % 3;   // channel of byte 0 of the dword
    const unsigned m0 = phase == 0 ? 0x01000001u : (phase == 1 ? 0x00010000u : 0x00000100u);
    const unsigned m1 = phase == 0 ? 0x00000100u : (phase == 1 ? 0x01000001u : 0x00010000u);
    const unsigned m2 = phase == 0 ? 0x00010000u : (phase == 1 ? 0x00000100u : 0x01000001u);
    s0 = __builtin_amdgcn_udot4(v, m0, s0, false);
    s1 = __builtin_amdgcn_udot4(v, m1, s1, false);
    s2 = __builtin_amdgcn_udot4(v, m2, s2, false);
}

// Bin edges are precomputed on the host (one packed (start | end<<16) word per output row / column of
// every level): the kernel does no 64-bit or repeated integer division.  grid = (blocks, frames).
// Three per-level modes (wave-uniform):
//   0  small bins (<= 5 px wide): one lane per pixel; each source row is 4-5 aligned dwords re-aligned to
//      the bin's first byte with v_alignbyte, so the BGR byte->channel masks are compile-time constants;
//   1  big bins, row pitch a multiple of 4 bytes: a lane owns one 12-byte group (4 whole pixels, constant
//      channel phase) of the bin for every (rl-th) source row: 3 coalesced loads + 9 v_dot4 per 12 bytes;
//   2  generic fallback (odd row pitch): flattened (row, dword) walk with per-dword masks.
__device__ __forceinline__ unsigned chan_mask(int p, int c) {   // dword whose byte 0 has channel p: bytes of channel c
    const int d = (c - p + 3) % 3;                               // byte index of the first byte of channel c
    return d == 0 ? 0x01000001u : (d == 1 ? 0x00000100u : 0x00010000u);
}
__device__ __forceinline__ unsigned valid_bytes(int rel, int nbytes) {   // 0xFF for bytes b of the dword with 0 <= rel+b < nbytes
    const int lo = rel < 0 ? -rel : 0;
    int hi = nbytes - rel; hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
    if (lo >= hi) return 0u;
    return (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
}

// The pyramid is written once and read once, much later, by the PNet kernel: stream it past the caches so the source
// frame (re-read by every level) keeps its L2 / Infinity Cache lines.
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pyr_store(PyrPx* dst, const float4& v) {
    static_assert(sizeof(PyrPx) == 12, "layout");
    f32x3_nt t = {v.x, v.y, v.z};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x3_nt*>(dst));
}

// Correctly rounded a / b from r = RN(1/b) (Markstein): q0 = RN(a r), e = a - b q0 (exact in one fma), q = RN(q0 + e r).
// For a = integer sums up to 255 kh kw and the bin sizes used here (kh, kw <= 96) the result equals the IEEE
// quotient for EVERY input -- checked exhaustively by the oracle's self test (oracle/trl_oracle.c:orc_selftest_recip_div);
// larger bins take the true division.  3 VALU ops instead of the ~11 of v_div_scale/v_rcp/v_div_fmas/v_div_fixup.
__device__ __forceinline__ float pyr_div(float a, float b, float r) {
    const float q0 = a * r;
    const float e = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e, r, q0);
}
__device__ __forceinline__ float pyr_norm(unsigned s, int kh, int kw, const PLevel& g) {
    const float a = (float)s, fkh = (float)kh, fkw = (float)kw;
    float q;
    if (g.fastdiv) {
        const float r1 = kh == g.khA ? g.rkh[0] : g.rkh[1], r2 = kw == g.kwA ? g.rkw[0] : g.rkw[1];
        q = pyr_div(pyr_div(a, fkh, r1), fkw, r2);
    } else {
        q = a / fkh / fkw;
    }
    return (q - 127.5f) * 0.0078125f;
}

struct PyrArgs { int H, W, n_frames, f0; long long pyr_stride; PLevel g; };

// threads q = q0, q0 + qstep, ... of level g of frame f
template <int MODE, typename A>
__device__ __forceinline__ void pyr_level(const uint8_t* __restrict__ frames, const A& a, const PLevel& g, const uint32_t* __restrict__ tab,
                                          PyrPx* __restrict__ pyr, int f, int q0, int qstep) {
    const int per_frame = g.pix_pad << g.gshift;            // threads of this level per frame
    constexpr int mode = MODE;
    const uint32_t* base32 = reinterpret_cast<const uint32_t*>(frames);
    const long long fbase = (long long)f * a.H * a.W * 3;
    const long long last_dw = ((long long)a.n_frames * a.H * a.W * 3 - 1) >> 2;
    const int row_bytes = a.W * 3;
    for (int q = q0; q < per_frame; q += qstep) {
        const int w = g.w, gsh = g.gshift, G = 1 << gsh;
        const int pixel = q >> gsh, sub = q & (G - 1);
        const bool valid = pixel < g.h * w;
        unsigned s0 = 0, s1 = 0, s2 = 0;
        int kh = 1, kw = 1;
        if (valid) {
            int oy = (int)__umulhi((unsigned)pixel, g.wmagic);       // floor(pixel / w) or one above it (large levels): fix up
            oy -= (oy * w > pixel) ? 1 : 0;
            const int ox = pixel - oy * w;
            const uint32_t ty = tab[g.ytab0 + oy], tx = tab[g.xtab0 + ox];
            const int ys = ty & 0xFFFF, ye = ty >> 16, xs = tx & 0xFFFF, xe = tx >> 16;
            kh = ye - ys; kw = xe - xs;
            const int nbytes = kw * 3;
            const long long o0 = fbase + (long long)ys * row_bytes + xs * 3;   // first byte of the bin
            if (mode == 1) {
                const int grsh = g.grshift, grp = sub & ((1 << grsh) - 1), rlane = sub >> grsh, rl = G >> grsh;
                const int sh = (int)(o0 & 3);
                const int rel = -sh + 12 * grp;                      // offset of this lane's group relative to the bin's first byte
                if (rel < nbytes) {
                    const int ph = (3 - sh % 3) % 3;                 // channel of the byte at the aligned start (12*grp keeps it)
                    unsigned mk[3][3];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const unsigned vb = valid_bytes(rel + 4 * j, nbytes) & 0x01010101u;
#pragma unroll
                        for (int c = 0; c < 3; c++) mk[j][c] = chan_mask((ph + j) % 3, c) & vb;
                    }
                    long long dw = ((o0 - sh) >> 2) + 3 * grp + (long long)rlane * (row_bytes >> 2);
                    const long long dstep = (long long)rl * (row_bytes >> 2);
                    for (int y = rlane; y < kh; y += 4 * rl, dw += 4 * dstep) {
                        unsigned w3[4][3];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool act = y + r * rl < kh;
                            const long long d = act ? dw + r * dstep : dw;
                            if (d + 2 <= last_dw) {
                                const u32x3_a4 v3 = *reinterpret_cast<const u32x3_a4*>(base32 + d);
                                w3[r][0] = v3.x; w3[r][1] = v3.y; w3[r][2] = v3.z;
                            } else {
                                w3[r][0] = base32[d]; w3[r][1] = base32[d + 1 <= last_dw ? d + 1 : last_dw]; w3[r][2] = base32[last_dw];
                            }
                            if (!act) { w3[r][0] = 0u; w3[r][1] = 0u; w3[r][2] = 0u; }
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            s0 = __builtin_amdgcn_udot4(w3[r][0], mk[0][0], s0, false); s1 = __builtin_amdgcn_udot4(w3[r][0], mk[0][1], s1, false);
                            s2 = __builtin_amdgcn_udot4(w3[r][0], mk[0][2], s2, false);
                            s0 = __builtin_amdgcn_udot4(w3[r][1], mk[1][0], s0, false); s1 = __builtin_amdgcn_udot4(w3[r][1], mk[1][1], s1, false);
                            s2 = __builtin_amdgcn_udot4(w3[r][1], mk[1][2], s2, false);
                            s0 = __builtin_amdgcn_udot4(w3[r][2], mk[2][0], s0, false); s1 = __builtin_amdgcn_udot4(w3[r][2], mk[2][1], s1, false);
                            s2 = __builtin_amdgcn_udot4(w3[r][2], mk[2][2], s2, false);
                        }
                    }
                }
            } else {
                const int ndw = (nbytes + 6) >> 2;            // dwords per row for the worst alignment
                int row = 0, d = sub;
                while (d >= ndw) { d -= ndw; row++; }
                while (row < kh) {
                    const long long o = o0 + (long long)row * row_bytes;
                    const long long al = (o & ~3ll) + 4 * d;
                    const int rel = (int)(al - o);
                    if (rel < nbytes) dword_sums(base32[al >> 2], rel, nbytes, s0, s1, s2);
                    d += G;
                    while (d >= ndw) { d -= ndw; row++; }
                }
            }
        }
        for (int off = G >> 1; off >= 1; off >>= 1) {
            s0 += __shfl_xor((int)s0, off, 64); s1 += __shfl_xor((int)s1, off, 64); s2 += __shfl_xor((int)s2, off, 64);
        }
        if (sub == 0 && pixel < g.pix_pad) {
            float4 o4;
            o4.x = valid ? pyr_norm(s0, kh, kw, g) : 0.f;
            o4.y = valid ? pyr_norm(s1, kh, kw, g) : 0.f;
            o4.z = valid ? pyr_norm(s2, kh, kw, g) : 0.f;
            o4.w = 0.f;
            pyr_store(pyr + ((long long)f * a.pyr_stride + g.pix0 + pixel), o4);
        }
    }
}

// Mode 0 (bins <= 5 px wide and <= 5 rows: the three finest levels = 85 % of the output pixels): one lane per pixel.
// VALU-bound, so everything per-row is pared down: the frame base is a scalar, the lane offset 32-bit; rows are
// unrolled to the level's khmax (scalar) and only the last one can be dead; the byte masks of the two possible bin
// widths are picked, not computed.
// ROWS = the level's khmax (3..5), FOUR = bins reach 5 px (a 5th dword per row).  Two pixels per thread per pass: the row
// loads of both are issued before either is consumed -- the kernel is bound by memory latency at 8 waves per SIMD, so
// loads in flight per wave are what counts (specialising on ROWS / FOUR keeps it under 64 VGPRs).
template <int ROWS, bool FOUR, typename A>
__device__ __forceinline__ void pyr_level0(const uint8_t* __restrict__ frames, const A& a, const PLevel& g, const uint32_t* __restrict__ tab,
                                           PyrPx* __restrict__ pyr, int f, int q0, int qstep) {
    constexpr int ND = FOUR ? 5 : 4;                                                 // dwords fetched per row
    const long long fbase = (long long)f * a.H * a.W * 3;
    const long long total = (long long)a.n_frames * a.H * a.W * 3;
    const int fb3 = (int)(fbase & 3);
    const char* fptr = reinterpret_cast<const char*>(frames) + (fbase - fb3);     // dword aligned, scalar
    const int row_bytes = a.W * 3;
    const bool lastf = f == a.n_frames - 1;                                          // other frames may read into their successor
    const unsigned avail = ((unsigned)fb3 + (unsigned)a.H * row_bytes + 3u) & ~3u;  // bytes from fptr to the end of the last dword
    struct Px { unsigned ww[ROWS][ND]; unsigned shv[ROWS]; int kh, kw; bool valid; };
    auto prep = [&](int pixel, Px& p) __attribute__((always_inline)) {
        p.valid = pixel < g.h * g.w;
        p.kh = 1; p.kw = 1;
        if (!p.valid) return;
        int oy = (int)__umulhi((unsigned)pixel, g.wmagic);       // floor(pixel / w) or one above it (large levels): fix up
        oy -= (oy * g.w > pixel) ? 1 : 0;
        const int ox = pixel - oy * g.w;
        int ys, kh, xs, kw;
        if (g.arith) {   // adaptive_avg_pool2d edges [floor(i*in/out), ceil((i+1)*in/out)) without touching memory
            ys = (int)__umulhi((unsigned)(oy * a.H), g.hmagic);
            kh = (int)__umulhi((unsigned)((oy + 1) * a.H + g.h - 1), g.hmagic) - ys;
            xs = (int)__umulhi((unsigned)(ox * a.W), g.wmagic);
            kw = (int)__umulhi((unsigned)((ox + 1) * a.W + g.w - 1), g.wmagic) - xs;
        } else {
            const uint32_t ty = tab[g.ytab0 + oy], tx = tab[g.xtab0 + ox];
            ys = ty & 0xFFFF; kh = (int)(ty >> 16) - ys; xs = tx & 0xFFFF; kw = (int)(tx >> 16) - xs;
        }
        p.kh = kh; p.kw = kw;
        const unsigned lo0 = (unsigned)(ys * row_bytes + xs * 3 + fb3);        // byte offset from fptr of the bin's first byte
        // the aligned 16/20-byte fetch of the LAST row may run past the end of the frame buffer only for the very
        // last pixels of the last frame: those take per-dword clamped loads
        const bool safe = !lastf || (lo0 & ~3u) + (unsigned)((kh - 1) * row_bytes) + 4u * ND <= avail;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const unsigned lo = lo0 + (unsigned)((r < kh ? r : kh - 1) * row_bytes);
            p.shv[r] = lo & 3u;
            const char* q = fptr + (lo & ~3u);
            if (safe) {
                const u32x4_a4 v4 = *reinterpret_cast<const u32x4_a4*>(q);
                p.ww[r][0] = v4[0]; p.ww[r][1] = v4[1]; p.ww[r][2] = v4[2]; p.ww[r][3] = v4[3];
                if (FOUR) p.ww[r][ND - 1] = *reinterpret_cast<const uint32_t*>(q + 16);
            } else {
                const long long lim = ((total - 1) >> 2) * 4 - (fbase - fb3);   // offset of the last dword holding frame bytes
#pragma unroll
                for (int j = 0; j < ND; j++) {
                    const long long o = (long long)(lo & ~3u) + 4 * j;
                    p.ww[r][j] = *reinterpret_cast<const uint32_t*>(fptr + (o < lim ? o : lim));
                }
            }
        }
    };
    auto finish = [&](int pixel, const Px& p) __attribute__((always_inline)) {
        if (pixel >= g.pix_pad) return;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.valid) {
            const bool wA = p.kw == g.kwA;                                      // a level has two bin widths: pick, don't compute
            const unsigned vm0 = wA ? g.vmA[0] : g.vmB[0], vm1 = wA ? g.vmA[1] : g.vmB[1], vm2 = wA ? g.vmA[2] : g.vmB[2],
                           vm3 = wA ? g.vmA[3] : g.vmB[3];
            unsigned s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const bool act = r < p.kh;                                       // only the last unrolled row can be dead
                const unsigned sh = p.shv[r];
                const unsigned d0 = __builtin_amdgcn_alignbyte(p.ww[r][1], p.ww[r][0], sh) & (act ? vm0 : 0u);
                const unsigned d1 = __builtin_amdgcn_alignbyte(p.ww[r][2], p.ww[r][1], sh) & (act ? vm1 : 0u);
                const unsigned d2 = __builtin_amdgcn_alignbyte(FOUR ? p.ww[r][3] : 0u, p.ww[r][2], sh) & (act ? vm2 : 0u);
                s0 = __builtin_amdgcn_udot4(d0, 0x01000001u, s0, false); s1 = __builtin_amdgcn_udot4(d0, 0x00000100u, s1, false);
                s2 = __builtin_amdgcn_udot4(d0, 0x00010000u, s2, false);
                s0 = __builtin_amdgcn_udot4(d1, 0x00010000u, s0, false); s1 = __builtin_amdgcn_udot4(d1, 0x01000001u, s1, false);
                s2 = __builtin_amdgcn_udot4(d1, 0x00000100u, s2, false);
                s0 = __builtin_amdgcn_udot4(d2, 0x00000100u, s0, false); s1 = __builtin_amdgcn_udot4(d2, 0x00010000u, s1, false);
                s2 = __builtin_amdgcn_udot4(d2, 0x01000001u, s2, false);
                if (FOUR) {
                    const unsigned d3 = __builtin_amdgcn_alignbyte(p.ww[r][4], p.ww[r][3], sh) & (act ? vm3 : 0u);
                    s0 = __builtin_amdgcn_udot4(d3, 0x01000001u, s0, false); s1 = __builtin_amdgcn_udot4(d3, 0x00000100u, s1, false);
                    s2 = __builtin_amdgcn_udot4(d3, 0x00010000u, s2, false);
                }
            }
            o4.x = pyr_norm(s0, p.kh, p.kw, g); o4.y = pyr_norm(s1, p.kh, p.kw, g); o4.z = pyr_norm(s2, p.kh, p.kw, g);
        }
        pyr_store(pyr + ((long long)f * a.pyr_stride + g.pix0 + pixel), o4);
    };
    for (int pixel = q0; pixel < g.pix_pad; pixel += 2 * qstep) {
        Px pa, pb;
        prep(pixel, pa);
        prep(pixel + qstep, pb);          // beyond pix_pad: invalid, nothing loaded, nothing stored
        finish(pixel, pa);
        finish(pixel + qstep, pb);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_pyramid(const uint8_t* __restrict__ frames, PyrArgs a, const uint32_t* __restrict__ tab,
                                                 PyrPx* __restrict__ pyr) {
    if (MODE == 0) pyr_level0<5, true>(frames, a, a.g, tab, pyr, a.f0 + blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    else pyr_level<MODE>(frames, a, a.g, tab, pyr, a.f0 + blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// mode 0, specialised on the level's row count / dword count (register budget = occupancy = loads in flight)
template <int ROWS, bool FOUR>
__global__ __launch_bounds__(256) void k_pyramid0(const uint8_t* __restrict__ frames, PyrArgs a, const uint32_t* __restrict__ tab,
                                                  PyrPx* __restrict__ pyr) {
    pyr_level0<ROWS, FOUR>(frames, a, a.g, tab, pyr, a.f0 + blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The three finest levels in ONE pass over the source (round 3).  A workgroup owns a source tile (a band of `band_cols` source
// columns x a strip of `strip_rows` source rows); wave L walks the tile's source rows for level L: its 64 lanes are the output
// columns of that level whose bins START in the band, its output rows those whose bins start in the strip (a bin may run past the
// tile: the wave simply reads on).  The three waves read the same source rows at about the same time, so the frame bytes come from
// HBM once and from the CU's L1 / the L2 for the other two levels -- the per-level kernels above re-read the whole frame chunk
// from the Infinity Cache for every level.  Inside a wave: a source row's three channel sums are formed once and added to the (at
// most two: H >= h) bins that contain it, column quantities are per-lane constants, row quantities wave-uniform (SALU); integer
// sums, the same division and normalisation (pyr_norm): bit-identical pixels.  D source rows in flight per lane.
struct PyrFineArgs {
    int H, W, n_frames, f0; long long pyr_stride;
    int nlev, band_cols, strip_rows, n_bands, n_strips;
    int own0;                      // offset in `tab` of the ownership table: [level][band] (ox_lo, ox_hi), then [level][strip] (oy_lo, oy_hi)
    PLevel g[3];
};
template <bool FOUR>
__device__ __forceinline__ void pyr_fine_wave(const uint8_t* __restrict__ frames, const PyrFineArgs& a, const PLevel& g, const uint32_t* __restrict__ tab,
                                              PyrPx* __restrict__ pyr, int f, int ox_lo, int ox_hi, int oy0, int oy1, int lane, bool pad_wave) {
    constexpr int ND = FOUR ? 5 : 4, D = 6;                                          // (4 and 8 rows in flight measured the same)
    PyrPx* const out = pyr + ((long long)f * a.pyr_stride + g.pix0);
    if (pad_wave && g.h * g.w + lane < g.pix_pad) pyr_store(out + (g.h * g.w + lane), make_float4(0.f, 0.f, 0.f, 0.f));   // the level's padding pixels: zeros, as ever
    if (oy0 >= oy1 || ox_lo >= ox_hi) return;                                        // wave-uniform
    const int ox = ox_lo + lane;
    const bool valid = ox < ox_hi;
    // ---- column quantities: constants of the lane for the whole strip ----
    const uint32_t tx = tab[g.xtab0 + (valid ? ox : ox_hi - 1)];
    const int xs = tx & 0xFFFF, kw = (int)(tx >> 16) - xs;
    const long long fbase = (long long)f * a.H * a.W * 3;
    const int fb3 = (int)(fbase & 3);
    const char* fptr = reinterpret_cast<const char*>(frames) + (fbase - fb3);      // dword aligned, scalar
    const unsigned row_bytes = (unsigned)a.W * 3u;
    const unsigned lx = (unsigned)(xs * 3 + fb3);                                   // the bin's first byte inside a source row, from fptr
    const bool wA = kw == g.kwA;
    const unsigned vm0 = wA ? g.vmA[0] : g.vmB[0], vm1 = wA ? g.vmA[1] : g.vmB[1], vm2 = wA ? g.vmA[2] : g.vmB[2], vm3 = wA ? g.vmA[3] : g.vmB[3];
    const bool lastf = f == a.n_frames - 1;
    const long long lim = ((((long long)a.n_frames * a.H * a.W * 3) - 1) >> 2) * 4 - (fbase - fb3);   // last dword holding frame bytes, from fptr
    // a source row: ND aligned dwords from the bin's first byte.  SLOW = a strip that reads the last row of the last frame, which
    // may not be read past the end of the buffer: clamped dwords (the pipelined loop stays single-path)
    auto fetch = [&](auto SLOW_T, int y, unsigned (&w)[ND]) __attribute__((always_inline)) {
        const unsigned lo = (unsigned)y * row_bytes + lx;
        const char* q = fptr + (lo & ~3u);
        if (!decltype(SLOW_T)::value) {
            const u32x4_a4 v4 = *reinterpret_cast<const u32x4_a4*>(q);
            w[0] = v4[0]; w[1] = v4[1]; w[2] = v4[2]; w[3] = v4[3];
            if (FOUR) w[ND - 1] = *reinterpret_cast<const uint32_t*>(q + 16);
        } else {
#pragma unroll
            for (int j = 0; j < ND; j++) {
                const long long o = (long long)(lo & ~3u) + 4 * j;
                w[j] = *reinterpret_cast<const uint32_t*>(fptr + (o < lim ? o : lim));
            }
        }
    };
    auto rowsum = [&](int y, const unsigned (&w)[ND], unsigned& r0, unsigned& r1, unsigned& r2) __attribute__((always_inline)) {
        const unsigned sh = ((unsigned)y * row_bytes + lx) & 3u;
        const unsigned d0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh) & vm0;
        const unsigned d1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh) & vm1;
        const unsigned d2 = __builtin_amdgcn_alignbyte(w[3], w[2], sh) & vm2;
        r0 = __builtin_amdgcn_udot4(d0, 0x01000001u, 0u, false); r1 = __builtin_amdgcn_udot4(d0, 0x00000100u, 0u, false);
        r2 = __builtin_amdgcn_udot4(d0, 0x00010000u, 0u, false);
        r0 = __builtin_amdgcn_udot4(d1, 0x00010000u, r0, false); r1 = __builtin_amdgcn_udot4(d1, 0x01000001u, r1, false);
        r2 = __builtin_amdgcn_udot4(d1, 0x00000100u, r2, false);
        r0 = __builtin_amdgcn_udot4(d2, 0x00000100u, r0, false); r1 = __builtin_amdgcn_udot4(d2, 0x00010000u, r1, false);
        r2 = __builtin_amdgcn_udot4(d2, 0x01000001u, r2, false);
        if (FOUR) {
            const unsigned d3 = __builtin_amdgcn_alignbyte(w[4], w[3], sh) & vm3;
            r0 = __builtin_amdgcn_udot4(d3, 0x01000001u, r0, false); r1 = __builtin_amdgcn_udot4(d3, 0x00000100u, r1, false);
            r2 = __builtin_amdgcn_udot4(d3, 0x00010000u, r2, false);
        }
    };
    // ---- row quantities: wave-uniform ----
    auto edges = [&](int oy, int& ys, int& ye) __attribute__((always_inline)) {     // bin [ys, ye) of output row oy; past the strip: never
        if (oy >= oy1) { ys = 0x7fffffff; ye = 0x7fffffff; return; }
        if (g.arith) {
            ys = (int)__umulhi((unsigned)(oy * a.H), g.hmagic);
            ye = (int)__umulhi((unsigned)((oy + 1) * a.H + g.h - 1), g.hmagic);
        } else {
            const uint32_t t = tab[g.ytab0 + oy];
            ys = (int)(t & 0xFFFF); ye = (int)(t >> 16);
        }
        ys = __builtin_amdgcn_readfirstlane(ys); ye = __builtin_amdgcn_readfirstlane(ye);
    };
    int ys_first, ye_first, ys_last, yend;
    edges(oy0, ys_first, ye_first);
    edges(oy1 - 1, ys_last, yend);                                                   // the wave's source rows: [ys_first, yend)
    auto run = [&](auto SLOW_T) __attribute__((always_inline)) {
        int oy = oy0, ys_c = ys_first, ye_c = ye_first, ys_n, ye_n;
        edges(oy + 1, ys_n, ye_n);
        unsigned c0 = 0, c1 = 0, c2 = 0, n0 = 0, n1 = 0, n2 = 0;                    // channel sums of the current bin and of the next one
        unsigned ring[D][ND];
        const int ycl = yend - 1;                                                    // rows past the wave's last bin are never touched
#pragma unroll
        for (int k = 0; k < D; k++) fetch(SLOW_T, ys_first + k < ycl ? ys_first + k : ycl, ring[k]);
        for (int yb = ys_first; yb < yend; yb += D) {
#pragma unroll
            for (int k = 0; k < D; k++) {
                const int y = yb + k;                                                // wave-uniform
                unsigned r0, r1, r2;
                rowsum(y, ring[k], r0, r1, r2);                                      // (rows at and past yend: a re-read row, sums unused)
                fetch(SLOW_T, y + D < ycl ? y + D : ycl, ring[k]);
                if (y < yend) {
                    c0 += r0; c1 += r1; c2 += r2;
                    if (y >= ys_n) { n0 += r0; n1 += r1; n2 += r2; }
                    if (y + 1 == ye_c) {                                           // the current bin is complete
                        if (valid) {
                            const int kh = ye_c - ys_c;
                            float4 o4;
                            o4.x = pyr_norm(c0, kh, kw, g); o4.y = pyr_norm(c1, kh, kw, g); o4.z = pyr_norm(c2, kh, kw, g); o4.w = 0.f;
                            pyr_store(out + (oy * g.w + ox), o4);
                        }
                        c0 = n0; c1 = n1; c2 = n2; n0 = 0; n1 = 0; n2 = 0;
                        oy++;
                        ys_c = ys_n; ye_c = ye_n;
                        edges(oy + 1, ys_n, ye_n);
                    }
                }
            }
        }
    };
    if (lastf && yend >= a.H) run(std::true_type{}); else run(std::false_type{});
}

__global__ __launch_bounds__(192) void k_pyramid_fine(const uint8_t* __restrict__ frames, PyrFineArgs a, const uint32_t* __restrict__ tab,
                                                      PyrPx* __restrict__ pyr) {
    const int lane = threadIdx.x & 63, lvl = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (lvl >= a.nlev) return;
    const int band = blockIdx.x, strip = blockIdx.z, f = a.f0 + blockIdx.y;
    const uint32_t* own = tab + a.own0;
    const uint32_t cb = own[lvl * a.n_bands + band], rb = own[3 * a.n_bands + lvl * a.n_strips + strip];
    const int ox_lo = __builtin_amdgcn_readfirstlane((int)(cb & 0xFFFF)), ox_hi = __builtin_amdgcn_readfirstlane((int)(cb >> 16));
    const int oy_lo = __builtin_amdgcn_readfirstlane((int)(rb & 0xFFFF)), oy_hi = __builtin_amdgcn_readfirstlane((int)(rb >> 16));
    const bool pad_wave = band == 0 && strip == 0;
    if (lvl == 0) { if (a.g[0].kwmax * 3 + 3 > 16) pyr_fine_wave<true>(frames, a, a.g[0], tab, pyr, f, ox_lo, ox_hi, oy_lo, oy_hi, lane, pad_wave); else pyr_fine_wave<false>(frames, a, a.g[0], tab, pyr, f, ox_lo, ox_hi, oy_lo, oy_hi, lane, pad_wave); }
    else if (lvl == 1) { if (a.g[1].kwmax * 3 + 3 > 16) pyr_fine_wave<true>(frames, a, a.g[1], tab, pyr, f, ox_lo, ox_hi, oy_lo, oy_hi, lane, pad_wave); else pyr_fine_wave<false>(frames, a, a.g[1], tab, pyr, f, ox_lo, ox_hi, oy_lo, oy_hi, lane, pad_wave); }
    else { if (a.g[2].kwmax * 3 + 3 > 16) pyr_fine_wave<true>(frames, a, a.g[2], tab, pyr, f, ox_lo, ox_hi, oy_lo, oy_hi, lane, pad_wave); else pyr_fine_wave<false>(frames, a, a.g[2], tab, pyr, f, ox_lo, ox_hi, oy_lo, oy_hi, lane, pad_wave); }
}

// ---- fused PNet --------------------------------------------------------------------------------------

// A-operand k offsets.  k = 4s+kq walks (tap, channel) of a [pixel][C] LDS tile whose rows are E floats
// further apart than D consecutive k: offset = k + E*(k/D).  With s a compile-time constant only the step
// whose four k straddle a multiple of D depends on the lane (kq >= thr), so no per-lane offset tables.
template <int D, int E>
__device__ __forceinline__ int koff(int s, int kq, int e1, int e2, int e3) {
    const int q0 = (4 * s) / D, r0 = (4 * s) % D, thr = D - r0;    // folded after unrolling
    const int add = thr == 1 ? e1 : (thr == 2 ? e2 : (thr == 3 ? e3 : 0));
    return 4 * s + E * q0 + kq + add;
}

// UNIT: no PReLU slope of the net exceeds 1 (negative slopes included); then prelu(v) == max(v, slope*v) exactly (trl_common.h).
// Weights with a slope above 1 (a trained checkpoint's are unconstrained) take the general instantiation, which costs the same
// two VALU instructions per value: med3(v, slope*v, +-inf) (trl_prelu_med3).  Both pool conv1 BEFORE its PReLU -- with the
// window's min next to its max when a conv1 slope is negative (the NEG1 instantiations, trl_prelu_pooled).
template <bool UNIT>
__device__ __forceinline__ float prelu_t(float v, float sl, float sel) { return UNIT ? vmax_nc(v, sl * v) : trl_prelu_med3(v, sl, sel); }
template <bool UNIT>
__device__ __forceinline__ float prelu_pooled_t(float m, float n, float sl, float sel) {
    return UNIT ? vmax_nc(m, sl * (sl < 0.f ? n : m)) : trl_prelu_pooled(m, n, sl, sel);
}

// ---- conv1 on v_mfma_f32_4x4x1_16B_f32 -----------------------------------------------------------------------------------------
// 16x16x4 tiles pad conv1's N = 10 to 16 (and K = 27 to 28): 40 % of their issue cycles multiply zeros.  The 4x4x1 block
// instruction computes sixteen independent 4x4 outer-product blocks, ONE k per instruction, and can broadcast block `abid` of the A
// register to all sixteen (cbsz = 4).  A UNIT = 16 pool cells x their 4 conv1 positions: block b = cell, lane-in-block j = position
// (dy, dx); the B operand is the lane's pixel value at k; the A operand is W[k][4 cg + i] for one of three channel groups, broadcast
// from block k & 15 of register k >> 4 -- the whole 27 x 12 weight matrix (+ the bias as a 28th "k" against a constant 1) lives in
// 6 VGPRs.  84 instructions x 8 cycles per 64 pixels instead of 4 x 7 x 32: -25 % issue cycles; the chain per output is still
// acc = bias, k ascending (fma(bias, 1, 0) == bias exactly).  The 2x2 pool window of a cell is the block's four lanes: two DPP max.
template <int CB, int AB>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CB, AB, 0); }
// max / min over the four lanes of each quad, four values at a time; one asm block per group so that no DPP instruction reads a
// register written by one of the two instructions in front of it (gfx9 DPP hazard) whatever the scheduler does around the block
__device__ __forceinline__ void quad_max4(const f32x4& v, float (&o)[4]) {
    float t0, t1, t2, t3;
    asm("v_max_f32_dpp %4, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %5, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %6, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %7, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
}
__device__ __forceinline__ void quad_min4(const f32x4& v, float (&o)[4]) {
    float t0, t1, t2, t3;
    asm("v_min_f32_dpp %4, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %5, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %6, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %7, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %1, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %2, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %3, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
}
// per-lane select on a wave-uniform 64-bit lane mask (one v_cndmask: hipcc turned the ?: on lane-index tests into exec-mask branches)
__device__ __forceinline__ float lane_sel(float if0, float if1, unsigned long long mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
    return r;
}
template <int... I, typename F>
__device__ __forceinline__ void pn_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// DBG: diagnostic instantiation -- records the launch's execution span on the device wall clock (TRL_PNET_CLOCK=1) and honours
// the timing-only phase ablation mask (TRL_PNET_SKIP, tools/pnet_phase_pmc.sh).  The production instantiation (DBG = false)
// carries neither: no instrumentation and no ablation tests on the hot path.
template <bool UNIT, bool NEG1, bool DBG>
__global__ __launch_bounds__(256, 2) void k_pnet_fused(PnetArgs a) {
    __shared__ __attribute__((aligned(16))) float RA[REGION_A];   // input tile [42][42][3]  ->  conv2 out [324][17]
    __shared__ __attribute__((aligned(16))) float RB[REGION_B];   // pooled [400][10] (+ the reach of conv2's zero-weight k padding)
    __shared__ __attribute__((aligned(16))) float T3all[72];      // conv3 bias[32], PReLU slopes[32] (a lane's 16 channels differ per register), head bias[8] (phase 3)
    extern __shared__ __attribute__((aligned(16))) float DYN[];  // DYN_LDS bytes, then whatever a tuning run pads (TRL_PNET_XLDS)
    float* const B3S = DYN;                                       // conv3 weights [k][cout]: read per k-chain batch, not held in VGPRs
    float* const CP0 = DYN + 144 * 32;                            // per band row: carried pooled columns [20][4][10], conv2 columns [18][2][17]
    float* const VP = CP0 + BAND * (CARRY_P + CARRY_C);           // carried pooled rows [4][20][10]
    float* const VC = VP + VCARRY_P;                              // carried conv2 rows  [2][18][17]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every M-tile index below is SALU work
    const int dbg_skip = DBG ? a.dbg_skip : 0;                   // a compile-time 0 in production
    if (DBG && tid == 0) atomicMin(&a.clk[0], (unsigned long long)wall_clock64());   // tuning build: execution span of the launch (first workgroup start .. last workgroup end)
    const int l15 = lane & 15, kq = lane >> 4;      // 16x16x4 operand coordinates
    const int l31 = lane & 31, hh = lane >> 5;      // 32x32x2 operand coordinates

    // ---- B operands: every weight matrix stays in registers for the whole launch -----------------------------
    for (int i = tid; i < 144 * 32; i += 256) B3S[i] = a.w3[i];          // conv3 B operand [k][32] in LDS (18 KB)
    float B2[23], WH[4];
#pragma unroll
    for (int s = 0; s < 23; s++) B2[s] = a.w2[(4 * s + kq) * 32 + l15];
#pragma unroll
    for (int r = 0; r < 4; r++) WH[r] = a.wh[(((lane >> 2) & 7) + 8 * r) * 32 + 4 * (lane >> 5) + (lane & 3)];   // heads: see phase 3
    // (vectors are zero padded to 128 floats)
    const float bias2 = a.b2[l15], slope2 = a.s2[l15];
    if (tid < 32) { T3all[tid] = a.b3[tid]; T3all[32 + tid] = a.s3[tid]; if (tid < 8) T3all[64 + tid] = a.bh[tid]; }   // (published by the barrier below)
    // general instantiation only: the per-channel med3 selector (+inf: max(v, s v), -inf: min(v, s v)); dead code when UNIT
    const float sel2 = trl_prelu_sel(slope2);
    const f32x4 bias2v = {bias2, bias2, bias2, bias2};

    // LDS beyond the live tiles is read by zero-weight k padding: it must hold finite values
    for (int i = tid; i < REGION_A; i += 256) RA[i] = 0.f;
    for (int i = tid; i < REGION_B; i += 256) RB[i] = 0.f;
    __syncthreads();

    const int e2_2 = kq >= 2 ? 170 : 0;                                                      // conv2: D = 30, E = 200-30
    // conv1 on 4x4x1 blocks: block xb = pool cell of the unit, xj = (dy, dx); weights + bias of channel group cg packed 16 k per register
    const int xb = lane >> 2, xj = lane & 3, xdy = xj >> 1, xdx = xj & 1;
    float WC[3][2];
#pragma unroll
    for (int cg = 0; cg < 3; cg++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = xb + 16 * h;
            WC[cg][h] = k < 27 ? a.w1[k * 32 + 4 * cg + xj] : (k == 27 ? a.b1[4 * cg + xj] : 0.f);
        }
    float xone = 1.0f;
    asm volatile("" : "+v"(xone));                        // a VGPR holding 1.0 (the bias "k"): not re-materialised per use
    const int xcg = xj < 3 ? xj : 2;                      // lane j < 3 finishes channel group j of its cell (lane 3 idles in the epilogue)
    const float sl4[4] = {a.s1[4 * xcg], a.s1[4 * xcg + 1], a.s1[4 * xcg + 2], a.s1[4 * xcg + 3]};
    // wide unit of pooled row wave + 4 i, columns 4..19: input lane base and store base (+ immediates per i)
    const int c1x_w = ((2 * wave + xdy) * IN_T + 8 + 2 * xb + xdx) * 3;
    const int c1x_sw = (wave * P1_T + 4 + xb) * 10 + 4 * xj;
    // SUPER unit: the 64 cells (row wave + 4 j, column 4 + xb), ONE conv1 position per pass: the pool window is then four
    // accumulators of the SAME lane (no cross-lane step), and every lane finishes its own cell
    const int c1s_l = ((2 * wave + 8 * xj) * IN_T + 8 + 2 * xb) * 3;
    const int c1s_s = ((wave + 4 * xj) * P1_T + 4 + xb) * 10;

    const int total_tiles = a.tiles_per_frame * a.n_frames;
    // XCD-aware persistent schedule: blocks sharing blockIdx%8 (one XCD) walk one contiguous 1/8 of the tiles
    const int xcd = blockIdx.x & 7;
    const int chunk = (total_tiles + 7) / 8;
    const int t_begin = xcd * chunk, t_end = (t_begin + chunk < total_tiles) ? t_begin + chunk : total_tiles;

    // tile -> (frame, level, ty, tx), all on the scalar unit: divisions are multiply-high by host magic numbers
    // (an integer division would expand into a VALU float-reciprocal sequence, paid in MFMA issue slots), the
    // level is a branch-free count over a first-tile table held in SGPRs.
    // the level of a tile index: lane i holds the first tile of level i (lanes past the last level: INT_MAX), so the level is one
    // compare and a ballot count -- sixteen thresholds in SGPRs pushed other uniform values into VGPRs and the whole decode onto the
    // vector unit (84 VALU per tile beside the other workgroup's MFMAs)
    const int lvl_t0v = lane < a.L ? a.lv[lane < 16 ? lane : 15].tile0 : 0x7fffffff;
    auto sdiv = [](int n, unsigned magic, int d) {
        int q = (int)__umulhi((unsigned)n, magic);
        if (q * d > n) q--;
        if (n - q * d >= d) q++;
        return q;
    };
    // A decoded tile carries the fields of its level that the loop reads: they are fetched HERE (scalar loads from the argument
    // block, issued while the previous tile is in phase 3), not at the top of the tile's own iteration -- left to itself hipcc
    // indexed a.lv[] with a VGPR and put a vector-memory round trip (~1,100 clocks per tile, measured with TRL_PNET_CLOCK) in
    // front of phase 0.
    struct TileId { int f, l, ty, tx, h, w, oh, ow, tiles_x, pix0, rib, rows; float scale; };   // rib: row inside the band, rows: tile rows of the band
    auto decode = [&](int tile) {
        TileId t;
        // (readfirstlane: at the SGPR limit hipcc parks uniform values in VGPRs, and everything computed from one runs on the vector
        // unit -- 84 VALU per tile for this decode; pinned to SGPRs here, it is scalar work again)
        const int tpf = __builtin_amdgcn_readfirstlane(a.tiles_per_frame);
        t.f = __builtin_amdgcn_readfirstlane(sdiv(tile, __builtin_amdgcn_readfirstlane(a.tpf_magic), tpf));
        const int tt = tile - t.f * tpf;
        t.l = __popcll(__ballot(tt >= lvl_t0v)) - 1;
        const PLevel& g = a.lv[t.l];
        t.h = g.h; t.w = g.w; t.oh = g.oh; t.ow = g.ow; t.tiles_x = g.tiles_x; t.pix0 = g.pix0; t.scale = g.scale;
        // band order: index inside the level -> (band, column, row inside the band); the last band of a level may be one row high
        const int tq = tt - g.tile0;
        const int band = sdiv(tq, g.bmagic, BAND * t.tiles_x);
        const int rb = tq - band * (BAND * t.tiles_x);
        t.rows = g.tiles_y - BAND * band < BAND ? g.tiles_y - BAND * band : BAND;
        static_assert(BAND == 2 || BAND == 3, "column of a band position: rb / rows for rows in 1..3");
        t.tx = __builtin_amdgcn_readfirstlane(t.rows == 3 ? (rb * 43691) >> 17 : (t.rows == 2 ? rb >> 1 : rb));   // (rb < 98304)
        t.rib = rb - t.tx * t.rows;
        t.ty = BAND * band + t.rib;
        return t;
    };
    // The next tile's 42x42 input pixels are fetched into registers while the current tile is in phase 3
    // (7 float4 per thread) and dropped into LDS at the top of the next iteration: the HBM/L2 latency of the
    // only global read of the kernel is off the critical path.  Thread t owns pixels p = t + 256 i; since
    // 256 = 6*42 + 4, (iy, ix) of pixel i follow from (iy0, ix0) with one conditional wrap, and the address is
    // a scalar tile base plus a 32-bit lane offset.  Interior tiles (the bulk) load without bounds tests.
    const int pin_iy0 = tid / IN_T, pin_ix0 = tid - pin_iy0 * IN_T;
    float4 pre[7];
    auto issue_input = [&](const TileId& t) {
        const TileId& g = t;
        const int gy0 = t.ty * 2 * TS, gx0 = t.tx * 2 * TS;
        const char* srcb = reinterpret_cast<const char*>(a.pyr + (long long)t.f * a.pyr_stride + g.pix0 + (long long)gy0 * g.w + gx0);
        const int w16 = g.w * 12;                  // bytes per pyramid row (12 B pixels)
        const int hrem = g.h - gy0, wrem = g.w - gx0;
        if (hrem >= IN_T && wrem >= IN_T) {
            const unsigned v0 = (unsigned)(pin_iy0 * w16 + pin_ix0 * 12);
#pragma unroll
            for (int i = 0; i < 7; i++) {
                unsigned vo = v0 + (unsigned)(i * (6 * w16 + 48)) + ((pin_ix0 >= IN_T - 4 * i) ? (unsigned)(w16 - IN_T * 12) : 0u);
                if (i == 6) vo = (tid < IN_T * IN_T - 6 * 256) ? vo : 0u;     // p >= 42*42: reload pixel 0 (lands in the RA tail)
                { const f32x3_nt v3 = *reinterpret_cast<const f32x3_nt*>(srcb + vo); pre[i] = make_float4(v3[0], v3[1], v3[2], 0.f); }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const bool wrap = pin_ix0 >= IN_T - 4 * i;
                const int iy = pin_iy0 + 6 * i + (wrap ? 1 : 0), ix = pin_ix0 + 4 * i - (wrap ? IN_T : 0);
                const bool ok = iy < hrem && ix < wrem && (i < 6 || tid < IN_T * IN_T - 6 * 256);
                pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) { const f32x3_nt v3 = *reinterpret_cast<const f32x3_nt*>(srcb + (unsigned)(iy * w16 + ix * 12)); pre[i] = make_float4(v3[0], v3[1], v3[2], 0.f); }
            }
        }
    };
    // Carry strips <-> tiles, a handful of instructions per tile (VALU beside the MFMAs is paid in matrix throughput): the pooled
    // strip is 20 rows x 40 floats = 10 float4 per row (16 threads per row, 10 active: rows 0..15 in one pass, 16..19 in a second);
    // the conv2 strip is 18 rows x 34 floats = 17 float2 per row (32 threads per row, 17 active: three passes).  (Splitting the
    // copies -- reads at the top of a phase, writes at its end -- hides their LDS round trips but holds ~20 registers across the
    // phases: the kernel then spills ~25 lane constants to scratch and loses 5 %: measured, reverted.)
    typedef float f32x2c __attribute__((ext_vector_type(2)));
    // (the copies index with `ctid`, the thread index laundered once per tile: their ~12 lane addresses are then recomputed, one VALU
    // each, instead of hoisted out of the tile loop and -- at 256 VGPRs -- spilled; a scratch reload waits on vmcnt(0), i.e. on
    // the next tile's input loads that are in flight for exactly that reason)
    int ctid = tid;
    auto copy_pooled = [&](auto SAVE_T, float* CP) {
        constexpr bool SAVE = decltype(SAVE_T)::value;          // tile -> strip (columns 16..19), or strip -> tile (columns 0..3)
        const int q = ctid & 15;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int r = (ctid >> 4) + 16 * pass;
            if (q < 10 && (pass == 0 || ctid < 64)) {
                f32x4* t = reinterpret_cast<f32x4*>(RB + r * (P1_T * 10) + (SAVE ? 16 * 10 : 0) + 4 * q);
                f32x4* c = reinterpret_cast<f32x4*>(CP + r * 40 + 4 * q);
                if (SAVE) *c = *t; else *t = *c;
            }
        }
    };
    auto copy_conv2 = [&](auto SAVE_T, float* CC) {
        constexpr bool SAVE = decltype(SAVE_T)::value;          // columns 16..17 -> strip, or strip -> columns 0..1
        const int q = ctid & 31;
#pragma unroll
        for (int pass = 0; pass < 3; pass++) {
            const int r = (ctid >> 5) + 8 * pass;
            if (q < 17 && (pass < 2 || ctid < 64)) {
                f32x2c* t = reinterpret_cast<f32x2c*>(RA + r * (C2_T * C2_LD) + (SAVE ? 16 * C2_LD : 0) + 2 * q);
                f32x2c* c = reinterpret_cast<f32x2c*>(CC + r * 34 + 2 * q);
                if (SAVE) *c = *t; else *t = *c;
            }
        }
    };
    // vertical strips: contiguous in both tiles (whole rows), one float4 per thread
    auto copy_rows = [&](auto SAVE_T) {                          // pooled rows 16..19 -> VP, or VP -> rows 0..3 (RB)
        constexpr bool SAVE = decltype(SAVE_T)::value;
        if (ctid < VCARRY_P / 4) {
            f32x4* t = reinterpret_cast<f32x4*>(RB + (SAVE ? 16 * P1_T * 10 : 0)) + ctid;
            f32x4* c = reinterpret_cast<f32x4*>(VP) + ctid;
            if (SAVE) *c = *t; else *t = *c;
        }
    };
    auto copy_rows2 = [&](auto SAVE_T) {                         // conv2 rows 16..17 -> VC, or VC -> rows 0..1 (RA)
        constexpr bool SAVE = decltype(SAVE_T)::value;
        static_assert((16 * C2_T * C2_LD) % 4 == 0 && VCARRY_C % 4 == 0, "float4 copies");
        if (ctid < VCARRY_C / 4) {
            f32x4* t = reinterpret_cast<f32x4*>(RA + (SAVE ? 16 * C2_T * C2_LD : 0)) + ctid;
            f32x4* c = reinterpret_cast<f32x4*>(VC) + ctid;
            if (SAVE) *c = *t; else *t = *c;
        }
    };
    // Dynamic schedule inside the XCD's chunk: a workgroup takes the next RUN of a.run consecutive tile indices (band order: see
    // decode) of its XCD from an atomic cursor, so one that starts late (another stream's kernel still on its CU) or loses time
    // simply processes fewer runs; inside a run the next tile is tile + 1 (the tile below, or the top of the next column: its
    // neighbours' carry strips are then in LDS); the cursor value for the tile after a run is fetched at the top of the run's
    // last tile and travels through LDS (the loop's own barriers order it).
    __shared__ int next_tile_s;
    int run_len = a.run;
    if (tid == 0) next_tile_s = t_begin + atomicAdd(&a.xcd_next[xcd], run_len);
    __syncthreads();
    int tile = __builtin_amdgcn_readfirstlane(next_tile_s);
    __syncthreads();
    TileId cur = decode(tile < t_end ? tile : 0), nxt = cur;
    if (tile < t_end) issue_input(cur);
    int rot = 0, tile_nxt = tile, run_pos = 0;
    // who left the strips: tile index of the last saver of each horizontal slot and of the vertical strips (a tile carries only
    // from exactly its neighbour's index, so a stale entry can never match)
    int hs_tile[BAND], vs_tile = -1;
#pragma unroll
    for (int i = 0; i < BAND; i++) hs_tile[i] = -1;
    // DBG + TRL_PNET_CLOCK: shader-clock time of every wave in each phase and at each barrier, summed over the launch (clk[2 + 8 wave + k])
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = 0;
    const bool prof = DBG && a.prof != 0;
    auto stamp = [&](int k) {
        if (DBG && prof) { const unsigned long long t = clock64(); pt[k] += t - pt_last; pt_last = t; }
    };
    if (DBG && prof) pt_last = clock64();
    for (; tile < t_end; tile = tile_nxt, cur = nxt, rot++) {
        const int f = cur.f, l = cur.l, ty = cur.ty, tx = cur.tx;
        const TileId& g = cur;
        asm volatile("" : "+v"(ctid));
        const int vrows = (g.oh - ty * TS < TS) ? g.oh - ty * TS : TS;      // valid output rows of this tile
        // left neighbour = `rows` positions back in the band order, upper neighbour = the previous position
        const bool carry = tx > 0 && (g.rib == 0 ? hs_tile[0] : (g.rib == 1 ? hs_tile[1] : hs_tile[BAND - 1])) == tile - g.rows;
        const bool vcarry = g.rib > 0 && vs_tile == tile - 1;
        const bool last_of_run = run_pos + 1 >= run_len;
        const bool feeds_next = run_pos + g.rows < run_len && tx + 1 < g.tiles_x;   // the right neighbour is a later tile of this run
        const bool feeds_down = !last_of_run && g.rib + 1 < g.rows;                   // the next tile of the run is the tile below
        float* const CP = CP0 + g.rib * (CARRY_P + CARRY_C);
        float* const CC = CP + CARRY_P;
        if (feeds_next) { if (g.rib == 0) hs_tile[0] = tile; else if (g.rib == 1) hs_tile[1] = tile; else hs_tile[BAND - 1] = tile; }
        if (feeds_down) vs_tile = tile;
        int cursor = tile + 1 - t_begin;
        // Guided self-scheduling: near the end of the XCD's chunk the runs shrink (down to single tiles), so the workgroups of an XCD
        // finish within a tile of each other instead of within a run.  Any run length is correct: a run is whatever atomicAdd hands out.
        int rem = (t_end - tile - 64 * run_len) >> 7;      // ~ tiles left per pair of workgroups (the cursor runs ~64 runs ahead of this tile)
        const int next_len = rem < 1 ? 1 : (rem > a.run ? a.run : rem);
        if (tid == 0 && last_of_run) cursor = atomicAdd(&a.xcd_next[xcd], next_len);   // in flight during phases 0-2, published at the barrier that ends phase 2
        run_pos = last_of_run ? 0 : run_pos + 1;
        run_len = last_of_run ? next_len : run_len;

        // ---- phase 0: prefetched input tile -> RA as [42][42][3] ---------------------------------------------
        stamp(7);
        if (!(dbg_skip & 16)) {
#pragma unroll
        for (int i = 0; i < 7; i++) {        // p < 1792: the 28 pixels past the tile land in RA's tail (finite, never a live operand)
            float* d = RA + 3 * tid + 768 * i;
            d[0] = pre[i].x; d[1] = pre[i].y; d[2] = pre[i].z;
        }
        }
        stamp(0);
        __syncthreads();
        stamp(1);

        // ---- phase 1: conv1 + 2x2 ceil max-pool + PReLU -> RB as [20][20][10] ------------------------------
        // conv1 (3 -> 10 channels, K = 27) runs on the 4x4x1 block instruction (see mfma4 above): wave w owns pooled rows w, w+4,
        // .., w+16.  Rows w .. w+12 x the 16 columns a carrying tile computes are ONE super unit of 64 cells (lane = cell, one
        // conv1 position per pass, the pool = an elementwise max of the four passes' accumulators); the fifth row and -- in a tile
        // without a left neighbour -- columns 0..3 run as quad units (block = cell, lanes = the four positions, pool = two DPP max).
        // Every LDS address is a per-lane base plus a compile-time offset; 420 block instructions of 8 cycles per wave and carry
        // tile against 20 x 7 16x16x4 instructions of 32 cycles before (N = 10 padded to 16, K = 27 to 28).
        if (!(dbg_skip & 2)) {
            const int vy = g.h - 2 - ty * 2 * TS, vx = g.w - 2 - tx * 2 * TS;   // valid conv1 extent inside the tile
            // max-pool commutes with PReLU when the slope is >= 0 (monotone): pool first, one PReLU per cell.  With a negative
            // slope somewhere the window's min is pooled as well and max_i prelu(v_i) = max(m, s n) (trl_prelu_pooled): still
            // one PReLU per pooled cell, never four before the pool.
            // Interior tiles (every conv1 pixel of the tile inside the level) also skip the ceil-mode masks.
            const bool fast = vy >= 2 * P1_T && vx >= 2 * P1_T;
            // CARRY: pooled columns 0..3 come from the left neighbour (CP, written in its phase 2): only column groups pg = 1..4 are
            // computed -- 20 M-tiles per wave instead of 25
            if (carry) copy_pooled(std::false_type{}, CP);
            if (vcarry) copy_rows(std::false_type{});      // (rows 0..3 of the strip columns arrive twice, with the same values)
            // Units of this wave: the wide units (columns 4..19) of its five pooled rows wave + 4 i; a tile that does not carry also
            // computes columns 0..3 as five NARROW units (4 rows x 4 columns each): narrow unit `wave` goes to this wave, narrow unit 4
            // to one wave in turn.  Units run in pairs (six independent accumulator chains keep the 8-cycle instruction issuing back to
            // back); the loop is not unrolled -- four instantiations of the body (edge masks x one / two units) instead of the old
            // path's four fully unrolled 25-tile sequences keeps the kernel's code inside the instruction cache.
            struct XU { int lb, sb, row, col; };
            auto wide_u = [&](int i) { XU u; u.lb = c1x_w + i * (8 * IN_T * 3); u.sb = c1x_sw + i * (4 * P1_T * 10); u.row = wave + 4 * i; u.col = 4 + xb; return u; };
            auto narrow_u = [&](int n) {
                XU u; u.row = 4 * n + (xb >> 2); u.col = xb & 3;
                u.lb = ((2 * u.row + xdy) * IN_T + 2 * u.col + xdx) * 3; u.sb = (u.row * P1_T + u.col) * 10 + 4 * xj;
                return u;
            };
            auto x_epi = [&](auto EDGE_T, const XU& u, const f32x4 (&acc)[3]) {
                constexpr bool EDGE = decltype(EDGE_T)::value;
                const float ninf = -__builtin_inff();
                const bool pv = !EDGE || ((2 * u.row + xdy < vy) && (2 * u.col + xdx < vx));     // this lane's conv1 pixel lies inside the level
                float m[3][4], n[3][4];
#pragma unroll
                for (int cg = 0; cg < 3; cg++) {
                    f32x4 v = acc[cg];
                    if (EDGE) { v[0] = pv ? v[0] : ninf; v[1] = pv ? v[1] : ninf; v[2] = pv ? v[2] : ninf; v[3] = pv ? v[3] : ninf; }
                    quad_max4(v, m[cg]);
                    if (NEG1) {
                        f32x4 w = acc[cg];
                        if (EDGE) { w[0] = pv ? w[0] : -ninf; w[1] = pv ? w[1] : -ninf; w[2] = pv ? w[2] : -ninf; w[3] = pv ? w[3] : -ninf; }
                        quad_min4(w, n[cg]);
                    }
                }
                const bool cv = !EDGE || ((2 * u.row < vy) && (2 * u.col < vx));                 // the cell exists (its (0,0) pixel does)
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pm = lane_sel(lane_sel(m[2][r], m[1][r], 0x2222222222222222ull), m[0][r], 0x1111111111111111ull);   // lane j: group j
                    const float sel = UNIT ? 0.f : trl_prelu_sel(sl4[r]);
                    if (NEG1) {
                        const float pn = lane_sel(lane_sel(n[2][r], n[1][r], 0x2222222222222222ull), n[0][r], 0x1111111111111111ull);
                        o[r] = prelu_pooled_t<UNIT>(pm, pn, sl4[r], sel);
                    } else {
                        o[r] = prelu_t<UNIT>(pm, sl4[r], sel);
                    }
                    if (EDGE) o[r] = cv ? o[r] : 0.f;
                }
                if (xj < 3) {                                    // lane j stores channels 4j .. 4j+3 of its cell (j = 2: channels 8, 9)
                    *reinterpret_cast<f32x2c*>(RB + u.sb) = f32x2c{o[0], o[1]};
                    if (xj < 2) *reinterpret_cast<f32x2c*>(RB + u.sb + 2) = f32x2c{o[2], o[3]};
                }
            };
            auto x_units = [&](auto EDGE_T, auto HASB_T, const XU& A, const XU& B) {
                constexpr bool HASB = decltype(HASB_T)::value;
                f32x4 accA[3], accB[3];
                float xa[2][9], xc[2][9];
                auto rd = [&](int t, float* xo, int lb) {        // k = 9 t + u sits at lb + 126 t + u of the [42][42][3] tile
#pragma unroll
                    for (int u = 0; u < 9; u++) xo[u] = RA[lb + 126 * t + u];
                };
                rd(0, xa[0], A.lb);
                if (HASB) rd(0, xc[0], B.lb);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cg = 0; cg < 3; cg++) {                 // the chain starts at the bias: slot k = 27 of the weight registers x 1.0
                    accA[cg] = mfma4<4, 11>(WC[cg][1], xone, z);
                    if (HASB) accB[cg] = mfma4<4, 11>(WC[cg][1], xone, z);
                }
                pn_static_for(std::make_integer_sequence<int, 3>{}, [&](auto T) __attribute__((always_inline)) {
                    constexpr int t = decltype(T)::value;
                    if (t < 2) { rd(t + 1, xa[(t + 1) & 1], A.lb); if (HASB) rd(t + 1, xc[(t + 1) & 1], B.lb); }
                    __builtin_amdgcn_sched_barrier(0);
                    pn_static_for(std::make_integer_sequence<int, 9>{}, [&](auto UU) __attribute__((always_inline)) {
                        constexpr int u = decltype(UU)::value, k = 9 * t + u;
#pragma unroll
                        for (int cg = 0; cg < 3; cg++) {
                            accA[cg] = mfma4<4, (k & 15)>(WC[cg][k >> 4], xa[t & 1][u], accA[cg]);
                            if (HASB) accB[cg] = mfma4<4, (k & 15)>(WC[cg][k >> 4], xc[t & 1][u], accB[cg]);
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                x_epi(EDGE_T, A, accA);
                if (HASB) x_epi(EDGE_T, B, accB);
            };
            // Rows wave, wave+4, wave+8, wave+12 x columns 4..19 as ONE super unit: pass p computes conv1 position (dy, dx) = (p >> 1, p & 1)
            // of all 64 cells (lane = cell); two passes run interleaved (six chains), the pool is an elementwise max of the passes'
            // accumulators -- 24 VALU per 64 cells where four quad units spend 96 DPP max and 32 selects -- and each lane applies the
            // PReLU to, and stores, its own cell's ten channels.
            auto x_super = [&](auto EDGE_T) {
                constexpr bool EDGE = decltype(EDGE_T)::value;
                const float ninf = -__builtin_inff();
                const int srow = wave + 4 * xj + (vcarry ? 4 : 0), scol = 4 + xb;   // a vertically carried tile computes rows 4..19
                const int sv_l = vcarry ? 8 * IN_T * 3 : 0, sv_s = vcarry ? 4 * P1_T * 10 : 0;
                f32x4 mx[3], mn[3];
                pn_static_for(std::make_integer_sequence<int, 2>{}, [&](auto PP) __attribute__((always_inline)) {
                    constexpr int pp = decltype(PP)::value;      // passes 2 pp (dx = 0) and 2 pp + 1 (dx = 1) of conv1 row parity dy = pp
                    const int lbA = c1s_l + sv_l + pp * (IN_T * 3), lbB = lbA + 3;
                    f32x4 accA[3], accB[3];
                    float xa[2][9], xc[2][9];
                    auto rd = [&](int t, float* xo, int lb) {
#pragma unroll
                        for (int u = 0; u < 9; u++) xo[u] = RA[lb + 126 * t + u];
                    };
                    rd(0, xa[0], lbA); rd(0, xc[0], lbB);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int cg = 0; cg < 3; cg++) { accA[cg] = mfma4<4, 11>(WC[cg][1], xone, z); accB[cg] = mfma4<4, 11>(WC[cg][1], xone, z); }
                    pn_static_for(std::make_integer_sequence<int, 3>{}, [&](auto T) __attribute__((always_inline)) {
                        constexpr int t = decltype(T)::value;
                        if (t < 2) { rd(t + 1, xa[(t + 1) & 1], lbA); rd(t + 1, xc[(t + 1) & 1], lbB); }
                        __builtin_amdgcn_sched_barrier(0);
                        pn_static_for(std::make_integer_sequence<int, 9>{}, [&](auto UU) __attribute__((always_inline)) {
                            constexpr int u = decltype(UU)::value, k = 9 * t + u;
#pragma unroll
                            for (int cg = 0; cg < 3; cg++) {
                                accA[cg] = mfma4<4, (k & 15)>(WC[cg][k >> 4], xa[t & 1][u], accA[cg]);
                                accB[cg] = mfma4<4, (k & 15)>(WC[cg][k >> 4], xc[t & 1][u], accB[cg]);
                            }
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    const bool pvA = !EDGE || ((2 * srow + pp < vy) && (2 * scol < vx)), pvB = !EDGE || ((2 * srow + pp < vy) && (2 * scol + 1 < vx));
#pragma unroll
                    for (int cg = 0; cg < 3; cg++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float va = EDGE ? (pvA ? accA[cg][r] : ninf) : accA[cg][r], vb = EDGE ? (pvB ? accB[cg][r] : ninf) : accB[cg][r];
                            mx[cg][r] = pp == 0 ? vmax_nc(va, vb) : vmax3_nc(mx[cg][r], va, vb);
                            if (NEG1) {
                                const float wa = EDGE ? (pvA ? accA[cg][r] : -ninf) : accA[cg][r], wb = EDGE ? (pvB ? accB[cg][r] : -ninf) : accB[cg][r];
                                mn[cg][r] = pp == 0 ? vmin_nc(wa, wb) : vmin3_nc(mn[cg][r], wa, wb);
                            }
                        }
                });
                const bool cv = !EDGE || ((2 * srow < vy) && (2 * scol < vx));
                float o[12];
#pragma unroll
                for (int c = 0; c < 10; c++) {
                    const float sl = a.s1[c];                    // uniform: a scalar load, the multiply takes it as an SGPR operand
                    const float sel = UNIT ? 0.f : trl_prelu_sel(sl);
                    o[c] = NEG1 ? prelu_pooled_t<UNIT>(mx[c >> 2][c & 3], mn[c >> 2][c & 3], sl, sel) : prelu_t<UNIT>(mx[c >> 2][c & 3], sl, sel);
                    if (EDGE) o[c] = cv ? o[c] : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 10; c += 2) *reinterpret_cast<f32x2c*>(RB + c1s_s + sv_s + c) = f32x2c{o[c], o[c + 1]};
            };
            if (fast) x_super(std::false_type{}); else x_super(std::true_type{});
            // ... the fifth row (wave + 16) and, in a tile that does not carry, the narrow units as quad units (pool = the block's lanes).
            // A vertically carried tile has no fifth row (rows 4..19 are the super unit); without a left neighbour its columns 0..3 of
            // rows 4..19 are narrow units 1..4, one per wave; with both carries there is nothing left.
            const int extra_wave = rot & 3;                      // the wave that takes narrow unit 4 of a tile that does not carry
#pragma unroll 1
            for (int pr = 2; pr < 4; pr++) {
                if (vcarry && (carry || pr == 3)) break;
                if (pr == 3 && (carry || wave != extra_wave)) break;
                const XU A = vcarry ? narrow_u(wave + 1) : (pr < 3 ? wide_u(4) : narrow_u(4));
                const XU B = narrow_u(wave);
                const bool two = pr == 2 && !carry && !vcarry;
                if (fast) { if (two) x_units(std::false_type{}, std::true_type{}, A, B); else x_units(std::false_type{}, std::false_type{}, A, B); }
                else { if (two) x_units(std::true_type{}, std::true_type{}, A, B); else x_units(std::true_type{}, std::false_type{}, A, B); }
            }
        }
        stamp(2);
        __syncthreads();
        stamp(3);

        // ---- phase 2: conv2 + PReLU -> RA as [324][17] ---------------------------------------------------------
        if (!(dbg_skip & 4)) {
            // 21 M-tiles of 16 rows (last one 4 rows); wave w takes pairs (w, w+4), (w+8, w+12), (w+16, w+20).
            // RA (the input tile) is dead since the barrier above, so the epilogue may overwrite it.
            // The A operands of pair j+1 are requested right after the MFMAs of pair j are issued, so their LDS
            // latency hides under that pair's tail and epilogue (same registers: MFMA sources are read at issue).
            // The 21 M-tiles split 6/5/5/5 over the waves; the 6-tile role rotates with the tile counter so that no
            // SIMD (wave i of both resident workgroups sits on SIMD i) is the long pole every time.
            const int w2 = (wave + rot) & 3;
            const int lim2 = C2_T * (vrows + 3 < C2_T ? vrows + 3 : C2_T);   // conv2 cells conv3 reads: rows 0 .. vrows+2
            // the pooled columns the right neighbour will not recompute (RB is read-only during this phase; CP was consumed
            // before the barrier that ended phase 1)
            if (feeds_next) copy_pooled(std::true_type{}, CP);
            if (feeds_down) copy_rows(std::true_type{});
            if (vcarry) copy_rows2(std::false_type{});       // conv2 rows 0..1 come from the upper neighbour (VC, written in its phase 3)
            if (carry) copy_conv2(std::false_type{}, CC);    // conv2 columns 0..1 from the left neighbour (CC, written in its phase 3)
            const int r0 = vcarry ? 2 : 0;                      // first conv2 row this tile computes
            float xa[23], xb[23];
            if (carry) {
                // conv2 columns 0..1 come from the left neighbour (CC, written in its phase 3); the 16 new columns of row y are
                // ONE M-tile: 18 M-tiles (instead of 21), wave role w2 takes rows w2, w2+4, .., every address a lane base + immediate
                const int rows2 = vrows + 3 < C2_T ? vrows + 3 : C2_T;          // conv2 rows conv3 reads
                const int cbase = (2 + l15) * 10;
                auto read_rows = [&](int j) {
                    int yA = r0 + w2 + 8 * j;
                    yA = yA < C2_T ? yA : C2_T - 1;                         // (rows past the tile are not computed: stay inside it)
                    const int yB = (yA + 4 < C2_T) ? yA + 4 : yA;
#pragma unroll
                    for (int s = 0; s < 23; s++) xa[s] = RB[yA * (P1_T * 10) + cbase + koff<30, 170>(s, kq, 0, e2_2, 0)];
                    if (yA + 4 < C2_T) {
#pragma unroll
                        for (int s = 0; s < 23; s++) xb[s] = RB[yB * (P1_T * 10) + cbase + koff<30, 170>(s, kq, 0, e2_2, 0)];
                    }
                };
                read_rows(0);
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int yA = r0 + w2 + 8 * j, yB = yA + 4;
                    const bool hasA = yA < rows2, hasB = yB < rows2;         // (rows2 <= 18: also bounds the row index)
                    f32x4 accA = bias2v, accB = bias2v;
                    __builtin_amdgcn_sched_barrier(0);
                    if (!hasA) {
                    } else if (hasB) {
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], B2[0], bias2v, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[0], B2[0], bias2v, 0, 0, 0);
#pragma unroll
                        for (int s = 1; s < 23; s++) {
                            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B2[s], accA, 0, 0, 0);
                            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[s], B2[s], accB, 0, 0, 0);
                        }
                    } else {
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], B2[0], bias2v, 0, 0, 0);
#pragma unroll
                        for (int s = 1; s < 23; s++) accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B2[s], accA, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 2) read_rows(j + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (hasA) RA[(yA * C2_T + 2 + kq * 4 + q) * C2_LD + l15] = prelu_t<UNIT>(accA[q], slope2, sel2);
                        if (hasB) RA[(yB * C2_T + 2 + kq * 4 + q) * C2_LD + l15] = prelu_t<UNIT>(accB[q], slope2, sel2);
                    }
                }
            } else {
            // (a vertically carried tile without a left neighbour: rows 2..17 = cells 36..323 = 18 M-tiles, the same walk shifted)
            const int c0 = r0 * C2_T, nmt2 = vcarry ? 18 : 21;
            auto read_pair2 = [&](int j) {
                const int mtA = w2 + 8 * j, mtB = (mtA + 4 < 21) ? mtA + 4 : mtA;
                int mA = c0 + mtA * 16 + l15; mA = mA < 324 ? mA : 323;
                int mB = c0 + mtB * 16 + l15; mB = mB < 324 ? mB : 323;
                const int yA = mA / C2_T, xA = mA - yA * C2_T, yB = mB / C2_T, xB = mB - yB * C2_T;
                const int baseA = (yA * P1_T + xA) * 10, baseB = (yB * P1_T + xB) * 10;
#pragma unroll
                for (int s = 0; s < 23; s++) xa[s] = RB[baseA + koff<30, 170>(s, kq, 0, e2_2, 0)];
                if (j < 2 || w2 == 0) {
#pragma unroll
                    for (int s = 0; s < 23; s++) xb[s] = RB[baseB + koff<30, 170>(s, kq, 0, e2_2, 0)];
                }
            };
            read_pair2(0);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int mtA = w2 + 8 * j, mtB = mtA + 4;
                // Tiles on the bottom edge of a level hold fewer than 16 valid output rows: conv2 cells below the rows conv3 will
                // read are skipped (M-tile m covers cells 16m.. of the 18-wide grid).  Uniform per wave, no barrier inside.
                const bool hasB = mtB < nmt2 && c0 + 16 * mtB < lim2;  // j == 2: only wave role 0 has a second tile (M-tile 20)
                const bool hasA = mtA < nmt2 && c0 + 16 * mtA < lim2;
                f32x4 accA = bias2v, accB = bias2v;
                __builtin_amdgcn_sched_barrier(0);
                if (!hasA) {
                } else if (hasB) {
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], B2[0], bias2v, 0, 0, 0);     // C operand = the bias quad (see conv1)
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[0], B2[0], bias2v, 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < 23; s++) {
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B2[s], accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[s], B2[s], accB, 0, 0, 0);
                    }
                } else {
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], B2[0], bias2v, 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < 23; s++) accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B2[s], accA, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j < 2) read_pair2(j + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ra = c0 + mtA * 16 + kq * 4 + q;
                    if (hasA && ra < 324) RA[ra * C2_LD + l15] = prelu_t<UNIT>(accA[q], slope2, sel2);
                    const int rb = c0 + mtB * 16 + kq * 4 + q;
                    if (hasB && rb < 324) RA[rb * C2_LD + l15] = prelu_t<UNIT>(accB[q], slope2, sel2);
                }
            }
            }   // !carry
        }
        if (tid == 0) next_tile_s = t_begin + cursor;   // the cursor's atomic has had phases 0-2 to return
        stamp(4);
        __syncthreads();
        stamp(5);

        // next tile's input: global loads into registers only (RA is still read by phase 3)
        tile_nxt = __builtin_amdgcn_readfirstlane(next_tile_s);
        if (tile_nxt < t_end) { nxt = decode(tile_nxt); if (!(dbg_skip & 1)) issue_input(nxt); }
        // the conv2 columns / rows the right / lower neighbour will not recompute (RA is read-only until the barrier that ends the
        // tile) are copied behind the wave's last conv3 M-tile, in front of the last head chain
        auto strips_get = [&]() __attribute__((always_inline)) {
            if (feeds_next) copy_conv2(std::true_type{}, CC);
            if (feeds_down) copy_rows2(std::true_type{});
        };

        // ---- phase 3: conv3 + PReLU -> heads -> candidates, register to register -------------------------------------
        if (!(dbg_skip & 8)) {
            const float fscale = g.scale;
            const f32x4* T3 = reinterpret_cast<const f32x4*>(T3all);
            // conv3 runs TRANSPOSED on mfma_f32_32x32x2: the A operand is the weight (rows = output channels), the B operand the
            // activation (columns = the 32 cells of the M-tile = 2 output rows), the same two LDS words per lane and k-step as the
            // other way round and the same chain per output (acc = bias, k ascending).  Lane l then holds 16 channels --
            // 8 (q >> 2) + (q & 3) + 4 hh -- of ONE cell (l & 31): exactly what the heads' B operand wants, so nothing is staged
            // through LDS between conv3 and the heads (round 3 measured that round trip and the idle chain behind it at 6 % of the
            // kernel).  `between(sx, u)` is called after MFMA u of sixth sx: the previous M-tile's head chain rides in those slots.
            auto conv3 = [&](int mt, f32x16& P, auto&& between) __attribute__((always_inline)) {
                const int y = mt * 2 + (l31 >> 4), x = l31 & 15;
                const int base = (y * C2_T + x) * C2_LD + hh;
                f32x16 acc;
#pragma unroll
                for (int qa = 0; qa < 4; qa++) {
                    const f32x4 b4 = T3[2 * qa + hh];                       // bias of channels 8 qa + 4 hh .. + 3
                    acc[4 * qa] = b4[0]; acc[4 * qa + 1] = b4[1]; acc[4 * qa + 2] = b4[2]; acc[4 * qa + 3] = b4[3];
                }
                // 6 x 12 k-steps, double buffered: the operands of sixth i+1 are requested before the MFMAs of sixth i
                // issue, so only the first sixth's LDS latency is exposed.
                float xa[2][12], wb[2][12];
                auto read_sixth = [&](int sx, float* xo, float* wo) {
#pragma unroll
                    for (int u = 0; u < 12; u++) {
                        const int s = sx * 12 + u, tap = s >> 3, ky = tap / 3, kx = tap - ky * 3;
                        xo[u] = RA[base + (ky * C2_T + kx) * C2_LD + 2 * (s & 7)];
                        wo[u] = B3S[(2 * s + hh) * 32 + l31];
                    }
                };
                read_sixth(0, xa[0], wb[0]);
                pn_static_for(std::make_integer_sequence<int, 6>{}, [&](auto SX) __attribute__((always_inline)) {
                    constexpr int sx = decltype(SX)::value;
                    if (sx < 5) read_sixth(sx + 1, xa[(sx + 1) & 1], wb[(sx + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    pn_static_for(std::make_integer_sequence<int, 12>{}, [&](auto UU) __attribute__((always_inline)) {
                        constexpr int u = decltype(UU)::value;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[sx & 1][u], xa[sx & 1][u], acc, 0, 0, 0);
                        between(SX, UU);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int qa = 0; qa < 4; qa++) {
                    const f32x4 s4 = T3[8 + 2 * qa + hh];                   // PReLU slopes of the same channels
#pragma unroll
                    for (int qb = 0; qb < 4; qb++) P[4 * qa + qb] = prelu_t<UNIT>(acc[4 * qa + qb], s4[qb], UNIT ? 0.f : trl_prelu_sel(s4[qb]));
                }
            };
            // heads (1x1, 32 -> 2 + 4) on v_mfma_f32_4x4x1_16B_f32: sixteen 4x4 blocks per instruction, ONE k per instruction.
            // Block b of lanes 4b..4b+3 = cells 4(b & 7) .. +3 (the B operand: the cell's conv3 value at k), outputs
            // 4(b >> 3) .. +3 (the A operand: weights W[k][4(b >> 3) + i], broadcast inside each group of eight blocks from block
            // k & 7 of register k >> 3 -- cbsz = 3, abid = k & 7): one instruction does both output groups of all 32 cells and
            // every lane ends up holding ITS cell's logits.  Same chain: acc = bias, k ascending.  Channel k of cell c sits in
            // lane c + 32 ((k >> 2) & 1), register 4 (k >> 3) + (k & 3): one v_permlane32_swap of the register with a copy of
            // itself yields the lower half's value in every lane (k & 4 == 0) and the upper half's (k & 4 == 4).
            unsigned hlo[4], hhi[4];
            auto head_step = [&](auto H_T, const f32x16& P, f32x4& hq) __attribute__((always_inline)) {
                constexpr int k = decltype(H_T)::value, ga = k >> 3, gb = k & 7;
                if constexpr (gb == 0) {
#pragma unroll
                    for (int qb = 0; qb < 4; qb++) {
                        const unsigned v = __float_as_uint(P[4 * ga + qb]);
                        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                        hlo[qb] = r[0]; hhi[qb] = r[1];
                    }
                }
                const float bk = __uint_as_float(gb < 4 ? hlo[gb & 3] : hhi[gb & 3]);
                hq = __builtin_amdgcn_mfma_f32_4x4x1f32(WH[ga], bk, hq, 3, gb, 0);
            };
            auto heads_all = [&](const f32x16& P, f32x4& hq) __attribute__((always_inline)) {
                pn_static_for(std::make_integer_sequence<int, 32>{}, [&](auto H_T) __attribute__((always_inline)) { head_step(H_T, P, hq); });
            };
            // lanes 0..31: {logit0, logit1, reg0, reg1} of cell = lane; lanes 32..63: {reg2, reg3, -, -} of cell = lane - 32
            // The softmax (two exps and a division: ~55 VALU beside the other workgroup's MFMAs) runs only for an M-tile in which
            // some cell's logit difference comes within reach of the threshold (a.dthr: a bound with a wide margin, fill_args) --
            // one compare and a scalar branch for the ~90 % of the M-tiles that hold no candidate.
            auto emit = [&](int mt, const f32x4& hq) __attribute__((always_inline)) {
                if (__builtin_amdgcn_ballot_w64(lane < 32 && hq[1] - hq[0] >= a.dthr) == 0) return;
                const float p = trl_softmax2_p1(hq[0], hq[1]);
                const auto u2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(hq[0]), __float_as_uint(hq[0]), false, false);
                const auto u3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(hq[1]), __float_as_uint(hq[1]), false, false);
                if (lane < 32) {                                // one lane per cell of the 32-row tile
                    const int oy = ty * TS + mt * 2 + (lane >> 4), ox = tx * TS + (lane & 15);
                    if (oy < g.oh && ox < g.ow) {
                        if (p >= a.thr && !(dbg_skip & 32)) {      // (bit 32: timing-only ablations emit no candidates)
                            const int seg = f * a.L + l;
                            const int sl = atomicAdd(&a.lvl_cnt[seg], 1);
                            if (sl < a.lv[l].cap) {            // (scalar loads inside the rare branch: the tile descriptor stays as small as it was)
                                Cand c;
                                c.x1 = floorf((2.f * (float)ox + 1.f) / fscale);
                                c.y1 = floorf((2.f * (float)oy + 1.f) / fscale);
                                c.x2 = floorf((2.f * (float)ox + 12.f) / fscale);
                                c.y2 = floorf((2.f * (float)oy + 12.f) / fscale);
                                c.score = p;
                                c.r0 = hq[2]; c.r1 = hq[3]; c.r2 = __uint_as_float(u2[1]); c.r3 = __uint_as_float(u3[1]);
                                c.cell = oy * g.ow + ox;
                                a.lvl_rec[(size_t)f * a.rec_stride + a.lv[l].rec0 + sl] = c;
                            } else {
                                a.flags[0] = 1;
                            }
                        }
                    }
                }
            };
            // 8 M-tiles of 32 cells = 2 output rows each; wave w takes M-tiles w and w + 4 (a bottom-edge tile may have neither or
            // only the first: those output rows lie below the level).  The head chain of the first rides in the MFMA stream of the
            // second -- one 8-cycle block instruction after every second 64-cycle one, so its 32 dependent steps never wait.
            const int mt0 = wave, mt1 = wave + 4;
            if (2 * mt0 < vrows) {
                f32x16 P0;
                f32x4 hq0 = T3[16 + hh];                        // head bias of this half's output group
                conv3(mt0, P0, [](auto, auto) {});
                if (2 * mt1 < vrows) {
                    f32x16 P1;
                    f32x4 hq1 = T3[16 + hh];
                    conv3(mt1, P1, [&](auto SX, auto UU) __attribute__((always_inline)) {
                        constexpr int sx = decltype(SX)::value, u = decltype(UU)::value, h = sx * 6 + (u >> 1);
                        if constexpr ((u & 1) == 1 && h < 32) {          // (pinned: the scheduler would bunch the block instructions)
                            __builtin_amdgcn_sched_barrier(0);
                            head_step(std::integral_constant<int, h>{}, P0, hq0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    strips_get();
                    heads_all(P1, hq1);
                    emit(mt0, hq0);
                    emit(mt1, hq1);
                } else {
                    strips_get();
                    heads_all(P0, hq0);
                    emit(mt0, hq0);
                }
            } else {
                strips_get();
            }
        } else {
            strips_get();
        }
        stamp(6);
        __syncthreads();   // RA / RB are rewritten by the next tile
        stamp(7);
    }
    if (DBG && prof && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) atomicAdd(&a.clk[2 + 8 * wave + k], pt[k]);
        if (wave == 0) atomicAdd(&a.clk[2 + 32], (unsigned long long)rot);
    }
    if (DBG && tid == 0) atomicMax(&a.clk[1], (unsigned long long)wall_clock64());
}

}  // namespace

// after a fused launch: [36] += its span, [37] += 1, [38] = its span; the start / end stamps are re-armed for the next launch
__global__ void k_pnet_span(unsigned long long* clk) {
    const unsigned long long t0 = clk[0], t1 = clk[1];
    const unsigned long long d = t1 > t0 ? t1 - t0 : 0ull;
    clk[36] += d; clk[37] += 1ull; clk[38] = d;
    clk[0] = ~0ull; clk[1] = 0ull;
}

int trl_pnet_prepare(trl_ctx* c) {
    for (const char* n : {"pnet.conv1.w", "pnet.conv2.w", "pnet.conv3.w", "pnet.heads.w"}) {
        const DevW* w = trl_w(c, n);
        if (!w || w->ld != 32) { trl_set_error("PNet weight %s missing or wrong shape", n); return TRL_ERR_WEIGHTS; }
    }
    const DevW *w1 = trl_w(c, "pnet.conv1.w"), *w2 = trl_w(c, "pnet.conv2.w"), *w3 = trl_w(c, "pnet.conv3.w"), *wh = trl_w(c, "pnet.heads.w");
    if (w1->K != 27 || w1->Cout != 10 || w2->K != 90 || w2->Cout != 16 || w3->K != 144 || w3->Cout != 32 || wh->K != 32 || wh->Cout != 6) {
        trl_set_error("PNet weights have unexpected shapes");
        return TRL_ERR_WEIGHTS;
    }
    float sl[10];
    TRL_HIP(hipMemcpy(sl, trl_v(c, "pnet.prelu1")->p, sizeof sl, hipMemcpyDeviceToHost));
    c->pnet_mono1 = 1;
    for (float v : sl) if (!(v >= 0.f)) c->pnet_mono1 = 0;
    c->pnet_unit = 1;
    for (const char* n : {"pnet.prelu1", "pnet.prelu2", "pnet.prelu3"}) {
        const DevV* v = trl_v(c, n);
        std::vector<float> h(v->n);
        TRL_HIP(hipMemcpy(h.data(), v->p, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (float x : h) if (!(x <= 1.f)) c->pnet_unit = 0;
    }
    return TRL_OK;
}

// ceil(2^32 / d) for the kernel's sdiv(); d == 1 would need 2^32: 2^32 - 1 gives q = n - 1 (or 0), which sdiv's upward correction
// step turns into n -- no special case on the device
static unsigned sdiv_magic(int d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

static int fill_args(trl_ctx* c, int n, int H, int W, PnetArgs& a, std::vector<uint32_t>* tab = nullptr) {
    const int L = trl_compute_levels(c, H, W);
    if (L > 16) { trl_set_error("more than 16 pyramid levels"); return TRL_ERR_INVALID; }
    if (L < 1) { trl_set_error("frame %dx%d has no pyramid level at min_face_size %d", W, H, c->cfg.min_face_size); return TRL_ERR_INVALID; }
    a.n_frames = n; a.L = L; a.H = H; a.W = W;
    int tiles = 0, ntab = 0; long long pix = 0, work = 0;
    for (int l = 0; l < L; l++) {
        const LevelGeom& g = c->lv[l];
        PLevel& p = a.lv[l];
        p.h = g.h; p.w = g.w; p.oh = g.oh; p.ow = g.ow;
        p.tiles_x = (g.ow + TS - 1) / TS;
        p.txmagic = sdiv_magic(p.tiles_x);
        p.tile0 = tiles;
        p.tiles_y = (g.oh + TS - 1) / TS;
        p.bmagic = sdiv_magic(BAND * p.tiles_x);
        tiles += p.tiles_x * ((g.oh + TS - 1) / TS);
        p.pix0 = (int)pix;
        p.pix_pad = (int)(((long long)g.h * g.w + 63) & ~63ll);
        pix += p.pix_pad;
        // pyramid kernel path and lanes per output pixel
        const int khmax = (H + g.h - 1) / g.h + 1, kwmax = (W + g.w - 1) / g.w + 1;   // upper bounds of the bin sizes
        p.khA = (H + g.h - 1) / g.h; p.kwA = (W + g.w - 1) / g.w;
        p.khmax = (H % g.h) ? p.khA + 1 : p.khA; p.kwmax = (W % g.w) ? p.kwA + 1 : p.kwA;
        p.rkh[0] = 1.0f / (float)p.khA; p.rkh[1] = 1.0f / (float)(p.khA + 1);
        p.rkw[0] = 1.0f / (float)p.kwA; p.rkw[1] = 1.0f / (float)(p.kwA + 1);
        p.fastdiv = (p.khA + 1 <= 96 && p.kwA + 1 <= 96) ? 1 : 0;
        for (int d = 0; d < 4; d++) {
            auto vb = [](int rel, int nbytes) { int hi = nbytes - rel; hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi); return hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u); };
            p.vmA[d] = vb(4 * d, 3 * p.kwA); p.vmB[d] = vb(4 * d, 3 * (p.kwA + 1));
        }
        p.wmagic = (unsigned)((0x100000000ull + g.w - 1) / g.w);
        p.hmagic = (unsigned)((0x100000000ull + g.h - 1) / g.h);
        // floor(n / d) == umulhi(n, ceil(2^32 / d)) for every n with n * d < 2^32 (n <= (in + 1) * out here)
        p.arith = ((unsigned long long)(H + 1) * g.h * g.h < 0x100000000ull && (unsigned long long)(W + 1) * g.w * g.w < 0x100000000ull &&
                   g.h > 1 && g.w > 1) ? 1 : 0;
        p.nd = 3; p.grshift = 0;
        if (kwmax * 3 <= 15 && khmax <= 5) {
            p.mode = 0; p.gshift = 0; p.nd = kwmax * 3 <= 9 ? 3 : 4;
        } else if ((W * 3) % 4 == 0) {
            p.mode = 1;
            const int ngr = (kwmax * 3 + 3 + 11) / 12;
            while ((1 << p.grshift) < ngr) p.grshift++;
            int rlsh = 0;
            while ((khmax >> rlsh) > 8 && p.grshift + rlsh < 6) rlsh++;
            p.gshift = p.grshift + rlsh;
            if (p.gshift > 6) { p.mode = 2; p.gshift = 6; }
        } else {
            p.mode = 2;
            const int dwords = khmax * ((kwmax * 3 + 6) / 4);
            p.gshift = dwords <= 40 ? 0 : (dwords <= 160 ? 2 : (dwords <= 640 ? 4 : 6));
        }
        p.work0 = (int)work;
        work += (long long)p.pix_pad << p.gshift;
        // adaptive_avg_pool2d bin edges: [floor(i*in/out), ceil((i+1)*in/out))
        p.ytab0 = ntab; ntab += g.h;
        p.xtab0 = ntab; ntab += g.w;
        if (tab) {
            for (int i = 0; i < g.h; i++) tab->push_back((uint32_t)(((long long)i * H) / g.h) | ((uint32_t)((((long long)i + 1) * H + g.h - 1) / g.h) << 16));
            for (int i = 0; i < g.w; i++) tab->push_back((uint32_t)(((long long)i * W) / g.w) | ((uint32_t)((((long long)i + 1) * W + g.w - 1) / g.w) << 16));
        }
        p.scale = (float)g.scale;
        p.cap = c->cb.lay.capl[l]; p.rec0 = c->cb.lay.rec0[l];   // (set by trl_cascade_detect before the fused launch)
    }
    a.tiles_per_frame = tiles;
    a.tpf_magic = sdiv_magic(tiles);
    a.pyr_stride = pix;
    a.work_per_frame = work;
    a.w1 = trl_w(c, "pnet.conv1.w")->p; a.w2 = trl_w(c, "pnet.conv2.w")->p; a.w3 = trl_w(c, "pnet.conv3.w")->p; a.wh = trl_w(c, "pnet.heads.w")->p;
    a.b1 = trl_v(c, "pnet.conv1.b")->p; a.b2 = trl_v(c, "pnet.conv2.b")->p; a.b3 = trl_v(c, "pnet.conv3.b")->p; a.bh = trl_v(c, "pnet.heads.b")->p;
    a.s1 = trl_v(c, "pnet.prelu1")->p; a.s2 = trl_v(c, "pnet.prelu2")->p; a.s3 = trl_v(c, "pnet.prelu3")->p;
    a.thr = c->cfg.thr0; a.rec_stride = c->cb.lay.S;
    // p = softmax(logit0, logit1)[1] >= thr needs logit1 - logit0 >= ln(thr / (1 - thr)) up to the rounding of the float softmax
    // (~1e-6 relative); 0.05 below that bound the probability is short of thr by 0.05 thr (1 - thr) >= 4.9e-4 for thr in [0.01, 0.99]
    a.dthr = (a.thr >= 0.01f && a.thr <= 0.99f) ? (float)(log((double)a.thr / (1.0 - (double)a.thr)) - 0.05) : -__builtin_inff();
    a.dbg_skip = trl_tune_int("TRL_PNET_SKIP", 0);
    a.lvl_cnt = c->cb.lvl_cnt; a.lvl_rec = c->cb.lvl_rec; a.flags = c->cb.flags;
    a.clk = c->pnet_clk; a.prof = c->pnet_prof ? 1 : 0;
    a.xcd_next = c->pnet_cursor;
    return TRL_OK;
}

size_t trl_pnet_fused_bytes(trl_ctx* c, int n, int H, int W) {
    PnetArgs a;
    if (fill_args(c, n, H, W, a) != TRL_OK) return 0;
    return (size_t)a.pyr_stride * n * sizeof(PyrPx) + 4096;
}

// All pyramid levels of all n frames: pyramid kernel + one persistent fused launch.
// ev[0..1] bracket the pyramid kernel, ev[2..3] the fused kernel (HIP events on the same stream).
namespace {

// ---- coarse levels in ONE streaming pass ------------------------------------------------------------------
// The per-level kernels above make every level re-read its whole source chunk; for the coarse levels (bins wider than
// 5 px: 8 of the 11 levels at 720p, 15 % of the pixels) that re-read IS the cost (~55 us per level and 64 frames,
// whatever the level's size).  k_pyramid_stream reads each source row ONCE for all of them:
//   * a workgroup owns a band of the frame (rows [R0,R1) x a column band of <= 4096 bytes) and walks its rows top down;
//     thread t owns 16 consecutive bytes of the row (one dword-aligned 16-byte load + 1 dword, re-aligned by a scalar shift);
//   * per level it keeps the column sums of the current output row's bin in registers (16 byte-columns, packed 16-bit);
//     at the bin's last source row the sums go to LDS, the horizontal bins are reduced, normalised (pyr_norm) and stored,
//     and the accumulators restart (with the current row when consecutive bins share it);
//   * a band computes the bins that START inside it and reads on past its end until they are complete (no atomics).
// Integer sums in any order are exact, so the result is bit-identical to the per-level kernels (tests: every level, 180p..4K).
constexpr int SMAXL = 12;               // coarse levels per launch
constexpr int SBYTES = 4096;            // bytes of a source row a workgroup covers (256 threads x 16)
struct SLevel { int h, w, pix0, pix_pad, ytab0, xtab0, khA, kwA, fastdiv; float rkh[2], rkw[2]; };
struct PyrStreamArgs {
    int H, W, n_frames, f0, nlev, rows_per_band, cols_per_band, row_bands, col_bands;
    long long pyr_stride;
    SLevel lv[SMAXL];
};

__device__ __forceinline__ float stream_norm(unsigned s, int kh, int kw, const SLevel& g) {
    const float a = (float)s, fkh = (float)kh, fkw = (float)kw;
    float q;
    if (g.fastdiv) {
        const float r1 = kh == g.khA ? g.rkh[0] : g.rkh[1], r2 = kw == g.kwA ? g.rkw[0] : g.rkw[1];
        q = pyr_div(pyr_div(a, fkh, r1), fkw, r2);
    } else {
        q = a / fkh / fkw;
    }
    return (q - 127.5f) * 0.0078125f;
}

constexpr int STAB = 6144;              // bin-edge words of the coarse levels kept in LDS (rows + columns of every level)
template <int NL>
__global__ __launch_bounds__(256) void k_pyramid_stream(const uint8_t* __restrict__ frames, PyrStreamArgs a, const uint32_t* __restrict__ gtab,
                                                        PyrPx* __restrict__ pyr) {
    __shared__ unsigned colbuf[SBYTES];
    __shared__ uint32_t tab[STAB];          // the levels' edge tables, re-based: level l rows at ty0[l], columns at tx0[l]
    const int tid = threadIdx.x;
    const int f = a.f0 + blockIdx.y;
    const int rb = blockIdx.x / a.col_bands, cb = blockIdx.x - rb * a.col_bands;
    const int R0 = rb * a.rows_per_band, R1 = (R0 + a.rows_per_band < a.H) ? R0 + a.rows_per_band : a.H;
    const int C0 = cb * a.cols_per_band, C1 = (C0 + a.cols_per_band < a.W) ? C0 + a.cols_per_band : a.W;
    const int row_bytes = a.W * 3;
    const long long fbase = (long long)f * a.H * row_bytes;
    const long long last_dw = ((long long)a.n_frames * a.H * row_bytes - 1) >> 2;
    const uint32_t* base32 = reinterpret_cast<const uint32_t*>(frames);
    // edge tables of the handled levels -> LDS (a flush would otherwise end in a dependent global load)
    int ty0[NL], tx0[NL];
    {
        int pos = 0;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            ty0[l] = tx0[l] = 0;
            if (l < a.nlev) {
                const SLevel& g = a.lv[l];
                ty0[l] = pos; tx0[l] = pos + g.h;
                for (int i = tid; i < g.h; i += 256) tab[pos + i] = gtab[g.ytab0 + i];
                for (int i = tid; i < g.w; i += 256) tab[pos + g.h + i] = gtab[g.xtab0 + i];
                pos += g.h + g.w;
            }
        }
    }
    __syncthreads();

    // per level (all scalar): owned output rows [j, jend), owned output columns [ox0, ox1), current bin rows [ys, ye)
    int j[NL], jend[NL], ys[NL], ye[NL], ox0[NL], ox1[NL];
    int yend = R0;
#pragma unroll
    for (int l = 0; l < NL; l++) {
        j[l] = jend[l] = 0; ys[l] = ye[l] = 0x7fffffff; ox0[l] = ox1[l] = 0;
        if (l < a.nlev) {
            const SLevel& g = a.lv[l];
            auto first_at_or_after = [&](int tab0, int n_out, int n_in, int pos) {   // first bin whose start >= pos
                if (pos >= n_in) return n_out;
                int q = (int)(((long long)pos * n_out + n_in - 1) / n_in);
                if (q > n_out) q = n_out;
                while (q > 0 && (int)(tab[tab0 + q - 1] & 0xFFFF) >= pos) q--;
                while (q < n_out && (int)(tab[tab0 + q] & 0xFFFF) < pos) q++;
                return q;
            };
            j[l] = first_at_or_after(ty0[l], g.h, a.H, R0);
            jend[l] = first_at_or_after(ty0[l], g.h, a.H, R1);
            ox0[l] = first_at_or_after(tx0[l], g.w, a.W, C0);
            ox1[l] = first_at_or_after(tx0[l], g.w, a.W, C1);
            if (j[l] < jend[l] && ox0[l] < ox1[l]) {
                const uint32_t t0 = tab[ty0[l] + j[l]], t1 = tab[ty0[l] + jend[l] - 1];
                ys[l] = t0 & 0xFFFF; ye[l] = t0 >> 16;
                yend = ((int)(t1 >> 16) > yend) ? (int)(t1 >> 16) : yend;
            } else {
                j[l] = jend[l];
            }
        }
    }
    // zero the 64-pixel padding behind each level once per frame
    if (blockIdx.x == 0) {
#pragma unroll
        for (int l = 0; l < NL; l++)
            if (l < a.nlev) {
                const SLevel& g = a.lv[l];
                for (int p = g.h * g.w + tid; p < g.pix_pad; p += 256)
                    pyr_store(pyr + ((long long)f * a.pyr_stride + g.pix0 + p), make_float4(0.f, 0.f, 0.f, 0.f));
            }
    }
    if (yend <= R0) return;

    unsigned ev[NL][4], od[NL][4];          // packed 16-bit column sums: bytes 0,2 / 1,3 of each of the thread's 4 dwords
#pragma unroll
    for (int l = 0; l < NL; l++)
#pragma unroll
        for (int d = 0; d < 4; d++) { ev[l][d] = 0; od[l][d] = 0; }

    // one source row: this thread's 16 bytes at byte offset C0*3 + 16*tid of row y
    auto load_row = [&](int y, unsigned (&w)[5], unsigned& sh) {
        const int yy = y < a.H ? y : a.H - 1;                                    // rows past the frame are never accumulated
        const long long o = fbase + (long long)yy * row_bytes + (long long)C0 * 3;   // scalar
        sh = (unsigned)(o & 3);
        const long long dw = (o >> 2) + 4 * tid;
        if (dw + 4 <= last_dw) {
            const u32x4_a4 v4 = *reinterpret_cast<const u32x4_a4*>(base32 + dw);
            w[0] = v4[0]; w[1] = v4[1]; w[2] = v4[2]; w[3] = v4[3];
            w[4] = base32[dw + 4];
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) w[k] = base32[dw + k <= last_dw ? dw + k : last_dw];
        }
    };
    auto consume = [&](int y, const unsigned (&w)[5], unsigned sh) {
        unsigned v[4];
#pragma unroll
        for (int d = 0; d < 4; d++) v[d] = __builtin_amdgcn_alignbyte(w[d + 1], w[d], sh);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            if (l < a.nlev && y >= ys[l] && y < ye[l]) {                         // uniform
#pragma unroll
                for (int d = 0; d < 4; d++) { ev[l][d] += v[d] & 0x00FF00FFu; od[l][d] += (v[d] >> 8) & 0x00FF00FFu; }
                if (y == ye[l] - 1) {
                    // ---- the bin row is complete: column sums -> LDS -> horizontal bins -> normalise -> store ----
                    const SLevel& g = a.lv[l];
                    const int kh = ye[l] - ys[l];
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        colbuf[16 * tid + 4 * d + 0] = ev[l][d] & 0xFFFFu; colbuf[16 * tid + 4 * d + 1] = od[l][d] & 0xFFFFu;
                        colbuf[16 * tid + 4 * d + 2] = ev[l][d] >> 16;     colbuf[16 * tid + 4 * d + 3] = od[l][d] >> 16;
                    }
                    __syncthreads();
                    for (int ox = ox0[l] + tid; ox < ox1[l]; ox += 256) {
                        const uint32_t tx = tab[tx0[l] + ox];
                        const int xs = tx & 0xFFFF, xe = tx >> 16;
                        unsigned s0 = 0, s1 = 0, s2 = 0;
                        for (int xx = xs; xx < xe; xx++) {
                            const unsigned* p = colbuf + (xx - C0) * 3;
                            s0 += p[0]; s1 += p[1]; s2 += p[2];
                        }
                        float4 o4;
                        o4.x = stream_norm(s0, kh, xe - xs, g); o4.y = stream_norm(s1, kh, xe - xs, g); o4.z = stream_norm(s2, kh, xe - xs, g);
                        o4.w = 0.f;
                        pyr_store(pyr + ((long long)f * a.pyr_stride + g.pix0 + (long long)j[l] * g.w + ox), o4);
                    }
                    __syncthreads();
                    // next owned bin of this level; consecutive bins may share this source row
                    j[l]++;
                    if (j[l] < jend[l]) {
                        const uint32_t t0 = tab[ty0[l] + j[l]];
                        ys[l] = t0 & 0xFFFF; ye[l] = t0 >> 16;
                    } else {
                        ys[l] = ye[l] = 0x7fffffff;
                    }
                    const bool again = ys[l] <= y;
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        ev[l][d] = again ? (v[d] & 0x00FF00FFu) : 0u;
                        od[l][d] = again ? ((v[d] >> 8) & 0x00FF00FFu) : 0u;
                    }
                }
            }
        }
    };

    // rows in groups of four: the loads of the next group are in flight while this one is consumed
    unsigned wb[4][5], shb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) load_row(R0 + u, wb[u], shb[u]);
    for (int y = R0; y < yend; y += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned wc[5];
#pragma unroll
            for (int k = 0; k < 5; k++) wc[k] = wb[u][k];
            const unsigned shc = shb[u];
            if (y + u + 4 < yend) load_row(y + u + 4, wb[u], shb[u]);
            if (y + u < yend) consume(y + u, wc, shc);
        }
    }
}

// ---- the same pass with WAVE-local ownership, for frames wider than one 4096-byte band (round 4) -------------------------------
// k_pyramid_stream covers a whole row with one workgroup and therefore stops at W = 1365; 1080p and 4K frames took the per-level
// kernels (55 launches per step at 4K, each re-reading the source).  Here a WAVE owns a segment of source columns for a band of rows
// and never talks to another wave: lane i holds 20 consecutive bytes of the row (5 dwords + 1 for the re-alignment), the segment is
// 64 x 20 = 1280 bytes = 426 pixels of which the last kwmax overlap the next segment, so that every bin that STARTS in the segment
// is covered by the wave's own loads.  A completed bin row is flushed through the wave's private LDS strip (LDS operations of one
// wave execute in order: no barrier), reduced, normalised and stored.  Integer sums: bit-identical.  (At 720p, where both apply,
// the block-wide pass is faster -- 0.63 vs 1.11 ms: every wave pays all ~165 flush round trips alone -- so it keeps the narrow frames.)
constexpr int SW_BYTES = 1280;           // bytes of a source row per wave
template <int NL>
__global__ __launch_bounds__(256) void k_pyramid_stream_w(const uint8_t* __restrict__ frames, PyrStreamArgs a, const uint32_t* __restrict__ gtab,
                                                          PyrPx* __restrict__ pyr) {
    __shared__ unsigned colbuf_all[4][SW_BYTES];
    __shared__ uint32_t tab[STAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = a.f0 + blockIdx.y;
    const int row_bytes = a.W * 3;
    const long long fbase = (long long)f * a.H * row_bytes;
    const long long last_dw = ((long long)a.n_frames * a.H * row_bytes - 1) >> 2;
    const uint32_t* base32 = reinterpret_cast<const uint32_t*>(frames);
    int ty0[NL], tx0[NL];
    {
        int pos = 0;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            ty0[l] = tx0[l] = 0;
            if (l < a.nlev) {
                const SLevel& g = a.lv[l];
                ty0[l] = pos; tx0[l] = pos + g.h;
                for (int i = tid; i < g.h; i += 256) tab[pos + i] = gtab[g.ytab0 + i];
                for (int i = tid; i < g.w; i += 256) tab[pos + g.h + i] = gtab[g.xtab0 + i];
                pos += g.h + g.w;
            }
        }
    }
    __syncthreads();                                     // the only block barrier: from here on the waves are on their own
    const int unit = blockIdx.x * 4 + wave;              // (row band, column segment) of this wave
    if (unit >= a.row_bands * a.col_bands) return;
    const int rb = unit / a.col_bands, cs = unit - rb * a.col_bands;
    const int R0 = rb * a.rows_per_band, R1 = (R0 + a.rows_per_band < a.H) ? R0 + a.rows_per_band : a.H;
    const int C0 = cs * a.cols_per_band, C1 = (C0 + a.cols_per_band < a.W) ? C0 + a.cols_per_band : a.W;
    unsigned* colbuf = colbuf_all[wave];

    int j[NL], jend[NL], ys[NL], ye[NL], ox0[NL], ox1[NL];
    int yend = R0;
#pragma unroll
    for (int l = 0; l < NL; l++) {
        j[l] = jend[l] = 0; ys[l] = ye[l] = 0x7fffffff; ox0[l] = ox1[l] = 0;
        if (l < a.nlev) {
            const SLevel& g = a.lv[l];
            auto first_at_or_after = [&](int tab0, int n_out, int n_in, int pos) {   // first bin whose start >= pos
                if (pos >= n_in) return n_out;
                int q = (int)(((long long)pos * n_out + n_in - 1) / n_in);
                if (q > n_out) q = n_out;
                while (q > 0 && (int)(tab[tab0 + q - 1] & 0xFFFF) >= pos) q--;
                while (q < n_out && (int)(tab[tab0 + q] & 0xFFFF) < pos) q++;
                return q;
            };
            j[l] = __builtin_amdgcn_readfirstlane(first_at_or_after(ty0[l], g.h, a.H, R0));
            jend[l] = __builtin_amdgcn_readfirstlane(first_at_or_after(ty0[l], g.h, a.H, R1));
            ox0[l] = __builtin_amdgcn_readfirstlane(first_at_or_after(tx0[l], g.w, a.W, C0));
            ox1[l] = __builtin_amdgcn_readfirstlane(first_at_or_after(tx0[l], g.w, a.W, C1));
            if (j[l] < jend[l] && ox0[l] < ox1[l]) {
                const uint32_t t0 = tab[ty0[l] + j[l]], t1 = tab[ty0[l] + jend[l] - 1];
                ys[l] = __builtin_amdgcn_readfirstlane((int)(t0 & 0xFFFF)); ye[l] = __builtin_amdgcn_readfirstlane((int)(t0 >> 16));
                const int e1 = __builtin_amdgcn_readfirstlane((int)(t1 >> 16));
                yend = e1 > yend ? e1 : yend;
            } else {
                j[l] = jend[l];
            }
        }
    }
    if (unit == 0) {                                     // zero the 64-pixel padding behind each level once per frame
#pragma unroll
        for (int l = 0; l < NL; l++)
            if (l < a.nlev) {
                const SLevel& g = a.lv[l];
                for (int p = g.h * g.w + lane; p < g.pix_pad; p += 64)
                    pyr_store(pyr + ((long long)f * a.pyr_stride + g.pix0 + p), make_float4(0.f, 0.f, 0.f, 0.f));
            }
    }
    if (yend <= R0) return;

    unsigned ev[NL][5], od[NL][5];          // packed 16-bit column sums: bytes 0,2 / 1,3 of each of the lane's 5 dwords
#pragma unroll
    for (int l = 0; l < NL; l++)
#pragma unroll
        for (int d = 0; d < 5; d++) { ev[l][d] = 0; od[l][d] = 0; }

    // one source row: this lane's 20 bytes at byte offset C0*3 + 20*lane of row y (+ the dword behind them for the re-alignment)
    auto load_row = [&](int y, unsigned (&w)[6], unsigned& sh) {
        const int yy = y < a.H ? y : a.H - 1;                                    // rows past the frame are never accumulated
        const long long o = fbase + (long long)yy * row_bytes + (long long)C0 * 3;   // scalar
        sh = (unsigned)(o & 3);
        const long long dw = (o >> 2) + 5 * lane;
        if (dw + 5 <= last_dw) {
            const u32x4_a4 v4 = *reinterpret_cast<const u32x4_a4*>(base32 + dw);
            w[0] = v4[0]; w[1] = v4[1]; w[2] = v4[2]; w[3] = v4[3];
            w[4] = base32[dw + 4]; w[5] = base32[dw + 5];
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = base32[dw + k <= last_dw ? dw + k : last_dw];
        }
    };
    auto consume = [&](int y, const unsigned (&w)[6], unsigned sh) {
        unsigned v[5];
#pragma unroll
        for (int d = 0; d < 5; d++) v[d] = __builtin_amdgcn_alignbyte(w[d + 1], w[d], sh);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            if (l < a.nlev && y >= ys[l] && y < ye[l]) {                         // wave-uniform
#pragma unroll
                for (int d = 0; d < 5; d++) { ev[l][d] += v[d] & 0x00FF00FFu; od[l][d] += (v[d] >> 8) & 0x00FF00FFu; }
                if (y == ye[l] - 1) {
                    // ---- the bin row is complete: column sums -> the wave's LDS strip -> horizontal bins -> normalise -> store ----
                    const SLevel& g = a.lv[l];
                    const int kh = ye[l] - ys[l];
#pragma unroll
                    for (int d = 0; d < 5; d++) {
                        colbuf[20 * lane + 4 * d + 0] = ev[l][d] & 0xFFFFu; colbuf[20 * lane + 4 * d + 1] = od[l][d] & 0xFFFFu;
                        colbuf[20 * lane + 4 * d + 2] = ev[l][d] >> 16;     colbuf[20 * lane + 4 * d + 3] = od[l][d] >> 16;
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int ox = ox0[l] + lane; ox < ox1[l]; ox += 64) {
                        const uint32_t tx = tab[tx0[l] + ox];
                        const int xs = tx & 0xFFFF, xe = tx >> 16;
                        unsigned s0 = 0, s1 = 0, s2 = 0;
                        for (int xx = xs; xx < xe; xx++) {
                            const unsigned* p = colbuf + (xx - C0) * 3;
                            s0 += p[0]; s1 += p[1]; s2 += p[2];
                        }
                        float4 o4;
                        o4.x = stream_norm(s0, kh, xe - xs, g); o4.y = stream_norm(s1, kh, xe - xs, g); o4.z = stream_norm(s2, kh, xe - xs, g);
                        o4.w = 0.f;
                        pyr_store(pyr + ((long long)f * a.pyr_stride + g.pix0 + (long long)j[l] * g.w + ox), o4);
                    }
                    __builtin_amdgcn_wave_barrier();
                    // next owned bin of this level; consecutive bins may share this source row
                    j[l]++;
                    if (j[l] < jend[l]) {
                        const uint32_t t0 = tab[ty0[l] + j[l]];
                        ys[l] = __builtin_amdgcn_readfirstlane((int)(t0 & 0xFFFF)); ye[l] = __builtin_amdgcn_readfirstlane((int)(t0 >> 16));
                    } else {
                        ys[l] = ye[l] = 0x7fffffff;
                    }
                    const bool again = ys[l] <= y;
#pragma unroll
                    for (int d = 0; d < 5; d++) {
                        ev[l][d] = again ? (v[d] & 0x00FF00FFu) : 0u;
                        od[l][d] = again ? ((v[d] >> 8) & 0x00FF00FFu) : 0u;
                    }
                }
            }
        }
    };

    // rows in groups of four: the loads of the next group are in flight while this one is consumed
    unsigned wb[4][6], shb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) load_row(R0 + u, wb[u], shb[u]);
    for (int y = R0; y < yend; y += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned wc[6];
#pragma unroll
            for (int k = 0; k < 6; k++) wc[k] = wb[u][k];
            const unsigned shc = shb[u];
            if (y + u + 4 < yend) load_row(y + u + 4, wb[u], shb[u]);
            if (y + u < yend) consume(y + u, wc, shc);
        }
    }
}

}  // namespace

// The pyramid of all n frames (production path of both the fused PNet and the debug export below).
static int build_pyramid(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, PnetArgs& a, hipEvent_t* ev, hipStream_t s) {
    if (((uintptr_t)d_frames & 3) != 0) { trl_set_error("frame buffer must be 4-byte aligned"); return TRL_ERR_INVALID; }
    if (H > 16383 || W > 16383) { trl_set_error("frame larger than 16383 px"); return TRL_ERR_INVALID; }
    std::vector<uint32_t> tab;
    const bool new_shape = (c->pyr_tab == nullptr || c->pyr_tab_H != H || c->pyr_tab_W != W);
    TRL_CHECK(fill_args(c, n, H, W, a, new_shape ? &tab : nullptr));
    if (new_shape) {   // bin-edge tables depend on (H, W) only: built once per frame shape
        // k_pyramid_fine: which output columns / rows of the (up to three) finest levels belong to which source tile -- those whose
        // bins START in it.  A band must not own more than 64 columns of any level (one wave = one level's columns of the band).
        c->pyr_fine.nlev = 0;
        int nfine = 0;
        while (nfine < 3 && nfine < a.L && a.lv[nfine].mode == 0 && a.lv[nfine].h <= H && a.lv[nfine].w <= W) nfine++;
        if (nfine >= 2) {
            static const int strip_env = trl_tune_int("TRL_PYR_FINE_STRIP", 0);   // tuning: source rows per tile
            const int strip_rows = strip_env >= 8 ? strip_env : 24;
            int band_cols = (int)(62.0 * W / a.lv[0].w);
            std::vector<uint32_t> own;
            for (; band_cols >= 16; band_cols--) {
                const int nb = (W + band_cols - 1) / band_cols, ns = (H + strip_rows - 1) / strip_rows;
                own.assign((size_t)3 * nb + (size_t)3 * ns, 0u);
                bool ok = true;
                for (int l = 0; l < nfine && ok; l++) {
                    const PLevel& g = a.lv[l];
                    int o = 0;
                    for (int b = 0; b < nb; b++) {                   // columns: bin starts are non-decreasing in ox
                        const int lo = o;
                        while (o < g.w && (int)(tab[g.xtab0 + o] & 0xFFFF) < (b + 1) * band_cols) o++;
                        if (o - lo > 64) { ok = false; break; }
                        own[(size_t)l * nb + b] = (uint32_t)lo | ((uint32_t)o << 16);
                    }
                    o = 0;
                    for (int t = 0; t < ns; t++) {
                        const int lo = o;
                        while (o < g.h && (int)(tab[g.ytab0 + o] & 0xFFFF) < (t + 1) * strip_rows) o++;
                        own[(size_t)3 * nb + (size_t)l * ns + t] = (uint32_t)lo | ((uint32_t)o << 16);
                    }
                }
                if (ok) {
                    c->pyr_fine.nlev = nfine; c->pyr_fine.own0 = (int)tab.size(); c->pyr_fine.band_cols = band_cols; c->pyr_fine.strip_rows = strip_rows;
                    c->pyr_fine.n_bands = nb; c->pyr_fine.n_strips = ns;
                    tab.insert(tab.end(), own.begin(), own.end());
                    break;
                }
            }
        }
        TRL_HIP(hipStreamSynchronize(s));
        if (c->pyr_tab) TRL_HIP(hipFree(c->pyr_tab));
        c->pyr_tab = nullptr;
        TRL_HIP(hipMalloc((void**)&c->pyr_tab, tab.size() * 4 + 64));
        TRL_HIP(hipMemcpy(c->pyr_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        c->pyr_tab_H = H; c->pyr_tab_W = W;
    }
    PyrPx* pyr = (PyrPx*)c->scratch.alloc((size_t)a.pyr_stride * n * sizeof(PyrPx));
    if (!pyr) { trl_set_error("pyramid workspace"); return TRL_ERR_STATE; }
    a.pyr = pyr;
    if (ev) TRL_HIP(hipEventRecord(ev[0], s));
    // Frames are resampled in chunks whose source bytes fit the 256 MiB Infinity Cache: every level re-reads the
    // whole source image, so the 2nd..11th level launches of a chunk are served on-die instead of from HBM.
    static const int chunk_env = trl_tune_int("TRL_PYR_CHUNK", 0);
    int chunk = chunk_env > 0 ? chunk_env : (int)((176ll << 20) / ((long long)H * W * 3));
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    // coarse levels: one streaming pass (k_pyramid_stream) when its preconditions hold, else the per-level kernels
    // Streaming pass (k_pyramid_stream) for the coarse levels (mode != 0); optionally (TRL_PYR_STREAM_FINE=1) a second one for
    // the fine levels.  A group that does not meet the kernel's preconditions falls back to the per-level kernels below.
    static const bool stream_off = trl_tune_int("TRL_PYR_STREAM", 1) == 0;
    static const bool fine_on = trl_tune_int("TRL_PYR_STREAM_FINE", 0) != 0;   // measured: 1.75 vs 1.83 ms, not worth a default
    static const int bands_env = trl_tune_int("TRL_PYR_BANDS", 0);
    static const int fbands_env = trl_tune_int("TRL_PYR_FINE_BANDS", 0);
    bool streamed[16] = {};
    static const int wide_env = trl_tune_int("TRL_PYR_STREAM_WIDE", 1);   // tuning: 0 = per-level kernels for frames wider than one band
    auto stream_group = [&](bool fine, int row_bands) -> int {
        // the levels of the group, at most 8 per launch (register budget of the per-level column sums): each launch reads the source once
        int lv_idx[16], nsel = 0;
        for (int l = 0; l < a.L; l++) if ((a.lv[l].mode == 0) == fine) lv_idx[nsel++] = l;
        const bool wide = W * 3 > SBYTES;                                       // one workgroup cannot cover the row: wave-local pass
        if (wide && !wide_env) return TRL_OK;
        for (int g0 = 0; g0 < nsel; g0 += 8) {
            const int gn = nsel - g0 < 8 ? nsel - g0 : 8;
            PyrStreamArgs sa;
            sa.nlev = 0;
            int stab_words = 0, kwm = 0;
            bool ok = true;
            for (int q = 0; q < gn; q++) {
                const PLevel& g = a.lv[lv_idx[g0 + q]];
                if (g.khmax > 256) ok = false;
                stab_words += g.h + g.w;
                kwm = g.kwmax > kwm ? g.kwmax : kwm;
                SLevel& t = sa.lv[sa.nlev++];
                t.h = g.h; t.w = g.w; t.pix0 = g.pix0; t.pix_pad = g.pix_pad; t.ytab0 = g.ytab0; t.xtab0 = g.xtab0; t.khA = g.khA; t.kwA = g.kwA;
                t.fastdiv = g.fastdiv; t.rkh[0] = g.rkh[0]; t.rkh[1] = g.rkh[1]; t.rkw[0] = g.rkw[0]; t.rkw[1] = g.rkw[1];
            }
            sa.H = H; sa.W = W; sa.n_frames = n; sa.pyr_stride = a.pyr_stride; sa.f0 = 0;   // every source row is read once: no Infinity-Cache chunking
            sa.row_bands = H >= 256 ? row_bands : 1;
            sa.rows_per_band = (H + sa.row_bands - 1) / sa.row_bands;
            if (!ok || stab_words > STAB || n > 65535) continue;                 // these levels take the per-level kernels
            if (!wide) {
                if ((kwm + 2) * 3 > SBYTES / 2) continue;
                sa.col_bands = 1; sa.cols_per_band = W;
                const dim3 sgrid(sa.row_bands * sa.col_bands, n);
                if (sa.nlev <= 4) k_pyramid_stream<4><<<sgrid, 256, 0, s>>>(d_frames, sa, c->pyr_tab, pyr);
                else k_pyramid_stream<8><<<sgrid, 256, 0, s>>>(d_frames, sa, c->pyr_tab, pyr);
            } else {
                // a wave covers 1280 bytes = 426 whole pixels of a row; the bins that start in its segment may reach kwmax further
                sa.cols_per_band = SW_BYTES / 3 - kwm;
                if (sa.cols_per_band < 64) continue;
                sa.col_bands = (W + sa.cols_per_band - 1) / sa.cols_per_band;
                sa.cols_per_band = (W + sa.col_bands - 1) / sa.col_bands;           // even segments
                // enough waves to fill the chip (~4 k) when the batch is small: more, shorter row bands (each reads on past its end
                // until its last bins are complete, so not shorter than 256 rows)
                int rbn = (4096 + n * sa.col_bands - 1) / (n * sa.col_bands);
                if (rbn > H / 256) rbn = H / 256;
                if (rbn > sa.row_bands) { sa.row_bands = rbn; sa.rows_per_band = (H + rbn - 1) / rbn; }
                const dim3 sgrid((sa.row_bands * sa.col_bands + 3) / 4, n);
                if (sa.nlev <= 4) k_pyramid_stream_w<4><<<sgrid, 256, 0, s>>>(d_frames, sa, c->pyr_tab, pyr);
                else k_pyramid_stream_w<8><<<sgrid, 256, 0, s>>>(d_frames, sa, c->pyr_tab, pyr);
            }
            TRL_LAUNCH_CHECK();
            for (int q = 0; q < gn; q++) streamed[lv_idx[g0 + q]] = true;
        }
        return TRL_OK;
    };
    if (!stream_off) {
        TRL_CHECK(stream_group(false, bands_env > 0 ? bands_env : 3));
        if (fine_on) TRL_CHECK(stream_group(true, fbands_env > 0 ? fbands_env : 8));
    }
    const bool stream_ok = true;
    // the finest levels in one pass over the source (TRL_PYR_FINE=0: the per-level kernels)
    static const bool fine_off = trl_tune_int("TRL_PYR_FINE", 1) == 0;
    if (!fine_off && c->pyr_fine.nlev >= 2 && n <= 65535 && c->pyr_fine.n_strips <= 65535) {
        PyrFineArgs fa;
        fa.H = H; fa.W = W; fa.n_frames = n; fa.f0 = 0; fa.pyr_stride = a.pyr_stride;
        fa.nlev = c->pyr_fine.nlev; fa.band_cols = c->pyr_fine.band_cols; fa.strip_rows = c->pyr_fine.strip_rows;
        fa.n_bands = c->pyr_fine.n_bands; fa.n_strips = c->pyr_fine.n_strips; fa.own0 = c->pyr_fine.own0;
        for (int l = 0; l < 3; l++) fa.g[l] = a.lv[l < fa.nlev ? l : 0];
        k_pyramid_fine<<<dim3(fa.n_bands, n, fa.n_strips), 192, 0, s>>>(d_frames, fa, c->pyr_tab, pyr);
        TRL_LAUNCH_CHECK();
        for (int l = 0; l < fa.nlev; l++) streamed[l] = true;
    }
    for (int f0 = 0; f0 < n; f0 += chunk) {
        const int nf = (n - f0 < chunk) ? n - f0 : chunk;
        for (int l = 0; l < a.L; l++) {
            if (stream_ok && streamed[l]) continue;
            PyrArgs pa;
            pa.H = H; pa.W = W; pa.n_frames = n; pa.f0 = f0; pa.pyr_stride = a.pyr_stride; pa.g = a.lv[l];
            const int threads = pa.g.pix_pad << pa.g.gshift;
            dim3 grid(pa.g.mode == 0 ? (threads + 511) / 512 : (threads + 255) / 256, nf);   // mode 0: two pixels per thread
            if (pa.g.mode == 0) {
                if (pa.g.khmax <= 3 && pa.g.nd <= 3) k_pyramid0<3, false><<<grid, 256, 0, s>>>(d_frames, pa, c->pyr_tab, pyr);
                else if (pa.g.khmax <= 4) k_pyramid0<4, true><<<grid, 256, 0, s>>>(d_frames, pa, c->pyr_tab, pyr);
                else k_pyramid0<5, true><<<grid, 256, 0, s>>>(d_frames, pa, c->pyr_tab, pyr);
            }
            else if (pa.g.mode == 1) k_pyramid<1><<<grid, 256, 0, s>>>(d_frames, pa, c->pyr_tab, pyr);
            else k_pyramid<2><<<grid, 256, 0, s>>>(d_frames, pa, c->pyr_tab, pyr);
            TRL_LAUNCH_CHECK();
        }
    }
    if (ev) TRL_HIP(hipEventRecord(ev[1], s));
    return TRL_OK;
}

// debug / test hook: level `level` of ONE frame's pyramid exactly as the fused PNet kernel reads it -> d_out [h][w][3]
__global__ void k_export_level(const PyrPx* __restrict__ pyr, int npix, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const PyrPx v = pyr[i];
    out[3 * i + 0] = v.b; out[3 * i + 1] = v.g; out[3 * i + 2] = v.r;
}
int trl_pyramid_export(trl_ctx* c, const uint8_t* d_frame, int H, int W, int level, float* d_out, int* h, int* w, hipStream_t s) {
    PnetArgs a;
    TRL_CHECK(build_pyramid(c, d_frame, 1, H, W, a, nullptr, s));
    if (level < 0 || level >= a.L) { trl_set_error("level %d out of range (%d levels)", level, a.L); return TRL_ERR_INVALID; }
    const PLevel& g = a.lv[level];
    k_export_level<<<(g.h * g.w + 255) / 256, 256, 0, s>>>(a.pyr + g.pix0, g.h * g.w, d_out);
    TRL_LAUNCH_CHECK();
    *h = g.h; *w = g.w;
    return TRL_OK;
}

int trl_pnet_fused_all(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, hipEvent_t* ev, hipStream_t s) {
    PnetArgs a;
    TRL_CHECK(build_pyramid(c, d_frames, n, H, W, a, ev, s));
    const int total_tiles = a.tiles_per_frame * n;
    static const int grid_env = trl_tune_int("TRL_PNET_GRID", 0);   // experiment: workgroups of the persistent launch
    int grid = grid_env > 0 ? grid_env : 256 * 2;   // 2 resident workgroups per CU (<= 256 VGPRs)
    if (grid > ((total_tiles + 7) / 8) * 8) grid = ((total_tiles + 7) / 8) * 8;
    if (grid < 8) grid = 8;
    TRL_HIP(hipMemsetAsync(c->pnet_cursor, 0, 8 * sizeof(int32_t), s));
    // The persistent grid fills every CU: it waits for the end of the device's previous call (another context's cascade tail or
    // embedder) and then runs alone, so the event pair below is its duration; the pyramid kernels queued in front of it are
    // memory-bound and DO overlap that call's narrow kernels (trl_gate_wait, trl_api.hip).
    TRL_CHECK(trl_gate_wait(c, s));
    if (ev) TRL_HIP(hipEventRecord(ev[2], s));   // the event pair brackets the kernel alone (HIP events on the launch's stream)
    static const int xlds = trl_tune_int("TRL_PNET_XLDS", 0);   // experiment: unused dynamic LDS, lowers the resident workgroups per CU
    // instantiation: slopes all <= 1 or not, a negative conv1 slope or not, diagnostics (TRL_PNET_CLOCK / TRL_PNET_SKIP) or not
#ifdef TRL_TUNING
    const bool dbg = c->pnet_prof || a.dbg_skip || trl_tune_set("TRL_PNET_SPAN");   // the instantiation with clock stamps / ablations
#else
    const bool dbg = false;                              // (not even instantiated in the shipped library)
#endif
    // Tiles per cursor fetch: a run of 24 = 8 columns of a 3-row band (7 of 8 columns carry horizontally, 2 of 3 rows vertically);
    // small batches keep shorter runs, down to single tiles, so that every CU gets work
    // (TRL_PNET_RUN / trl_debug_pnet_run override: tuning, and tests that exercise the carry path on small frames)
    static const int run_env = trl_tune_int("TRL_PNET_RUN", 0);
    const int auto_run = (total_tiles / 8) / 128;
    a.run = c->pnet_run > 0 ? c->pnet_run : (run_env > 0 ? run_env : (auto_run < 1 ? 1 : (auto_run > 8 * BAND ? 8 * BAND : auto_run)));
    auto launch = [&](auto kern) {
        // static + dynamic LDS exceed the default 64 KB per workgroup
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS + xlds) != hipSuccess) return false;
        kern<<<grid, 256, DYN_LDS + xlds, s>>>(a);
        return true;
    };
    bool launched = false;
#ifdef TRL_TUNING
#define TRL_PK(U, N) do { launched = dbg ? launch(k_pnet_fused<U, N, true>) : launch(k_pnet_fused<U, N, false>); } while (0)
#else
#define TRL_PK(U, N) do { launched = launch(k_pnet_fused<U, N, false>); } while (0)
#endif
    if (c->pnet_unit) { if (c->pnet_mono1) TRL_PK(true, false); else TRL_PK(true, true); }
    else { if (c->pnet_mono1) TRL_PK(false, false); else TRL_PK(false, true); }
#undef TRL_PK
    if (!launched) { trl_set_error("k_pnet_fused: %d bytes of dynamic LDS refused", DYN_LDS + xlds); return TRL_ERR_HIP; }
    TRL_LAUNCH_CHECK();
    if (ev) TRL_HIP(hipEventRecord(ev[3], s));
    if (dbg) {
        k_pnet_span<<<1, 1, 0, s>>>(c->pnet_clk);   // fold the launch's span into the running sums, re-arm the two stamps
        TRL_LAUNCH_CHECK();
    }
    return TRL_OK;
}
