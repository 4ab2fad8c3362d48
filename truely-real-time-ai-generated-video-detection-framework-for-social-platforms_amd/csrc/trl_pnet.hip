// trl_pnet.hip -- MTCNN stage 1 (PNet over the image pyramid) for gfx950, the dominant kernel of the
// hot path (83 % of the conv FLOPs at 720p; server/model.py:47 -> detect_face stage 1).
//
// Two launches per batch of frames:
//
//  k_pyramid      u8 BGR frames -> every pyramid level, imresample (F.interpolate mode="area") +
//                 (x-127.5)*0.0078125, stored as float4 {b,g,r,0} per pixel.  One thread per output
//                 pixel, aligned dword loads + v_dot4 byte sums (integer-exact, order independent).
//
//  k_pnet_fused   ONE persistent launch over all (frame, level, 16x16-cell tile) work items.  Per tile,
//                 entirely in LDS / registers:
//                   42x42x3 input tile -> conv1 3x3 (3->10) + PReLU + 2x2 ceil max-pool (in the MFMA
//                   epilogue: the 4 rows a lane holds ARE one pool window) -> conv2 3x3 (10->16) + PReLU
//                   -> conv3 3x3 (16->32) + PReLU -> 1x1 heads (32->2+4) -> softmax -> thr0 ->
//                   generateBoundingBox record appended to the (frame, level) candidate list.
//                 All four layers run on the f32 matrix cores (v_mfma_f32_16x16x4_f32 for N<=16,
//                 v_mfma_f32_32x32x2_f32 for conv3), k ascending, accumulator seeded with the bias: the
//                 same fmaf chain as the oracle, so maps and candidates are bit-identical.
//                 Every weight matrix lives in registers for the whole launch (B operands: 7+23+72+8
//                 VGPRs per lane), activations never leave the CU: HBM traffic is the pyramid read only.
//                 blockIdx -> tile mapping keeps an XCD on a contiguous run of tiles (halo rows of
//                 neighbouring tiles hit the same L2).
#include "trl_ctx.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TS = 16;            // output cells per tile side
constexpr int IN_T = 2 * TS + 10; // 42 input pixels
constexpr int P1_T = TS + 4;      // 20 pooled cells
constexpr int C2_T = TS + 2;      // 18 conv2 cells
constexpr int C2_LD = 17;         // padded channel stride of the conv2 tile (bank spread for conv3 reads)
constexpr int ST_LD = 33;         // padded channel stride of the per-wave conv3 staging tile
constexpr int REGION_A = C2_T * C2_T * C2_LD;        // 5508 floats: input tile, later conv2 output
constexpr int REGION_B = 4 * 32 * ST_LD;             // 4224 floats: pooled conv1, later conv3 staging
static_assert(IN_T * IN_T * 3 <= REGION_A, "input tile fits region A");
static_assert(P1_T * P1_T * 10 <= REGION_B, "pooled tile fits region B");

struct PLevel {
    int h, w, oh, ow;        // level size, PNet map size
    int tiles_x, tile0;      // tiles per row, first tile index of the level inside a frame
    int pix0;                // float4 offset of the level inside a frame's pyramid
    int pix_pad;             // h*w rounded up to 64 (pyramid slots of the level)
    int gshift, work0;       // pyramid kernel: log2(lanes per pixel), first thread of the level inside a frame
    int ytab0, xtab0;        // offsets of the level's row / column bin-edge tables
    float scale;
};
struct PnetArgs {
    const float4* pyr; long long pyr_stride;   // float4 per frame
    int n_frames, L, tiles_per_frame, H, W;
    long long work_per_frame;                  // pyramid kernel threads per frame
    PLevel lv[16];
    const float *w1, *w2, *w3, *wh;            // [Kpad][32] zero padded
    const float *b1, *b2, *b3, *bh, *s1, *s2, *s3;
    float thr; int cap;
    int dbg_skip;                              // timing-only ablation mask (TRL_PNET_SKIP); 0 in production
    int32_t* lvl_cnt; Cand* lvl_rec; int32_t* flags;
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
__global__ __launch_bounds__(256) void k_pyramid(const uint8_t* __restrict__ frames, PnetArgs a, const uint32_t* __restrict__ tab,
                                                 float4* __restrict__ pyr) {
    const int per_frame = (int)a.work_per_frame;           // threads per frame, multiple of 64
    const int f = blockIdx.y;
    const uint32_t* base32 = reinterpret_cast<const uint32_t*>(frames);
    const long long fbase = (long long)f * a.H * a.W * 3;
    const int row_bytes = a.W * 3;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < per_frame; p += gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < a.L && p >= a.lv[l + 1].work0) l++;   // wave-uniform: level boundaries are multiples of 64
        const PLevel& g = a.lv[l];
        const int w = g.w, gsh = g.gshift, G = 1 << gsh;
        const int q = p - g.work0;
        const int pixel = q >> gsh, sub = q & (G - 1);
        const bool valid = pixel < g.h * w;
        unsigned s0 = 0, s1 = 0, s2 = 0;
        int kh = 1, kw = 1;
        if (valid) {
            const int oy = pixel / w, ox = pixel - oy * w;
            const uint32_t ty = tab[g.ytab0 + oy], tx = tab[g.xtab0 + ox];
            const int ys = ty & 0xFFFF, ye = ty >> 16, xs = tx & 0xFFFF, xe = tx >> 16;
            kh = ye - ys; kw = xe - xs;
            const int nbytes = kw * 3;
            const long long o0 = fbase + (long long)ys * row_bytes + xs * 3;   // first byte of the bin
            if (G == 1) {
                long long o = o0;
                for (int y = 0; y < kh; y++, o += row_bytes) {
                    long long al = o & ~3ll;
                    for (int rel = (int)(al - o); rel < nbytes; rel += 4, al += 4) dword_sums(base32[al >> 2], rel, nbytes, s0, s1, s2);
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
            const float fkh = (float)kh, fkw = (float)kw;
            float4 o4;
            o4.x = valid ? ((float)s0 / fkh / fkw - 127.5f) * 0.0078125f : 0.f;
            o4.y = valid ? ((float)s1 / fkh / fkw - 127.5f) * 0.0078125f : 0.f;
            o4.z = valid ? ((float)s2 / fkh / fkw - 127.5f) * 0.0078125f : 0.f;
            o4.w = 0.f;
            pyr[(long long)f * a.pyr_stride + g.pix0 + pixel] = o4;
        }
    }
}

// ---- fused PNet --------------------------------------------------------------------------------------
__device__ __forceinline__ float prelu(float v, float sl) { return v > 0.f ? v : sl * v; }

// A-operand k offsets.  k = 4s+kq walks (tap, channel) of a [pixel][C] LDS tile whose rows are E floats
// further apart than D consecutive k: offset = k + E*(k/D).  With s a compile-time constant only the step
// whose four k straddle a multiple of D depends on the lane (kq >= thr), so no per-lane offset tables.
template <int D, int E>
__device__ __forceinline__ int koff(int s, int kq, int e1, int e2, int e3) {
    const int q0 = (4 * s) / D, r0 = (4 * s) % D, thr = D - r0;    // folded after unrolling
    const int add = thr == 1 ? e1 : (thr == 2 ? e2 : (thr == 3 ? e3 : 0));
    return 4 * s + E * q0 + kq + add;
}

__global__ __launch_bounds__(256, 2) void k_pnet_fused(PnetArgs a) {
    __shared__ __attribute__((aligned(16))) float RA[REGION_A];   // input tile [42][42][3]  ->  conv2 out [324][17]
    __shared__ __attribute__((aligned(16))) float RB[REGION_B];   // pooled [400][10]        ->  conv3 staging [4][32][33]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;      // 16x16x4 operand coordinates
    const int l31 = lane & 31, hh = lane >> 5;      // 32x32x2 operand coordinates

    // ---- B operands: every weight matrix stays in registers for the whole launch -----------------------------
    float B3[72];
#pragma unroll
    for (int s = 0; s < 72; s++) B3[s] = a.w3[(2 * s + hh) * 32 + l31];
    float B1[7], B2[23], BH[8];
#pragma unroll
    for (int s = 0; s < 7; s++) B1[s] = a.w1[(4 * s + kq) * 32 + l15];
#pragma unroll
    for (int s = 0; s < 23; s++) B2[s] = a.w2[(4 * s + kq) * 32 + l15];
#pragma unroll
    for (int s = 0; s < 8; s++) BH[s] = a.wh[(4 * s + kq) * 32 + l15];
    const float bias1 = a.b1[l15], slope1 = a.s1[l15];   // vectors are zero padded to 128 floats
    const float bias2 = a.b2[l15], slope2 = a.s2[l15];
    const float bias3 = a.b3[l31], slope3 = a.s3[l31];
    const float biash = a.bh[l15];

    // LDS beyond the live tiles is read by zero-weight k padding: it must hold finite values
    for (int i = tid; i < REGION_A; i += 256) RA[i] = 0.f;
    for (int i = tid; i < REGION_B; i += 256) RB[i] = 0.f;
    __syncthreads();

    // conv1 rows: i = pool cell (i>>2) x sub-position (i&3) of a 4-cell group
    const int c1_pc = l15 >> 2, c1_dy = (l15 >> 1) & 1, c1_dx = l15 & 1;
    const int e1_1 = kq >= 1 ? 117 : 0, e1_2 = kq >= 2 ? 117 : 0, e1_3 = kq >= 3 ? 117 : 0;   // conv1: D = 9,  E = 126-9
    const int e2_2 = kq >= 2 ? 170 : 0;                                                      // conv2: D = 30, E = 200-30

    const int total_tiles = a.tiles_per_frame * a.n_frames;
    // XCD-aware persistent schedule: blocks sharing blockIdx%8 (one XCD) walk one contiguous 1/8 of the tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
    const int chunk = (total_tiles + 7) / 8;
    const int t_begin = xcd * chunk, t_end = (t_begin + chunk < total_tiles) ? t_begin + chunk : total_tiles;

    for (int tile = t_begin + slot; tile < t_end; tile += per_xcd_blocks) {
        const int f = tile / a.tiles_per_frame;
        const int tt = tile - f * a.tiles_per_frame;
        int l = 0;
        while (l + 1 < a.L && tt >= a.lv[l + 1].tile0) l++;
        const PLevel& g = a.lv[l];
        const int tq = tt - g.tile0;
        const int ty = tq / g.tiles_x, tx = tq - ty * g.tiles_x;

        // ---- phase 0: input tile -> RA as [42][42][3] ------------------------------------------------------
        if (!(a.dbg_skip & 1)) {
            const float4* src = a.pyr + (long long)f * a.pyr_stride + g.pix0;
            const int gy0 = ty * 2 * TS, gx0 = tx * 2 * TS;
            for (int p = tid; p < IN_T * IN_T; p += 256) {
                const int iy = p / IN_T, ix = p - iy * IN_T;
                const int gy = gy0 + iy, gx = gx0 + ix;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy < g.h && gx < g.w) v = src[(long long)gy * g.w + gx];
                RA[3 * p + 0] = v.x; RA[3 * p + 1] = v.y; RA[3 * p + 2] = v.z;
            }
        }
        __syncthreads();

        // ---- phase 1: conv1 + PReLU + 2x2 ceil max-pool -> RB as [20][20][10] ------------------------------
        if (!(a.dbg_skip & 2)) {
            const int vy = g.h - 2 - ty * 2 * TS, vx = g.w - 2 - tx * 2 * TS;   // valid conv1 extent inside the tile
            // 100 M-tiles (5 groups of 4 pool cells per pooled row); two independent accumulators per wave
#pragma unroll 1
            for (int j = 0; j < 13; j++) {
                const int mtA = wave + 8 * j, mtB = mtA + 4;
                const bool hasB = mtB < 100;
                const int pyA = mtA / 5, pgA = mtA - pyA * 5;
                const int pyB = hasB ? mtB / 5 : pyA, pgB = hasB ? mtB - (mtB / 5) * 5 : pgA;
                const int baseA = ((2 * pyA + c1_dy) * IN_T + 2 * (4 * pgA + c1_pc) + c1_dx) * 3;
                const int baseB = ((2 * pyB + c1_dy) * IN_T + 2 * (4 * pgB + c1_pc) + c1_dx) * 3;
                f32x4 accA = {bias1, bias1, bias1, bias1}, accB = accA;
                float xa[7], xb[7];     // all A operands of the pair are in flight before the first MFMA
#pragma unroll
                for (int s = 0; s < 7; s++) {
                    const int ko = koff<9, 117>(s, kq, e1_1, e1_2, e1_3);
                    xa[s] = RA[baseA + ko];
                    xb[s] = RA[baseB + ko];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMA chain (counted lgkmcnt waits follow)
#pragma unroll
                for (int s = 0; s < 7; s++) {
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B1[s], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[s], B1[s], accB, 0, 0, 0);
                }
                // epilogue: lane holds channel l15 of pool cell kq, the 4 registers are its 2x2 window
                if (l15 < 10) {
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        if (t == 1 && !hasB) break;
                        const f32x4 acc = t ? accB : accA;
                        const int py = t ? pyB : pyA, px = 4 * (t ? pgB : pgA) + kq;
                        float m = -INFINITY;
                        bool any = false;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
                            if (y < vy && x < vx) { const float v = prelu(acc[q], slope1); m = v > m ? v : m; any = true; }
                        }
                        RB[(py * P1_T + px) * 10 + l15] = any ? m : 0.f;
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: conv2 + PReLU -> RA as [324][17] ---------------------------------------------------------
        if (!(a.dbg_skip & 4)) {
            // 21 M-tiles of 16 rows (last one 4 rows); wave w takes pairs (w, w+4), (w+8, w+12), (w+16, w+20).
            // RA (the input tile) is dead since the barrier above, so the epilogue may overwrite it.
#pragma unroll 1
            for (int j = 0; j < 3; j++) {
                const int mtA = wave + 8 * j, mtB = mtA + 4;
                const bool hasB = mtB < 21;
                int mA = mtA * 16 + l15; mA = mA < 324 ? mA : 323;
                int mB = (hasB ? mtB : mtA) * 16 + l15; mB = mB < 324 ? mB : 323;
                const int yA = mA / C2_T, xA = mA - yA * C2_T, yB = mB / C2_T, xB = mB - yB * C2_T;
                const int baseA = (yA * P1_T + xA) * 10, baseB = (yB * P1_T + xB) * 10;
                f32x4 accA = {bias2, bias2, bias2, bias2}, accB = accA;
                float xa[23], xb[23];
#pragma unroll
                for (int s = 0; s < 23; s++) {
                    const int ko = koff<30, 170>(s, kq, 0, e2_2, 0);
                    xa[s] = RB[baseA + ko];
                    xb[s] = RB[baseB + ko];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 23; s++) {
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], B2[s], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[s], B2[s], accB, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ra = mtA * 16 + kq * 4 + q;
                    if (ra < 324) RA[ra * C2_LD + l15] = prelu(accA[q], slope2);
                    const int rb = mtB * 16 + kq * 4 + q;
                    if (hasB && rb < 324) RA[rb * C2_LD + l15] = prelu(accB[q], slope2);
                }
            }
        }
        __syncthreads();

        // ---- phase 3: conv3 + PReLU -> per-wave staging -> heads -> candidates -------------------------------
        if (!(a.dbg_skip & 8)) {
            float* ST = RB + wave * 32 * ST_LD;
            const float fscale = g.scale;
#pragma unroll 1
            for (int it = 0; it < 2; it++) {
                const int mt = wave + 4 * it;               // 8 M-tiles of 32 rows = 2 output rows each
                const int y = mt * 2 + (l31 >> 4), x = l31 & 15;
                const int base = (y * C2_T + x) * C2_LD + hh;
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; q++) acc[q] = bias3;
#pragma unroll
                for (int half = 0; half < 3; half++) {      // 3 x 24 k-steps: the operands of a third are all in flight first
                    float xa[24];
#pragma unroll
                    for (int u = 0; u < 24; u++) {
                        const int s = half * 24 + u, tap = s >> 3, ky = tap / 3, kx = tap - ky * 3;
                        xa[u] = RA[base + (ky * C2_T + kx) * C2_LD + 2 * (s & 7)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 24; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u], B3[half * 24 + u], acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
                    ST[row * ST_LD + l31] = prelu(acc[q], slope3);
                }
                __builtin_amdgcn_wave_barrier();
                // heads: two 16-row M-tiles, K = 32
                f32x4 hA = {biash, biash, biash, biash}, hB = hA;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const float xa = ST[l15 * ST_LD + 4 * s + kq];
                    const float xb = ST[(16 + l15) * ST_LD + 4 * s + kq];
                    hA = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, BH[s], hA, 0, 0, 0);
                    hB = __builtin_amdgcn_mfma_f32_16x16x4f32(xb, BH[s], hB, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
                // lane (n = l15, rows kq*4+q): n=0,1 class logits, n=2..5 box regression
                const int lbase = lane & ~15;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const f32x4 hv = t ? hB : hA;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float v = hv[q];
                        const float l0 = __shfl(v, lbase + 0, 64);
                        const float r0 = __shfl(v, lbase + 2, 64), r1 = __shfl(v, lbase + 3, 64);
                        const float r2 = __shfl(v, lbase + 4, 64), r3 = __shfl(v, lbase + 5, 64);
                        if (l15 == 1) {
                            const int row = t * 16 + kq * 4 + q;            // row inside the 32-row tile
                            const int oy = ty * TS + mt * 2 + (row >> 4), ox = tx * TS + (row & 15);
                            if (oy < g.oh && ox < g.ow) {
                                const float p = trl_softmax2_p1(l0, v);
                                if (p >= a.thr) {
                                    const int seg = f * a.L + l;
                                    const int sl = atomicAdd(&a.lvl_cnt[seg], 1);
                                    if (sl < a.cap) {
                                        Cand c;
                                        c.x1 = floorf((2.f * (float)ox + 1.f) / fscale);
                                        c.y1 = floorf((2.f * (float)oy + 1.f) / fscale);
                                        c.x2 = floorf((2.f * (float)ox + 12.f) / fscale);
                                        c.y2 = floorf((2.f * (float)oy + 12.f) / fscale);
                                        c.score = p;
                                        c.r0 = r0; c.r1 = r1; c.r2 = r2; c.r3 = r3;
                                        c.cell = oy * g.ow + ox;
                                        a.lvl_rec[(size_t)seg * a.cap + sl] = c;
                                    } else {
                                        a.flags[0] = 1;
                                    }
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();   // RA / RB are rewritten by the next tile
    }
}

}  // namespace

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
    return TRL_OK;
}

static int fill_args(trl_ctx* c, int n, int H, int W, PnetArgs& a, std::vector<uint32_t>* tab = nullptr) {
    const int L = trl_compute_levels(c, H, W);
    if (L > 16) { trl_set_error("more than 16 pyramid levels"); return TRL_ERR_INVALID; }
    a.n_frames = n; a.L = L; a.H = H; a.W = W;
    int tiles = 0, ntab = 0; long long pix = 0, work = 0;
    for (int l = 0; l < L; l++) {
        const LevelGeom& g = c->lv[l];
        PLevel& p = a.lv[l];
        p.h = g.h; p.w = g.w; p.oh = g.oh; p.ow = g.ow;
        p.tiles_x = (g.ow + TS - 1) / TS;
        p.tile0 = tiles;
        tiles += p.tiles_x * ((g.oh + TS - 1) / TS);
        p.pix0 = (int)pix;
        p.pix_pad = (int)(((long long)g.h * g.w + 63) & ~63ll);
        pix += p.pix_pad;
        // lanes per output pixel: keep a lane's share of the bin near <= 32 dwords
        const int kh = (H + g.h - 1) / g.h + 1, ndw = (((W + g.w - 1) / g.w + 1) * 3 + 6) / 4;
        const int dwords = kh * ndw;
        p.gshift = dwords <= 40 ? 0 : (dwords <= 160 ? 2 : (dwords <= 640 ? 4 : 6));
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
    }
    a.tiles_per_frame = tiles;
    a.pyr_stride = pix;
    a.work_per_frame = work;
    a.w1 = trl_w(c, "pnet.conv1.w")->p; a.w2 = trl_w(c, "pnet.conv2.w")->p; a.w3 = trl_w(c, "pnet.conv3.w")->p; a.wh = trl_w(c, "pnet.heads.w")->p;
    a.b1 = trl_v(c, "pnet.conv1.b")->p; a.b2 = trl_v(c, "pnet.conv2.b")->p; a.b3 = trl_v(c, "pnet.conv3.b")->p; a.bh = trl_v(c, "pnet.heads.b")->p;
    a.s1 = trl_v(c, "pnet.prelu1")->p; a.s2 = trl_v(c, "pnet.prelu2")->p; a.s3 = trl_v(c, "pnet.prelu3")->p;
    a.thr = c->cfg.thr0; a.cap = c->cfg.cap_level;
    { const char* e = getenv("TRL_PNET_SKIP"); a.dbg_skip = e ? atoi(e) : 0; }
    a.lvl_cnt = c->cb.lvl_cnt; a.lvl_rec = c->cb.lvl_rec; a.flags = c->cb.flags;
    return TRL_OK;
}

size_t trl_pnet_fused_bytes(trl_ctx* c, int n, int H, int W) {
    PnetArgs a;
    if (fill_args(c, n, H, W, a) != TRL_OK) return 0;
    return (size_t)a.pyr_stride * n * sizeof(float4) + 4096;
}

// All pyramid levels of all n frames: pyramid kernel + one persistent fused launch.
// ev[0..1] bracket the pyramid kernel, ev[2..3] the fused kernel (HIP events on the same stream).
int trl_pnet_fused_all(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, hipEvent_t* ev, hipStream_t s) {
    if (((uintptr_t)d_frames & 3) != 0) { trl_set_error("frame buffer must be 4-byte aligned"); return TRL_ERR_INVALID; }
    if (H > 16383 || W > 16383) { trl_set_error("frame larger than 16383 px"); return TRL_ERR_INVALID; }
    PnetArgs a;
    std::vector<uint32_t> tab;
    const bool new_shape = (c->pyr_tab == nullptr || c->pyr_tab_H != H || c->pyr_tab_W != W);
    TRL_CHECK(fill_args(c, n, H, W, a, new_shape ? &tab : nullptr));
    if (new_shape) {   // bin-edge tables depend on (H, W) only: built once per frame shape
        TRL_HIP(hipStreamSynchronize(s));
        if (c->pyr_tab) TRL_HIP(hipFree(c->pyr_tab));
        c->pyr_tab = nullptr;
        TRL_HIP(hipMalloc((void**)&c->pyr_tab, tab.size() * 4 + 64));
        TRL_HIP(hipMemcpy(c->pyr_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        c->pyr_tab_H = H; c->pyr_tab_W = W;
    }
    float4* pyr = (float4*)c->scratch.alloc((size_t)a.pyr_stride * n * sizeof(float4));
    if (!pyr) { trl_set_error("pyramid workspace"); return TRL_ERR_STATE; }
    a.pyr = pyr;
    int blocks = (int)((a.work_per_frame + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (ev) TRL_HIP(hipEventRecord(ev[0], s));
    k_pyramid<<<dim3(blocks, n), 256, 0, s>>>(d_frames, a, c->pyr_tab, pyr);
    TRL_LAUNCH_CHECK();
    if (ev) { TRL_HIP(hipEventRecord(ev[1], s)); TRL_HIP(hipEventRecord(ev[2], s)); }
    const int total_tiles = a.tiles_per_frame * n;
    int grid = 256 * 2;                       // 2 resident workgroups per CU (<= 256 VGPRs)
    if (grid > ((total_tiles + 7) / 8) * 8) grid = ((total_tiles + 7) / 8) * 8;
    if (grid < 8) grid = 8;
    k_pnet_fused<<<grid, 256, 0, s>>>(a);
    TRL_LAUNCH_CHECK();
    if (ev) TRL_HIP(hipEventRecord(ev[3], s));
    return TRL_OK;
}
