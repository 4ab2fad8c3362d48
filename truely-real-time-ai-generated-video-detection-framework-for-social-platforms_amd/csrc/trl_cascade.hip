// trl_cascade.hip -- the data-dependent part of MTCNN.detect on the device (gfx950).
//
// Restates facenet_pytorch 2.6.0 utils/detect_face.py::detect_face + MTCNN.detect
// (select_largest=True) as called at server/model.py:47, and model.py:49-58 (int cast, clamp,
// crop, cv2.resize INTER_LINEAR u8, to_tensor).  Everything stays on the GPU; candidates live in
// fixed-capacity per-frame lists.  One workgroup owns one (frame, level) or one frame segment:
//   sort   : bitonic sort in LDS on a 64-bit key  (~score | tie-break index) -> exactly the order of
//            a stable descending sort, so results do not depend on the order candidates were
//            appended by the PNet kernel's atomics;
//   NMS    : greedy suppression in sorted order, all lanes test one kept box against the rest;
//   compact: ordered compaction by block scan, so list order equals the reference's `pick` order.
// All float expressions follow the reference's operation order (one rounding per op,
// -ffp-contract=off), so boxes / keep masks are bit-identical to the oracle.
#include "trl_ctx.h"
#include <stdlib.h>
#include <string.h>

// R-/O-Net candidates per launch set: one set covers a 256-frame batch's CAPACITY (160 / 48 candidates per frame + 64: 41 k / 12.4 k;
// 31 k / 8.5 k are live), so no dead second chunk is launched (8 launches x 4.7 us); scratch per chunk = 100 KB / 640 KB per
// candidate (5 / 10.7 GB at the cap, of 288 GB)
// (trl_debug_option("rnet_chunk" / "onet_chunk") shrinks a launch set for tests, so that the multi-chunk path runs on small inputs)

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {   // ascending-order preserving map
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- block primitives ------------------------------------------------------------------------
__device__ void block_bitonic(uint64_t* key, uint32_t* id, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P; t += blockDim.x) {
                const int u = t ^ j;
                if (u > t) {
                    const bool up = ((t & k) == 0);
                    const uint64_t a = key[t], b = key[u];
                    if ((a > b) == up) {
                        key[t] = b; key[u] = a;
                        const uint32_t ia = id[t]; id[t] = id[u]; id[u] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ int next_pow2(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

// Does kept box i suppress box j?  MIN_MODE = nms_numpy 'Min' (+1 extents, inter / min(area), keep o <= thr); otherwise torchvision
// nms (inter / (a_i + a_j - inter) > thr suppresses).  One rounding per operation, in the reference's order.
template <bool MIN_MODE>
__device__ __forceinline__ bool nms_suppresses(const float4 bi, const float ai, const float4 bj, const float aj, const float thr) {
    const float xx1 = bi.x > bj.x ? bi.x : bj.x;
    const float yy1 = bi.y > bj.y ? bi.y : bj.y;
    const float xx2 = bi.z < bj.z ? bi.z : bj.z;
    const float yy2 = bi.w < bj.w ? bi.w : bj.w;
    if (MIN_MODE) {
        float w = xx2 - xx1 + 1.f; w = w > 0.f ? w : 0.f;
        float h = yy2 - yy1 + 1.f; h = h > 0.f ? h : 0.f;
        const float inter = w * h;
        const float mn = ai < aj ? ai : aj;
        const float o = inter / mn;
        return !(o <= thr);
    } else {
        float w = xx2 - xx1; w = w > 0.f ? w : 0.f;
        float h = yy2 - yy1; h = h > 0.f ? h : 0.f;
        const float inter = w * h;
        const float ovr = inter / (ai + aj - inter);
        return ovr > thr;
    }
}

// Greedy NMS over boxes already in descending-score order.  MIN_MODE = facenet_pytorch
// nms_numpy(..., 'Min') (+1 areas, inter/min(area), keep o <= thr); otherwise torchvision nms
// (inter/(a_i+a_j-inter) > thr suppresses).  keep[] receives positions in pick order.
// `preset`: sup[] already holds suppressions by boxes that precede this list (the spill tier's earlier chunks).
// The list is walked in sub-blocks of 64 boxes: wave 0 resolves a sub-block by itself -- lane = box, the next live box is
// broadcast with v_readlane, no barrier between the 64 dependent steps -- then every thread applies the sub-block's kept boxes
// to its share of the later boxes.  Two barriers per 64 boxes instead of one per kept box; a box is kept iff no earlier kept box
// suppresses it, exactly the sequential rule.
template <bool MIN_MODE>
__device__ int block_nms(const float4* box, const float* area, int n, float thr, uint8_t* sup, int* keep, int* nkeep, bool preset = false) {
    __shared__ unsigned long long kept_s;
    if (!preset) for (int t = threadIdx.x; t < n; t += blockDim.x) sup[t] = 0;
    if (threadIdx.x == 0) *nkeep = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int b0 = 0; b0 < n; b0 += 64) {
        const int nb = n - b0 < 64 ? n - b0 : 64;
        if (threadIdx.x < 64) {
            const bool have = lane < nb;
            const float4 bm = have ? box[b0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float am = have ? area[b0 + lane] : 0.f;
            bool dead = !have || sup[b0 + lane];
            const int base = *nkeep;
            unsigned long long kept = 0;
            for (int q = 0; q < nb; q++) {
                if ((__ballot(dead) >> q) & 1) continue;          // wave-uniform: box q was suppressed (or preset)
                kept |= 1ull << q;
                const float4 bq = make_float4(__shfl(bm.x, q), __shfl(bm.y, q), __shfl(bm.z, q), __shfl(bm.w, q));
                const float aq = __shfl(am, q);
                if (lane > q && !dead && nms_suppresses<MIN_MODE>(bq, aq, bm, am, thr)) dead = true;
            }
            if ((kept >> lane) & 1) keep[base + __popcll(kept & ((1ull << lane) - 1ull))] = b0 + lane;
            if (lane == 0) { kept_s = kept; *nkeep = base + __popcll(kept); }
        }
        __syncthreads();
        const unsigned long long kept = kept_s;
        for (int j = b0 + 64 + threadIdx.x; j < n; j += blockDim.x) {
            if (sup[j]) continue;
            const float4 bj = box[j];
            const float aj = area[j];
            bool s = false;
            for (unsigned long long m = kept; m; m &= m - 1) {
                const int q = b0 + __ffsll((long long)m) - 1;
                s |= nms_suppresses<MIN_MODE>(box[q], area[q], bj, aj, thr);
            }
            if (s) sup[j] = 1;
        }
        __syncthreads();
    }
    return *nkeep;
}

// exclusive scan of 0/1 flags (n <= capacity); returns total, pos[] = output slot
__device__ int block_compact_positions(const uint8_t* flag, int n, int* pos, int* part /*[blockDim.x + 1]*/) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int per = (n + T - 1) / T;
    const int b = tid * per, e = (b + per < n) ? b + per : n;
    int cnt = 0;
    for (int i = b; i < e; i++) cnt += flag[i];
    part[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < T; i++) { const int v = part[i]; part[i] = run; run += v; }
        part[T] = run;
    }
    __syncthreads();
    int o = part[tid];
    for (int i = b; i < e; i++) { pos[i] = o; o += flag[i]; }
    __syncthreads();
    return part[T];
}

struct Smem {   // carve the dynamic LDS for capacity `cap`
    uint64_t* key; uint32_t* id; float4* box; float* area; float* aux; int* keep; int* pos; uint8_t* sup; uint8_t* flg;
    int* part; int* scal;
    __device__ Smem(unsigned char* base, int cap, int threads = 256) {
        key = (uint64_t*)base;                 // 8
        box = (float4*)(key + cap);            // 16
        id = (uint32_t*)(box + cap);           // 4
        area = (float*)(id + cap);             // 4
        aux = area + cap;                      // 4
        keep = (int*)(aux + cap);              // 4
        pos = keep + cap;                      // 4
        part = pos + cap;                      // threads + 4 ints
        scal = part + threads + 4;             // 8 ints
        sup = (uint8_t*)(scal + 8);            // 1
        flg = sup + cap;                       // 1
    }
    static size_t bytes(int cap, int threads = 256) { return (size_t)cap * 46 + (size_t)(threads + 12) * 4 + 64; }
};

// ---- spill tier: lists longer than the LDS tier ----------------------------------------------------------------------------
// detect_face() has no candidate limit (torchvision's nms and nms_numpy take any number of boxes), so neither has this cascade:
// a list that does not fit the LDS tier is sorted and suppressed by the SAME workgroup in global memory --
//   sort   : the same bitonic network on the same 64-bit keys, compare-exchange distances below the LDS chunk run in LDS,
//            the longer ones in global memory;
//   NMS    : greedy suppression in chunks of the sorted order: a chunk is first tested against every box kept from earlier
//            chunks, then suppressed in LDS like a short list.  A box is kept iff no earlier kept box suppresses it -- the
//            sequential rule, so the pick order and every keep decision equal the LDS tier's (and the reference's).
// Workspace comes from a bump pool in the cascade arena (one atomic per list); a pool that runs out raises a flag and the call is
// re-run with a larger one, like every other capacity (trl_cascade_check).
struct Spill { char* base; unsigned long long cap; int32_t* flags; };
enum { FLG_LEVEL = 0, FLG_FRAME = 1, FLG_T2 = 2, FLG_T3 = 3, FLG_T2N = 4, FLG_T3N = 5, FLG_FRAME_MAX = 6, FLG_SPILL = 7,
       FLG_SPILL_CUR = 8 /* u64 */, FLG_SPILL_LISTS = 10, FLG_LEVEL_MAX = 16 /* [32] */ };

__device__ char* spill_alloc(const Spill& sp, size_t bytes) {   // every thread of the workgroup calls it; uniform result
    __shared__ unsigned long long off_s;
    bytes = (bytes + 255) & ~(size_t)255;
    if (threadIdx.x == 0) {
        off_s = atomicAdd(reinterpret_cast<unsigned long long*>(sp.flags + FLG_SPILL_CUR), (unsigned long long)bytes);
        atomicAdd(&sp.flags[FLG_SPILL_LISTS], 1);
        if (off_s + bytes > sp.cap) sp.flags[FLG_SPILL] = 1;      // the cursor keeps counting: the host learns the total need
    }
    __syncthreads();
    const unsigned long long o = off_s;
    __syncthreads();
    return o + bytes <= sp.cap ? sp.base + o : nullptr;
}
__device__ __forceinline__ int pow2_floor(int n) {
    int p = 1;
    while (2 * p <= n) p <<= 1;
    return p;
}
// compare-exchange distances j0, j0/2, .., 1 of merge width k on the LDS-resident chunk [base, base + C) of the list
__device__ void lds_bitonic_steps(uint64_t* key, uint32_t* id, int C, int base, int k, int j0) {
    for (int j = j0; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < C; t += blockDim.x) {
            const int u = t ^ j;
            if (u > t) {
                const bool up = (((base + t) & k) == 0);
                const uint64_t a = key[t], b = key[u];
                if ((a > b) == up) {
                    key[t] = b; key[u] = a;
                    const uint32_t ia = id[t]; id[t] = id[u]; id[u] = ia;
                }
            }
        }
        __syncthreads();
    }
}
// ascending sort of gk[0..P) (P a power of two, padded with ~0 keys) with payload gi, by one workgroup; lk / li: LDS for C entries
__device__ void big_bitonic(uint64_t* gk, uint32_t* gi, int P, uint64_t* lk, uint32_t* li, int C) {
    if (C > P) C = P;
    for (int base = 0; base < P; base += C) {
        for (int t = threadIdx.x; t < C; t += blockDim.x) { lk[t] = gk[base + t]; li[t] = gi[base + t]; }
        __syncthreads();
        for (int k = 2; k <= C; k <<= 1) lds_bitonic_steps(lk, li, C, base, k, k >> 1);
        for (int t = threadIdx.x; t < C; t += blockDim.x) { gk[base + t] = lk[t]; gi[base + t] = li[t]; }
        __syncthreads();
    }
    for (int k = 2 * C; k <= P && k > 0; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) {
            for (int t = threadIdx.x; t < P; t += blockDim.x) {
                const int u = t ^ j;
                if (u > t) {
                    const bool up = ((t & k) == 0);
                    const uint64_t a = gk[t], b = gk[u];
                    if ((a > b) == up) {
                        gk[t] = b; gk[u] = a;
                        const uint32_t ia = gi[t]; gi[t] = gi[u]; gi[u] = ia;
                    }
                }
            }
            __syncthreads();
        }
        if (C > 1) {
            for (int base = 0; base < P; base += C) {
                for (int t = threadIdx.x; t < C; t += blockDim.x) { lk[t] = gk[base + t]; li[t] = gi[base + t]; }
                __syncthreads();
                lds_bitonic_steps(lk, li, C, base, k, C >> 1);
                for (int t = threadIdx.x; t < C; t += blockDim.x) { gk[base + t] = lk[t]; gi[base + t] = li[t]; }
                __syncthreads();
            }
        }
    }
}
// Greedy NMS of the first m entries of the sorted list (payload gi -> box through box_of).  kbox / karea / kid (global, m entries)
// receive the kept boxes, their areas and payloads in pick order.  S: LDS carve with room for C entries.  Returns the kept count.
//
// A chunk is first tested against the boxes kept from earlier chunks.  IoU mode (grid != null): through a uniform grid over the
// kept boxes' CENTRES (32-px cells, linked lists in global memory: head[cell], next[kept]).  A kept box i can only suppress a
// candidate j if they intersect AND inter > thr * area_i, hence w_i < w_j / thr and h_i < h_j / thr, hence
// |cx_i - cx_j| < w_j (1 + 1/thr) / 2 (same in y): the candidate walks the cells of that window only -- the pairs it skips could
// not have suppressed it, the pairs it tests are tested exactly as before, so every decision is the brute-force one.  Cell
// indices are clamped to the grid on both sides (boxes regressed past the frame land in the border cells).  'Min' mode has no such
// bound (a box of any size suppresses the boxes inside it): it tests against every kept box, staged through LDS in tiles.
struct KeptGrid { int* head; int* next; int gx, gy; };
constexpr int GRID_SHIFT = 5;
__device__ __forceinline__ int grid_clamp(float v, int n) {
    int c = (int)floorf(v * (1.0f / (1 << GRID_SHIFT)));
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
template <bool MIN_MODE, class BoxOf>
__device__ int big_greedy(const Smem& S, int C, const uint32_t* gi, int m, float thr, BoxOf box_of, float4* kbox, float* karea,
                          uint32_t* kid, KeptGrid grid) {
    // 'Min' mode: kept boxes of earlier chunks are staged through LDS in tiles (the key area is idle during NMS: 8 C bytes = TK boxes + areas)
    const int TK = (C * 8) / 20;
    float4* tb = (float4*)S.key;
    float* ta = (float*)(tb + TK);
    constexpr int E = 2;                                  // entries of a chunk per thread: the spill tier runs with 1024 threads (C <= 2048)
    const bool use_grid = !MIN_MODE && grid.head != nullptr;
    if (use_grid) {
        for (int t = threadIdx.x; t < grid.gx * grid.gy; t += blockDim.x) grid.head[t] = -1;
        __syncthreads();
    }
    const float reach = 0.5f * (1.0f + 1.0f / thr);       // window half-size in units of the candidate's own extent (+ 1 px below)
    int K = 0;
    for (int base = 0; base < m; base += C) {
        const int nc = m - base < C ? m - base : C;
        float4 bt[E]; float at[E]; bool sp[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int t = threadIdx.x + e * blockDim.x;
            sp[e] = true;                                 // (no entry)
            if (t < nc) {
                const uint32_t id = gi[base + t];
                const float4 b = box_of(id);
                bt[e] = b;
                at[e] = MIN_MODE ? (b.z - b.x + 1.f) * (b.w - b.y + 1.f) : (b.z - b.x) * (b.w - b.y);
                S.box[t] = b; S.area[t] = at[e]; S.id[t] = id;
                sp[e] = false;
            }
        }
        if (use_grid) {
            if (K > 0) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (sp[e]) continue;
                    const float4 b = bt[e];
                    const float w = b.z - b.x, h = b.w - b.y;
                    if (!(w > 0.f) || !(h > 0.f)) continue;                   // an empty box intersects nothing: never suppressed
                    const float cx = 0.5f * (b.x + b.z), cy = 0.5f * (b.y + b.w), rx = w * reach + 1.f, ry = h * reach + 1.f;
                    const int x0 = grid_clamp(cx - rx, grid.gx), x1 = grid_clamp(cx + rx, grid.gx);
                    const int y0 = grid_clamp(cy - ry, grid.gy), y1 = grid_clamp(cy + ry, grid.gy);
                    bool s = false;
                    for (int gy = y0; gy <= y1 && !s; gy++)
                        for (int gx = x0; gx <= x1 && !s; gx++)
                            for (int k = __hip_atomic_load(&grid.head[gy * grid.gx + gx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); k >= 0 && !s; k = grid.next[k])   // (head is updated by atomics at L2: read it there)
                                s = nms_suppresses<false>(kbox[k], karea[k], b, at[e], thr);
                    sp[e] = s;
                }
            }
        } else {
            // boxes kept from earlier chunks: every thread tests its own entries of the chunk against a tile of them at a time
            for (int k0 = 0; k0 < K; k0 += TK) {
                const int nt = K - k0 < TK ? K - k0 : TK;
                __syncthreads();
                for (int i = threadIdx.x; i < nt; i += blockDim.x) { tb[i] = kbox[k0 + i]; ta[i] = karea[k0 + i]; }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (sp[e]) continue;
                    bool s = false;
#pragma unroll 4
                    for (int k = 0; k < nt; k++) s |= nms_suppresses<MIN_MODE>(tb[k], ta[k], bt[e], at[e], thr);
                    sp[e] = s;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int t = threadIdx.x + e * blockDim.x;
            if (t < nc) S.sup[t] = sp[e] ? 1 : 0;
        }
        __syncthreads();
        const int nk = block_nms<MIN_MODE>(S.box, S.area, nc, thr, S.sup, S.keep, S.scal, true);
        for (int r = threadIdx.x; r < nk; r += blockDim.x) {
            const int t = S.keep[r];
            const float4 b = S.box[t];
            kbox[K + r] = b; karea[K + r] = S.area[t]; kid[K + r] = S.id[t];
            if (use_grid) {                               // register the kept box under its centre's cell
                const int cell = grid_clamp(0.5f * (b.y + b.w), grid.gy) * grid.gx + grid_clamp(0.5f * (b.x + b.z), grid.gx);
                grid.next[K + r] = atomicExch(&grid.head[cell], K + r);
            }
        }
        K += nk;
        __syncthreads();
    }
    return K;
}

// rerec() of one box
__device__ __forceinline__ void rerec1(float& x1, float& y1, float& x2, float& y2) {
    const float h = y2 - y1, w = x2 - x1;
    const float l = w > h ? w : h;
    x1 = x1 + w * 0.5f - l * 0.5f;
    y1 = y1 + h * 0.5f - l * 0.5f;
    x2 = x1 + l;
    y2 = y1 + l;
}
// pad(): trunc, clamp.  ok = the reference's `ey > y-1 and ex > x-1`
__device__ __forceinline__ bool pad1(float x1, float y1, float x2, float y2, int W, int H, int& y, int& ey, int& x, int& ex) {
    const int bx = (int)truncf(x1), by = (int)truncf(y1), bex = (int)truncf(x2), bey = (int)truncf(y2);
    x = bx < 1 ? 1 : bx;
    y = by < 1 ? 1 : by;
    ex = bex > W ? W : bex;
    ey = bey > H ? H : bey;
    return (ey > y - 1) && (ex > x - 1);
}

// ---- imresample (F.interpolate mode="area") of whole frames to one pyramid level --------------
__global__ void k_area_level(const uint8_t* __restrict__ frames, int nf, int H, int W, int h, int w, float* __restrict__ out) {
    const size_t total = (size_t)nf * h * w;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % w);
        const int oy = (int)((idx / w) % h);
        const int f = (int)(idx / ((size_t)w * h));
        const int ys = (int)(((long long)oy * H) / h), ye = (int)((((long long)oy + 1) * H + h - 1) / h);
        const int xs = (int)(((long long)ox * W) / w), xe = (int)((((long long)ox + 1) * W + w - 1) / w);
        unsigned s0 = 0, s1 = 0, s2 = 0;
        const uint8_t* fp = frames + (size_t)f * H * W * 3;
        for (int y = ys; y < ye; y++) {
            const uint8_t* p = fp + ((size_t)y * W + xs) * 3;
            for (int x = xs; x < xe; x++, p += 3) { s0 += p[0]; s1 += p[1]; s2 += p[2]; }
        }
        const float kh = (float)(ye - ys), kw = (float)(xe - xs);
        float* o = out + idx * 3;
        o[0] = ((float)s0 / kh / kw - 127.5f) * 0.0078125f;
        o[1] = ((float)s1 / kh / kw - 127.5f) * 0.0078125f;
        o[2] = ((float)s2 / kh / kw - 127.5f) * 0.0078125f;
    }
}

// generateBoundingBox on the generic path: heads [nf][oh][ow][6] -> candidate records
__global__ void k_pnet_collect(const float* __restrict__ heads, int nf, int f0, int oh, int ow, float scale, float thr,
                               int L, int l, int cap, int rec0, int S, int32_t* __restrict__ lvl_cnt, Cand* __restrict__ lvl_rec,
                               int32_t* __restrict__ flags) {
    const size_t total = (size_t)nf * oh * ow;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const float* hd = heads + idx * 6;
        const float p = trl_softmax2_p1(hd[0], hd[1]);
        if (!(p >= thr)) continue;
        const int cell = (int)(idx % ((size_t)oh * ow));
        const int f = f0 + (int)(idx / ((size_t)oh * ow));
        const int y = cell / ow, x = cell - y * ow;
        const int slot = atomicAdd(&lvl_cnt[f * L + l], 1);
        if (slot >= cap) { flags[0] = 1; continue; }
        Cand c;
        c.x1 = floorf((2.f * (float)x + 1.f) / scale);
        c.y1 = floorf((2.f * (float)y + 1.f) / scale);
        c.x2 = floorf((2.f * (float)x + 12.f) / scale);
        c.y2 = floorf((2.f * (float)y + 12.f) / scale);
        c.score = p;
        c.r0 = hd[2]; c.r1 = hd[3]; c.r2 = hd[4]; c.r3 = hd[5];
        c.cell = cell;
        lvl_rec[(size_t)f * S + rec0 + slot] = c;
    }
}

__global__ void k_heads_to_maps(const float* __restrict__ heads, int cells, float* __restrict__ prob, float* __restrict__ reg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const float* hd = heads + (size_t)i * 6;
    prob[i] = trl_softmax2_p1(hd[0], hd[1]);
    reg[4 * i + 0] = hd[2]; reg[4 * i + 1] = hd[3]; reg[4 * i + 2] = hd[4]; reg[4 * i + 3] = hd[5];
}

// ---- stage 1a: per (frame, level) batched_nms(0.5) ------------------------------------------
// Two launches share the segments: the SMALL tier (LDS sized for `lds_cap` = 512 candidates: ~24 KB, six workgroups per CU) takes
// every segment with at most lds_cap candidates -- practically all of them: a level holds ~0.2 % of its cells -- and the FULL tier
// (LDS for 2048 candidates: 94 KB, one workgroup per CU) the crowded ones, through LDS up to its capacity and through the spill
// tier (global memory) beyond; a workgroup whose segment belongs to the other tier exits at once.  One tier for everything ran
// the 2,816 segments of a 256-frame batch in eleven rounds of 256.
__global__ __launch_bounds__(1024) void k_nms_level(LvLayout G, int lds_cap, int min_cnt, int spill_tier, int W, int H, const int32_t* __restrict__ lvl_cnt,
                                                   const Cand* __restrict__ lvl_rec,
                                                   int32_t* __restrict__ keep_cnt, int32_t* __restrict__ keep_idx,
                                                   int32_t* __restrict__ flags, Spill sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem S(smem_raw, lds_cap, blockDim.x);
    const int seg = blockIdx.x;
    const int f = seg / G.L, l = seg - f * G.L;
    const int cap = G.capl[l];
    int cnt = lvl_cnt[seg];
    // the small tier visits every segment: it reports the level's largest count (what the lists must hold: trl_cascade_check)
    if (min_cnt == 0 && threadIdx.x == 0 && cnt > 0) atomicMax(&flags[FLG_LEVEL_MAX + l], cnt);
    if (cnt > cap) { cnt = cap; if (threadIdx.x == 0) flags[FLG_LEVEL] = 1; }   // the call is re-run with longer lists
    if (cnt < min_cnt || (cnt > lds_cap && !spill_tier)) return;                  // the other tier's segment
    if (cnt == 0) { if (threadIdx.x == 0) keep_cnt[seg] = 0; return; }
    const Cand* recs = lvl_rec + (size_t)f * G.S + G.rec0[l];
    int32_t* kout = keep_idx + (size_t)f * G.S + G.rec0[l];
    if (cnt > lds_cap) {                                                          // ---- spill tier
        const int P = next_pow2(cnt);
        const int gx = (W >> GRID_SHIFT) + 1, gy = (H >> GRID_SHIFT) + 1;
        char* w = spill_alloc(sp, (size_t)P * 12 + (size_t)cnt * 24 + (size_t)gx * gy * 4);
        if (!w) { if (threadIdx.x == 0) keep_cnt[seg] = 0; return; }
        uint64_t* gk = (uint64_t*)w; uint32_t* gi = (uint32_t*)(gk + P);
        float4* kbox = (float4*)(gi + P); float* karea = (float*)(kbox + cnt);
        const KeptGrid grid{(int*)(karea + cnt) + cnt, (int*)(karea + cnt), gx, gy};
        for (int t = threadIdx.x; t < P; t += blockDim.x) {
            gk[t] = t < cnt ? (((uint64_t)(~f2ord(recs[t].score))) << 32) | (uint32_t)recs[t].cell : ~0ull;
            gi[t] = t;
        }
        __syncthreads();
        big_bitonic(gk, gi, P, S.key, S.id, pow2_floor(lds_cap));
        const int nk = big_greedy<false>(S, pow2_floor(lds_cap), gi, cnt, 0.5f,
                                         [&](uint32_t id) { const Cand& c = recs[id]; return make_float4(c.x1, c.y1, c.x2, c.y2); },
                                         kbox, karea, (uint32_t*)kout, grid);
        if (threadIdx.x == 0) keep_cnt[seg] = nk;
        return;
    }
    const int P = next_pow2(cnt);
    for (int t = threadIdx.x; t < P; t += blockDim.x) {
        S.key[t] = t < cnt ? (((uint64_t)(~f2ord(recs[t].score))) << 32) | (uint32_t)recs[t].cell : ~0ull;
        S.id[t] = t;
    }
    __syncthreads();
    block_bitonic(S.key, S.id, P);
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
        const Cand& c = recs[S.id[t]];
        S.box[t] = make_float4(c.x1, c.y1, c.x2, c.y2);
        S.area[t] = (c.x2 - c.x1) * (c.y2 - c.y1);
    }
    __syncthreads();
    const int nk = block_nms<false>(S.box, S.area, cnt, 0.5f, S.sup, S.keep, S.scal);
    for (int r = threadIdx.x; r < nk; r += blockDim.x) kout[r] = (int)S.id[S.keep[r]];
    if (threadIdx.x == 0) keep_cnt[seg] = nk;
}

// ---- stage 1b: per frame batched_nms(0.7) over all levels, regress, rerec ----------------------
// regress with the PNet offsets (w, h WITHOUT +1), rerec; ok = the clipped box is not empty
__device__ __forceinline__ bool stage1_row(const Cand& c, int W, int H, float4& box) {
    const float regw = c.x2 - c.x1, regh = c.y2 - c.y1;
    float x1 = c.x1 + c.r0 * regw, y1 = c.y1 + c.r1 * regh, x2 = c.x2 + c.r2 * regw, y2 = c.y2 + c.r3 * regh;
    rerec1(x1, y1, x2, y2);
    int y, ey, x, ex;
    box = make_float4(x1, y1, x2, y2);
    return pad1(x1, y1, x2, y2, W, H, y, ey, x, ex);
}
__global__ __launch_bounds__(1024) void k_nms_frame(LvLayout G, int lds_cap, int capF, int W, int H, const Cand* __restrict__ lvl_rec,
                                                   const int32_t* __restrict__ keep_cnt, const int32_t* __restrict__ keep_idx,
                                                   int32_t* __restrict__ n1, float* __restrict__ s1_box, int32_t* __restrict__ flags, Spill sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem S(smem_raw, lds_cap, blockDim.x);
    __shared__ int offs[33];
    const int f = blockIdx.x, L = G.L;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int l = 0; l < L; l++) { offs[l] = run; run += keep_cnt[f * L + l]; }
        offs[L] = run;
        if (run > 0) atomicMax(&flags[FLG_FRAME_MAX], run);
        if (run > capF) flags[FLG_FRAME] = 1;                                    // the call is re-run with longer per-frame lists
    }
    __syncthreads();
    const int total = offs[L];
    if (total == 0 || total > capF) { if (threadIdx.x == 0) n1[f] = 0; return; }
    const Cand* frec = lvl_rec + (size_t)f * G.S;
    const int32_t* fkeep = keep_idx + (size_t)f * G.S;
    auto fill = [&](int e, uint64_t& key, uint32_t& gid) {   // entry e of the concatenated per-level pick lists
        int l = 0;
        while (e >= offs[l + 1]) l++;
        gid = (uint32_t)(G.rec0[l] + fkeep[G.rec0[l] + (e - offs[l])]);
        key = (((uint64_t)(~f2ord(frec[gid].score))) << 32) | (uint32_t)e;   // e = position in the concatenated list
    };
    float* fout = s1_box + (size_t)f * capF * 5;
    if (total > lds_cap) {                                                        // ---- spill tier
        const int P = next_pow2(total), C = pow2_floor(lds_cap);
        const int gx = (W >> GRID_SHIFT) + 1, gy = (H >> GRID_SHIFT) + 1;
        char* w = spill_alloc(sp, (size_t)P * 12 + (size_t)total * 28 + (size_t)gx * gy * 4);
        if (!w) { if (threadIdx.x == 0) n1[f] = 0; return; }
        uint64_t* gk = (uint64_t*)w; uint32_t* gi = (uint32_t*)(gk + P);
        float4* kbox = (float4*)(gi + P); float* karea = (float*)(kbox + total); uint32_t* kid = (uint32_t*)(karea + total);
        const KeptGrid grid{(int*)(kid + total) + total, (int*)(kid + total), gx, gy};
        for (int e = threadIdx.x; e < P; e += blockDim.x) {
            uint64_t k = ~0ull; uint32_t g = 0;
            if (e < total) fill(e, k, g);
            gk[e] = k; gi[e] = g;
        }
        __syncthreads();
        big_bitonic(gk, gi, P, S.key, S.id, C);
        const int nk = big_greedy<false>(S, C, gi, total, 0.7f,
                                         [&](uint32_t id) { const Cand& c = frec[id]; return make_float4(c.x1, c.y1, c.x2, c.y2); },
                                         kbox, karea, kid, grid);
        int out = 0;                                                              // rows written so far (ordered compaction, chunk by chunk)
        for (int base = 0; base < nk; base += C) {
            const int nc = nk - base < C ? nk - base : C;
            for (int r = threadIdx.x; r < nc; r += blockDim.x) {
                const Cand& c = frec[kid[base + r]];
                float4 b;
                S.flg[r] = stage1_row(c, W, H, b) ? 1 : 0;
                S.box[r] = b; S.aux[r] = c.score;
            }
            __syncthreads();
            const int m = block_compact_positions(S.flg, nc, S.pos, S.part);
            for (int r = threadIdx.x; r < nc; r += blockDim.x) {
                if (!S.flg[r]) continue;
                float* o = fout + (size_t)(out + S.pos[r]) * 5;
                const float4 b = S.box[r];
                o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = S.aux[r];
            }
            out += m;
            __syncthreads();
        }
        if (threadIdx.x == 0) n1[f] = out;
        return;
    }
    const int P = next_pow2(total);
    for (int e = threadIdx.x; e < P; e += blockDim.x) {
        uint64_t k = ~0ull; uint32_t g = 0;
        if (e < total) fill(e, k, g);
        S.key[e] = k; S.id[e] = g;
    }
    __syncthreads();
    block_bitonic(S.key, S.id, P);
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
        const Cand& c = frec[S.id[t]];
        S.box[t] = make_float4(c.x1, c.y1, c.x2, c.y2);
        S.area[t] = (c.x2 - c.x1) * (c.y2 - c.y1);
    }
    __syncthreads();
    const int nk = block_nms<false>(S.box, S.area, total, 0.7f, S.sup, S.keep, S.scal);
    // regress with the PNet offsets (w,h WITHOUT +1), rerec, drop empty clipped boxes
    for (int r = threadIdx.x; r < nk; r += blockDim.x) {
        const Cand& c = frec[S.id[S.keep[r]]];
        float4 b;
        S.flg[r] = stage1_row(c, W, H, b) ? 1 : 0;
        S.box[r] = b;   // box[] no longer needed by NMS
        S.aux[r] = c.score;
    }
    __syncthreads();
    const int m = block_compact_positions(S.flg, nk, S.pos, S.part);
    for (int r = threadIdx.x; r < nk; r += blockDim.x) {
        if (!S.flg[r]) continue;
        float* o = fout + (size_t)S.pos[r] * 5;
        const float4 b = S.box[r];
        o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = S.aux[r];
    }
    if (threadIdx.x == 0) n1[f] = m;
}

// exclusive scan of per-frame counts; off[n] = total.  Also the candidate -> (frame, local) map.
__global__ __launch_bounds__(256) void k_scan_counts(const int32_t* __restrict__ cnt, int n, int32_t* __restrict__ off, int cap_total,
                                                     int32_t* __restrict__ flags, int slot) {
    // thread t owns the contiguous run [t per, (t+1) per) of frames: fixed LDS whatever n is (any batch check_call admits)
    __shared__ int part[257];
    const int t = threadIdx.x, per = (n + 255) / 256;
    const int b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
    int s = 0;
    for (int i = b; i < e; i++) s += cnt[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = run; run += v; }
        part[256] = run;
    }
    __syncthreads();
    int run = part[t];
    for (int i = b; i < e; i++) { off[i] = run; run += cnt[i]; }
    // the batch the next stage processes was sized by an optimistic capacity: report the real total and whether it fits
    if (t == 0) { const int total = part[256]; off[n] = total; flags[4 + slot] = total; if (total > cap_total) flags[2 + slot] = 1; }
}
// Candidate list of a stage in launch order: candidate t = off[f] + i is box i of frame f.  The record holds what the front kernel
// needs to start its crop -- frame index and pad()'s clamped window (detect_face.py pad(): x = max(x1, 1), ex = min(x2, W), crop
// [y-1:ey, x-1:ex]) -- so that kernel reaches its first pixel load after ONE dependent read instead of three (map -> box -> pixels).
__global__ void k_build_map(const int32_t* __restrict__ cnt, const int32_t* __restrict__ off, const float* __restrict__ boxes, int capF,
                            int W, int H, int32_t* __restrict__ cbox) {
    const int f = blockIdx.x;
    const int c = cnt[f], o = off[f];
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        const float* b = boxes + ((size_t)f * capF + i) * 5;
        const int bx = (int)truncf(b[0]), by = (int)truncf(b[1]), bex = (int)truncf(b[2]), bey = (int)truncf(b[3]);
        const int x = bx < 1 ? 1 : bx, y = by < 1 ? 1 : by, ex = bex > W ? W : bex, ey = bey > H ? H : bey;
        int4* r = reinterpret_cast<int4*>(cbox + (size_t)(o + i) * 8);
        r[0] = make_int4(f, y - 1, x - 1, ey - (y - 1));
        r[1] = make_int4(ex - (x - 1), 0, 0, 0);
    }
}

// ---- stage 2 tail: thr1, batched_nms(0.7), bbreg, rerec ------------------------------------------
// bbreg (+1 widths) with the R-Net offsets g, rerec; ok = the clipped box is not empty
__device__ __forceinline__ bool stage2_row(const float4 b, const float* g, int W, int H, float4& out) {
    const float w = b.z - b.x + 1.f, h = b.w - b.y + 1.f;
    float x1 = b.x + g[0] * w, y1 = b.y + g[1] * h, x2 = b.z + g[2] * w, y2 = b.w + g[3] * h;
    rerec1(x1, y1, x2, y2);
    int y, ey, x, ex;
    out = make_float4(x1, y1, x2, y2);
    return pad1(x1, y1, x2, y2, W, H, y, ey, x, ex);
}
__global__ __launch_bounds__(1024) void k_stage2_post(int lds_cap, int capF, int cap_total, int W, int H, float thr, const int32_t* __restrict__ n1,
                                                     const float* __restrict__ s1_box, const int32_t* __restrict__ off2,
                                                     const float* __restrict__ out6, int32_t* __restrict__ n2,
                                                     float* __restrict__ s2_box, Spill sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem S(smem_raw, lds_cap, blockDim.x);
    const int f = blockIdx.x;
    const int cnt = n1[f];
    // total past the launch capacity: out6 is incomplete, the call is re-run with a larger capacity (flag set by k_scan_counts)
    if (cnt == 0 || off2[gridDim.x] > cap_total) { if (threadIdx.x == 0) n2[f] = 0; return; }
    const float* logits = out6 + (size_t)off2[f] * 6;
    const float* fb = s1_box + (size_t)f * capF * 5;
    float* fout = s2_box + (size_t)f * capF * 5;
    const int P = next_pow2(cnt);
    if (threadIdx.x == 0) S.scal[4] = 0;
    __syncthreads();
    if (cnt > lds_cap) {                                                          // ---- spill tier
        const int C = pow2_floor(lds_cap);
        const int gx = (W >> GRID_SHIFT) + 1, gy = (H >> GRID_SHIFT) + 1;
        char* w = spill_alloc(sp, (size_t)P * 12 + (size_t)cnt * 28 + (size_t)gx * gy * 4);
        if (!w) { if (threadIdx.x == 0) n2[f] = 0; return; }
        uint64_t* gk = (uint64_t*)w; uint32_t* gi = (uint32_t*)(gk + P);
        float4* kbox = (float4*)(gi + P); float* karea = (float*)(kbox + cnt); uint32_t* kid = (uint32_t*)(karea + cnt);
        const KeptGrid grid{(int*)(kid + cnt) + cnt, (int*)(kid + cnt), gx, gy};
        int mine = 0;
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            uint64_t k = ~0ull;
            if (i < cnt) {
                const float p = trl_softmax2_p1(logits[6 * i], logits[6 * i + 1]);
                if (p > thr) { k = (((uint64_t)(~f2ord(p))) << 32) | (uint32_t)i; mine++; }
            }
            gk[i] = k; gi[i] = i;
        }
        if (mine) atomicAdd(&S.scal[4], mine);
        __syncthreads();
        const int m = S.scal[4];
        __syncthreads();                                                          // (scal is reused by the NMS below)
        if (m == 0) { if (threadIdx.x == 0) n2[f] = 0; return; }
        big_bitonic(gk, gi, P, S.key, S.id, C);
        const int nk = big_greedy<false>(S, C, gi, m, 0.7f,
                                         [&](uint32_t i) { const float* b = fb + 5 * i; return make_float4(b[0], b[1], b[2], b[3]); },
                                         kbox, karea, kid, grid);
        int out = 0;
        for (int base = 0; base < nk; base += C) {
            const int nc = nk - base < C ? nk - base : C;
            for (int r = threadIdx.x; r < nc; r += blockDim.x) {
                const int i = (int)kid[base + r];
                float4 b;
                S.flg[r] = stage2_row(kbox[base + r], logits + 6 * i + 2, W, H, b) ? 1 : 0;
                S.box[r] = b; S.aux[r] = trl_softmax2_p1(logits[6 * i], logits[6 * i + 1]);
            }
            __syncthreads();
            const int mm = block_compact_positions(S.flg, nc, S.pos, S.part);
            for (int r = threadIdx.x; r < nc; r += blockDim.x) {
                if (!S.flg[r]) continue;
                float* o = fout + (size_t)(out + S.pos[r]) * 5;
                const float4 b = S.box[r];
                o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = S.aux[r];
            }
            out += mm;
            __syncthreads();
        }
        if (threadIdx.x == 0) n2[f] = out;
        return;
    }
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < cnt) {
            const float p = trl_softmax2_p1(logits[6 * i], logits[6 * i + 1]);
            S.aux[i] = p;
            if (p > thr) { k = (((uint64_t)(~f2ord(p))) << 32) | (uint32_t)i; atomicAdd(&S.scal[4], 1); }
        }
        S.key[i] = k; S.id[i] = i;
    }
    __syncthreads();
    const int m = S.scal[4];
    if (m == 0) { if (threadIdx.x == 0) n2[f] = 0; return; }
    block_bitonic(S.key, S.id, P);
    for (int t = threadIdx.x; t < m; t += blockDim.x) {
        const float* b = fb + 5 * S.id[t];
        S.box[t] = make_float4(b[0], b[1], b[2], b[3]);
        S.area[t] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    __syncthreads();
    const int nk = block_nms<false>(S.box, S.area, m, 0.7f, S.sup, S.keep, S.scal);
    // bbreg (+1 widths), rerec; rows go to s2_box provisionally at r, then are compacted in place
    for (int r = threadIdx.x; r < nk; r += blockDim.x) {
        const int i = (int)S.id[S.keep[r]];
        float4 b;
        S.flg[r] = stage2_row(S.box[S.keep[r]], logits + 6 * i + 2, W, H, b) ? 1 : 0;
        float* o = fout + (size_t)r * 5;
        o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = S.aux[i];
    }
    __syncthreads();
    const int mm = block_compact_positions(S.flg, nk, S.pos, S.part);
    // in-place ordered compaction (pos[r] <= r): rounds of blockDim rows, reads precede writes
    float row[5];
    for (int base = 0; base < nk; base += blockDim.x) {
        const int r = base + threadIdx.x;
        bool live = r < nk && S.flg[r];
        if (live) { const float* o = fout + (size_t)r * 5; for (int q = 0; q < 5; q++) row[q] = o[q]; }
        __syncthreads();
        if (live) { float* o = fout + (size_t)S.pos[r] * 5; for (int q = 0; q < 5; q++) o[q] = row[q]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) n2[f] = mm;
}

// ---- stage 3 tail: thr2, landmarks, bbreg, nms 'Min' 0.7 --------------------------------------
__device__ __forceinline__ float4 stage3_box(const float* b, const float* g) {   // bbreg (+1 widths) with the O-Net offsets
    const float w = b[2] - b[0] + 1.f, h = b[3] - b[1] + 1.f;
    return make_float4(b[0] + g[0] * w, b[1] + g[1] * h, b[2] + g[2] * w, b[3] + g[3] * h);
}
__device__ __forceinline__ void stage3_points(const float* b, const float* pt, float* po) {
    const float w_i = b[2] - b[0] + 1.f, h_i = b[3] - b[1] + 1.f;
    for (int j = 0; j < 5; j++) {
        po[j] = w_i * pt[j] + b[0] - 1.f;
        po[5 + j] = h_i * pt[5 + j] + b[1] - 1.f;
    }
}
__global__ __launch_bounds__(1024) void k_stage3_post(int lds_cap, int capF, int cap_total, float thr, const int32_t* __restrict__ n2, const float* __restrict__ s2_box,
                                                     const int32_t* __restrict__ off3, const float* __restrict__ out16,
                                                     int32_t* __restrict__ n3, float* __restrict__ s3_box, float* __restrict__ s3_pts, Spill sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem S(smem_raw, lds_cap, blockDim.x);
    const int f = blockIdx.x;
    const int cnt = n2[f];
    if (cnt == 0 || off3[gridDim.x] > cap_total) { if (threadIdx.x == 0) n3[f] = 0; return; }   // see k_stage2_post
    const float* logits = out16 + (size_t)off3[f] * 16;
    const float* fb = s2_box + (size_t)f * capF * 5;
    const int P = next_pow2(cnt);
    if (threadIdx.x == 0) S.scal[4] = 0;
    __syncthreads();
    if (cnt > lds_cap) {                                                          // ---- spill tier
        const int C = pow2_floor(lds_cap);
        char* w = spill_alloc(sp, (size_t)P * 12 + (size_t)cnt * 24);
        if (!w) { if (threadIdx.x == 0) n3[f] = 0; return; }
        uint64_t* gk = (uint64_t*)w; uint32_t* gi = (uint32_t*)(gk + P);
        float4* kbox = (float4*)(gi + P); float* karea = (float*)(kbox + cnt); uint32_t* kid = (uint32_t*)(karea + cnt);
        int mine = 0;
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            uint64_t k = ~0ull;
            if (i < cnt) {
                const float p = trl_softmax2_p1(logits[16 * i], logits[16 * i + 1]);
                if (p > thr) { k = (((uint64_t)(~f2ord(p))) << 32) | (uint32_t)(~(uint32_t)i); mine++; }
            }
            gk[i] = k; gi[i] = i;
        }
        if (mine) atomicAdd(&S.scal[4], mine);
        __syncthreads();
        const int m = S.scal[4];
        __syncthreads();
        if (m == 0) { if (threadIdx.x == 0) n3[f] = 0; return; }
        big_bitonic(gk, gi, P, S.key, S.id, C);
        const int nk = big_greedy<true>(S, C, gi, m, 0.7f,
                                        [&](uint32_t i) { return stage3_box(fb + 5 * i, logits + 16 * i + 2); }, kbox, karea, kid,
                                        KeptGrid{nullptr, nullptr, 0, 0});
        for (int r = threadIdx.x; r < nk; r += blockDim.x) {
            const int i = (int)kid[r];
            const float4 bb = kbox[r];
            float* o = s3_box + ((size_t)f * capF + r) * 5;
            o[0] = bb.x; o[1] = bb.y; o[2] = bb.z; o[3] = bb.w; o[4] = trl_softmax2_p1(logits[16 * i], logits[16 * i + 1]);
            stage3_points(fb + 5 * i, logits + 16 * i + 6, s3_pts + ((size_t)f * capF + r) * 10);
        }
        if (threadIdx.x == 0) n3[f] = nk;
        return;
    }
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < cnt) {
            const float p = trl_softmax2_p1(logits[16 * i], logits[16 * i + 1]);
            S.aux[i] = p;
            // np.argsort ascending then take-from-the-end: descending score, ties -> higher index first
            if (p > thr) { k = (((uint64_t)(~f2ord(p))) << 32) | (uint32_t)(~(uint32_t)i); atomicAdd(&S.scal[4], 1); }
        }
        S.key[i] = k; S.id[i] = i;
    }
    __syncthreads();
    const int m = S.scal[4];
    if (m == 0) { if (threadIdx.x == 0) n3[f] = 0; return; }
    block_bitonic(S.key, S.id, P);
    for (int t = threadIdx.x; t < m; t += blockDim.x) {
        const int i = (int)S.id[t];
        const float4 bb = stage3_box(fb + 5 * i, logits + 16 * i + 2);
        S.box[t] = bb;
        S.area[t] = (bb.z - bb.x + 1.f) * (bb.w - bb.y + 1.f);
    }
    __syncthreads();
    const int nk = block_nms<true>(S.box, S.area, m, 0.7f, S.sup, S.keep, S.scal);
    for (int r = threadIdx.x; r < nk; r += blockDim.x) {
        const int t = S.keep[r], i = (int)S.id[t];
        const float4 bb = S.box[t];
        float* o = s3_box + ((size_t)f * capF + r) * 5;
        o[0] = bb.x; o[1] = bb.y; o[2] = bb.z; o[3] = bb.w; o[4] = S.aux[i];
        stage3_points(fb + 5 * i, logits + 16 * i + 6, s3_pts + ((size_t)f * capF + r) * 10);
    }
    if (threadIdx.x == 0) n3[f] = nk;
}

// MTCNN.detect(select_largest=True) ordering + model.py:49-54
__global__ __launch_bounds__(64) void k_select(int capF, int max_faces, int W, int H, const int32_t* __restrict__ n3,
                                               const float* __restrict__ s3_box, const float* __restrict__ s3_pts,
                                               float* __restrict__ boxes, float* __restrict__ probs, float* __restrict__ points,
                                               int32_t* __restrict__ counts, float* __restrict__ box0, float* __restrict__ prob0,
                                               int32_t* __restrict__ rect, uint8_t* __restrict__ valid, float* __restrict__ pts0) {
    const int f = blockIdx.x;
    const int n = n3[f];
    const float* b = s3_box + (size_t)f * capF * 5;
    // rank of r in np.argsort(area)[::-1] with stable-sort tie semantics: descending area, ties -> higher index first
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const float ar = (b[5 * r + 2] - b[5 * r]) * (b[5 * r + 3] - b[5 * r + 1]);
        int rank = 0;
        for (int q = 0; q < n; q++) {
            const float aq = (b[5 * q + 2] - b[5 * q]) * (b[5 * q + 3] - b[5 * q + 1]);
            rank += (aq > ar) || (aq == ar && q > r);
        }
        if (boxes && rank < max_faces) {
            float* o = boxes + ((size_t)f * max_faces + rank) * 4;
            o[0] = b[5 * r]; o[1] = b[5 * r + 1]; o[2] = b[5 * r + 2]; o[3] = b[5 * r + 3];
            probs[(size_t)f * max_faces + rank] = b[5 * r + 4];
            if (points) {   // detect(landmarks=True): points[..., j] = (x_j, y_j), stored x0..x4,y0..y4 like the stage-3 rows
                const float* ps = s3_pts + ((size_t)f * capF + r) * 10;
                float* po = points + ((size_t)f * max_faces + rank) * 10;
                for (int q = 0; q < 10; q++) po[q] = ps[q];
            }
        }
        if (rank == 0 && box0) {
            box0[4 * f] = b[5 * r]; box0[4 * f + 1] = b[5 * r + 1]; box0[4 * f + 2] = b[5 * r + 2]; box0[4 * f + 3] = b[5 * r + 3];
            prob0[f] = b[5 * r + 4];
            // model.py:49-53: boxes[0].astype(int) (toward zero), clamp
            long long x0 = (long long)b[5 * r], y0 = (long long)b[5 * r + 1], x1 = (long long)b[5 * r + 2], y1 = (long long)b[5 * r + 3];
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 > W) x1 = W;
            if (y1 > H) y1 = H;
            rect[4 * f] = (int)x0; rect[4 * f + 1] = (int)y0; rect[4 * f + 2] = (int)x1; rect[4 * f + 3] = (int)y1;
            valid[f] = (x1 > x0 && y1 > y0) ? 1 : 0;   // model.py:54
            if (pts0) {                                 // embedding mode 3: the largest face's five landmarks
                const float* ps = s3_pts + ((size_t)f * capF + r) * 10;
                for (int q = 0; q < 10; q++) pts0[10 * f + q] = ps[q];
            }
        }
    }
    if (threadIdx.x == 0) {
        if (counts) counts[f] = n < max_faces ? n : max_faces;
        if (n == 0 && box0) {
            for (int q = 0; q < 4; q++) { box0[4 * f + q] = 0.f; rect[4 * f + q] = 0; }
            prob0[f] = 0.f; valid[f] = 0;
            if (pts0) for (int q = 0; q < 10; q++) pts0[10 * f + q] = 0.f;
        }
    }
}

// model.py:55-58: frame[y0:y1, x0:x1] -> cv2.resize(.., (80,80)) INTER_LINEAR (u8 fixed point) -> /255
__device__ __forceinline__ int sat_short_round(float v) {
    int r = (int)__builtin_rintf(v);
    return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}
__global__ __launch_bounds__(256) void k_crop_resize80(const uint8_t* __restrict__ frames, int H, int W, const int32_t* __restrict__ rect,
                                                       const uint8_t* __restrict__ valid, float* __restrict__ out) {
    constexpr int O = 80;
    __shared__ int xofs[O], xofs1[O], a0[O], a1[O], ys0[O], ys1[O], b0[O], b1[O];
    const int f = blockIdx.x;
    float* o = out + (size_t)f * O * O * 3;
    if (!valid[f]) {
        for (int p = threadIdx.x; p < O * O * 3; p += blockDim.x) o[p] = 0.f;
        return;
    }
    const int x0 = rect[4 * f], y0 = rect[4 * f + 1], sw = rect[4 * f + 2] - x0, sh = rect[4 * f + 3] - y0;
    const double scale_x = 1. / ((double)O / sw), scale_y = 1. / ((double)O / sh);
    if (threadIdx.x < O) {
        const int d = threadIdx.x;
        float fx = (float)((d + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[d] = sx; xofs1[d] = sx + 1 > sw - 1 ? sw - 1 : sx + 1;
        a0[d] = sat_short_round((1.f - fx) * 2048.f); a1[d] = sat_short_round(fx * 2048.f);
    } else if (threadIdx.x < 2 * O) {
        const int d = threadIdx.x - O;
        float fy = (float)((d + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        b0[d] = sat_short_round((1.f - fy) * 2048.f); b1[d] = sat_short_round(fy * 2048.f);
        ys0[d] = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        ys1[d] = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
    }
    __syncthreads();
    const uint8_t* fp = frames + ((size_t)f * H + y0) * W * 3 + (size_t)x0 * 3;
    for (int p = threadIdx.x; p < O * O * 3; p += blockDim.x) {
        const int c = p % 3, dx = (p / 3) % O, dy = p / (3 * O);
        const uint8_t* S0 = fp + (size_t)ys0[dy] * W * 3;
        const uint8_t* S1 = fp + (size_t)ys1[dy] * W * 3;
        const int r0 = S0[xofs[dx] * 3 + c] * a0[dx] + S0[xofs1[dx] * 3 + c] * a1[dx];
        const int r1 = S1[xofs[dx] * 3 + c] * a0[dx] + S1[xofs1[dx] * 3 + c] * a1[dx];
        int v = (((b0[dy] * (r0 >> 4)) >> 16) + ((b1[dy] * (r1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[p] = (float)v / 255.0f;   // to_tensor
    }
}

// SURVEY 8(f)-4 native embedding mode: facenet-pytorch extract_face() for tensor input = crop -> imresample (area)
// to SxS -> .byte() (truncation) -> fixed_image_standardization (x-127.5)/128, optionally BGR -> RGB.
__global__ __launch_bounds__(256) void k_crop_area_std(const uint8_t* __restrict__ frames, int H, int W, const int32_t* __restrict__ rect,
                                                       const uint8_t* __restrict__ valid, int S, int rgb, float* __restrict__ out) {
    const int f = blockIdx.y;
    float* o = out + (size_t)f * S * S * 3;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S * S) return;
    if (!valid[f]) { o[3 * p] = 0.f; o[3 * p + 1] = 0.f; o[3 * p + 2] = 0.f; return; }
    const int x0 = rect[4 * f], y0 = rect[4 * f + 1], iw = rect[4 * f + 2] - x0, ih = rect[4 * f + 3] - y0;
    const int oy = p / S, ox = p - oy * S;
    const int ys = (int)(((long long)oy * ih) / S), ye = (int)((((long long)oy + 1) * ih + S - 1) / S);
    const int xs = (int)(((long long)ox * iw) / S), xe = (int)((((long long)ox + 1) * iw + S - 1) / S);
    unsigned s0 = 0, s1 = 0, s2 = 0;
    const uint8_t* fp = frames + (size_t)f * H * W * 3;
    for (int y = ys; y < ye; y++) {
        const uint8_t* q = fp + ((size_t)(y0 + y) * W + x0 + xs) * 3;
        for (int x = xs; x < xe; x++, q += 3) { s0 += q[0]; s1 += q[1]; s2 += q[2]; }
    }
    const float kh = (float)(ye - ys), kw = (float)(xe - xs);
    const float b0 = (float)(unsigned char)((float)s0 / kh / kw), b1 = (float)(unsigned char)((float)s1 / kh / kw),
                b2 = (float)(unsigned char)((float)s2 / kh / kw);
    o[3 * p + (rgb ? 2 : 0)] = (b0 - 127.5f) / 128.0f;
    o[3 * p + 1] = (b1 - 127.5f) / 128.0f;
    o[3 * p + (rgb ? 0 : 2)] = (b2 - 127.5f) / 128.0f;
}

// SURVEY 8(f)-4 "landmark-aligned" embedding mode (trl_config.embed_mode 3; this project's own definition, restated in
// oracle/trl_oracle.c orc_crop_aligned): least-squares similarity from the scaled 112x112 five-point template to the face's
// O-Net landmarks, estimated as the inverse map (double, fixed operation order), bilinear sample of the u8 frame with
// replicated borders (float), (v-127.5)/128, optional BGR -> RGB.  One thread per output pixel; every thread of a face
// recomputes the six transform parameters (60 flops) rather than paying a second launch.
__constant__ double TPL_X[5] = {54.706571428571436, 105.04542857142857, 80.036, 59.35614285714286, 101.04271428571428};
__constant__ double TPL_Y[5] = {73.85185714285714, 73.57342857142856, 102.48085714285713, 131.9507142857143, 131.72014285714286};
__global__ __launch_bounds__(256) void k_crop_aligned(const uint8_t* __restrict__ frames, int H, int W, const float* __restrict__ pts0,
                                                      const uint8_t* __restrict__ valid, int S, int rgb, float* __restrict__ out) {
    const int f = blockIdx.y;
    float* o = out + (size_t)f * S * S * 3;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S * S) return;
    if (!valid[f]) { o[3 * p] = 0.f; o[3 * p + 1] = 0.f; o[3 * p + 2] = 0.f; return; }
    const float* pts = pts0 + 10 * f;
    double tx = 0., ty = 0., px = 0., py = 0.;
#pragma unroll
    for (int j = 0; j < 5; j++) { tx += TPL_X[j]; ty += TPL_Y[j]; px += (double)pts[j]; py += (double)pts[5 + j]; }
    tx = tx / 5.; ty = ty / 5.; px = px / 5.; py = py / 5.;
    double sdd = 0., sde = 0., scr = 0.;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const double dx = TPL_X[j] - tx, dy = TPL_Y[j] - ty, ex = (double)pts[j] - px, ey = (double)pts[5 + j] - py;
        sdd = sdd + (dx * dx + dy * dy);
        sde = sde + (dx * ex + dy * ey);
        scr = scr + (dx * ey - dy * ex);
    }
    const double a = sde / sdd, b = scr / sdd;
    const int v = p / S, u = p - v * S;
    const double du = (double)u - tx, dv = (double)v - ty;
    const double x = (a * du - b * dv) + px, y = (b * du + a * dv) + py;
    const double xf = floor(x), yf = floor(y);
    const float fx = (float)(x - xf), fy = (float)(y - yf);
    const double xc = (xf >= -1.) ? (xf > (double)W ? (double)W : xf) : -1., yc = (yf >= -1.) ? (yf > (double)H ? (double)H : yf) : -1.;   // NaN -> -1
    int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    const uint8_t* fp = frames + (size_t)f * H * W * 3;
    const uint8_t *r0 = fp + (size_t)y0 * W * 3, *r1 = fp + (size_t)y1 * W * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float p00 = (float)r0[x0 * 3 + c], p01 = (float)r0[x1 * 3 + c], p10 = (float)r1[x0 * 3 + c], p11 = (float)r1[x1 * 3 + c];
        const float top = p00 + fx * (p01 - p00), bot = p10 + fx * (p11 - p10);
        const float val = top + fy * (bot - top);
        o[3 * p + (rgb ? 2 - c : c)] = (val - 127.5f) / 128.0f;
    }
}

template <typename K>
int set_dyn_smem(K kernel, size_t bytes) {
    TRL_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TRL_OK;
}

}  // namespace

// ---- host helpers -------------------------------------------------------------------------------
int trl_compute_levels(trl_ctx* c, int H, int W) {
    // detect_face.py: m = 12/minsize; minl = min(h,w)*m; while minl >= 12: scales += [scale_i]; scale_i *= factor
    const double m = 12.0 / c->cfg.min_face_size;
    double minl = (H < W ? H : W) * m;
    double scale_i = m;
    int L = 0;
    while (minl >= 12 && L < 32) {
        LevelGeom& g = c->lv[L];
        g.scale = scale_i;
        g.h = (int)(H * scale_i + 1);
        g.w = (int)(W * scale_i + 1);
        const int ph = (g.h - 2 + 1) / 2, pw = (g.w - 2 + 1) / 2;
        g.oh = ph - 4; g.ow = pw - 4;
        L++;
        scale_i = scale_i * c->cfg.factor;
        minl = minl * c->cfg.factor;
    }
    return L;
}

int trl_launch_area_level(const uint8_t* d_frames, int nf, int H, int W, int h, int w, float* d_level, hipStream_t s) {
    const size_t total = (size_t)nf * h * w;
    size_t blocks = (total + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    k_area_level<<<(unsigned)blocks, 256, 0, s>>>(d_frames, nf, H, W, h, w, d_level);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
int trl_launch_heads_to_maps(const float* d_heads, int cells, float* d_prob, float* d_reg, hipStream_t s) {
    k_heads_to_maps<<<(cells + 255) / 256, 256, 0, s>>>(d_heads, cells, d_prob, d_reg);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
int trl_launch_crop_area_std(const uint8_t* d_frames, int n, int H, int W, const int32_t* d_rect, const uint8_t* d_valid, int S,
                             bool rgb, float* d_faces, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    k_crop_area_std<<<dim3((S * S + 255) / 256, n), 256, 0, s>>>(d_frames, H, W, d_rect, d_valid, S, rgb ? 1 : 0, d_faces);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
int trl_launch_crop_aligned(const uint8_t* d_frames, int n, int H, int W, const float* d_pts0, const uint8_t* d_valid, int S, bool rgb,
                            float* d_faces, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    k_crop_aligned<<<dim3((S * S + 255) / 256, n), 256, 0, s>>>(d_frames, H, W, d_pts0, d_valid, S, rgb ? 1 : 0, d_faces);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}
int trl_launch_crop_resize80(const uint8_t* d_frames, int n, int H, int W, const int32_t* d_rect, const uint8_t* d_valid,
                             float* d_faces, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    k_crop_resize80<<<n, 256, 0, s>>>(d_frames, H, W, d_rect, d_valid, d_faces);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}



// Record slots per level and rows per frame of this call: the configured start values, raised to what recent calls needed
// (trl_cascade_check), never beyond what the geometry can produce (a level has oh x ow cells; a frame's stage-1 list is a subset
// of its cells) -- so a frame small enough cannot overflow at all.
static void plan_lists(trl_ctx* c, int L) {
    CascadeBufs& B = c->cb;
    LvLayout& G = B.lay;
    G = LvLayout();
    G.L = L;
    long long cells_total = 0;
    int S = 0;
    for (int l = 0; l < L; l++) {
        const long long cells = (long long)c->lv[l].oh * c->lv[l].ow;
        long long want = c->cfg.cap_level;
        if ((long long)c->lvl_hint[l] > want) want = (long long)c->lvl_hint[l];
        if (want > cells) want = cells;
        if (want < 4) want = 4;
        G.capl[l] = (int)((want + 3) & ~3ll);
        G.rec0[l] = S;
        S += G.capl[l];
        cells_total += cells;
    }
    G.S = S;
    long long wantF = c->cfg.cap_frame;
    if ((long long)c->frame_hint > wantF) wantF = (long long)c->frame_hint;
    if (wantF > cells_total) wantF = cells_total;
    if (wantF < 4) wantF = 4;
    B.capF = (int)((wantF + 3) & ~3ll);
}
static size_t spill_need(long long cnt, int W, int H) {   // workspace of one spilled list of cnt entries (k_nms_level .. k_stage3_post)
    long long P = 2;
    while (P < cnt) P <<= 1;
    return (size_t)(P * 12 + cnt * 28 + ((long long)(W >> 5) + 1) * ((H >> 5) + 1) * 4 + 256);
}

// detect_face() stages 1-3 for n frames; results stay in c->cb
// `resume` = 2 / 3: re-run of a call whose R-Net / O-Net batch capacity was too small -- everything up to that stage is intact in
// the cascade arena (the lists, the stage boxes, the counts), so the attempt starts at stage 2 / 3 instead of at the pyramid.
int trl_cascade_detect(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, hipStream_t s, int resume) {
    const int L = trl_compute_levels(c, H, W);
    CascadeBufs& B = c->cb;
    if (resume >= 2 && !(B.flags && B.n == n && B.L == L && B.H == H && B.W == W && L > 0)) resume = 0;   // nothing to resume from
    B.n = n; B.L = L; B.H = H; B.W = W;
    if (!resume) plan_lists(c, L);
    const LvLayout& G = B.lay;
    const int capF = B.capF;
    const int lds_full = c->nms_full, lds_small = c->nms_small < c->nms_full ? c->nms_small : c->nms_full;
    // spill workspace: lists that can outgrow the LDS tier get theirs up front (bounded; a pool that still runs out is grown by the re-run)
    size_t spill = c->spill_hint;
    {
        size_t per_frame = 0;
        for (int l = 0; l < L; l++) if (G.capl[l] > lds_full) per_frame += spill_need(G.capl[l], W, H);
        if (capF > lds_full) per_frame += 3 * spill_need(capF, W, H);
        size_t up_front = per_frame * (size_t)n;
        if (up_front > (8ull << 30)) up_front = 8ull << 30;
        if (up_front > spill) spill = up_front;
    }
    Arena& A = c->arena;      // cascade lists: live for the whole call (and for the debug hooks after it)
    Arena& X = c->scratch;    // activations: reset between stages
    if (resume) {
        A.off = B.arena_mark;                                // the API layer's per-frame outputs are allocated again behind the lists
        // the stages that run again report afresh (the stage-2 total of a resume at stage 3 stays: it is complete)
        TRL_HIP(hipMemsetAsync(B.flags + FLG_T3, 0, 4, s));
        TRL_HIP(hipMemsetAsync(B.flags + FLG_T3N, 0, 4, s));
        if (resume == 2) { TRL_HIP(hipMemsetAsync(B.flags + FLG_T2, 0, 4, s)); TRL_HIP(hipMemsetAsync(B.flags + FLG_T2N, 0, 4, s)); }
    }
    const size_t need = (size_t)n * ((size_t)G.S * (sizeof(Cand) + 4) + (size_t)L * 8) + (size_t)n * capF * (5 * 3 + 10 + 8) * 4 +
                        (size_t)n * 1024 + spill + (1u << 20);   // + the API layer's per-frame outputs (box0, prob0, rect, valid, pts0)
    if (!resume) {
    TRL_CHECK(trl_ensure(c, A, need));
    A.reset();
    B.lvl_cnt = (int32_t*)A.alloc((size_t)n * L * 4);
    B.lvl_keep_cnt = (int32_t*)A.alloc((size_t)n * L * 4);
    B.lvl_rec = (Cand*)A.alloc((size_t)n * G.S * sizeof(Cand));
    B.lvl_keep_idx = (int32_t*)A.alloc((size_t)n * G.S * 4);
    B.n1 = (int32_t*)A.alloc((size_t)n * 4); B.n2 = (int32_t*)A.alloc((size_t)n * 4); B.n3 = (int32_t*)A.alloc((size_t)n * 4);
    B.s1_box = (float*)A.alloc((size_t)n * capF * 20); B.s2_box = (float*)A.alloc((size_t)n * capF * 20);
    B.s3_box = (float*)A.alloc((size_t)n * capF * 20); B.s3_pts = (float*)A.alloc((size_t)n * capF * 40);
    B.off2 = (int32_t*)A.alloc((size_t)(n + 1) * 4); B.off3 = (int32_t*)A.alloc((size_t)(n + 1) * 4);
    B.cbox = (int32_t*)A.alloc((size_t)n * capF * 32);
    B.flags = (int32_t*)A.alloc(TRL_NFLAGS * 4);
    B.spill = spill ? (char*)A.alloc(spill) : nullptr;
    B.spill_cap = B.spill ? spill : 0;
    if (!B.flags || (spill && !B.spill)) { trl_set_error("cascade workspace allocation failed"); return TRL_ERR_STATE; }
    B.arena_mark = A.off;
    TRL_HIP(hipMemsetAsync(B.flags, 0, TRL_NFLAGS * 4, s));
    }
    const Spill sp{B.spill, (unsigned long long)B.spill_cap, B.flags};
    if (L == 0) {
        // min(H, W) * 12 / min_face_size < 12: detect_face() builds no scale at all and returns no boxes (the frame is smaller
        // than the smallest face looked for).  Every stage count is zero; k_select then reports "no face" for every frame.
        TRL_HIP(hipMemsetAsync(B.n1, 0, (size_t)n * 4, s));
        TRL_HIP(hipMemsetAsync(B.n2, 0, (size_t)n * 4, s));
        TRL_HIP(hipMemsetAsync(B.n3, 0, (size_t)n * 4, s));
        TRL_HIP(hipMemsetAsync(B.off2, 0, (size_t)(n + 1) * 4, s));
        TRL_HIP(hipMemsetAsync(B.off3, 0, (size_t)(n + 1) * 4, s));
        if (c->scratch_after_cascade) TRL_CHECK(trl_ensure(c, X, c->scratch_after_cascade));   // the crops + embedder of the same call
        return TRL_OK;
    }
    if (!resume) TRL_HIP(hipMemsetAsync(B.lvl_cnt, 0, (size_t)n * L * 4, s));

    // ---- stage 1: PNet over the pyramid ----------------------------------------------------------
    if (!resume) c->pnet_ev_used = 0;
    X.reset();
    int chunk[32];
    const bool fused = c->cfg.pnet_mode == 0 && L <= 16;   // the fused launch carries 16 level descriptors; taller pyramids (a
                                                           // 16 K frame at min_face_size 12) take the per-level path
    if (!fused) {
        size_t mx = 0;
        for (int l = 0; l < L; l++) {
            const LevelGeom& g = c->lv[l];
            const size_t per = trl_pnet_generic_bytes(1, g.h, g.w) + (size_t)g.h * g.w * 12 + (size_t)g.oh * g.ow * 24 + 4096;
            int ch = (int)((size_t)(2048ull << 20) / per);
            if (ch < 1) ch = 1;
            if (ch > n) ch = n;
            chunk[l] = ch;
            if (per * ch > mx) mx = per * ch;
        }
        TRL_CHECK(trl_ensure(c, X, mx + (4u << 20)));
    }
    auto next_ev = [&](std::pair<hipEvent_t, hipEvent_t>*& out) -> int {
        if ((int)c->pnet_ev.size() <= c->pnet_ev_used) {
            hipEvent_t e0, e1;
            TRL_HIP(hipEventCreate(&e0)); TRL_HIP(hipEventCreate(&e1));
            c->pnet_ev.push_back({e0, e1});
        }
        out = &c->pnet_ev[c->pnet_ev_used++];
        return TRL_OK;
    };
    // capacities of the R-/O-Net candidate batches of this call, and ONE workspace big enough for every stage: it is only
    // ever grown here, before anything is queued (growing frees the old block)
    {
        long long c2 = (long long)(c->t2_per_frame * n) + 64, c3 = (long long)(c->t3_per_frame * n) + 64;
        const long long lim = (long long)n * capF;
        c->cap_t2 = (int)(c2 < lim ? c2 : lim);
        c->cap_t3 = (int)(c3 < lim ? c3 : lim);
        const int TRL_CH2 = c->rnet_chunk, TRL_CH3 = c->onet_chunk;
        const int ch2 = c->cap_t2 < TRL_CH2 ? c->cap_t2 : TRL_CH2, ch3 = c->cap_t3 < TRL_CH3 ? c->cap_t3 : TRL_CH3;
        // per candidate: the front kernel's pooled map + every activation of the tail (trl_run_rnet_tail / trl_run_onet_tail:
        // R-Net 3388 + 3888 + 768 + 576 + 128 floats = 35 KB; O-Net 16928 + 28224 + 6400 + 4096 + 1024 + 1152 + 256 = 227 KB)
        size_t need_x = (size_t)c->cap_t2 * 24 + (size_t)ch2 * (40 * 1024) + (1u << 20);
        const size_t need3 = (size_t)c->cap_t3 * 64 + (size_t)ch3 * (240 * 1024) + (1u << 20);
        if (need3 > need_x) need_x = need3;
        if (fused) { const size_t p = trl_pnet_fused_bytes(c, n, H, W) + (1u << 20); if (p > need_x) need_x = p; }
        if (c->scratch_after_cascade > need_x) need_x = c->scratch_after_cascade;   // the embedder that follows in the same call
        TRL_CHECK(trl_ensure(c, X, need_x));
    }
    if (resume) {
        // stage 1 is intact
    } else if (fused) {
        // fused path: pyramid kernel + ONE persistent PNet launch over every (frame, level, tile)
        std::pair<hipEvent_t, hipEvent_t>*pa, *pb;
        c->pnet_ev.reserve(64);   // next_ev hands out pointers into the vector: no reallocation below
        TRL_CHECK(next_ev(pa)); TRL_CHECK(next_ev(pb));
        hipEvent_t ev[4] = {pa->first, pa->second, pb->first, pb->second};
        TRL_CHECK(trl_pnet_fused_all(c, d_frames, n, H, W, ev, s));
    } else {
        for (int l = 0; l < L; l++) {
            const LevelGeom& g = c->lv[l];
            if (g.oh < 1 || g.ow < 1) continue;
            std::pair<hipEvent_t, hipEvent_t>* pe;
            TRL_CHECK(next_ev(pe));
            TRL_HIP(hipEventRecord(pe->first, s));
            // generic layer path: materialise the level for a chunk of frames
            for (int f0 = 0; f0 < n; f0 += chunk[l]) {
                const int nf = (n - f0 < chunk[l]) ? n - f0 : chunk[l];
                X.reset();
                float* lvl = (float*)X.alloc((size_t)nf * g.h * g.w * 12);
                float* heads = (float*)X.alloc((size_t)nf * g.oh * g.ow * 24);
                if (!lvl || !heads) { trl_set_error("pnet generic workspace"); return TRL_ERR_STATE; }
                TRL_CHECK(trl_launch_area_level(d_frames + (size_t)f0 * H * W * 3, nf, H, W, g.h, g.w, lvl, s));
                TRL_CHECK(trl_run_pnet_generic(c, lvl, nf, g.h, g.w, heads, s));
                const size_t total = (size_t)nf * g.oh * g.ow;
                size_t blocks = (total + 255) / 256;
                if (blocks > 16384) blocks = 16384;
                k_pnet_collect<<<(unsigned)blocks, 256, 0, s>>>(heads, nf, f0, g.oh, g.ow, (float)g.scale, c->cfg.thr0, L, l, G.capl[l],
                                                                 G.rec0[l], G.S, B.lvl_cnt, B.lvl_rec, B.flags);
                TRL_LAUNCH_CHECK();
            }
            TRL_HIP(hipEventRecord(pe->second, s));
        }
    }
    // LDS tiers: no list is longer than its capacity, so the carve never exceeds what the call can produce
    int max_capl = 4;
    for (int l = 0; l < L; l++) if (G.capl[l] > max_capl) max_capl = G.capl[l];
    const int full_l = max_capl < lds_full ? max_capl : lds_full, full_f = capF < lds_full ? capF : lds_full;
    // lists that can take the spill tier run with 1024 threads per workgroup: its cost is pair tests (candidates x boxes kept so far)
    const int th_l = max_capl > lds_full ? 1024 : 256, th_f = capF > lds_full ? 1024 : 256;
    const size_t sm_l = Smem::bytes(full_l, th_l), sm_f = Smem::bytes(full_f, th_f);
    TRL_CHECK(set_dyn_smem(k_nms_level, sm_l));
    TRL_CHECK(set_dyn_smem(k_nms_frame, sm_f));
    TRL_CHECK(set_dyn_smem(k_stage2_post, sm_f));
    TRL_CHECK(set_dyn_smem(k_stage3_post, sm_f));
    if (!resume) {
        const int small_cap = full_l < lds_small ? full_l : lds_small;
        const size_t sm_s = Smem::bytes(small_cap);
        k_nms_level<<<n * L, 256, sm_s, s>>>(G, small_cap, 0, small_cap == max_capl ? 1 : 0, W, H, B.lvl_cnt, B.lvl_rec, B.lvl_keep_cnt, B.lvl_keep_idx, B.flags, sp);
        TRL_LAUNCH_CHECK();
        if (small_cap < max_capl) {
            k_nms_level<<<n * L, th_l, sm_l, s>>>(G, full_l, small_cap + 1, 1, W, H, B.lvl_cnt, B.lvl_rec, B.lvl_keep_cnt, B.lvl_keep_idx, B.flags, sp);
            TRL_LAUNCH_CHECK();
        }
    }
    if (!resume) {
        k_nms_frame<<<n, th_f, sm_f, s>>>(G, full_f, capF, W, H, B.lvl_rec, B.lvl_keep_cnt, B.lvl_keep_idx, B.n1, B.s1_box, B.flags, sp);
        TRL_LAUNCH_CHECK();
    }

    // ---- stage 2: RNet ------------------------------------------------------------------------------
    // No host round trip: the candidate total stays on the device (off2[n]).  Launches are sized by an optimistic capacity
    // (c->cap_t2, from earlier calls) and workgroups past the real total exit at once; if the total exceeds the capacity a
    // flag is raised and the caller re-runs the call with a larger one (trl_cascade_run).
    const int cap2 = c->cap_t2, cap3 = c->cap_t3;
    if (resume < 3) {
    k_scan_counts<<<1, 256, 0, s>>>(B.n1, n, B.off2, cap2, B.flags, 0);
    TRL_LAUNCH_CHECK();
    k_build_map<<<n, 64, 0, s>>>(B.n1, B.off2, B.s1_box, capF, W, H, B.cbox);
    TRL_LAUNCH_CHECK();
    X.reset();   // stream order keeps the PNet workspace alive until its kernels are done: reuse needs no host sync
    float* out6 = (float*)X.alloc((size_t)cap2 * 24);
    {
        const int CH = c->rnet_chunk;
        const size_t mk = X.off;
        for (int t0 = 0; t0 < cap2; t0 += CH) {
            const int nc = (cap2 - t0 < CH) ? cap2 - t0 : CH;
            X.off = mk;
            float* pool1 = (float*)X.alloc((size_t)nc * 11 * 11 * 28 * 4);
            if (!pool1 || !out6) { trl_set_error("rnet workspace"); return TRL_ERR_STATE; }
            TRL_CHECK(trl_launch_rnet_front(c, d_frames, H, W, B.off2 + n, t0, nc, pool1, s));   // crop + conv1 + pool1 in LDS
            TRL_CHECK(trl_run_rnet_tail(c, pool1, nc, out6 + (size_t)t0 * 6, s, B.off2 + n, t0));
        }
    }
    k_stage2_post<<<n, th_f, sm_f, s>>>(full_f, capF, cap2, W, H, c->cfg.thr1, B.n1, B.s1_box, B.off2, out6, B.n2, B.s2_box, sp);
    TRL_LAUNCH_CHECK();
    }

    // ---- stage 3: ONet --------------------------------------------------------------------------------
    k_scan_counts<<<1, 256, 0, s>>>(B.n2, n, B.off3, cap3, B.flags, 1);
    TRL_LAUNCH_CHECK();
    k_build_map<<<n, 64, 0, s>>>(B.n2, B.off3, B.s2_box, capF, W, H, B.cbox);
    TRL_LAUNCH_CHECK();
    X.reset();
    float* out16 = (float*)X.alloc((size_t)cap3 * 64);
    {
        const int CH = c->onet_chunk;
        const size_t mk = X.off;
        for (int t0 = 0; t0 < cap3; t0 += CH) {
            const int nc = (cap3 - t0 < CH) ? cap3 - t0 : CH;
            X.off = mk;
            float* pool1 = (float*)X.alloc((size_t)nc * 23 * 23 * 32 * 4);
            if (!pool1 || !out16) { trl_set_error("onet workspace"); return TRL_ERR_STATE; }
            TRL_CHECK(trl_launch_onet_front(c, d_frames, H, W, B.off3 + n, t0, nc, pool1, s));   // crop + conv1 + pool1 in LDS
            TRL_CHECK(trl_run_onet_tail(c, pool1, nc, out16 + (size_t)t0 * 16, s, B.off3 + n, t0));
        }
    }
    k_stage3_post<<<n, th_f, sm_f, s>>>(full_f, capF, cap3, c->cfg.thr2, B.n2, B.s2_box, B.off3, out16, B.n3, B.s3_box, B.s3_pts, sp);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

int trl_cascade_finish(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_boxes, float* d_probs, float* d_points,
                       int32_t* d_counts, float* d_box0, float* d_prob0, int32_t* d_rect, uint8_t* d_valid, float* d_pts0, hipStream_t s) {
    (void)d_frames;
    CascadeBufs& B = c->cb;
    k_select<<<n, 64, 0, s>>>(B.capF, c->cfg.max_faces, W, H, B.n3, B.s3_box, B.s3_pts, d_boxes, d_probs, d_points, d_counts, d_box0, d_prob0,
                              d_rect, d_valid, d_pts0);
    TRL_LAUNCH_CHECK();
    // overflow flags + stage totals travel to pinned host memory behind the kernels; trl_cascade_check reads them after the
    // call's ONE stream synchronisation (no host round trip inside the call)
    TRL_HIP(hipMemcpyAsync(c->h_pinned + 4, B.flags, TRL_NFLAGS * 4, hipMemcpyDeviceToHost, s));
    return TRL_OK;
}

// After the stream has been synchronised.  *retry = 1 when a capacity of this attempt was too small -- a level's record list, a
// frame's box list, the spill workspace, or the optimistic R-/O-Net batch -- : the capacities have been raised to what the
// attempt measured and the caller runs the call again (results of the attempt are incomplete, not wrong-but-plausible, and are
// never delivered).  No input can make this fail: detect_face() (server/model.py:47) has no candidate limit.
int trl_cascade_check(trl_ctx* c, int n, int* retry) {
    const int32_t* f = c->h_pinned + 4;
    const int L = c->cb.L;
    *retry = 0;
    c->resume_stage = 0;
    unsigned long long spill_used = 0;
    memcpy(&spill_used, f + FLG_SPILL_CUR, 8);
    if (f[FLG_LEVEL]) {                                  // every later stage ran on truncated lists: their totals mean nothing
        long long sum = 0;
        for (int l = 0; l < L; l++) {
            const float want = 1.25f * (float)f[FLG_LEVEL_MAX + l] + 4.f;
            if (f[FLG_LEVEL_MAX + l] > c->cb.lay.capl[l] && want > c->lvl_hint[l]) c->lvl_hint[l] = want;
            sum += f[FLG_LEVEL_MAX + l];
        }
        // a frame's stage-1 list is a subset of its candidates: when the per-frame lists of the batch stay under 8 GB at that bound,
        // grow them in the same step (one attempt less for crowded content; otherwise the next attempt measures the real total)
        if ((float)sum > c->frame_hint && (double)sum * 132.0 * (double)n < 8e9) c->frame_hint = (float)sum;
        *retry = 1;
        return TRL_OK;
    }
    if (f[FLG_SPILL]) {                                  // lists that found no workspace were dropped: later totals are incomplete
        c->spill_hint = (size_t)spill_used * 2 + (16u << 20);
        *retry = 1;
        if (f[FLG_FRAME]) c->frame_hint = 1.25f * (float)f[FLG_FRAME_MAX] + 4.f;
        return TRL_OK;
    }
    if (f[FLG_FRAME]) { c->frame_hint = 1.25f * (float)f[FLG_FRAME_MAX] + 4.f; *retry = 1; return TRL_OK; }
    // keep ~25 % headroom over the largest batch seen, so a drifting clip rarely needs a second attempt
    const float want2 = 1.25f * (float)f[FLG_T2N] / (float)n + 1.f, want3 = 1.25f * (float)f[FLG_T3N] / (float)n + 1.f;
    if (f[FLG_T2]) { c->t2_per_frame = want2; *retry = 1; c->resume_stage = 2; }       // stage 3 ran on an incomplete stage 2: its total is meaningless
    else if (f[FLG_T3]) { c->t3_per_frame = want3; *retry = 1; c->resume_stage = 3; }  // (stages 1 / 1-2 are intact: the re-run starts behind them)
    if (!*retry) {
        // follow the content: grow at once, decay 3 % per call towards what recent batches needed (never below the start values),
        // so one crowded batch does not inflate every later call's launches and workspace for the life of the context
        const float d2 = 0.97f * c->t2_per_frame, d3 = 0.97f * c->t3_per_frame;
        c->t2_per_frame = want2 > d2 ? want2 : (d2 > 160.f ? d2 : (c->t2_per_frame < 160.f ? c->t2_per_frame : 160.f));
        c->t3_per_frame = want3 > d3 ? want3 : (d3 > 48.f ? d3 : (c->t3_per_frame < 48.f ? c->t3_per_frame : 48.f));
        for (int l = 0; l < L; l++) {                    // list capacities decay the same way (values below the configured start are ignored)
            const float want = 1.25f * (float)f[FLG_LEVEL_MAX + l] + 4.f, d = 0.97f * c->lvl_hint[l];
            c->lvl_hint[l] = want > d ? want : d;
        }
        {
            const float want = 1.25f * (float)f[FLG_FRAME_MAX] + 4.f, d = 0.97f * c->frame_hint;
            c->frame_hint = want > d ? want : d;
        }
        {
            const size_t want = (size_t)spill_used + (spill_used >> 2), d = c->spill_hint - (c->spill_hint >> 5);
            c->spill_hint = spill_used ? (want > d ? want : d) : (d > (1u << 20) ? d : 0);
        }
    }
    return TRL_OK;
}
