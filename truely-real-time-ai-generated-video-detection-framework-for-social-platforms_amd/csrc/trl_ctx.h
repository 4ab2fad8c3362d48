// trl_ctx.h -- the opaque context behind the C ABI.
#pragma once
#include "trl_common.h"

// One PNet candidate (generateBoundingBox row + its cell index).  40 bytes.
struct Cand {
    float x1, y1, x2, y2, score;
    float r0, r1, r2, r3;
    int cell;
};

struct LevelGeom {
    double scale;
    int h, w;        // pyramid level size
    int oh, ow;      // PNet output map size
};

// Candidate lists of one call.  Level l of every frame owns capl[l] record slots (never more than the level has cells); a frame's
// records are contiguous: record (f, l, slot) = lvl_rec[f * S + rec0[l] + slot], S = sum of capl.  The per-frame box lists of
// stages 1-3 hold capF rows.  Both capacities follow the content: a call that overflows one is re-run with larger lists
// (trl_cascade_check), so no input can make the detector fail -- detect_face() has no candidate limit either.
struct LvLayout {
    int L = 0, S = 0;
    int capl[32] = {0}, rec0[32] = {0};
};
struct CascadeBufs {   // device pointers into the arena, valid until the next call
    int n = 0, L = 0, H = 0, W = 0;
    LvLayout lay;                    // record slots per level
    int capF = 0;                    // rows per frame of s1_box / s2_box / s3_box / s3_pts
    int32_t* lvl_cnt = nullptr;      // [n][L]
    Cand* lvl_rec = nullptr;         // [n][S]
    int32_t* lvl_keep_cnt = nullptr; // [n][L]
    int32_t* lvl_keep_idx = nullptr; // [n][S]
    int32_t* n1 = nullptr; float* s1_box = nullptr;   // [n], [n][capF][5]
    int32_t* n2 = nullptr; float* s2_box = nullptr;   // [n], [n][capF][5]
    int32_t* n3 = nullptr; float* s3_box = nullptr;   // [n], [n][capF][5]
    float* s3_pts = nullptr;                           // [n][capF][10]
    int32_t* off2 = nullptr; int32_t* off3 = nullptr; // [n+1] exclusive scans of n1 / n2
    int32_t* cbox = nullptr;     // [n*capF][8]: candidate t of the current stage = {frame, y0, x0, ih, iw, 0, 0, 0} (pad()'s crop window)
    int32_t* flags = nullptr;        // [TRL_NFLAGS]: see trl_cascade.hip
    size_t arena_mark = 0;           // arena offset behind the cascade's own blocks (where a resumed attempt re-allocates the API outputs)
    char* spill = nullptr;           // global-memory workspace of the NMS spill tier (lists longer than the LDS tier)
    size_t spill_cap = 0;
};
enum { TRL_NFLAGS = 64 };            // int32 words of CascadeBufs::flags (copied to pinned host memory behind every call)

struct trl_ctx {
    trl_config cfg;
    bool have_weights = false;
    char* wdev = nullptr;            // all weights, device
    size_t wbytes = 0;
    std::unordered_map<std::string, DevW> W;
    std::unordered_map<std::string, DevV> V;
    Arena arena;                     // per-call persistent blocks (cascade lists)
    Arena scratch;                   // transient activations; only ever grown while empty
    Arena sims_tmp;                  // similarities when trl_drift_score is called with d_sims == NULL
    int32_t* h_pinned = nullptr;     // small pinned host scratch
    CascadeBufs cb;
    LevelGeom lv[32];
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pnet_ev;
    int pnet_ev_used = 0;
    float last_ms[4] = {0, 0, 0, 0};
    int pnet_mono1 = 0;              // conv1 PReLU slopes all >= 0
    int pnet_unit = 0;               // no PNet PReLU slope above 1 (negative ones allowed): prelu(v) == max(v, s v)
    int pnet_run = 0;                // > 0: tiles per cursor fetch of the fused PNet launch (trl_debug_pnet_run); 0 = automatic
    int32_t* pnet_cursor = nullptr;           // device: 8 per-XCD tile cursors of the fused PNet launch
    unsigned long long* pnet_clk = nullptr;   // device: [0] first-start / [1] last-end wall clock of the fused PNet launch in flight, [2..35] phase clocks (DBG), [36] summed spans, [37] launches
    bool pnet_prof = false;                   // TRL_PNET_CLOCK: the DBG instantiation with per-phase wave clocks
    float pnet_kernel_ms = 0.f;      // its span in ms (collect_timings)
    int dbg_poison = -1;             // >= 0 after trl_debug_poison: byte written into every newly allocated workspace
    // Optimistic capacities of the R-Net / O-Net candidate batches (candidates per frame): launches are sized by them and
    // exit early on the device; a call that overflows one is re-run with a larger value (trl_cascade.hip)
    float t2_per_frame = 160.f, t3_per_frame = 48.f;
    int cap_t2 = 0, cap_t3 = 0;      // capacities of the call in progress
    // Candidate-list capacities follow the content the same way: the largest per-level count / per-frame stage-1 total / spill
    // workspace recent calls needed (with head-room, decaying towards the configured start values)
    float lvl_hint[32] = {0}, frame_hint = 0.f;
    size_t spill_hint = 0;
    int rnet_chunk = 49152, onet_chunk = 16384;   // R-/O-Net candidates per launch set (trl_cascade.hip; trl_debug_option shrinks them for tests)
    int nms_small = 512, nms_full = 2048;   // LDS tiers of the sort + NMS kernels (candidates); longer lists take the spill tier
                                            // (trl_debug_nms_tiers lowers them so that small test inputs reach every tier)
    int last_attempts = 0;           // attempts the last call took (test hook)
    int resume_stage = 0;            // 2 / 3: the next attempt of the call in progress starts at that stage (trl_cascade_check)
    size_t scratch_after_cascade = 0;   // scratch bytes the rest of the call needs (crops + FaceNet): sized with the cascade's
    int rnet_front_mode = -1, onet_front_mode = -1;   // conv1 PReLU slope class (trl_front.hip), -1 = not yet classified
    // the call queued by trl_detect_embed_begin / trl_detect_crop_begin and not yet finished by trl_detect_embed_end
    struct Pending {
        bool active = false; int attempt = 0;
        const uint8_t* frames = nullptr; int n = 0, H = 0, W = 0;
        float* box = nullptr; float* prob = nullptr; int32_t* rect = nullptr; uint8_t* valid = nullptr; float* emb = nullptr;
        float* faces_out = nullptr; void* stream = nullptr;
    } pend;
    uint32_t* pyr_tab = nullptr;     // pyramid bin-edge tables for the last (H, W)
    int pyr_tab_H = 0, pyr_tab_W = 0;
    // one-pass kernel for the finest levels (k_pyramid_fine): ownership table inside pyr_tab, source tile shape; nlev == 0: not usable for this shape
    struct { int nlev = 0, own0 = 0, band_cols = 0, strip_rows = 0, n_bands = 0, n_strips = 0; } pyr_fine;
};

// Optional device-wide ordering of the wide phases of different contexts (trl_api.hip; OFF by default, trl_debug_option
// "pnet_gate"): a fused PNet launch waits for the END of the most recent call queued on the device, whichever context queued it,
// so it runs alone (its HIP event pair is its duration) while the memory-bound pyramid kernels in front of it overlap the previous
// call's narrow R-/O-Net / embedder kernels.  Measured with two contexts in flight: 18.8 k frames/s and a clean clock, against
// 19.1 k with the launches left to the hardware queues and 18.6 k with one context (profiles/round4_pnet_gate_ab.txt).
int trl_gate_wait(trl_ctx* c, hipStream_t s);      // before the fused PNet launch
int trl_gate_record(trl_ctx* c, hipStream_t s);    // behind the last kernel of a call
extern int g_trl_pnet_gate;                        // trl_debug_option("pnet_gate"): 0 (default) / 1
int trl_ensure(trl_ctx* c, Arena& a, size_t bytes);   // grow (never while blocks of `a` are live)
const DevW* trl_w(trl_ctx* c, const std::string& name);
const DevV* trl_v(trl_ctx* c, const std::string& name);

// networks (trl_nets.hip)
int trl_run_facenet(trl_ctx* c, const float* d_faces, int n, int h, int w, const uint8_t* d_valid, float* d_emb, hipStream_t s);
int trl_run_rnet(trl_ctx* c, const float* d_crops, int n, float* d_out6, hipStream_t s);
int trl_run_onet(trl_ctx* c, const float* d_crops, int n, float* d_out16, hipStream_t s);
// n = CAPACITY of the launch; the candidates that exist are clamp(*n_dev - n_base, 0, n) (device-sized, no host sync)
int trl_run_rnet_tail(trl_ctx* c, const float* d_pool1, int n, float* d_out6, hipStream_t s, const int32_t* n_dev = nullptr, int n_base = 0);
int trl_run_onet_tail(trl_ctx* c, const float* d_pool1, int n, float* d_out16, hipStream_t s, const int32_t* n_dev = nullptr, int n_base = 0);
int trl_launch_rnet_front(trl_ctx* c, const uint8_t* d_frames, int H, int W, const int32_t* d_total, int t0, int nc, float* d_pool, hipStream_t s);
int trl_launch_onet_front(trl_ctx* c, const uint8_t* d_frames, int H, int W, const int32_t* d_total, int t0, int nc, float* d_pool, hipStream_t s);
// PNet on one materialised level for nf frames: heads [nf][oh][ow][6]
int trl_run_pnet_generic(trl_ctx* c, const float* d_level, int nf, int h, int w, float* d_heads, hipStream_t s);
size_t trl_pnet_generic_bytes(int nf, int h, int w);

// cascade (trl_cascade.hip)
int trl_cascade_detect(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, hipStream_t s, int resume = 0);
int trl_cascade_finish(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_boxes, float* d_probs, float* d_points,
                       int32_t* d_counts, float* d_box0, float* d_prob0, int32_t* d_rect, uint8_t* d_valid, float* d_pts0, hipStream_t s);
int trl_cascade_check(trl_ctx* c, int n, int* retry);   // after the call's stream synchronisation
int trl_launch_crop_resize80(const uint8_t* d_frames, int n, int H, int W, const int32_t* d_rect, const uint8_t* d_valid,
                             float* d_faces, hipStream_t s);
int trl_launch_crop_aligned(const uint8_t* d_frames, int n, int H, int W, const float* d_pts0, const uint8_t* d_valid, int S, bool rgb,
                            float* d_faces, hipStream_t s);
int trl_launch_crop_area_std(const uint8_t* d_frames, int n, int H, int W, const int32_t* d_rect, const uint8_t* d_valid, int S,
                             bool rgb, float* d_faces, hipStream_t s);
int trl_launch_area_level(const uint8_t* d_frames, int nf, int H, int W, int h, int w, float* d_level, hipStream_t s);
int trl_launch_heads_to_maps(const float* d_heads, int cells, float* d_prob, float* d_reg, hipStream_t s);
int trl_compute_levels(trl_ctx* c, int H, int W);
// fused PNet (trl_pnet.hip)
int trl_pnet_prepare(trl_ctx* c);
size_t trl_pnet_fused_bytes(trl_ctx* c, int n, int H, int W);
int trl_pnet_fused_all(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, hipEvent_t* ev, hipStream_t s);
int trl_pyramid_export(trl_ctx* c, const uint8_t* d_frame, int H, int W, int level, float* d_out, int* h, int* w, hipStream_t s);
