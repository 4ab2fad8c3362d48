/*
 * truely_hip.h -- C ABI of libtruely_hip.so: the MI355X (gfx950) implementation of Truely's
 * per-frame visual hot path.
 *
 * The reference has no FFI / plugin layer: its boundary is the Python function
 *     run(video_path_one, video_path_two) -> int          (server/model.py:11-14)
 * imported by the FastAPI handlers (server/server.py:35,611,856), which internally makes two
 * library calls per sampled frame:
 *     boxes, _ = mtcnn.detect(frame)                       (server/model.py:47)
 *     emb = facenet_model(face_tensor)                     (server/model.py:59)
 * followed by the cosine-drift state machine (server/model.py:60-66,86-95).
 * Every entry point below names the reference lines it replaces.  Plain pointers and sizes only:
 * no torch types.  All `d_*` pointers are DEVICE pointers owned by the caller (e.g. the
 * data_ptr() of torch-ROCm tensors); `stream` is a hipStream_t passed as void*.
 *
 * Error convention: every function returns TRL_OK (0) or a negative trl_status; a thread-local
 * message is available from trl_last_error().  (model.run() itself maps failures to the
 * reference's `return 0`, server/model.py:20-34,83-88.)
 *
 * Threading: a context is bound to one device and one in-flight call; use one context per GPU /
 * rank.  Calls on distinct contexts are independent.
 */
#ifndef TRUELY_HIP_H
#define TRUELY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRL_ABI_VERSION 7

typedef enum {
    TRL_OK = 0,
    TRL_ERR_INVALID = -1,   /* bad argument / shape */
    TRL_ERR_HIP = -2,       /* HIP runtime error (message has hipGetErrorString) */
    TRL_ERR_WEIGHTS = -3,   /* blob malformed or tensor missing */
    TRL_ERR_CAPACITY = -4,  /* (ABI <= 6: a candidate list exceeded its configured capacity.  Since ABI 7 no input can
                             *  produce it: lists grow to what the content needs, as detect_face() has no limit) */
    TRL_ERR_STATE = -5      /* call order (e.g. no weights loaded) */
} trl_status;

typedef struct trl_ctx trl_ctx;

/* MTCNN() constructor arguments that affect detect() (facenet_pytorch MTCNN.__init__ defaults,
 * as instantiated at server/model.py:18) plus capacities of the device-side candidate lists. */
typedef struct {
    int    device;          /* HIP device ordinal */
    int    min_face_size;   /* 20 */
    float  thr0, thr1, thr2;/* 0.6 0.7 0.7 */
    double factor;          /* 0.709 */
    int    cap_level;       /* START capacity of the per-(frame, pyramid level) candidate lists   [2048] */
    int    cap_frame;       /* START capacity of the per-frame box lists of stages 1-3           [2048]
                             * Lists are never longer than the geometry allows (a level has oh x ow cells) and grow with
                             * the content: a call whose frames need more is re-run internally with larger lists (and
                             * sorted / suppressed through a global-memory tier beyond the LDS tier), never refused. */
    int    max_faces;       /* max boxes returned per frame by trl_mtcnn_detect          [64]   */
    int    pnet_mode;       /* 0 = fused PNet kernel, 1 = generic layer path (validation) */
    int    embed_mode;      /* 0 = reference: 80x80 INTER_LINEAR crop, BGR, /255 (model.py:41,57-58)  [default]
                             * 1 = SURVEY 8(f)-4 native mode: facenet-pytorch extract_face (160x160 area
                             *     resample, .byte(), (x-127.5)/128), channel order kept; 2 = same, BGR->RGB
                             * 3 = 8(f)-4 "landmark-aligned": 160x160 similarity warp of the largest face from its five
                             *     O-Net landmarks to the scaled 112x112 five-point template (bilinear, replicated
                             *     borders), (x-127.5)/128, RGB -- this project's definition, oracle: orc_crop_aligned */
    int    embed_precision; /* 0 = f32, bit-exact with the oracle                                        [default]
                             * 1 = bf16 activations/weights on the bf16 matrix cores for InceptionResnetV1 only
                             *     (BASELINE configs[2]); detector, crops and valid mask stay f32-exact; embeddings
                             *     agree with the f32 path to ~1e-2 (cosine > 0.999), see tests/test_gpu_api.py
                             * 2 = the same on fp16 (BASELINE configs[4]): three more mantissa bits, cosine > 0.99999 */
} trl_config;

int  trl_abi_version(void);
const char* trl_last_error(void);
int  trl_default_config(trl_config* cfg);

/* Replaces the per-call model construction at server/model.py:18-19: create once, load the
 * packed weights (TRLW0001 blob from weights.pack_state_dicts) once, reuse for every clip. */
int  trl_create(const trl_config* cfg, trl_ctx** out);
int  trl_destroy(trl_ctx* ctx);
int  trl_load_weights(trl_ctx* ctx, const void* host_blob, size_t nbytes);

/* server/model.py:47  `boxes, probs = mtcnn.detect(frame)` for a batch of n frames.
 *   d_frames : u8  [n][H][W][3]  BGR as cv2.VideoCapture.read() yields (model.py:43)
 *   d_boxes  : f32 [n][max_faces][4]   x1,y1,x2,y2, largest area first (select_largest=True)
 *   d_probs  : f32 [n][max_faces]
 *   d_counts : i32 [n]                 number of faces (0 <=> detect() returned None) */
int  trl_mtcnn_detect(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                      float* d_boxes, float* d_probs, int32_t* d_counts, void* stream);

/* `mtcnn.detect(frame, landmarks=True)`: the same call returning the 5 facial landmarks O-Net predicts as well.  The
 * reference discards them (`boxes, _ = mtcnn.detect(frame)`, server/model.py:47); facenet-pytorch computes them on every
 * call (detect_face.py stage 3), so they are exported for callers that align faces.
 *   d_points : f32 [n][max_faces][10]  x0..x4, y0..y4 of each face, same order as d_boxes */
int  trl_mtcnn_detect_landmarks(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                                float* d_boxes, float* d_probs, float* d_points, int32_t* d_counts, void* stream);

/* server/model.py:59  `facenet_model(face_tensor)`:  InceptionResnetV1(...).eval() forward.
 *   d_faces : f32 [n][h][w][3] NHWC, already scaled as model.py:58 does (to_tensor: /255)
 *   d_emb   : f32 [n][512], L2-normalised */
int  trl_facenet_embed(trl_ctx* ctx, const float* d_faces, int n, int h, int w, float* d_emb, void* stream);

/* server/model.py:47-59 fused for a batch of sampled frames: detect, take boxes[0], int-cast +
 * clamp (model.py:49-53), crop, cv2.resize(...,(80,80)) (model.py:55-57), to_tensor (model.py:58),
 * embed (model.py:59).
 *   d_box   : f32 [n][4]   boxes[0] (zeros if none)
 *   d_prob  : f32 [n]
 *   d_rect  : i32 [n][4]   clamped integer crop rectangle x0,y0,x1,y1
 *   d_valid : u8  [n]      1 iff an embedding was produced for the frame (model.py:48,54,56)
 *   d_emb   : f32 [n][512] (zero rows where !valid) */
int  trl_detect_embed(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                      float* d_box, float* d_prob, int32_t* d_rect, uint8_t* d_valid, float* d_emb,
                      void* stream);

/* The two halves of trl_detect_embed, for callers that embed the faces of SEVERAL frame batches in one embedder call
 * (the embedder's 100 small launches amortise over more faces: 2.22 ms per 256 faces alone, 1.77 ms at 768 per call).
 * trl_detect_crop = model.py:47-58: d_faces [n][S][S][3] f32 receives the crops (S = 80, or 160 in the native embed modes; zero
 * where !valid).  trl_facenet_embed_masked = model.py:59 with zero rows where !valid.  Results are bit-identical to
 * trl_detect_embed whatever the grouping. */
int  trl_detect_crop(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                     float* d_box, float* d_prob, int32_t* d_rect, uint8_t* d_valid, float* d_faces, void* stream);
int  trl_facenet_embed_masked(trl_ctx* ctx, const float* d_faces, const uint8_t* d_valid, int n, int h, int w, float* d_emb, void* stream);

/* The same two calls split into "queue" and "finish", so ONE host thread can keep several contexts (one per batch in flight, or
 * one per GPU) busy without a thread per context: *_begin validates, queues every kernel of the call on `stream` and returns
 * without synchronising; trl_detect_embed_end is the call's one host synchronisation, checks the candidate capacities and -- rarely,
 * when a candidate list or an optimistic R-/O-Net batch capacity was too small for the content -- re-runs the call with larger
 * ones before returning.  Outputs are valid after _end.  A
 * context holds at most one call in flight (TRL_ERR_STATE from every entry point that would touch its workspaces meanwhile); the buffers must stay alive until _end returns.
 * server/model.py:42-59 is strictly one frame at a time; this is the batched, overlapped form of the same two library calls. */
int  trl_detect_embed_begin(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                            float* d_box, float* d_prob, int32_t* d_rect, uint8_t* d_valid, float* d_emb, void* stream);
int  trl_detect_crop_begin(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                           float* d_box, float* d_prob, int32_t* d_rect, uint8_t* d_valid, float* d_faces, void* stream);
int  trl_detect_embed_end(trl_ctx* ctx);

/* server/model.py:60-66,70,75,86-95: cosine similarity against the last embedded frame, the
 * run-length counter, and the 0..100 score.  n = number of sampled frames (in time order),
 * frame_count = frames decoded, fps as int(cap.get(CAP_PROP_FPS)) (model.py:28).
 *   d_sims   : f32 [n]  (2.0 where no comparison happened)        may be NULL
 *   d_flags  : u8  [n]  1 where the frame was counted "AI detected" (model.py:66)  may be NULL
 *   d_result : i32 [4]  {score, final run length, hits, total sampled frames} */
int  trl_drift_score(trl_ctx* ctx, const float* d_emb, const uint8_t* d_valid, int n,
                     long long frame_count, int fps, float* d_sims, uint8_t* d_flags,
                     int32_t* d_result, void* stream);

/* The same state machine continued across the WINDOWS of one clip, for callers that stream (model.run): d_state
 * (TRL_DRIFT_STATE_BYTES device bytes, zero-filled before the clip's first window) carries what model.py's loop variables
 * `previous_embedding`, `consecutive_count`, `ai_detected_frames` hold between two sampled frames (model.py:60-75).  The n
 * embeddings of the window are compared in order, the first one with the carried embedding; d_sims / d_flags (may be NULL)
 * cover the window; d_result as above, computed for `frame_count` = frames decoded so far (n = 0 with the clip's final count
 * gives the final score).  Windows of any sizes give the similarities, flags and score of one trl_drift_score over the clip. (ABI v7) */
#define TRL_DRIFT_STATE_BYTES 2064
int  trl_drift_update(trl_ctx* ctx, void* d_state, const float* d_emb, const uint8_t* d_valid, int n,
                      long long frame_count, int fps, float* d_sims, uint8_t* d_flags, int32_t* d_result, void* stream);

/* SURVEY 8(f)-1, device-side ingest in place of the CPU decode + sampling at server/model.py:43,46:
 * d_nv12 holds n_in decoder-output frames (NV12: H*W luma bytes, then H/2 rows of interleaved U,V);
 * frames 0, step, 2*step, ... are converted to u8 BGR [n_out][H][W][3] (OpenCV's integer BT.601
 * limited-range arithmetic) ready for trl_detect_embed.  step = max(1, int(fps / 7)) (model.py:40). */
int  trl_ingest_nv12(trl_ctx* ctx, const uint8_t* d_nv12, int n_in, int H, int W, int step,
                     uint8_t* d_bgr, int* n_out, void* stream);
/* the same for PLANAR 4:2:0 (I420: Y plane, U plane, V plane -- YUV4MPEG2 files, software decoders): no host-side repacking (ABI v7) */
int  trl_ingest_i420(trl_ctx* ctx, const uint8_t* d_i420, int n_in, int H, int W, int step,
                     uint8_t* d_bgr, int* n_out, void* stream);

/* ---- inspection hooks used by the parity tests (stage-by-stage vs the oracle) ------------- */
/* Boxes of one frame after cascade stage 1/2/3 of the LAST trl_mtcnn_detect / trl_detect_embed
 * call (host output, rows of 5: x1,y1,x2,y2,score).  Returns the count in *n_out. */
int  trl_debug_stage_boxes(trl_ctx* ctx, int stage, int frame, float* h_boxes, int max_rows, int* n_out);
/* Per-level PNet candidate / kept counts of one frame (host output, up to 32 levels each). */
int  trl_debug_level_counts(trl_ctx* ctx, int frame, int32_t* h_cand, int32_t* h_keep, int* n_levels);
/* Candidate records of one (frame, pyramid level) of the last call, as the PNet kernel appended them (arbitrary order):
 * rows of 40 bytes {x1,y1,x2,y2,score,r0,r1,r2,r3 : f32; cell : i32}, cell = y*ow + x of the PNet output map.  With
 * thr0 = 0 every cell is a candidate, so this reads the fused kernel's own probability / regression maps. */
int  trl_debug_level_cands(trl_ctx* ctx, int frame, int level, void* h_rows, int max_rows, int* n_out);
/* The per-level NMS picks (detect_face.py stage 1: batched_nms 0.5 per scale) of one (frame, level) of the last call: indices into
 * the rows trl_debug_level_cands returns (append order), in pick order.  With it a test can check the keep set of a list of ANY
 * length against the definition of greedy NMS (a box is kept iff no kept box of higher priority overlaps it by more than the
 * threshold) without an O(n^2) reference run. (ABI v7) */
int  trl_debug_level_keep(trl_ctx* ctx, int frame, int level, int32_t* h_idx, int max_rows, int* n_out);
/* test hook: LDS tiers of the sort + NMS kernels in candidates per list (0 keeps a value; multiples of 4 in [16, 3072]; defaults
 * 512 / 2048).  Lists longer than `full_tier` are sorted and suppressed in global memory (the spill tier); lowering the tiers
 * lets small inputs reach it.  Results never depend on the tiers. (ABI v7) */
int  trl_debug_nms_tiers(trl_ctx* ctx, int small_tier, int full_tier);
/* candidate-list statistics of the last call: h_out8 = {attempts, lists that took the spill tier, spill bytes used, spill bytes
 * available, rows per frame of the stage lists, record slots per frame over all levels, largest per-level candidate count,
 * largest per-frame stage-1 total} (ABI v7) */
int  trl_debug_list_stats(trl_ctx* ctx, long long* h_out8);
/* test hooks that used to be environment variables (the library reads none; experiment switches exist only in a `make TUNING=1`
 * build): key "rnet_chunk" / "onet_chunk" = candidates per R-/O-Net launch set of this context (>= 16: small inputs then run
 * the multi-chunk path); "no_fnconv" (process-wide, ctx may be NULL) = FaceNet's small maps through the generic conv kernels,
 * value 0 restores the default; "pnet_gate" (process-wide, default 0) = 1: a fused PNet launch waits for the end of the previous
 * call queued on its device by ANY context, so that with several contexts in flight it runs alone (a timing aid: the HIP event
 * pair of trl_debug_timings is then the launch's duration).  Results never depend on them. (ABI v7) */
int  trl_debug_option(trl_ctx* ctx, const char* key, int value);
/* test hook: the R-/O-Net launches are sized by optimistic per-frame candidate capacities; set them (<= 0 keeps a value) and read
 * how many attempts the last call took (a too-small capacity makes the call re-run itself with a larger one) */
int  trl_debug_batch_capacity(trl_ctx* ctx, float t2_per_frame, float t3_per_frame, int* last_attempts);
/* test hook: level `level` of one frame's image pyramid as the fused PNet kernel reads it; d_out holds h*w*3 floats
 * (capacity: at least (int(H*m+1))*(int(W*m+1))*3 with m = 12/min_face_size) */
int  trl_debug_pyramid_level(trl_ctx* ctx, const uint8_t* d_frame, int H, int W, int level, float* d_out, int* h, int* w, void* stream);
/* test hook: fill the activation workspaces with a byte pattern (0xFF -> NaNs) before the next call */
int  trl_debug_poison(trl_ctx* ctx, int byte);
/* PNet on one pyramid level of frame 0: face-prob map and regression map (device outputs). */
int  trl_debug_pnet_level(trl_ctx* ctx, const uint8_t* d_frame, int H, int W, int level,
                          float* d_prob, float* d_reg, int* oh, int* ow, void* stream);
/* R-Net / O-Net on prepared crops: d_crops f32 [n][S][S][3]; d_out f32 [n][6] / [n][16]
 * = {logit0, logit1, reg[4], (landmarks[10])}. */
int  trl_debug_rnet(trl_ctx* ctx, const float* d_crops, int n, float* d_out, void* stream);
int  trl_debug_onet(trl_ctx* ctx, const float* d_crops, int n, float* d_out, void* stream);
/* The production stage-2 / stage-3 network path (fused front kernel + layer tail) on caller-chosen boxes of ONE frame:
 * h_boxes = nb host rows x1,y1,x2,y2 (as after rerec); net = 24 (R-Net, d_out [nb][6]) or 48 (O-Net, d_out [nb][16]). */
int  trl_debug_front_net(trl_ctx* ctx, const uint8_t* d_frame, int H, int W, const float* h_boxes, int nb, int net,
                         float* d_out, void* stream);
/* model.py:55-58 alone: crop rect (x0,y0,x1,y1 per frame, i32) -> f32 [n][80][80][3] in [0,1] */
int  trl_debug_crop_resize(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W,
                           const int32_t* d_rect, const uint8_t* d_valid, float* d_faces, void* stream);
/* embed_mode 3's crop alone: d_pts [n][10] = x0..x4, y0..y4 per frame -> f32 [n][S][S][3] */
int  trl_debug_crop_aligned(trl_ctx* ctx, const uint8_t* d_frames, int n, int H, int W, const float* d_pts,
                            const uint8_t* d_valid, int S, int rgb, float* d_faces, void* stream);
/* Time (ms, HIP events on the call's stream) of the last trl_detect_embed call:
 * out[0] = PNet kernel (fused: the one persistent launch; generic: sum over levels),
 * out[1] = whole call, out[2] = number of PNet launches timed, out[3] = pyramid kernel. */
int  trl_debug_timings(trl_ctx* ctx, float* out4);
/* R-Net / O-Net candidate totals of the last call over the whole batch: h_out2[0] = boxes that entered stage 2, [1] = stage 3 */
int  trl_debug_stage_totals(trl_ctx* ctx, int32_t* h_out2);
/* test / tuning hook: consecutive tiles (band order: three tile rows, column by column) a workgroup of the fused PNet launch takes
 * per cursor fetch (0 = automatic: 24 for large batches, down to 1 for small ones).  With runs > 1 a tile reuses the halo columns /
 * rows its left / upper neighbour computed; same results. */
int  trl_debug_pnet_run(trl_ctx* ctx, int run);
/* TUNING builds only (0 otherwise): execution span (first workgroup start -> last workgroup end, device wall clock) of the last
 * fused PNet launch, in ms */
int  trl_debug_pnet_kernel_ms(trl_ctx* ctx, float* ms);
/* TUNING builds with TRL_PNET_SPAN set only (0 launches otherwise): the same span summed on the device over every fused PNet
 * launch of this context since the last reset.  The context must be idle.  The shipped library times the launch with the HIP
 * event pair on its stream (trl_debug_timings out[0]); fused launches of different contexts on one device are ordered one after
 * the other, so that pair measures execution, not queueing. (ABI v6) */
int  trl_debug_pnet_span(trl_ctx* ctx, int reset, double* ms_sum, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif
