/*
 * trl_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the visual hot path of the reference
 *   server/model.py::run  (/root/reference/server/model.py:11-95)
 * and of the third-party numerics it calls (facenet_pytorch==2.6.0 MTCNN +
 * InceptionResnetV1, torchvision 0.17.2 nms/to_tensor, OpenCV 4.x resize), which
 * are NOT present under /root/reference nor installed in the build container.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and its
 * dependencies cannot be imported here.  This oracle is
 * pinned only by (a) hand-derived known-answer tests for the score state machine
 * (model.py:60-66,86-95), (b) primitive-level cross-checks against torch CPU ops
 * (oracle/torch_ref.py), (c) its own committed fixtures under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path never calls it.
 *
 * Arithmetic contract (what makes GPU parity BIT-exact rather than "close"):
 *   - every convolution / linear layer accumulates as ONE f32 fused-multiply-add
 *     chain per output element:  acc = bias (or 0);  for k ascending:
 *     acc = fmaf(x[k], w[k], acc),  k = (ky*KW + kx)*Cin + c   (NHWC, HWIO).
 *     gfx950's v_mfma_f32_32x32x2_f32 / 16x16x4_f32 compute exactly this chain.
 *     Exception (layer rule, batch independent): when OH*OW <= 9 and K >= 512 the chain is cut into four
 *     consecutive quarters of k, combined as (c0 + c1) + (c2 + c3), c0 seeded with the bias.
 *   - elementwise steps mirror the torch / numpy expression order, one IEEE
 *     rounding per operation, no contraction (-ffp-contract=off on both sides).
 *   - exp() is a fixed fmaf polynomial (orc_expf) shared verbatim with the device code.
 *   - 512-long dot products use a fixed 64-lane strided + butterfly order (orc_dot512).
 */
#ifndef TRL_ORACLE_H
#define TRL_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* MTCNN() defaults used by model.py:18 (facenet_pytorch MTCNN.__init__). */
typedef struct {
    int    min_face_size;  /* 20 */
    float  thr0, thr1, thr2; /* 0.6, 0.7, 0.7 */
    double factor;         /* 0.709 */
} orc_params;

/* Per-frame trace of the cascade, for stage-by-stage GPU parity tests. */
typedef struct {
    int   max_boxes;     /* capacity of each array below (rows) */
    int   n_scales;
    int   n_cand_scale[32];   /* PNet cells passing thr0, per scale */
    int   n_keep_scale[32];   /* after per-scale NMS 0.5 */
    int   n1;  float* boxes1; /* [n1][5] after cross-scale NMS 0.7 + regress + rerec */
    int   n2;  float* boxes2; /* [n2][5] after RNet thr, NMS 0.7, bbreg, rerec */
    int   n3;  float* boxes3; /* [n3][5] final (pick order of the 'Min' NMS) */
    float* points3;           /* [n3][10] landmarks x0..x4,y0..y4 (may be NULL) */
} orc_trace;

orc_ctx* orc_create(const void* blob, size_t nbytes);
void     orc_destroy(orc_ctx*);
const char* orc_last_error(void);
void     orc_set_threads(int n);
orc_params orc_default_params(void);

/* ---- primitives ---------------------------------------------------------- */
float orc_expf(float x);
float orc_dot512(const float* a, const float* b);
int   orc_scales(int H, int W, int minsize, double factor, double* scales, int* hs, int* ws, int max);
/* adaptive-avg-pool (interpolate mode="area") of the u8 HWC crop rows [y0,y1) cols [x0,x1)
 * to (oh,ow), then (v-127.5)*0.0078125; out is HWC f32. */
void  orc_area_resample_norm(const uint8_t* img, int H, int W, int y0, int y1, int x0, int x1,
                             int oh, int ow, float* out);
/* PNet on one normalised level (HWC f32, h x w x 3). prob: [oh*ow] face prob, reg: [oh*ow][4]. */
void  orc_pnet_level(const orc_ctx*, const float* in, int h, int w, float* prob, float* reg, int* oh, int* ow);
void  orc_rnet(const orc_ctx*, const float* crops /*[n][24][24][3]*/, int n, float* prob, float* reg);
void  orc_onet(const orc_ctx*, const float* crops /*[n][48][48][3]*/, int n, float* prob, float* reg, float* pts);
/* greedy IoU NMS as torchvision.ops.nms: returns #kept, keep[] = indices in descending-score order */
int   orc_nms_iou(const float* boxes /*[n][4]*/, const float* scores, int n, float thr, int* keep);
/* facenet_pytorch nms_numpy(..., 'Min'), +1 areas; ties resolved as a stable ascending argsort */
int   orc_nms_min(const float* boxes, const float* scores, int n, float thr, int* keep);
/* OpenCV INTER_LINEAR u8 fixed-point resize of frame[y0:y1, x0:x1] to 80x80 (model.py:55-57) */
void  orc_resize_linear_u8(const uint8_t* img, int H, int W, int y0, int y1, int x0, int x1,
                           int oh, int ow, uint8_t* out);
/* InceptionResnetV1.eval() forward, input NHWC f32 [n][H][W][3] already /255 (model.py:58-59) */
void  orc_facenet(const orc_ctx*, const float* in, int n, int H, int W, float* emb /*[n][512]*/);

/* ---- the cascade (MTCNN.detect, select_largest=True) --------------------- */
/* boxes_out [max_out][4], probs_out [max_out], sorted largest-area first. returns #boxes. */
int   orc_detect(const orc_ctx*, const uint8_t* frame, int H, int W, const orc_params*,
                 float* boxes_out, float* probs_out, int max_out, orc_trace* trace);

/* model.py:47-59 for a batch of already-sampled frames (u8 BGR HWC).
 *   box_out  [n][4] float  : boxes[0] of MTCNN.detect (largest face) or zeros
 *   prob_out [n]
 *   rect_out [n][4] int32  : model.py:49-53 int-cast + clamped crop rectangle x0,y0,x1,y1
 *   valid_out[n] u8        : 1 iff a face was embedded (model.py:48,54,56)
 *   emb_out  [n][512]      : embedding (zeros when !valid)
 *   face_out [n][80][80][3] u8 resized crop (may be NULL) */
int   orc_detect_embed(const orc_ctx*, const uint8_t* frames, int n, int H, int W, const orc_params*,
                       float* box_out, float* prob_out, int32_t* rect_out, uint8_t* valid_out,
                       float* emb_out, uint8_t* face_out);

/* model.py:60-66,70,75,86-95.  emb [n][512], valid [n]; n = number of SAMPLED frames,
 * frame_count = total decoded frames, step = max(1,int(fps/7)).
 * sims_out [n] (NaN-free: 2.0f where no comparison happened), flag_out [n] u8 = 1 when the
 * sampled frame is counted as deepfake (model.py:66). Returns the 0..100 score. */
int   orc_drift_score(const float* emb, const uint8_t* valid, int n, long frame_count, int fps,
                      float* sims_out, uint8_t* flag_out, int* final_run, int* hits);

/* mode 0 = reference (80x80 INTER_LINEAR, BGR, /255); 1 = facenet-pytorch extract_face for tensor
 * input (160x160 area resample, .byte(), (x-127.5)/128), BGR kept; 2 = the same with BGR->RGB. */
void  orc_crop_area_std(const uint8_t* img, int H, int W, int x0, int y0, int x1, int y1, int S, int rgb, float* out);
/* mode 3 of orc_detect_embed_mode: five-point similarity alignment (see trl_oracle.c); prm = {a, b, Tx, Ty, Px, Py} */
void  orc_align_params(const float* pts, double* prm);
void  orc_crop_aligned(const uint8_t* img, int H, int W, const float* pts, int S, int rgb, float* out);
int   orc_detect_embed_mode(const orc_ctx*, const uint8_t* frames, int n, int H, int W, const orc_params*, int mode,
                            float* box_out, float* prob_out, int32_t* rect_out, uint8_t* valid_out, float* emb_out);

/* exhaustive check of the device's reciprocal division against IEEE division for bins up to kmax x kmax
 * (csrc/trl_pnet.hip:pyr_div); returns the number of mismatching (kh, kw, sum) triples */
long  orc_selftest_recip_div(int kmax);

/* one NV12 frame (H*W luma + H/2 x W interleaved UV) -> BGR, OpenCV integer BT.601 */
void  orc_nv12_to_bgr(const uint8_t* nv12, int H, int W, uint8_t* bgr);

#ifdef __cplusplus
}
#endif
#endif
