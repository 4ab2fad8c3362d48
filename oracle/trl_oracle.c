/*
 * trl_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See trl_oracle.h.
 *
 * Restates /root/reference/server/model.py:11-95 plus the internals of its third-party
 * calls.  "RECALLED" = restated from the published behaviour of the pinned package
 * (requirements.txt:1,6,11: facenet_pytorch==2.6.0, opencv 4.x, torchvision==0.17.2),
 * which is absent from /root/reference; see SURVEY.md Appendix A.  PARITY UNPINNED.
 *
 * Build: make -C oracle   (gcc -O2 -mfma -mavx2 -ffp-contract=off -fopenmp)
 */
#include "trl_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* weights blob ("TRLW0001"): header, entries, then f32 data.                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    char     name[56];
    uint32_t ndim;
    uint32_t dims[4];
    uint64_t offset;
    uint64_t nbytes;
} trlw_entry; /* 96 bytes */

struct orc_ctx {
    uint8_t*    blob;
    size_t      nbytes;
    uint32_t    n_tensors;
    trlw_entry* entries;
};

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int g_threads = 0;
void orc_set_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

orc_ctx* orc_create(const void* blob, size_t nbytes) {
    if (nbytes < 16 || memcmp(blob, "TRLW0001", 8) != 0) {
        snprintf(g_err, sizeof g_err, "bad weights blob magic");
        return NULL;
    }
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
    c->blob = (uint8_t*)malloc(nbytes);
    memcpy(c->blob, blob, nbytes);
    c->nbytes = nbytes;
    memcpy(&c->n_tensors, c->blob + 8, 4);
    c->entries = (trlw_entry*)(c->blob + 16);
    if (16 + (size_t)c->n_tensors * sizeof(trlw_entry) > nbytes) {
        snprintf(g_err, sizeof g_err, "truncated weights blob");
        orc_destroy(c);
        return NULL;
    }
    return c;
}
void orc_destroy(orc_ctx* c) {
    if (!c) return;
    free(c->blob);
    free(c);
}

static const float* wt(const orc_ctx* c, const char* name, int* d0, int* d1) {
    for (uint32_t i = 0; i < c->n_tensors; i++) {
        if (strncmp(c->entries[i].name, name, 56) == 0) {
            if (d0) *d0 = (int)c->entries[i].dims[0];
            if (d1) *d1 = (int)c->entries[i].dims[1];
            return (const float*)(c->blob + c->entries[i].offset);
        }
    }
    fprintf(stderr, "oracle: missing tensor %s\n", name);
    abort();
}
static const float* wtf(const orc_ctx* c, const char* prefix, const char* suffix, int* d0, int* d1) {
    char buf[128];
    snprintf(buf, sizeof buf, "%s%s", prefix, suffix);
    return wt(c, buf, d0, d1);
}

orc_params orc_default_params(void) {
    orc_params p; /* RECALLED: facenet_pytorch MTCNN.__init__ defaults (model.py:18 passes none) */
    p.min_face_size = 20;
    p.thr0 = 0.6f; p.thr1 = 0.7f; p.thr2 = 0.7f;
    p.factor = 0.709;
    return p;
}

/* ------------------------------------------------------------------------- */
/* scalar primitives                                                          */
/* ------------------------------------------------------------------------- */
/* Cephes-style expf as a fixed fmaf sequence; shared verbatim with the device code so the
 * 2-way softmax is bit-identical.  Domain: x <= 0 (softmax subtracts the max first). */
float orc_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int e = (int)n;
    uint32_t bits = (uint32_t)(e + 127) << 23;
    float s;
    memcpy(&s, &bits, 4);
    return y * s;
}

/* softmax over 2 logits, returns P(class 1).  RECALLED: nn.Softmax(dim=1) = exp(x-max)/sum. */
static float softmax2_p1(float a0, float a1) {
    float m = a0 > a1 ? a0 : a1;
    float e0 = orc_expf(a0 - m), e1 = orc_expf(a1 - m);
    return e1 / (e0 + e1);
}

/* 512-long dot in the fixed "64 lanes x 8 strided, then xor butterfly" order. */
float orc_dot512(const float* a, const float* b) {
    float p[64], q[64];
    for (int j = 0; j < 64; j++) {
        float s = 0.f;
        for (int i = 0; i < 8; i++) s = fmaf(a[j + 64 * i], b[j + 64 * i], s);
        p[j] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int j = 0; j < 64; j++) q[j] = p[j] + p[j ^ off];
        memcpy(p, q, sizeof p);
    }
    return p[0];
}

/* ------------------------------------------------------------------------- */
/* tensors (NHWC f32)                                                         */
/* ------------------------------------------------------------------------- */
typedef struct { int n, h, w, c; float* d; } T;
static T talloc(int n, int h, int w, int c) {
    T t = {n, h, w, c, NULL};
    size_t e = (size_t)n * h * w * c;
    t.d = (float*)malloc((e ? e : 1) * sizeof(float));
    return t;
}
static void tfree(T* t) { free(t->d); t->d = NULL; }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2 };

/* Generic conv, NHWC x HWIO.  One fmaf chain per output, k = (ky*KW+kx)*Cin + c ascending,
 * acc starts at bias[co] (or 0).  Zero-padded taps are skipped (fmaf(0,w,acc)==acc).
 * Epilogue: v = acc; if scale: v = fmaf(v, scale[co], shift[co]);
 *           if res:   v = v * res_scale + res[...]   (torch: out * self.scale + x, two roundings)
 *           act.
 * Output written at y[((n*OH+oy)*OW+ox)*ldy + coff + co]. */
static void conv2d(const T* x, const float* w, const float* bias, int KH, int KW, int sh, int sw,
                   int ph, int pw, int Cout, const float* scale, const float* shift,
                   const float* res, int ldres, float res_scale, int act, const float* slope,
                   float* y, int ldy, int coff, int OH, int OW) {
    const int Cin = x->c, H = x->h, W = x->w;
    const long rows = (long)x->n * OH;
    /* Long reductions over tiny maps (OH*OW <= 9, K >= 512: the 3x3 / 1x1-spatial tail of FaceNet and the
     * dense layers of R/O-Net) are accumulated as FOUR chains over consecutive quarters of k, combined as
     * (c0 + c1) + (c2 + c3); chain 0 starts at the bias.  The rule depends on the layer only, never on the
     * batch, and is what lets the GPU give each quarter to its own wave (see conv_splitk4 in trl_layers.hip). */
    const int Ktot = KH * KW * Cin;
    const int nseg = (OH * OW <= 9 && Ktot >= 512 && (Ktot & 15) == 0) ? 4 : 1;
    const int segK = Ktot / nseg;
#pragma omp parallel for schedule(static) if (rows * OW * Cout * KH * KW * Cin > 2000000L)
    for (long row = 0; row < rows; row++) {
        int n = (int)(row / OH), oy = (int)(row % OH);
        float accs[4][2048];
        float* acc = accs[0];
        for (int ox = 0; ox < OW; ox++) {
            for (int sg = 0; sg < nseg; sg++)
                for (int co = 0; co < Cout; co++) accs[sg][co] = (sg == 0 && bias) ? bias[co] : 0.f;
            for (int ky = 0; ky < KH; ky++) {
                int iy = oy * sh - ph + ky;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < KW; kx++) {
                    int ix = ox * sw - pw + kx;
                    if (ix < 0 || ix >= W) continue;
                    const float* xp = x->d + (((size_t)n * H + iy) * W + ix) * Cin;
                    const float* wp = w + (size_t)((ky * KW + kx) * Cin) * Cout;
                    for (int c = 0; c < Cin; c++) {
                        const float a = xp[c];
                        const float* wr = wp + (size_t)c * Cout;
                        float* ac = accs[nseg == 1 ? 0 : ((ky * KW + kx) * Cin + c) / segK];
                        for (int co = 0; co < Cout; co++) ac[co] = fmaf(a, wr[co], ac[co]);
                    }
                }
            }
            if (nseg == 4)
                for (int co = 0; co < Cout; co++) acc[co] = (accs[0][co] + accs[1][co]) + (accs[2][co] + accs[3][co]);
            size_t pix = ((size_t)n * OH + oy) * OW + ox;
            float* yp = y + pix * ldy + coff;
            for (int co = 0; co < Cout; co++) {
                float v = acc[co];
                if (scale) v = fmaf(v, scale[co], shift[co]);
                if (res) { v = v * res_scale; v = v + res[pix * ldres + co]; }
                if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (act == ACT_PRELU) v = v > 0.f ? v : slope[co] * v;
                yp[co] = v;
            }
        }
    }
}

static int pool_out(int L, int k, int s, int ceil_mode) {
    int o;
    if (ceil_mode) {
        o = (L - k + s - 1) / s + 1;
        if ((o - 1) * s >= L) o--; /* last window must start inside the input */
    } else {
        o = (L - k) / s + 1;
    }
    return o;
}
/* MaxPool2d(k, s, ceil_mode), no padding; windows are clipped to the input. */
static T maxpool(const T* x, int k, int s, int ceil_mode) {
    int OH = pool_out(x->h, k, s, ceil_mode), OW = pool_out(x->w, k, s, ceil_mode);
    T y = talloc(x->n, OH, OW, x->c);
    for (int n = 0; n < x->n; n++)
        for (int oy = 0; oy < OH; oy++)
            for (int ox = 0; ox < OW; ox++) {
                float* yp = y.d + (((size_t)n * OH + oy) * OW + ox) * x->c;
                for (int c = 0; c < x->c; c++) yp[c] = -INFINITY;
                for (int ky = 0; ky < k; ky++) {
                    int iy = oy * s + ky;
                    if (iy >= x->h) break;
                    for (int kx = 0; kx < k; kx++) {
                        int ix = ox * s + kx;
                        if (ix >= x->w) break;
                        const float* xp = x->d + (((size_t)n * x->h + iy) * x->w + ix) * x->c;
                        for (int c = 0; c < x->c; c++) yp[c] = xp[c] > yp[c] ? xp[c] : yp[c];
                    }
                }
            }
    return y;
}

/* MTCNN-style conv (bias, valid padding, stride 1) + PReLU */
static T conv_prelu(const orc_ctx* c, const char* net, const char* conv, const char* prelu,
                    const T* x, int k) {
    char nm[96];
    int K, Cout;
    snprintf(nm, sizeof nm, "%s.%s.w", net, conv);
    const float* w = wt(c, nm, &K, &Cout);
    if (K != k * k * x->c) { fprintf(stderr, "oracle: K mismatch %s\n", nm); abort(); }
    snprintf(nm, sizeof nm, "%s.%s.b", net, conv);
    const float* b = wt(c, nm, NULL, NULL);
    const float* sl = NULL;
    if (prelu) { snprintf(nm, sizeof nm, "%s.%s", net, prelu); sl = wt(c, nm, NULL, NULL); }
    int OH = x->h - k + 1, OW = x->w - k + 1;
    T y = talloc(x->n, OH, OW, Cout);
    conv2d(x, w, b, k, k, 1, 1, 0, 0, Cout, NULL, NULL, NULL, 0, 0.f, prelu ? ACT_PRELU : ACT_NONE,
           sl, y.d, Cout, 0, OH, OW);
    return y;
}

/* ------------------------------------------------------------------------- */
/* MTCNN networks  (RECALLED: facenet_pytorch/models/mtcnn.py PNet/RNet/ONet)  */
/* ------------------------------------------------------------------------- */
void orc_pnet_level(const orc_ctx* c, const float* in, int h, int w, float* prob, float* reg,
                    int* oh, int* ow) {
    T x = {1, h, w, 3, (float*)in};
    T a = conv_prelu(c, "pnet", "conv1", "prelu1", &x, 3);
    T p = maxpool(&a, 2, 2, 1);
    tfree(&a);
    T b = conv_prelu(c, "pnet", "conv2", "prelu2", &p, 3);
    tfree(&p);
    T d = conv_prelu(c, "pnet", "conv3", "prelu3", &b, 3);
    tfree(&b);
    T cls = conv_prelu(c, "pnet", "conv4_1", NULL, &d, 1);
    T box = conv_prelu(c, "pnet", "conv4_2", NULL, &d, 1);
    tfree(&d);
    int n = cls.h * cls.w;
    for (int i = 0; i < n; i++) prob[i] = softmax2_p1(cls.d[2 * i], cls.d[2 * i + 1]);
    memcpy(reg, box.d, (size_t)n * 4 * sizeof(float));
    *oh = cls.h; *ow = cls.w;
    tfree(&cls); tfree(&box);
}

/* dense layers are stored as [K][Cout] with K already permuted by the packer from torch's
 * x.permute(0,3,2,1) (W,H,C) flatten order to NHWC (H,W,C) order, so dense4/dense5 are a
 * KHxKW "valid" conv over the whole 3x3 map. */
void orc_rnet(const orc_ctx* c, const float* crops, int n, float* prob, float* reg) {
    if (n <= 0) return;
    T x = {n, 24, 24, 3, (float*)crops};
    T a = conv_prelu(c, "rnet", "conv1", "prelu1", &x, 3);
    T p1 = maxpool(&a, 3, 2, 1); tfree(&a);
    T b = conv_prelu(c, "rnet", "conv2", "prelu2", &p1, 3); tfree(&p1);
    T p2 = maxpool(&b, 3, 2, 1); tfree(&b);
    T d = conv_prelu(c, "rnet", "conv3", "prelu3", &p2, 2); tfree(&p2);
    T f = conv_prelu(c, "rnet", "dense4", "prelu4", &d, 3); tfree(&d);
    T cls = conv_prelu(c, "rnet", "dense5_1", NULL, &f, 1);
    T box = conv_prelu(c, "rnet", "dense5_2", NULL, &f, 1);
    tfree(&f);
    for (int i = 0; i < n; i++) prob[i] = softmax2_p1(cls.d[2 * i], cls.d[2 * i + 1]);
    memcpy(reg, box.d, (size_t)n * 4 * sizeof(float));
    tfree(&cls); tfree(&box);
}

void orc_onet(const orc_ctx* c, const float* crops, int n, float* prob, float* reg, float* pts) {
    if (n <= 0) return;
    T x = {n, 48, 48, 3, (float*)crops};
    T a = conv_prelu(c, "onet", "conv1", "prelu1", &x, 3);
    T p1 = maxpool(&a, 3, 2, 1); tfree(&a);
    T b = conv_prelu(c, "onet", "conv2", "prelu2", &p1, 3); tfree(&p1);
    T p2 = maxpool(&b, 3, 2, 1); tfree(&b);
    T d = conv_prelu(c, "onet", "conv3", "prelu3", &p2, 3); tfree(&p2);
    T p3 = maxpool(&d, 2, 2, 1); tfree(&d);
    T e = conv_prelu(c, "onet", "conv4", "prelu4", &p3, 2); tfree(&p3);
    T f = conv_prelu(c, "onet", "dense5", "prelu5", &e, 3); tfree(&e);
    T cls = conv_prelu(c, "onet", "dense6_1", NULL, &f, 1);
    T box = conv_prelu(c, "onet", "dense6_2", NULL, &f, 1);
    T lmk = conv_prelu(c, "onet", "dense6_3", NULL, &f, 1);
    tfree(&f);
    for (int i = 0; i < n; i++) prob[i] = softmax2_p1(cls.d[2 * i], cls.d[2 * i + 1]);
    memcpy(reg, box.d, (size_t)n * 4 * sizeof(float));
    if (pts) memcpy(pts, lmk.d, (size_t)n * 10 * sizeof(float));
    tfree(&cls); tfree(&box); tfree(&lmk);
}

/* ------------------------------------------------------------------------- */
/* detect_face helpers (RECALLED: facenet_pytorch/models/utils/detect_face.py) */
/* ------------------------------------------------------------------------- */
int orc_scales(int H, int W, int minsize, double factor, double* scales, int* hs, int* ws, int max) {
    double m = 12.0 / minsize;
    double minl = (H < W ? H : W) * m;
    double scale_i = m;
    int n = 0;
    while (minl >= 12 && n < max) {
        scales[n] = scale_i;
        hs[n] = (int)(H * scale_i + 1);
        ws[n] = (int)(W * scale_i + 1);
        n++;
        scale_i = scale_i * factor;
        minl = minl * factor;
    }
    return n;
}

/* imresample = F.interpolate(mode="area") = adaptive_avg_pool2d: cell i covers
 * [floor(i*in/out), ceil((i+1)*in/out)).  u8 sums are exact integers in f32 at every size
 * used here (< 2^24), so the result is independent of the summation order. */
void orc_area_resample_norm(const uint8_t* img, int H, int W, int y0, int y1, int x0, int x1,
                            int oh, int ow, float* out) {
    (void)H;
    int ih = y1 - y0, iw = x1 - x0;
    for (int oy = 0; oy < oh; oy++) {
        int ys = (int)(((long)oy * ih) / oh), ye = (int)((((long)oy + 1) * ih + oh - 1) / oh);
        for (int ox = 0; ox < ow; ox++) {
            int xs = (int)(((long)ox * iw) / ow), xe = (int)((((long)ox + 1) * iw + ow - 1) / ow);
            uint32_t s[3] = {0, 0, 0};
            for (int y = ys; y < ye; y++) {
                const uint8_t* p = img + ((size_t)(y0 + y) * W + x0 + xs) * 3;
                for (int x = xs; x < xe; x++, p += 3) { s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; }
            }
            /* ATen's CPU adaptive_avg_pool2d divides twice: sum / kh / kw (checked against
             * torch.nn.functional.interpolate(mode="area") bit-for-bit in tests/test_oracle.py) */
            float kh = (float)(ye - ys), kw = (float)(xe - xs);
            for (int c = 0; c < 3; c++) {
                float v = (float)s[c] / kh / kw;
                out[((size_t)oy * ow + ox) * 3 + c] = (v - 127.5f) * 0.0078125f;
            }
        }
    }
}

typedef struct { float s; int i; } sidx;
static int cmp_desc(const void* a, const void* b) {
    const sidx *x = (const sidx*)a, *y = (const sidx*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i - y->i; /* stable */
}
static int cmp_asc(const void* a, const void* b) {
    const sidx *x = (const sidx*)a, *y = (const sidx*)b;
    if (x->s < y->s) return -1;
    if (x->s > y->s) return 1;
    return x->i - y->i; /* stable */
}

/* RECALLED: torchvision/csrc/ops/cpu/nms_kernel.cpp (stable descending sort, areas without +1,
 * suppress when inter/(a_i+a_j-inter) > thr; thr is a double there, equivalent in f32). */
int orc_nms_iou(const float* boxes, const float* scores, int n, float thr, int* keep) {
    if (n <= 0) return 0;
    sidx* order = (sidx*)malloc(sizeof(sidx) * n);
    float* area = (float*)malloc(sizeof(float) * n);
    uint8_t* sup = (uint8_t*)calloc(n, 1);
    for (int i = 0; i < n; i++) {
        order[i].s = scores[i]; order[i].i = i;
        area[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    qsort(order, n, sizeof(sidx), cmp_desc);
    int nk = 0;
    for (int _i = 0; _i < n; _i++) {
        int i = order[_i].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float ia = area[i];
        for (int _j = _i + 1; _j < n; _j++) {
            int j = order[_j].i;
            if (sup[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1; w = w > 0.f ? w : 0.f;
            float h = yy2 - yy1; h = h > 0.f ? h : 0.f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    free(order); free(area); free(sup);
    return nk;
}

/* RECALLED: detect_face.py nms_numpy(boxes, scores, 0.7, 'Min'): +1 areas, I = argsort(s)
 * ascending, repeatedly take the last, keep those with inter/min(area) <= thr.  numpy's default
 * argsort is not stable; ties are resolved here as a stable sort would (documented choice). */
int orc_nms_min(const float* boxes, const float* scores, int n, float thr, int* keep) {
    if (n <= 0) return 0;
    sidx* I = (sidx*)malloc(sizeof(sidx) * n);
    float* area = (float*)malloc(sizeof(float) * n);
    for (int i = 0; i < n; i++) {
        I[i].s = scores[i]; I[i].i = i;
        area[i] = (boxes[4 * i + 2] - boxes[4 * i] + 1.f) * (boxes[4 * i + 3] - boxes[4 * i + 1] + 1.f);
    }
    qsort(I, n, sizeof(sidx), cmp_asc);
    int m = n, nk = 0;
    while (m > 0) {
        int i = I[m - 1].i;
        keep[nk++] = i;
        int m2 = 0;
        for (int t = 0; t < m - 1; t++) {
            int j = I[t].i;
            float xx1 = boxes[4 * i] > boxes[4 * j] ? boxes[4 * i] : boxes[4 * j];
            float yy1 = boxes[4 * i + 1] > boxes[4 * j + 1] ? boxes[4 * i + 1] : boxes[4 * j + 1];
            float xx2 = boxes[4 * i + 2] < boxes[4 * j + 2] ? boxes[4 * i + 2] : boxes[4 * j + 2];
            float yy2 = boxes[4 * i + 3] < boxes[4 * j + 3] ? boxes[4 * i + 3] : boxes[4 * j + 3];
            float w = xx2 - xx1 + 1.f; w = w > 0.f ? w : 0.f;
            float h = yy2 - yy1 + 1.f; h = h > 0.f ? h : 0.f;
            float inter = w * h;
            float mn = area[i] < area[j] ? area[i] : area[j];
            float o = inter / mn;
            if (o <= thr) I[m2++] = I[t];
        }
        m = m2;
    }
    free(I); free(area);
    return nk;
}

static void rerec(float* b, int n) { /* rows of 5 */
    for (int i = 0; i < n; i++) {
        float* r = b + 5 * i;
        float h = r[3] - r[1], w = r[2] - r[0];
        float l = w > h ? w : h;
        r[0] = r[0] + w * 0.5f - l * 0.5f;
        r[1] = r[1] + h * 0.5f - l * 0.5f;
        r[2] = r[0] + l;
        r[3] = r[1] + l;
    }
}
static void bbreg(float* b, const float* reg, int n) {
    for (int i = 0; i < n; i++) {
        float* r = b + 5 * i;
        const float* g = reg + 4 * i;
        float w = r[2] - r[0] + 1.f, h = r[3] - r[1] + 1.f;
        float b1 = r[0] + g[0] * w, b2 = r[1] + g[1] * h, b3 = r[2] + g[2] * w, b4 = r[3] + g[3] * h;
        r[0] = b1; r[1] = b2; r[2] = b3; r[3] = b4;
    }
}
/* pad(): trunc to int, clamp; returns y,ey,x,ex (1-based start, inclusive end) */
static void pad1(const float* r, int W, int H, int* y, int* ey, int* x, int* ex) {
    int bx = (int)truncf(r[0]), by = (int)truncf(r[1]), bex = (int)truncf(r[2]), bey = (int)truncf(r[3]);
    *x = bx < 1 ? 1 : bx;
    *y = by < 1 ? 1 : by;
    *ex = bex > W ? W : bex;
    *ey = bey > H ? H : bey;
}

static void trace_put(float* dst, int max, const float* src, int n, int cols) {
    if (!dst) return;
    int m = n < max ? n : max;
    memcpy(dst, src, (size_t)m * cols * sizeof(float));
}

int orc_detect(const orc_ctx* c, const uint8_t* frame, int H, int W, const orc_params* P,
               float* boxes_out, float* probs_out, int max_out, orc_trace* tr) {
    double scales[32];
    int hs[32], ws[32];
    int ns = orc_scales(H, W, P->min_face_size, P->factor, scales, hs, ws, 32);
    if (tr) { tr->n_scales = ns; tr->n1 = tr->n2 = tr->n3 = 0; }

    /* ---- stage 1: PNet over the pyramid ---------------------------------- */
    int cap = 0, nall = 0;
    float* all = NULL; /* rows of 9 */
    for (int s = 0; s < ns; s++) {
        int h = hs[s], w = ws[s];
        float* lvl = (float*)malloc((size_t)h * w * 3 * sizeof(float));
        orc_area_resample_norm(frame, H, W, 0, H, 0, W, h, w, lvl);
        int oh0 = (h - 2 + 1) / 2 - 4, ow0 = (w - 2 + 1) / 2 - 4; /* ceil((h-2)/2) - 4 */
        if (oh0 < 1 || ow0 < 1) { free(lvl); if (tr) tr->n_cand_scale[s] = tr->n_keep_scale[s] = 0; continue; }
        float* prob = (float*)malloc((size_t)oh0 * ow0 * sizeof(float));
        float* reg = (float*)malloc((size_t)oh0 * ow0 * 4 * sizeof(float));
        int oh, ow;
        orc_pnet_level(c, lvl, h, w, prob, reg, &oh, &ow);
        free(lvl);
        /* generateBoundingBox: mask = probs >= thresh; nonzero() is row-major (y, x) */
        float sc = (float)scales[s];
        int nc = 0;
        for (int i = 0; i < oh * ow; i++) nc += prob[i] >= P->thr0;
        float* bb = (float*)malloc((size_t)(nc ? nc : 1) * 9 * sizeof(float));
        float* b4 = (float*)malloc((size_t)(nc ? nc : 1) * 4 * sizeof(float));
        float* sco = (float*)malloc((size_t)(nc ? nc : 1) * sizeof(float));
        int k = 0;
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++) {
                float p = prob[y * ow + x];
                if (!(p >= P->thr0)) continue;
                float* r = bb + 9 * k;
                r[0] = floorf((2.f * (float)x + 1.f) / sc);
                r[1] = floorf((2.f * (float)y + 1.f) / sc);
                r[2] = floorf((2.f * (float)x + 12.f) / sc);
                r[3] = floorf((2.f * (float)y + 12.f) / sc);
                r[4] = p;
                memcpy(r + 5, reg + 4 * (y * ow + x), 4 * sizeof(float));
                memcpy(b4 + 4 * k, r, 4 * sizeof(float));
                sco[k] = p;
                k++;
            }
        int* keep = (int*)malloc(sizeof(int) * (nc ? nc : 1));
        int nk = orc_nms_iou(b4, sco, nc, 0.5f, keep);
        if (tr) { tr->n_cand_scale[s] = nc; tr->n_keep_scale[s] = nk; }
        if (nall + nk > cap) { cap = (nall + nk) * 2 + 64; all = (float*)realloc(all, (size_t)cap * 9 * sizeof(float)); }
        for (int i = 0; i < nk; i++) memcpy(all + 9 * (nall + i), bb + 9 * keep[i], 9 * sizeof(float));
        nall += nk;
        free(prob); free(reg); free(bb); free(b4); free(sco); free(keep);
    }
    if (nall == 0) { free(all); return 0; }

    /* cross-scale NMS 0.7, regress (w,h WITHOUT +1), rerec */
    int n1;
    float* B1; /* rows of 5 */
    {
        float* b4 = (float*)malloc((size_t)nall * 4 * sizeof(float));
        float* sco = (float*)malloc((size_t)nall * sizeof(float));
        int* keep = (int*)malloc(sizeof(int) * nall);
        for (int i = 0; i < nall; i++) { memcpy(b4 + 4 * i, all + 9 * i, 16); sco[i] = all[9 * i + 4]; }
        n1 = orc_nms_iou(b4, sco, nall, 0.7f, keep);
        B1 = (float*)malloc((size_t)n1 * 5 * sizeof(float));
        for (int i = 0; i < n1; i++) {
            const float* r = all + 9 * keep[i];
            float regw = r[2] - r[0], regh = r[3] - r[1];
            B1[5 * i + 0] = r[0] + r[5] * regw;
            B1[5 * i + 1] = r[1] + r[6] * regh;
            B1[5 * i + 2] = r[2] + r[7] * regw;
            B1[5 * i + 3] = r[3] + r[8] * regh;
            B1[5 * i + 4] = r[4];
        }
        rerec(B1, n1);
        free(b4); free(sco); free(keep); free(all);
    }

    /* ---- stage 2: RNet ---------------------------------------------------- */
    /* The reference skips candidates whose clipped box is empty when building im_data but does
     * not drop them from `boxes` (it would then raise on the mask shape mismatch); here they
     * are dropped from both -- documented deviation, cannot occur for in-frame boxes. */
    {
        int m = 0;
        for (int i = 0; i < n1; i++) {
            int y, ey, x, ex;
            pad1(B1 + 5 * i, W, H, &y, &ey, &x, &ex);
            if (ey > y - 1 && ex > x - 1) { if (m != i) memcpy(B1 + 5 * m, B1 + 5 * i, 20); m++; }
        }
        n1 = m;
    }
    if (tr) { tr->n1 = n1; trace_put(tr->boxes1, tr->max_boxes, B1, n1, 5); }
    int n2 = 0;
    float* B2 = NULL;
    if (n1 > 0) {
        float* crops = (float*)malloc((size_t)n1 * 24 * 24 * 3 * sizeof(float));
        for (int i = 0; i < n1; i++) {
            int y, ey, x, ex;
            pad1(B1 + 5 * i, W, H, &y, &ey, &x, &ex);
            orc_area_resample_norm(frame, H, W, y - 1, ey, x - 1, ex, 24, 24, crops + (size_t)i * 24 * 24 * 3);
        }
        float* prob = (float*)malloc(sizeof(float) * n1);
        float* reg = (float*)malloc(sizeof(float) * n1 * 4);
        orc_rnet(c, crops, n1, prob, reg);
        free(crops);
        float* bb = (float*)malloc(sizeof(float) * n1 * 5);
        float* b4 = (float*)malloc(sizeof(float) * n1 * 4);
        float* sco = (float*)malloc(sizeof(float) * n1);
        float* mv = (float*)malloc(sizeof(float) * n1 * 4);
        int m = 0;
        for (int i = 0; i < n1; i++) {
            if (!(prob[i] > P->thr1)) continue;
            memcpy(bb + 5 * m, B1 + 5 * i, 16); bb[5 * m + 4] = prob[i];
            memcpy(b4 + 4 * m, B1 + 5 * i, 16);
            sco[m] = prob[i];
            memcpy(mv + 4 * m, reg + 4 * i, 16);
            m++;
        }
        int* keep = (int*)malloc(sizeof(int) * (m ? m : 1));
        n2 = orc_nms_iou(b4, sco, m, 0.7f, keep);
        B2 = (float*)malloc(sizeof(float) * (n2 ? n2 : 1) * 5);
        float* mv2 = (float*)malloc(sizeof(float) * (n2 ? n2 : 1) * 4);
        for (int i = 0; i < n2; i++) { memcpy(B2 + 5 * i, bb + 5 * keep[i], 20); memcpy(mv2 + 4 * i, mv + 4 * keep[i], 16); }
        bbreg(B2, mv2, n2);
        rerec(B2, n2);
        free(prob); free(reg); free(bb); free(b4); free(sco); free(mv); free(keep); free(mv2);
    }
    free(B1);
    /* same empty-crop rule before stage 3 */
    {
        int m = 0;
        for (int i = 0; i < n2; i++) {
            int y, ey, x, ex;
            pad1(B2 + 5 * i, W, H, &y, &ey, &x, &ex);
            if (ey > y - 1 && ex > x - 1) { if (m != i) memcpy(B2 + 5 * m, B2 + 5 * i, 20); m++; }
        }
        n2 = m;
    }
    if (tr) { tr->n2 = n2; trace_put(tr->boxes2, tr->max_boxes, B2, n2, 5); }

    /* ---- stage 3: ONet ---------------------------------------------------- */
    int n3 = 0;
    float* B3 = NULL;
    float* P3 = NULL;
    if (n2 > 0) {
        float* crops = (float*)malloc((size_t)n2 * 48 * 48 * 3 * sizeof(float));
        for (int i = 0; i < n2; i++) {
            int y, ey, x, ex;
            pad1(B2 + 5 * i, W, H, &y, &ey, &x, &ex);
            orc_area_resample_norm(frame, H, W, y - 1, ey, x - 1, ex, 48, 48, crops + (size_t)i * 48 * 48 * 3);
        }
        float* prob = (float*)malloc(sizeof(float) * n2);
        float* reg = (float*)malloc(sizeof(float) * n2 * 4);
        float* pts = (float*)malloc(sizeof(float) * n2 * 10);
        orc_onet(c, crops, n2, prob, reg, pts);
        free(crops);
        float* bb = (float*)malloc(sizeof(float) * n2 * 5);
        float* mv = (float*)malloc(sizeof(float) * n2 * 4);
        float* pp = (float*)malloc(sizeof(float) * n2 * 10);
        int m = 0;
        for (int i = 0; i < n2; i++) {
            if (!(prob[i] > P->thr2)) continue;
            memcpy(bb + 5 * m, B2 + 5 * i, 16); bb[5 * m + 4] = prob[i];
            memcpy(mv + 4 * m, reg + 4 * i, 16);
            float w_i = bb[5 * m + 2] - bb[5 * m + 0] + 1.f, h_i = bb[5 * m + 3] - bb[5 * m + 1] + 1.f;
            for (int j = 0; j < 5; j++) {
                pp[10 * m + j] = w_i * pts[10 * i + j] + bb[5 * m + 0] - 1.f;
                pp[10 * m + 5 + j] = h_i * pts[10 * i + 5 + j] + bb[5 * m + 1] - 1.f;
            }
            m++;
        }
        bbreg(bb, mv, m);
        float* b4 = (float*)malloc(sizeof(float) * (m ? m : 1) * 4);
        float* sco = (float*)malloc(sizeof(float) * (m ? m : 1));
        for (int i = 0; i < m; i++) { memcpy(b4 + 4 * i, bb + 5 * i, 16); sco[i] = bb[5 * i + 4]; }
        int* keep = (int*)malloc(sizeof(int) * (m ? m : 1));
        n3 = orc_nms_min(b4, sco, m, 0.7f, keep);
        B3 = (float*)malloc(sizeof(float) * (n3 ? n3 : 1) * 5);
        P3 = (float*)malloc(sizeof(float) * (n3 ? n3 : 1) * 10);
        for (int i = 0; i < n3; i++) { memcpy(B3 + 5 * i, bb + 5 * keep[i], 20); memcpy(P3 + 10 * i, pp + 10 * keep[i], 40); }
        free(prob); free(reg); free(pts); free(bb); free(mv); free(pp); free(b4); free(sco); free(keep);
    }
    free(B2);
    if (tr) { tr->n3 = n3; trace_put(tr->boxes3, tr->max_boxes, B3, n3, 5); trace_put(tr->points3, tr->max_boxes, P3, n3, 10); }

    /* MTCNN.detect(select_largest=True): order = np.argsort(area)[::-1] (ties as stable sort) */
    int nout = 0;
    if (n3 > 0) {
        sidx* o = (sidx*)malloc(sizeof(sidx) * n3);
        for (int i = 0; i < n3; i++) {
            o[i].s = (B3[5 * i + 2] - B3[5 * i]) * (B3[5 * i + 3] - B3[5 * i + 1]);
            o[i].i = i;
        }
        qsort(o, n3, sizeof(sidx), cmp_asc);
        for (int t = n3 - 1; t >= 0 && nout < max_out; t--, nout++) {
            memcpy(boxes_out + 4 * nout, B3 + 5 * o[t].i, 16);
            probs_out[nout] = B3[5 * o[t].i + 4];
        }
        free(o);
    }
    free(B3); free(P3);
    return n3;
}

/* ------------------------------------------------------------------------- */
/* OpenCV INTER_LINEAR u8 (RECALLED: modules/imgproc/src/resize.cpp: 11-bit fixed-point      */
/* coefficients, HResizeLinear into int32, VResizeLinear's (((b*(S>>4))>>16)+..+2)>>2).       */
/* ------------------------------------------------------------------------- */
static inline short sat_short_round(float v) {
    long r = lrintf(v); /* cvRound: nearest-even */
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}
void orc_resize_linear_u8(const uint8_t* img, int H, int W, int y0, int y1, int x0, int x1,
                          int oh, int ow, uint8_t* out) {
    (void)H;
    int sh = y1 - y0, sw = x1 - x0;
    double inv_x = (double)ow / sw, inv_y = (double)oh / sh;
    double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    int* xofs = (int*)malloc(sizeof(int) * ow);
    short* ialpha = (short*)malloc(sizeof(short) * ow * 2);
    for (int dx = 0; dx < ow; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short_round((1.f - fx) * 2048.f);
        ialpha[2 * dx + 1] = sat_short_round(fx * 2048.f);
    }
    int* row0 = (int*)malloc(sizeof(int) * ow * 3);
    int* row1 = (int*)malloc(sizeof(int) * ow * 3);
    for (int dy = 0; dy < oh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = sat_short_round((1.f - fy) * 2048.f), b1 = sat_short_round(fy * 2048.f);
        int sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t* S0 = img + ((size_t)(y0 + sy0) * W + x0) * 3;
        const uint8_t* S1 = img + ((size_t)(y0 + sy1) * W + x0) * 3;
        for (int dx = 0; dx < ow; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 > sw - 1 ? sw - 1 : sx + 1;
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            for (int c = 0; c < 3; c++) {
                row0[dx * 3 + c] = S0[sx * 3 + c] * a0 + S0[sx1 * 3 + c] * a1;
                row1[dx * 3 + c] = S1[sx * 3 + c] * a0 + S1[sx1 * 3 + c] * a1;
            }
        }
        for (int i = 0; i < ow * 3; i++) {
            int v = (((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2;
            out[(size_t)dy * ow * 3 + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

/* ------------------------------------------------------------------------- */
/* InceptionResnetV1 (RECALLED: facenet_pytorch/models/inception_resnet_v1.py) */
/* ------------------------------------------------------------------------- */
/* BasicConv2d = conv(bias=False) -> BatchNorm2d(eps=1e-3, eval) -> ReLU.  The packer folds BN
 * into per-channel (scale, shift) exactly as ATen's CPU batch_norm does (alpha = w*invstd,
 * beta = b - mean*alpha); here y = relu(fmaf(acc, scale, shift)). */
static void bconv(const orc_ctx* c, const char* name, const T* x, int kh, int kw, int sh, int sw,
                  int ph, int pw, T* y, int coff) {
    int K, Cout;
    const float* w = wtf(c, name, ".w", &K, &Cout);
    if (K != kh * kw * x->c) { fprintf(stderr, "oracle: K mismatch %s (%d vs %d)\n", name, K, kh * kw * x->c); abort(); }
    const float* sc = wtf(c, name, ".scale", NULL, NULL);
    const float* sf = wtf(c, name, ".shift", NULL, NULL);
    int OH = (x->h + 2 * ph - kh) / sh + 1, OW = (x->w + 2 * pw - kw) / sw + 1;
    if (!y->d) { *y = talloc(x->n, OH, OW, Cout); coff = 0; }
    conv2d(x, w, NULL, kh, kw, sh, sw, ph, pw, Cout, sc, sf, NULL, 0, 0.f, ACT_RELU, NULL, y->d, y->c, coff, OH, OW);
}
static T bconv_new(const orc_ctx* c, const char* name, const T* x, int kh, int kw, int sh, int sw, int ph, int pw) {
    T y = {0, 0, 0, 0, NULL};
    bconv(c, name, x, kh, kw, sh, sw, ph, pw, &y, 0);
    return y;
}
/* residual tail: out = relu?( conv1x1(cat)+bias ) * scale + x ) */
static T resid(const orc_ctx* c, const char* name, const T* cat, const T* x, float scale, int relu) {
    int K, Cout;
    const float* w = wtf(c, name, ".w", &K, &Cout);
    const float* b = wtf(c, name, ".b", NULL, NULL);
    T y = talloc(x->n, x->h, x->w, Cout);
    conv2d(cat, w, b, 1, 1, 1, 1, 0, 0, Cout, NULL, NULL, x->d, x->c, scale, relu ? ACT_RELU : ACT_NONE, NULL,
           y.d, Cout, 0, x->h, x->w);
    return y;
}
static void maxpool_into(const T* x, int k, int s, T* y, int coff) {
    T p = maxpool(x, k, s, 0);
    for (size_t i = 0; i < (size_t)p.n * p.h * p.w; i++) memcpy(y->d + i * y->c + coff, p.d + i * p.c, sizeof(float) * p.c);
    tfree(&p);
}

static T block35(const orc_ctx* c, const char* pre, const T* x) {
    char nm[96];
    T cat = talloc(x->n, x->h, x->w, 96);
    snprintf(nm, sizeof nm, "%s.branch0", pre); bconv(c, nm, x, 1, 1, 1, 1, 0, 0, &cat, 0);
    snprintf(nm, sizeof nm, "%s.branch1.0", pre); T a = bconv_new(c, nm, x, 1, 1, 1, 1, 0, 0);
    snprintf(nm, sizeof nm, "%s.branch1.1", pre); bconv(c, nm, &a, 3, 3, 1, 1, 1, 1, &cat, 32); tfree(&a);
    snprintf(nm, sizeof nm, "%s.branch2.0", pre); T b = bconv_new(c, nm, x, 1, 1, 1, 1, 0, 0);
    snprintf(nm, sizeof nm, "%s.branch2.1", pre); T b2 = bconv_new(c, nm, &b, 3, 3, 1, 1, 1, 1); tfree(&b);
    snprintf(nm, sizeof nm, "%s.branch2.2", pre); bconv(c, nm, &b2, 3, 3, 1, 1, 1, 1, &cat, 64); tfree(&b2);
    snprintf(nm, sizeof nm, "%s.conv2d", pre);
    T y = resid(c, nm, &cat, x, 0.17f, 1);
    tfree(&cat);
    return y;
}
static T block17(const orc_ctx* c, const char* pre, const T* x) {
    char nm[96];
    T cat = talloc(x->n, x->h, x->w, 256);
    snprintf(nm, sizeof nm, "%s.branch0", pre); bconv(c, nm, x, 1, 1, 1, 1, 0, 0, &cat, 0);
    snprintf(nm, sizeof nm, "%s.branch1.0", pre); T a = bconv_new(c, nm, x, 1, 1, 1, 1, 0, 0);
    snprintf(nm, sizeof nm, "%s.branch1.1", pre); T a2 = bconv_new(c, nm, &a, 1, 7, 1, 1, 0, 3); tfree(&a);
    snprintf(nm, sizeof nm, "%s.branch1.2", pre); bconv(c, nm, &a2, 7, 1, 1, 1, 3, 0, &cat, 128); tfree(&a2);
    snprintf(nm, sizeof nm, "%s.conv2d", pre);
    T y = resid(c, nm, &cat, x, 0.10f, 1);
    tfree(&cat);
    return y;
}
static T block8(const orc_ctx* c, const char* pre, const T* x, float scale, int relu) {
    char nm[96];
    T cat = talloc(x->n, x->h, x->w, 384);
    snprintf(nm, sizeof nm, "%s.branch0", pre); bconv(c, nm, x, 1, 1, 1, 1, 0, 0, &cat, 0);
    snprintf(nm, sizeof nm, "%s.branch1.0", pre); T a = bconv_new(c, nm, x, 1, 1, 1, 1, 0, 0);
    snprintf(nm, sizeof nm, "%s.branch1.1", pre); T a2 = bconv_new(c, nm, &a, 1, 3, 1, 1, 0, 1); tfree(&a);
    snprintf(nm, sizeof nm, "%s.branch1.2", pre); bconv(c, nm, &a2, 3, 1, 1, 1, 1, 0, &cat, 192); tfree(&a2);
    snprintf(nm, sizeof nm, "%s.conv2d", pre);
    T y = resid(c, nm, &cat, x, scale, relu);
    tfree(&cat);
    return y;
}

void orc_facenet(const orc_ctx* c, const float* in, int n, int H, int W, float* emb) {
    if (n <= 0) return;
    char nm[96];
    T x0 = {n, H, W, 3, (float*)in};
    T x = bconv_new(c, "facenet.conv2d_1a", &x0, 3, 3, 2, 2, 0, 0);
    T t = bconv_new(c, "facenet.conv2d_2a", &x, 3, 3, 1, 1, 0, 0); tfree(&x); x = t;
    t = bconv_new(c, "facenet.conv2d_2b", &x, 3, 3, 1, 1, 1, 1); tfree(&x); x = t;
    t = maxpool(&x, 3, 2, 0); tfree(&x); x = t;
    t = bconv_new(c, "facenet.conv2d_3b", &x, 1, 1, 1, 1, 0, 0); tfree(&x); x = t;
    t = bconv_new(c, "facenet.conv2d_4a", &x, 3, 3, 1, 1, 0, 0); tfree(&x); x = t;
    t = bconv_new(c, "facenet.conv2d_4b", &x, 3, 3, 2, 2, 0, 0); tfree(&x); x = t;
    for (int i = 0; i < 5; i++) { snprintf(nm, sizeof nm, "facenet.repeat_1.%d", i); t = block35(c, nm, &x); tfree(&x); x = t; }
    { /* Mixed_6a: cat(b0 384, b1 256, pool 256) */
        int OH = (x.h - 3) / 2 + 1, OW = (x.w - 3) / 2 + 1;
        T cat = talloc(n, OH, OW, 896);
        bconv(c, "facenet.mixed_6a.branch0", &x, 3, 3, 2, 2, 0, 0, &cat, 0);
        T a = bconv_new(c, "facenet.mixed_6a.branch1.0", &x, 1, 1, 1, 1, 0, 0);
        T a2 = bconv_new(c, "facenet.mixed_6a.branch1.1", &a, 3, 3, 1, 1, 1, 1); tfree(&a);
        bconv(c, "facenet.mixed_6a.branch1.2", &a2, 3, 3, 2, 2, 0, 0, &cat, 384); tfree(&a2);
        maxpool_into(&x, 3, 2, &cat, 640);
        tfree(&x); x = cat;
    }
    for (int i = 0; i < 10; i++) { snprintf(nm, sizeof nm, "facenet.repeat_2.%d", i); t = block17(c, nm, &x); tfree(&x); x = t; }
    { /* Mixed_7a: cat(b0 384, b1 256, b2 256, pool 896) */
        int OH = (x.h - 3) / 2 + 1, OW = (x.w - 3) / 2 + 1;
        T cat = talloc(n, OH, OW, 1792);
        T a = bconv_new(c, "facenet.mixed_7a.branch0.0", &x, 1, 1, 1, 1, 0, 0);
        bconv(c, "facenet.mixed_7a.branch0.1", &a, 3, 3, 2, 2, 0, 0, &cat, 0); tfree(&a);
        a = bconv_new(c, "facenet.mixed_7a.branch1.0", &x, 1, 1, 1, 1, 0, 0);
        bconv(c, "facenet.mixed_7a.branch1.1", &a, 3, 3, 2, 2, 0, 0, &cat, 384); tfree(&a);
        a = bconv_new(c, "facenet.mixed_7a.branch2.0", &x, 1, 1, 1, 1, 0, 0);
        T a2 = bconv_new(c, "facenet.mixed_7a.branch2.1", &a, 3, 3, 1, 1, 1, 1); tfree(&a);
        bconv(c, "facenet.mixed_7a.branch2.2", &a2, 3, 3, 2, 2, 0, 0, &cat, 640); tfree(&a2);
        maxpool_into(&x, 3, 2, &cat, 896);
        tfree(&x); x = cat;
    }
    for (int i = 0; i < 5; i++) { snprintf(nm, sizeof nm, "facenet.repeat_3.%d", i); t = block8(c, nm, &x, 0.20f, 1); tfree(&x); x = t; }
    t = block8(c, "facenet.block8", &x, 1.0f, 0); tfree(&x); x = t;
    /* avgpool_1a = AdaptiveAvgPool2d(1): row-major sequential sum / count */
    T g = talloc(n, 1, 1, x.c);
    for (int i = 0; i < n; i++)
        for (int ch = 0; ch < x.c; ch++) {
            float s = 0.f;
            for (int p = 0; p < x.h * x.w; p++) s = s + x.d[((size_t)i * x.h * x.w + p) * x.c + ch];
            g.d[(size_t)i * x.c + ch] = s / (float)(x.h * x.w);
        }
    tfree(&x);
    /* last_linear (no bias) -> last_bn (BatchNorm1d eval, folded) -> F.normalize(p=2, dim=1) */
    int K, Cout;
    const float* w = wt(c, "facenet.last_linear.w", &K, &Cout);
    const float* sc = wt(c, "facenet.last_bn.scale", NULL, NULL);
    const float* sf = wt(c, "facenet.last_bn.shift", NULL, NULL);
    T e = talloc(n, 1, 1, Cout);
    conv2d(&g, w, NULL, 1, 1, 1, 1, 0, 0, Cout, sc, sf, NULL, 0, 0.f, ACT_NONE, NULL, e.d, Cout, 0, 1, 1);
    tfree(&g);
    for (int i = 0; i < n; i++) {
        const float* v = e.d + (size_t)i * 512;
        float nrm = sqrtf(orc_dot512(v, v));
        if (nrm < 1e-12f) nrm = 1e-12f;
        for (int j = 0; j < 512; j++) emb[(size_t)i * 512 + j] = v[j] / nrm;
    }
    tfree(&e);
}

/* ------------------------------------------------------------------------- */
/* model.py:47-59 on a batch of sampled frames                                */
/* ------------------------------------------------------------------------- */
/* SURVEY 8(f)-4 native mode: facenet_pytorch extract_face() for tensor input (RECALLED: crop_resize ->
 * imresample(area) -> .byte() truncation), then fixed_image_standardization (x-127.5)/128, optional BGR->RGB. */
void orc_crop_area_std(const uint8_t* img, int H, int W, int x0, int y0, int x1, int y1, int S, int rgb, float* out) {
    (void)H;
    const int ih = y1 - y0, iw = x1 - x0;
    for (int oy = 0; oy < S; oy++) {
        const int ys = (int)(((long)oy * ih) / S), ye = (int)((((long)oy + 1) * ih + S - 1) / S);
        for (int ox = 0; ox < S; ox++) {
            const int xs = (int)(((long)ox * iw) / S), xe = (int)((((long)ox + 1) * iw + S - 1) / S);
            uint32_t s[3] = {0, 0, 0};
            for (int y = ys; y < ye; y++) {
                const uint8_t* p = img + ((size_t)(y0 + y) * W + x0 + xs) * 3;
                for (int x = xs; x < xe; x++, p += 3) { s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; }
            }
            const float kh = (float)(ye - ys), kw = (float)(xe - xs);
            for (int c = 0; c < 3; c++) {
                const float mean = (float)s[c] / kh / kw;
                const float byte = (float)(uint8_t)mean;              /* .byte(): truncation toward zero */
                out[((size_t)oy * S + ox) * 3 + (rgb ? 2 - c : c)] = (byte - 127.5f) / 128.0f;
            }
        }
    }
}

/* SURVEY 8(f)-4, "landmark-aligned": this project's own definition (facenet-pytorch does not align; the reference discards the
 * landmarks, server/model.py:47).  The five O-Net points (left eye, right eye, nose, mouth corners) are matched to the 112x112
 * five-point template everyone uses for face embedders, scaled to SxS = 160: least-squares SIMILARITY (scale, rotation,
 * translation) from template coordinates to frame coordinates, estimated directly as the inverse map the warp needs:
 *     d_j = T_j - mean(T),  e_j = P_j - mean(P),  a = sum(d.e)/sum|d|^2,  b = sum(dx*ey - dy*ex)/sum|d|^2,
 *     x(u,v) = a*(u-Tx) - b*(v-Ty) + Px,   y(u,v) = b*(u-Tx) + a*(v-Ty) + Py          (all in double, this order)
 * then a bilinear sample of the u8 frame with replicated borders (float: p00 + fx*(p01-p00), rows likewise), no byte
 * quantisation, fixed_image_standardization (v-127.5)/128, optional BGR -> RGB. */
static const double ORC_TPL_X[5] = {54.706571428571436, 105.04542857142857, 80.036, 59.35614285714286, 101.04271428571428};
static const double ORC_TPL_Y[5] = {73.85185714285714, 73.57342857142856, 102.48085714285713, 131.9507142857143, 131.72014285714286};
void orc_align_params(const float* pts, double* prm) {
    double tx = 0., ty = 0., px = 0., py = 0.;
    for (int j = 0; j < 5; j++) { tx += ORC_TPL_X[j]; ty += ORC_TPL_Y[j]; px += (double)pts[j]; py += (double)pts[5 + j]; }
    tx = tx / 5.; ty = ty / 5.; px = px / 5.; py = py / 5.;
    double sdd = 0., sde = 0., scr = 0.;
    for (int j = 0; j < 5; j++) {
        const double dx = ORC_TPL_X[j] - tx, dy = ORC_TPL_Y[j] - ty, ex = (double)pts[j] - px, ey = (double)pts[5 + j] - py;
        sdd = sdd + (dx * dx + dy * dy);
        sde = sde + (dx * ex + dy * ey);
        scr = scr + (dx * ey - dy * ex);
    }
    prm[0] = sde / sdd; prm[1] = scr / sdd; prm[2] = tx; prm[3] = ty; prm[4] = px; prm[5] = py;
}
void orc_crop_aligned(const uint8_t* img, int H, int W, const float* pts, int S, int rgb, float* out) {
    double prm[6];
    orc_align_params(pts, prm);
    const double a = prm[0], b = prm[1];
    for (int v = 0; v < S; v++) {
        for (int u = 0; u < S; u++) {
            const double du = (double)u - prm[2], dv = (double)v - prm[3];
            const double x = (a * du - b * dv) + prm[4], y = (b * du + a * dv) + prm[5];
            const double xf = floor(x), yf = floor(y);
            const float fx = (float)(x - xf), fy = (float)(y - yf);
            /* clamp in double first: a degenerate point set can send the coordinates anywhere */
            const double xc = (xf >= -1.) ? (xf > (double)W ? (double)W : xf) : -1., yc = (yf >= -1.) ? (yf > (double)H ? (double)H : yf) : -1.;   /* NaN -> -1 */
            int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
            x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
            y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
            const uint8_t *r0 = img + (size_t)y0 * W * 3, *r1 = img + (size_t)y1 * W * 3;
            for (int c = 0; c < 3; c++) {
                const float p00 = (float)r0[x0 * 3 + c], p01 = (float)r0[x1 * 3 + c], p10 = (float)r1[x0 * 3 + c], p11 = (float)r1[x1 * 3 + c];
                const float top = p00 + fx * (p01 - p00), bot = p10 + fx * (p11 - p10);
                const float val = top + fy * (bot - top);
                out[((size_t)v * S + u) * 3 + (rgb ? 2 - c : c)] = (val - 127.5f) / 128.0f;
            }
        }
    }
}

int orc_detect_embed_mode(const orc_ctx* c, const uint8_t* frames, int n, int H, int W, const orc_params* P, int mode,
                          float* box_out, float* prob_out, int32_t* rect_out, uint8_t* valid_out, float* emb_out) {
    if (mode == 0) return orc_detect_embed(c, frames, n, H, W, P, box_out, prob_out, rect_out, valid_out, emb_out, NULL);
    const int S = 160;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        const uint8_t* fr = frames + (size_t)i * H * W * 3;
        float bx[4 * 64], pr[64];
        orc_trace tr;
        memset(&tr, 0, sizeof tr);
        float *tb3 = NULL, *tp3 = NULL;
        if (mode == 3) {          /* the landmarks of the largest face come out of the stage-3 trace */
            tr.max_boxes = 4096;
            tb3 = (float*)malloc(sizeof(float) * 5 * tr.max_boxes); tp3 = (float*)malloc(sizeof(float) * 10 * tr.max_boxes);
            tr.boxes3 = tb3; tr.points3 = tp3;
        }
        int k = orc_detect(c, fr, H, W, P, bx, pr, 64, mode == 3 ? &tr : NULL);
        memset(box_out + 4 * i, 0, 16); memset(rect_out + 4 * i, 0, 16);
        prob_out[i] = 0.f; valid_out[i] = 0;
        memset(emb_out + (size_t)i * 512, 0, 2048);
        float pts0[10] = {0};
        if (mode == 3 && k > 0) {  /* rank 0 of argsort(area)[::-1] with stable ties: largest area, ties -> higher index */
            int best = 0; float ba = -1.f;
            const int m = tr.n3 < tr.max_boxes ? tr.n3 : tr.max_boxes;
            for (int q = 0; q < m; q++) {
                const float ar = (tb3[5 * q + 2] - tb3[5 * q]) * (tb3[5 * q + 3] - tb3[5 * q + 1]);
                if (ar >= ba) { ba = ar; best = q; }
            }
            memcpy(pts0, tp3 + 10 * best, sizeof pts0);
        }
        free(tb3); free(tp3);
        if (k <= 0) continue;
        memcpy(box_out + 4 * i, bx, 16);
        prob_out[i] = pr[0];
        long b0 = (long)bx[0], b1 = (long)bx[1], b2 = (long)bx[2], b3 = (long)bx[3];   /* extract_face, margin 0 */
        if (b0 < 0) b0 = 0;
        if (b1 < 0) b1 = 0;
        if (b2 > W) b2 = W;
        if (b3 > H) b3 = H;
        rect_out[4 * i] = (int32_t)b0; rect_out[4 * i + 1] = (int32_t)b1; rect_out[4 * i + 2] = (int32_t)b2; rect_out[4 * i + 3] = (int32_t)b3;
        if (!(b2 > b0 && b3 > b1)) continue;
        float* face = (float*)malloc((size_t)S * S * 3 * sizeof(float));
        if (mode == 3) orc_crop_aligned(fr, H, W, pts0, S, 1, face);
        else orc_crop_area_std(fr, H, W, (int)b0, (int)b1, (int)b2, (int)b3, S, mode == 2, face);
        orc_facenet(c, face, 1, S, S, emb_out + (size_t)i * 512);
        free(face);
        valid_out[i] = 1;
    }
    return 0;
}

int orc_detect_embed(const orc_ctx* c, const uint8_t* frames, int n, int H, int W, const orc_params* P,
                     float* box_out, float* prob_out, int32_t* rect_out, uint8_t* valid_out,
                     float* emb_out, uint8_t* face_out) {
    float* faces = (float*)calloc((size_t)n * 80 * 80 * 3, sizeof(float));
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        const uint8_t* fr = frames + (size_t)i * H * W * 3;
        float bx[4 * 64], pr[64];
        int k = orc_detect(c, fr, H, W, P, bx, pr, 64, NULL);
        memset(box_out + 4 * i, 0, 16);
        memset(rect_out + 4 * i, 0, 16);
        prob_out[i] = 0.f;
        valid_out[i] = 0;
        if (face_out) memset(face_out + (size_t)i * 19200, 0, 19200);
        if (k <= 0) continue;
        memcpy(box_out + 4 * i, bx, 16);
        prob_out[i] = pr[0];
        /* model.py:49-53: boxes[0].astype(int) truncates toward zero, then clamp */
        long b0 = (long)bx[0], b1 = (long)bx[1], b2 = (long)bx[2], b3 = (long)bx[3];
        if (b0 < 0) b0 = 0;
        if (b1 < 0) b1 = 0;
        if (b2 > W) b2 = W;
        if (b3 > H) b3 = H;
        rect_out[4 * i] = (int32_t)b0; rect_out[4 * i + 1] = (int32_t)b1;
        rect_out[4 * i + 2] = (int32_t)b2; rect_out[4 * i + 3] = (int32_t)b3;
        if (!(b2 > b0 && b3 > b1)) continue; /* model.py:54 */
        uint8_t f80[19200];
        orc_resize_linear_u8(fr, H, W, (int)b1, (int)b3, (int)b0, (int)b2, 80, 80, f80); /* model.py:55-57 */
        if (face_out) memcpy(face_out + (size_t)i * 19200, f80, 19200);
        /* model.py:58 to_tensor: u8 -> f32 / 255 (channel order untouched: BGR, no standardisation) */
        for (int j = 0; j < 19200; j++) faces[(size_t)i * 19200 + j] = (float)f80[j] / 255.0f;
        valid_out[i] = 1;
    }
    /* embed all rows (invalid rows are zero images whose outputs are discarded) */
    float* emb = (float*)malloc((size_t)n * 512 * sizeof(float));
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        if (valid_out[i]) orc_facenet(c, faces + (size_t)i * 19200, 1, 80, 80, emb + (size_t)i * 512);
    }
    for (int i = 0; i < n; i++) {
        if (valid_out[i]) memcpy(emb_out + (size_t)i * 512, emb + (size_t)i * 512, 2048);
        else memset(emb_out + (size_t)i * 512, 0, 2048);
    }
    free(emb); free(faces);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* model.py:60-66,70,75,86-95                                                  */
/* ------------------------------------------------------------------------- */
int orc_drift_score(const float* emb, const uint8_t* valid, int n, long frame_count, int fps,
                    float* sims_out, uint8_t* flag_out, int* final_run, int* hits_out) {
    const float thr_sim = 0.99f; /* model.py:16 */
    const int thr_frames = 15;   /* model.py:17 */
    int run = 0, hits = 0;
    const float* prev = NULL;
    for (int i = 0; i < n; i++) {
        if (sims_out) sims_out[i] = 2.0f;
        if (flag_out) flag_out[i] = 0;
        if (!valid[i]) continue;
        const float* cur = emb + (size_t)i * 512;
        if (prev) {
            /* model.py:61: np.dot(a,b) / (np.linalg.norm(a) * np.linalg.norm(b)), all float32 */
            float d = orc_dot512(cur, prev);
            float na = sqrtf(orc_dot512(cur, cur)), nb = sqrtf(orc_dot512(prev, prev));
            float sim = d / (na * nb);
            if (sims_out) sims_out[i] = sim;
            if (sim < thr_sim) run += 1; else run = 0; /* model.py:62-65 */
            if (run > thr_frames) { hits += 1; if (flag_out) flag_out[i] = 1; } /* model.py:66,70 */
        }
        prev = cur; /* model.py:75 */
    }
    if (final_run) *final_run = run;
    if (hits_out) *hits_out = hits;
    if (frame_count == 0 || fps <= 0) return 0; /* model.py:30-34,83-85 */
    int step = (int)(fps / 7.0); /* model.py:40 */
    if (step < 1) step = 1;
    long total = (frame_count + step - 1) / step; /* model.py:86 */
    if (total == 0) return 0;
    double pct = ((double)hits / (double)total) * 100.0; /* model.py:89 */
    double conf = pct * ((double)run / (double)thr_frames); /* model.py:90 */
    if (conf > 100.0) conf = 100.0;
    double w = (frame_count > (long)fps * 30) ? 0.5 : 0.3; /* model.py:91-94 */
    double ws = pct + conf * w;
    if (ws > 100.0) ws = 100.0;
    int r = (int)ws;
    if (r < 0) r = 0;
    if (r > 100) r = 100;
    return r; /* model.py:95 */
}

/* ------------------------------------------------------------------------- */
/* SURVEY 8(f)-1 ingest: NV12 -> BGR, OpenCV cvtColor(COLOR_YUV2BGR_NV12) integer BT.601            */
/* (RECALLED: modules/imgproc/src/color_yuv.simd.hpp uvToRGBuv / yRGBuvToRGBA, 20-bit fixed point). */
/* ------------------------------------------------------------------------- */
void orc_nv12_to_bgr(const uint8_t* nv12, int H, int W, uint8_t* bgr) {
    const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SH = 20;
    const uint8_t* yp = nv12;
    const uint8_t* uvp = nv12 + (size_t)H * W;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int u = uvp[(size_t)(y / 2) * W + (x / 2) * 2] - 128, v = uvp[(size_t)(y / 2) * W + (x / 2) * 2 + 1] - 128;
            const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
            int yy = yp[(size_t)y * W + x] - 16;
            yy = (yy > 0 ? yy : 0) * CY;
            int b = (yy + buv) >> SH, g = (yy + guv) >> SH, r = (yy + ruv) >> SH;
            uint8_t* o = bgr + ((size_t)y * W + x) * 3;
            o[0] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
            o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
            o[2] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}

/* ---- self test backing a DEVICE-side shortcut (csrc/trl_pnet.hip:pyr_div) ---------------------------------
 * The pyramid kernel divides by the bin height and width with q0 = RN(a r), e = fma(-b, q0, a), q = fma(e, r, q0),
 * r = RN(1/b), instead of two IEEE divisions.  The oracle itself keeps the true divisions (orc_area_resample_norm);
 * this routine checks, for every bin size kh, kw <= kmax and every possible byte sum s <= 255 kh kw, that both
 * routes give bit-identical results.  Returns the number of mismatches (0 expected). */
static inline float selftest_rdiv(float a, float b, float r) {
    const float q0 = a * r;
    const float e = fmaf(-b, q0, a);
    return fmaf(e, r, q0);
}
long orc_selftest_recip_div(int kmax) {
    long bad = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : bad)
    for (int kh = 1; kh <= kmax; kh++) {
        const float fkh = (float)kh, rkh = 1.0f / fkh;
        for (int kw = 1; kw <= kmax; kw++) {
            const float fkw = (float)kw, rkw = 1.0f / fkw;
            const long smax = 255L * kh * kw;
            for (long s = 0; s <= smax; s++) {
                const float a = (float)s;
                const float t = a / fkh / fkw;
                const float u = selftest_rdiv(selftest_rdiv(a, fkh, rkh), fkw, rkw);
                if (t != u) bad++;   /* finite, non-negative: value equality is bit equality */
            }
        }
    }
    return bad;
}
