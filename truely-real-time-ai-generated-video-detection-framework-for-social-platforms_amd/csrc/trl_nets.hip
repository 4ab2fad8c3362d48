// trl_nets.hip -- host-side walkers that issue the layer kernels of the four networks.
//
// Architecture restated from the published facenet_pytorch 2.6.0 modules (absent from
// /root/reference), call sites server/model.py:18-19,47,59.
// Activations are NHWC f32 in the context arena; a concat is a channel slice of a wider
// buffer (Act.coff / Act.ld), so torch.cat costs nothing.
#include "trl_ctx.h"
#include <stdlib.h>

namespace {

struct Runner {
    trl_ctx* c;
    hipStream_t s;
    int err = TRL_OK;
    const int32_t* m_dev = nullptr;   // device-sized batch (candidate lists): item count lives on the device, see ConvArgs
    int m_base = 0;

    Act alloc(int n, int h, int w, int ch, bool bf = false) {
        Act a;
        a.n = n; a.h = h; a.w = w; a.c = ch; a.ld = ch; a.coff = 0; a.bf = bf;
        a.p = (float*)c->scratch.alloc((size_t)n * h * w * ch * (bf ? sizeof(uint16_t) : sizeof(float)) + 64);
        if (!a.p && err == TRL_OK) {
            trl_set_error("activation scratch exhausted (%zu of %zu bytes used)", c->scratch.off, c->scratch.cap);
            err = TRL_ERR_STATE;
        }
        return a;
    }
    static Act slice(const Act& buf, int coff, int ch) {
        Act a = buf;
        a.coff = buf.coff + coff;
        a.c = ch;
        return a;
    }

    // Deferred launches: convs queued between begin_group() and end_group() are independent of each other and go out as ONE
    // launch when the small-map family takes them all (trl_launch_fn_group), else one by one in order.
    std::vector<ConvArgs> pending;
    bool grouping = false;
    int ysplit = 1 << 30, yskip = 0;   // destination scatter of the next conv: output column n >= ysplit lands yskip channels further
    void begin_group() { grouping = true; pending.clear(); }
    void end_group() {
        grouping = false;
        if (pending.empty() || err != TRL_OK) { pending.clear(); return; }
        bool all = pending.size() <= 3;
        for (auto& a : pending) all = all && trl_fn_eligible(a);
        if (all) for (auto& a : pending) all = all && (trl_fn_split4_rule(a) == trl_fn_split4_rule(pending[0]));
        int st = TRL_OK;
        if (all) st = trl_launch_fn_group(pending.data(), (int)pending.size(), s);
        else for (auto& a : pending) { st = trl_launch_conv(a, s); if (st != TRL_OK) break; }
        if (st != TRL_OK) err = st;
        pending.clear();
    }

    // generic conv launcher; `into` selects a pre-allocated (concat) destination view
    Act conv(const Act& x, const DevW* w, const DevV* bias, const DevV* scale, const DevV* shift, const DevV* slope,
             int kh, int kw, int sh, int sw, int ph, int pw, int act, const Act* into, const Act* res, float res_scale) {
        const int OH = (x.h + 2 * ph - kh) / sh + 1, OW = (x.w + 2 * pw - kw) / sw + 1;
        Act y = into ? *into : alloc(x.n, OH, OW, w ? w->Cout : 0, x.bf);
        if (err != TRL_OK) return y;
        if (!w || w->K != kh * kw * x.c || (into && (into->h != OH || into->w != OW || into->c != w->Cout))) {
            trl_set_error("conv shape mismatch (K=%d expected %d)", w ? w->K : -1, kh * kw * x.c);
            err = TRL_ERR_WEIGHTS;
            return y;
        }
        ConvArgs a;
        a.x = x.p; a.N = x.n; a.H = x.h; a.W = x.w; a.Cin = x.c; a.ldx = x.ld; a.xoff = x.coff;
        a.w = w->p; a.ldw = w->ld; a.K = w->K;
        a.bias = bias ? bias->p : nullptr;
        a.scale = scale ? scale->p : nullptr; a.shift = shift ? shift->p : nullptr;
        a.slope = slope ? slope->p : nullptr;
        a.res = res ? res->p + res->coff : nullptr; a.ldres = res ? res->ld : 0; a.res_scale = res_scale;
        a.y = y.p; a.ldy = y.ld; a.yoff = y.coff;
        a.KH = kh; a.KW = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
        a.Cout = w->Cout; a.OH = OH; a.OW = OW; a.act = act;
        a.M = x.n * OH * OW;
        a.m_dev = m_dev; a.m_base = m_base; a.m_per = OH * OW;
        a.ysplit = ysplit; a.yskip = yskip;
        if (yskip && !trl_fn_eligible(a)) { trl_set_error("scattered destination needs the small-map conv family"); err = TRL_ERR_STATE; return y; }
        if (x.bf) {   // reduced-precision embedder: bf16 in / out / residual, transposed bf16 weights
            if (!w->pt || y.bf != true || (res && !res->bf) || act == TRL_ACT_PRELU) {
                trl_set_error("bf16 conv without bf16 weights / destination");
                err = TRL_ERR_STATE;
                return y;
            }
            a.lowp = c->cfg.embed_precision; a.wt = w->pt; a.ldwt = w->ldt;
            if (res) a.res = reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(res->p) + res->coff);
            int st = trl_launch_conv_bf16(a, s);
            if (st != TRL_OK) err = st;
            return y;
        }
        if (grouping) { pending.push_back(a); return y; }
        int st = trl_launch_conv(a, s);
        if (st != TRL_OK) err = st;
        return y;
    }
    // BasicConv2d: conv(no bias) + folded BN + ReLU
    Act bconv(const Act& x, const std::string& name, int kh, int kw, int sh, int sw, int ph, int pw, const Act* into = nullptr) {
        return conv(x, trl_w(c, name + ".w"), nullptr, trl_v(c, name + ".scale"), trl_v(c, name + ".shift"), nullptr,
                    kh, kw, sh, sw, ph, pw, TRL_ACT_RELU, into, nullptr, 0.f);
    }
    // MTCNN conv (bias) + optional PReLU, valid padding
    Act mconv(const Act& x, const std::string& net, const std::string& name, const char* prelu, int k) {
        return conv(x, trl_w(c, net + "." + name + ".w"), trl_v(c, net + "." + name + ".b"), nullptr, nullptr,
                    prelu ? trl_v(c, net + "." + prelu) : nullptr, k, k, 1, 1, 0, 0, prelu ? TRL_ACT_PRELU : TRL_ACT_NONE,
                    nullptr, nullptr, 0.f);
    }
    Act resid(const Act& cat, const Act& x, const std::string& name, float scale, bool relu) {
        return conv(cat, trl_w(c, name + ".w"), trl_v(c, name + ".b"), nullptr, nullptr, nullptr, 1, 1, 1, 1, 0, 0,
                    relu ? TRL_ACT_RELU : TRL_ACT_NONE, nullptr, &x, scale);
    }
    Act pool(const Act& x, int k, int st, int ceil_mode, const Act* into = nullptr) {
        const int OH = trl_pool_out(x.h, k, st, ceil_mode), OW = trl_pool_out(x.w, k, st, ceil_mode);
        Act y = into ? *into : alloc(x.n, OH, OW, x.c, x.bf);
        if (err != TRL_OK) return y;
        if (x.bf) {
            int e = trl_launch_maxpool_bf16(reinterpret_cast<const uint16_t*>(x.p), x.n, x.h, x.w, x.c, x.ld, x.coff, k, st,
                                            reinterpret_cast<uint16_t*>(y.p), y.ld, y.coff, OH, OW, s, c->cfg.embed_precision);
            if (e != TRL_OK) err = e;
            return y;
        }
        int e = trl_launch_maxpool(x.p, x.n, x.h, x.w, x.c, x.ld, x.coff, k, st, ceil_mode, y.p, y.ld, y.coff, OH, OW, s, m_dev, m_base);
        if (e != TRL_OK) err = e;
        return y;
    }
};

// The 1x1 branches that read the block input run as one fused conv (weights concatenated at load time);
// the concat buffer doubles as their scratch: a slice is only overwritten after its last reader has run.
Act block35(Runner& R, const Act& x, const std::string& p) {
    Act cat = R.alloc(x.n, x.h, x.w, 96, x.bf);
    Act s1 = Runner::slice(cat, 32, 32), s2 = Runner::slice(cat, 64, 32);
    R.bconv(x, p + ".fused", 1, 1, 1, 1, 0, 0, &cat);                 // [branch0 | branch2.0 | branch1.0]
    Act b2 = R.bconv(s1, p + ".branch2.1", 3, 3, 1, 1, 1, 1);         // reads branch2.0 (cols 32:64)
    R.bconv(s2, p + ".branch1.1", 3, 3, 1, 1, 1, 1, &s1);             // reads branch1.0 (64:96) -> final branch1 at 32:64
    R.bconv(b2, p + ".branch2.2", 3, 3, 1, 1, 1, 1, &s2);             // final branch2 at 64:96
    return R.resid(cat, x, p + ".conv2d", 0.17f, true);
}
// The same block in 4 launches for the small-map conv family (f32, M <= 16384): the fused 1x1 scatters its columns into a
// 128-wide buffer W = [branch0 | (free) | branch2.0 | branch1.0], so the two independent 3x3s can share ONE launch without a
// hazard -- branch1.1 reads W[96:128] and writes the free slot W[32:64] (nobody reads it), branch2.1 reads W[64:96] into a
// temporary -- then branch2.2 writes W[64:96] and the up-projection reads W[0:96] = [branch0 | branch1 | branch2].
Act block35_grouped(Runner& R, const Act& x, const std::string& p) {
    Act W = R.alloc(x.n, x.h, x.w, 128, false);
    Act w96 = Runner::slice(W, 0, 96);
    Act f1 = Runner::slice(W, 32, 32), f2 = Runner::slice(W, 64, 32), f3 = Runner::slice(W, 96, 32);
    R.ysplit = 32; R.yskip = 32;                                      // columns >= 32 land 32 channels further right
    R.bconv(x, p + ".fused", 1, 1, 1, 1, 0, 0, &w96);                 // [branch0 | - | branch2.0 | branch1.0]
    R.ysplit = 1 << 30; R.yskip = 0;
    R.begin_group();
    Act b2 = R.bconv(f2, p + ".branch2.1", 3, 3, 1, 1, 1, 1);         // W[64:96] -> temporary
    R.bconv(f3, p + ".branch1.1", 3, 3, 1, 1, 1, 1, &f1);             // W[96:128] -> W[32:64]
    R.end_group();
    R.bconv(b2, p + ".branch2.2", 3, 3, 1, 1, 1, 1, &f2);             // -> W[64:96] (branch2.0 is dead)
    return R.resid(w96, x, p + ".conv2d", 0.17f, true);
}
Act block17(Runner& R, const Act& x, const std::string& p) {
    Act cat = R.alloc(x.n, x.h, x.w, 256, x.bf);
    Act s1 = Runner::slice(cat, 128, 128);
    R.bconv(x, p + ".fused", 1, 1, 1, 1, 0, 0, &cat);                 // [branch0 | branch1.0]
    Act a2 = R.bconv(s1, p + ".branch1.1", 1, 7, 1, 1, 0, 3);
    R.bconv(a2, p + ".branch1.2", 7, 1, 1, 1, 3, 0, &s1);
    return R.resid(cat, x, p + ".conv2d", 0.10f, true);
}
Act block8(Runner& R, const Act& x, const std::string& p, float scale, bool relu) {
    Act cat = R.alloc(x.n, x.h, x.w, 384, x.bf);
    Act s1 = Runner::slice(cat, 192, 192);
    R.bconv(x, p + ".fused", 1, 1, 1, 1, 0, 0, &cat);                 // [branch0 | branch1.0]
    Act a2 = R.bconv(s1, p + ".branch1.1", 1, 3, 1, 1, 0, 1);
    R.bconv(a2, p + ".branch1.2", 3, 1, 1, 1, 1, 0, &s1);
    return R.resid(cat, x, p + ".conv2d", scale, relu);
}

}  // namespace

// InceptionResnetV1.eval().forward (server/model.py:59)
int trl_run_facenet(trl_ctx* c, const float* d_faces, int n, int h, int w, const uint8_t* d_valid, float* d_emb, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    Runner R{c, s};
    Act x0; x0.p = const_cast<float*>(d_faces); x0.n = n; x0.h = h; x0.w = w; x0.c = 3; x0.ld = 3; x0.coff = 0;
    const std::string f = "facenet.";
    Act x = R.bconv(x0, f + "conv2d_1a", 3, 3, 2, 2, 0, 0);
    if (c->cfg.embed_precision >= 1) {   // everything after the 3-channel stem conv runs on 16-bit activations (trl_bf16.hip)
        Act xb = R.alloc(x.n, x.h, x.w, x.c, true);
        if (R.err != TRL_OK) return R.err;
        TRL_CHECK(trl_launch_to_bf16(x.p, x.pixels() * x.c, reinterpret_cast<uint16_t*>(xb.p), s, c->cfg.embed_precision));
        x = xb;
    }
    x = R.bconv(x, f + "conv2d_2a", 3, 3, 1, 1, 0, 0);
    x = R.bconv(x, f + "conv2d_2b", 3, 3, 1, 1, 1, 1);
    x = R.pool(x, 3, 2, 0);
    x = R.bconv(x, f + "conv2d_3b", 1, 1, 1, 1, 0, 0);
    x = R.bconv(x, f + "conv2d_4a", 3, 3, 1, 1, 0, 0);
    x = R.bconv(x, f + "conv2d_4b", 3, 3, 2, 2, 0, 0);
    if (R.err != TRL_OK) return R.err;
    if (x.h < 3 || x.w < 3) { trl_set_error("face crop %dx%d too small for InceptionResnetV1", h, w); return TRL_ERR_INVALID; }
    const bool small_f32 = !x.bf && (long long)x.n * x.h * x.w <= 16384 && !(g_trl_no_fnconv != 0 || trl_tune_set("TRL_NO_FNCONV"));
    for (int i = 0; i < 5; i++) x = small_f32 ? block35_grouped(R, x, f + "repeat_1." + std::to_string(i)) : block35(R, x, f + "repeat_1." + std::to_string(i));
    {   // Mixed_6a
        const int OH = (x.h - 3) / 2 + 1, OW = (x.w - 3) / 2 + 1;
        Act cat = R.alloc(n, OH, OW, 896, x.bf);
        Act s0 = Runner::slice(cat, 0, 384), s1 = Runner::slice(cat, 384, 256), s2 = Runner::slice(cat, 640, 256);
        R.bconv(x, f + "mixed_6a.branch0", 3, 3, 2, 2, 0, 0, &s0);
        Act a = R.bconv(x, f + "mixed_6a.branch1.0", 1, 1, 1, 1, 0, 0);
        Act a2 = R.bconv(a, f + "mixed_6a.branch1.1", 3, 3, 1, 1, 1, 1);
        R.bconv(a2, f + "mixed_6a.branch1.2", 3, 3, 2, 2, 0, 0, &s1);
        R.pool(x, 3, 2, 0, &s2);
        x = cat;
    }
    if (R.err != TRL_OK) return R.err;
    if (x.h < 3 || x.w < 3) { trl_set_error("face crop %dx%d too small for InceptionResnetV1", h, w); return TRL_ERR_INVALID; }
    for (int i = 0; i < 10; i++) x = block17(R, x, f + "repeat_2." + std::to_string(i));
    {   // Mixed_7a
        const int OH = (x.h - 3) / 2 + 1, OW = (x.w - 3) / 2 + 1;
        Act cat = R.alloc(n, OH, OW, 1792, x.bf);
        Act s0 = Runner::slice(cat, 0, 384), s1 = Runner::slice(cat, 384, 256), s2 = Runner::slice(cat, 640, 256),
            s3 = Runner::slice(cat, 896, 896);
        Act t = R.bconv(x, f + "mixed_7a.fused", 1, 1, 1, 1, 0, 0);   // [branch0.0 | branch1.0 | branch2.0]
        Act t0 = Runner::slice(t, 0, 256), t1 = Runner::slice(t, 256, 256), t2 = Runner::slice(t, 512, 256);
        R.begin_group();                                              // three independent convs over slices of t: one launch
        R.bconv(t0, f + "mixed_7a.branch0.1", 3, 3, 2, 2, 0, 0, &s0);
        R.bconv(t1, f + "mixed_7a.branch1.1", 3, 3, 2, 2, 0, 0, &s1);
        Act a2 = R.bconv(t2, f + "mixed_7a.branch2.1", 3, 3, 1, 1, 1, 1);
        R.end_group();
        R.bconv(a2, f + "mixed_7a.branch2.2", 3, 3, 2, 2, 0, 0, &s2);
        R.pool(x, 3, 2, 0, &s3);
        x = cat;
    }
    for (int i = 0; i < 5; i++) x = block8(R, x, f + "repeat_3." + std::to_string(i), 0.20f, true);
    x = block8(R, x, f + "block8", 1.0f, false);
    if (R.err != TRL_OK) return R.err;
    // avgpool_1a -> last_linear (no bias) -> last_bn (folded) -> F.normalize
    Act g;
    if (!x.bf && x.h * x.w == 1 && x.coff == 0 && x.ld == x.c) {
        g = x;                                   // a 1x1 map IS its average (sum of one element / 1.0f, exact): no kernel
    } else {
        g = R.alloc(n, 1, 1, x.c);
        if (R.err != TRL_OK) return R.err;
        if (x.bf) TRL_CHECK(trl_launch_gap_bf16(reinterpret_cast<const uint16_t*>(x.p), n, x.h * x.w, x.c, g.p, s, c->cfg.embed_precision));
        else TRL_CHECK(trl_launch_gap(x.p, n, x.h * x.w, x.c, g.p, s));
    }
    Act e = R.conv(g, trl_w(c, f + "last_linear.w"), nullptr, trl_v(c, f + "last_bn.scale"), trl_v(c, f + "last_bn.shift"),
                   nullptr, 1, 1, 1, 1, 0, 0, TRL_ACT_NONE, nullptr, nullptr, 0.f);
    if (R.err != TRL_OK) return R.err;
    return trl_launch_l2norm512(e.p, d_valid, n, d_emb, s);
}

// RNet: d_out6[n][6] = {logit0, logit1, reg0..3}
int trl_run_rnet(trl_ctx* c, const float* d_crops, int n, float* d_out6, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    Runner R{c, s};
    Act x0; x0.p = const_cast<float*>(d_crops); x0.n = n; x0.h = 24; x0.w = 24; x0.c = 3; x0.ld = 3; x0.coff = 0;
    Act x = R.mconv(x0, "rnet", "conv1", "prelu1", 3);
    x = R.pool(x, 3, 2, 1);
    x = R.mconv(x, "rnet", "conv2", "prelu2", 3);
    x = R.pool(x, 3, 2, 1);
    x = R.mconv(x, "rnet", "conv3", "prelu3", 2);
    x = R.mconv(x, "rnet", "dense4", "prelu4", 3);
    Act out; out.p = d_out6; out.n = n; out.h = 1; out.w = 1; out.c = 6; out.ld = 6; out.coff = 0;
    R.conv(x, trl_w(c, "rnet.heads.w"), trl_v(c, "rnet.heads.b"), nullptr, nullptr, nullptr, 1, 1, 1, 1, 0, 0,
           TRL_ACT_NONE, &out, nullptr, 0.f);
    return R.err;
}

// ONet: d_out16[n][16] = {logit0, logit1, reg0..3, landmarks0..9}
int trl_run_onet(trl_ctx* c, const float* d_crops, int n, float* d_out16, hipStream_t s) {
    if (n <= 0) return TRL_OK;
    Runner R{c, s};
    Act x0; x0.p = const_cast<float*>(d_crops); x0.n = n; x0.h = 48; x0.w = 48; x0.c = 3; x0.ld = 3; x0.coff = 0;
    Act x = R.mconv(x0, "onet", "conv1", "prelu1", 3);
    x = R.pool(x, 3, 2, 1);
    x = R.mconv(x, "onet", "conv2", "prelu2", 3);
    x = R.pool(x, 3, 2, 1);
    x = R.mconv(x, "onet", "conv3", "prelu3", 3);
    x = R.pool(x, 2, 2, 1);
    x = R.mconv(x, "onet", "conv4", "prelu4", 2);
    x = R.mconv(x, "onet", "dense5", "prelu5", 3);
    Act out; out.p = d_out16; out.n = n; out.h = 1; out.w = 1; out.c = 16; out.ld = 16; out.coff = 0;
    R.conv(x, trl_w(c, "onet.heads.w"), trl_v(c, "onet.heads.b"), nullptr, nullptr, nullptr, 1, 1, 1, 1, 0, 0,
           TRL_ACT_NONE, &out, nullptr, 0.f);
    return R.err;
}

// R-Net from the fused front end's pooled map [n][11][11][28]
int trl_run_rnet_tail(trl_ctx* c, const float* d_pool1, int n, float* d_out6, hipStream_t s, const int32_t* n_dev, int n_base) {
    if (n <= 0) return TRL_OK;
    Runner R{c, s};
    R.m_dev = n_dev; R.m_base = n_base;
    Act x; x.p = const_cast<float*>(d_pool1); x.n = n; x.h = 11; x.w = 11; x.c = 28; x.ld = 28; x.coff = 0;
    x = R.mconv(x, "rnet", "conv2", "prelu2", 3);
    x = R.pool(x, 3, 2, 1);
    x = R.mconv(x, "rnet", "conv3", "prelu3", 2);
    x = R.mconv(x, "rnet", "dense4", "prelu4", 3);
    Act out; out.p = d_out6; out.n = n; out.h = 1; out.w = 1; out.c = 6; out.ld = 6; out.coff = 0;
    R.conv(x, trl_w(c, "rnet.heads.w"), trl_v(c, "rnet.heads.b"), nullptr, nullptr, nullptr, 1, 1, 1, 1, 0, 0,
           TRL_ACT_NONE, &out, nullptr, 0.f);
    return R.err;
}
// O-Net from the fused front end's pooled map [n][23][23][32]
int trl_run_onet_tail(trl_ctx* c, const float* d_pool1, int n, float* d_out16, hipStream_t s, const int32_t* n_dev, int n_base) {
    if (n <= 0) return TRL_OK;
    Runner R{c, s};
    R.m_dev = n_dev; R.m_base = n_base;
    Act x; x.p = const_cast<float*>(d_pool1); x.n = n; x.h = 23; x.w = 23; x.c = 32; x.ld = 32; x.coff = 0;
    x = R.mconv(x, "onet", "conv2", "prelu2", 3);
    x = R.pool(x, 3, 2, 1);
    x = R.mconv(x, "onet", "conv3", "prelu3", 3);
    x = R.pool(x, 2, 2, 1);
    x = R.mconv(x, "onet", "conv4", "prelu4", 2);
    x = R.mconv(x, "onet", "dense5", "prelu5", 3);
    Act out; out.p = d_out16; out.n = n; out.h = 1; out.w = 1; out.c = 16; out.ld = 16; out.coff = 0;
    R.conv(x, trl_w(c, "onet.heads.w"), trl_v(c, "onet.heads.b"), nullptr, nullptr, nullptr, 1, 1, 1, 1, 0, 0,
           TRL_ACT_NONE, &out, nullptr, 0.f);
    return R.err;
}

size_t trl_pnet_generic_bytes(int nf, int h, int w) {
    const size_t c1 = (size_t)(h - 2) * (w - 2) * 10, ph = (h - 2 + 1) / 2, pw = (w - 2 + 1) / 2;
    const size_t p1 = ph * pw * 10, c2 = (ph - 2) * (pw - 2) * 16, c3 = (ph - 4) * (pw - 4) * 32;
    return (size_t)nf * (c1 + p1 + c2 + c3) * sizeof(float) + 4096;
}

// PNet through the generic layer kernels (validation path / fallback): heads [nf][oh][ow][6]
int trl_run_pnet_generic(trl_ctx* c, const float* d_level, int nf, int h, int w, float* d_heads, hipStream_t s) {
    Runner R{c, s};
    Act x0; x0.p = const_cast<float*>(d_level); x0.n = nf; x0.h = h; x0.w = w; x0.c = 3; x0.ld = 3; x0.coff = 0;
    Act x = R.mconv(x0, "pnet", "conv1", "prelu1", 3);
    x = R.pool(x, 2, 2, 1);
    x = R.mconv(x, "pnet", "conv2", "prelu2", 3);
    x = R.mconv(x, "pnet", "conv3", "prelu3", 3);
    Act out; out.p = d_heads; out.n = nf; out.h = x.h; out.w = x.w; out.c = 6; out.ld = 6; out.coff = 0;
    R.conv(x, trl_w(c, "pnet.heads.w"), trl_v(c, "pnet.heads.b"), nullptr, nullptr, nullptr, 1, 1, 1, 1, 0, 0,
           TRL_ACT_NONE, &out, nullptr, 0.f);
    return R.err;
}
