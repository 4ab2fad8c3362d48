// trl_api.hip -- C ABI of libtruely_hip.so (see include/truely_hip.h for the contract and the
// reference lines each entry point replaces).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "trl_ctx.h"

static thread_local char g_err[512] = "";
int g_trl_no_fnconv = 0;
int g_trl_pnet_gate = 0;

#include <mutex>
namespace {
std::mutex g_gate_mu;
hipEvent_t g_gate_ev[64] = {};
bool g_gate_set[64] = {};
}
int trl_gate_wait(trl_ctx* c, hipStream_t s) {
    const int d = c->cfg.device;
    if (!g_trl_pnet_gate || d < 0 || d >= 64) return TRL_OK;
    std::lock_guard<std::mutex> lk(g_gate_mu);
    if (g_gate_set[d]) TRL_HIP(hipStreamWaitEvent(s, g_gate_ev[d], 0));
    return TRL_OK;
}
int trl_gate_record(trl_ctx* c, hipStream_t s) {
    const int d = c->cfg.device;
    if (!g_trl_pnet_gate || d < 0 || d >= 64) return TRL_OK;
    std::lock_guard<std::mutex> lk(g_gate_mu);
    if (!g_gate_ev[d]) TRL_HIP(hipEventCreateWithFlags(&g_gate_ev[d], hipEventDisableTiming));
    TRL_HIP(hipEventRecord(g_gate_ev[d], s));
    g_gate_set[d] = true;
    return TRL_OK;
}
void trl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" {

int trl_abi_version(void) { return TRL_ABI_VERSION; }
const char* trl_last_error(void) { return g_err; }

int trl_default_config(trl_config* cfg) {
    if (!cfg) return TRL_ERR_INVALID;
    cfg->device = 0;
    cfg->min_face_size = 20;           // facenet_pytorch MTCNN.__init__ defaults (server/model.py:18)
    cfg->thr0 = 0.6f; cfg->thr1 = 0.7f; cfg->thr2 = 0.7f;
    cfg->factor = 0.709;
    cfg->cap_level = 2048;
    cfg->cap_frame = 2048;
    cfg->max_faces = 64;
    cfg->pnet_mode = 0;
    cfg->embed_mode = 0;
    cfg->embed_precision = 0;
    return TRL_OK;
}

int trl_create(const trl_config* cfg, trl_ctx** out) {
    if (!cfg || !out) { trl_set_error("null argument"); return TRL_ERR_INVALID; }
    if (cfg->cap_level < 64 || cfg->cap_level > (1 << 24) || cfg->cap_frame < 64 || cfg->cap_frame > (1 << 24) || (cfg->cap_level & 3) ||
        (cfg->cap_frame & 3) || cfg->min_face_size < 12 || cfg->max_faces < 1 || !(cfg->factor > 0.1 && cfg->factor < 0.99) || cfg->embed_mode < 0 || cfg->embed_mode > 3 ||
        cfg->embed_precision < 0 || cfg->embed_precision > 2) {
        trl_set_error("bad trl_config (list start capacities must be multiples of 4 in [64, 2^24], min_face_size >= 12)");
        return TRL_ERR_INVALID;
    }
    TRL_HIP(hipSetDevice(cfg->device));
    trl_ctx* c = new trl_ctx();
    c->cfg = *cfg;
    // execution span of every fused PNet launch (two atomics per workgroup, summed on the device: trl_debug_pnet_span);
    // TRL_PNET_CLOCK additionally selects the DBG instantiation with per-phase wave clocks
    c->pnet_prof = trl_tune_set("TRL_PNET_CLOCK");
    if (hipMalloc((void**)&c->pnet_clk, 8 * 40) != hipSuccess || hipMemset(c->pnet_clk, 0, 8 * 40) != hipSuccess || hipMemset(c->pnet_clk, 0xFF, 8) != hipSuccess ||
        hipMalloc((void**)&c->pnet_cursor, 64) != hipSuccess || hipHostMalloc((void**)&c->h_pinned, 1024) != hipSuccess ||
        hipEventCreate(&c->ev_call0) != hipSuccess || hipEventCreate(&c->ev_call1) != hipSuccess) {
        trl_set_error("context allocation failed: %s", hipGetErrorString(hipGetLastError()));
        trl_destroy(c);                      // frees whatever was created
        return TRL_ERR_HIP;
    }
    memset(c->h_pinned, 0, 1024);
    *out = c;
    return TRL_OK;
}

int trl_destroy(trl_ctx* c) {
    if (!c) return TRL_OK;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();            // (also ends a call that was queued and never finished: its kernels are done)
    c->pend.active = false;
    for (auto& kv : c->W) if (kv.second.pt) (void)hipFree(kv.second.pt);
    if (c->wdev) (void)hipFree(c->wdev);
    if (c->arena.base) (void)hipFree(c->arena.base);
    if (c->scratch.base) (void)hipFree(c->scratch.base);
    if (c->sims_tmp.base) (void)hipFree(c->sims_tmp.base);
    if (c->pyr_tab) (void)hipFree(c->pyr_tab);
    if (c->pnet_clk) {
        // TRL_PNET_CLOCK: where the waves of the fused PNet launches spent their time (shader clocks per tile and wave, DBG instantiation)
        unsigned long long t[40];
        if (c->pnet_prof && hipMemcpy(t, c->pnet_clk, sizeof t, hipMemcpyDeviceToHost) == hipSuccess && t[2 + 32] > 0) {
            static const char* nm[8] = {"phase0", "barrier0", "phase1", "barrier1", "phase2", "barrier2", "phase3", "barrier3"};
            const double tiles = (double)t[2 + 32];
            fprintf(stderr, "[TRL_PNET_CLOCK] shader clocks per tile (wave 0..3), %.0f workgroup-tiles\n", tiles);
            for (int k = 0; k < 8; k++)
                fprintf(stderr, "[TRL_PNET_CLOCK] %-9s %8.0f %8.0f %8.0f %8.0f\n", nm[k], t[2 + k] / tiles, t[2 + 8 + k] / tiles, t[2 + 16 + k] / tiles, t[2 + 24 + k] / tiles);
        }
        (void)hipFree(c->pnet_clk);
    }
    if (c->pnet_cursor) (void)hipFree(c->pnet_cursor);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->ev_call0) (void)hipEventDestroy(c->ev_call0);
    if (c->ev_call1) (void)hipEventDestroy(c->ev_call1);
    for (auto& e : c->pnet_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete c;
    return TRL_OK;
}

}  // extern "C"

// ---- weights ---------------------------------------------------------------------------------------
namespace {
struct Entry {
    char name[56];
    uint32_t ndim;
    uint32_t dims[4];
    uint64_t offset;
    uint64_t nbytes;
};
static_assert(sizeof(Entry) == 96, "TRLW entry layout");
}  // namespace

const DevW* trl_w(trl_ctx* c, const std::string& name) {
    auto it = c->W.find(name);
    if (it == c->W.end()) { trl_set_error("weight matrix '%s' not loaded", name.c_str()); return nullptr; }
    return &it->second;
}
const DevV* trl_v(trl_ctx* c, const std::string& name) {
    auto it = c->V.find(name);
    if (it == c->V.end()) { trl_set_error("weight vector '%s' not loaded", name.c_str()); return nullptr; }
    return &it->second;
}

int trl_ensure(trl_ctx* c, Arena& a, size_t bytes) {
    if (bytes <= a.cap) return TRL_OK;
    // Growing frees the old block, so it is only called while nothing allocated from `a` is live:
    // at the start of a call (arena) or between cascade stages (scratch).
    const size_t ncap = bytes + (bytes >> 3) + (32u << 20);
    TRL_HIP(hipDeviceSynchronize());
    if (a.base) { TRL_HIP(hipFree(a.base)); a.base = nullptr; a.cap = 0; }
    TRL_HIP(hipMalloc((void**)&a.base, ncap));
    a.cap = ncap;
    a.off = 0;
    if (c->dbg_poison >= 0) TRL_HIP(hipMemset(a.base, c->dbg_poison, ncap));   // test hook: workspaces that grow stay poisoned
    return TRL_OK;
}

extern "C" int trl_load_weights(trl_ctx* c, const void* blob, size_t nbytes) {
    if (!c || !blob) { trl_set_error("null argument"); return TRL_ERR_INVALID; }
    const uint8_t* b = (const uint8_t*)blob;
    if (nbytes < 16 || memcmp(b, "TRLW0001", 8) != 0) { trl_set_error("bad weights blob magic"); return TRL_ERR_WEIGHTS; }
    uint32_t nt;
    memcpy(&nt, b + 8, 4);
    if (16 + (size_t)nt * sizeof(Entry) > nbytes) { trl_set_error("truncated weights blob"); return TRL_ERR_WEIGHTS; }
    const Entry* ent = (const Entry*)(b + 16);
    TRL_HIP(hipSetDevice(c->cfg.device));

    struct Pending { std::string name; bool mat; int K, Cout, Kpad, ld, n; size_t off; const float* src; };
    std::vector<Pending> items;
    size_t total = 0;
    auto add_mat = [&](const std::string& name, int K, int Cout, const float* src) {
        Pending p; p.name = name; p.mat = true; p.K = K; p.Cout = Cout; p.Kpad = (K + 15) / 16 * 16; p.ld = (Cout + 31) / 32 * 32;
        p.n = 0; p.src = src; p.off = total;
        total += ((size_t)p.Kpad * p.ld * 4 + 255) & ~(size_t)255;
        items.push_back(p);
    };
    auto add_vec = [&](const std::string& name, int n, const float* src) {
        Pending p; p.name = name; p.mat = false; p.K = p.Cout = p.Kpad = p.ld = 0; p.n = n; p.src = src; p.off = total;
        total += ((size_t)((n + 127) / 128 * 128) * 4 + 255) & ~(size_t)255;
        items.push_back(p);
    };
    std::unordered_map<std::string, const Entry*> idx;
    for (uint32_t i = 0; i < nt; i++) {
        const Entry& e = ent[i];
        if (e.offset > nbytes || e.nbytes > nbytes - e.offset || (e.offset & 3)) { trl_set_error("tensor out of blob bounds"); return TRL_ERR_WEIGHTS; }
        std::string name(e.name, strnlen(e.name, 56));
        // the declared shape must account for exactly the bytes the entry owns (the copies below trust the shape)
        const uint64_t elems = e.ndim == 2 ? (uint64_t)e.dims[0] * e.dims[1] : (e.ndim == 1 ? (uint64_t)e.dims[0] : 0);
        if ((e.ndim == 1 || e.ndim == 2) && (elems == 0 || elems > (1ull << 31) || elems * 4 != e.nbytes)) {
            trl_set_error("tensor '%s': shape and byte count disagree", name.c_str());
            return TRL_ERR_WEIGHTS;
        }
        idx[name] = &e;
        const float* src = (const float*)(b + e.offset);
        if (e.ndim == 2) add_mat(name, (int)e.dims[0], (int)e.dims[1], src);
        else if (e.ndim == 1) add_vec(name, (int)e.dims[0], src);
    }
    // merged heads: the 1x1 class and regression (and landmark) convs share their input, so they run as
    // one [K][6] / [K][16] matrix; each output column is still its own fmaf chain (bit-identical).
    std::vector<std::vector<float>> keep_alive;
    auto merge_heads = [&](const std::string& net, const std::vector<std::string>& parts) -> int {
        int K = -1, tot = 0;
        for (auto& p : parts) {
            auto it = idx.find(net + "." + p + ".w");
            auto ib = idx.find(net + "." + p + ".b");
            if (it == idx.end() || ib == idx.end()) { trl_set_error("missing head %s.%s", net.c_str(), p.c_str()); return TRL_ERR_WEIGHTS; }
            if (K < 0) K = (int)it->second->dims[0];
            if (K != (int)it->second->dims[0] || ib->second->dims[0] != it->second->dims[1]) {
                trl_set_error("head %s.%s has an unexpected shape", net.c_str(), p.c_str());
                return TRL_ERR_WEIGHTS;
            }
            tot += (int)it->second->dims[1];
        }
        keep_alive.emplace_back((size_t)K * tot);
        keep_alive.emplace_back((size_t)tot);       // (keep_alive is reserved below: the references stay valid)
        std::vector<float>& bias = keep_alive[keep_alive.size() - 1];
        std::vector<float>& wm = keep_alive[keep_alive.size() - 2];
        int col = 0;
        for (auto& p : parts) {
            const Entry* e = idx[net + "." + p + ".w"];
            const Entry* eb = idx[net + "." + p + ".b"];
            const float* src = (const float*)(b + e->offset);
            const float* sb = (const float*)(b + eb->offset);
            const int co = (int)e->dims[1];
            for (int k = 0; k < K; k++) for (int j = 0; j < co; j++) wm[(size_t)k * tot + col + j] = src[(size_t)k * co + j];
            for (int j = 0; j < co; j++) bias[col + j] = sb[j];
            col += co;
        }
        add_mat(net + ".heads.w", K, tot, wm.data());
        add_vec(net + ".heads.b", tot, bias.data());
        return TRL_OK;
    };
    keep_alive.reserve(512);
    // FaceNet: 1x1 BasicConv2d branches that read the same input run as ONE conv with concatenated output
    // columns (each column keeps its own fmaf chain, so results are unchanged): fewer, wider launches.
    auto fuse_bconv = [&](const std::string& out, const std::vector<std::string>& parts) -> int {
        int K = -1, tot = 0;
        for (auto& p : parts) {
            auto it = idx.find(p + ".w");
            if (it == idx.end() || idx.find(p + ".scale") == idx.end() || idx.find(p + ".shift") == idx.end()) {
                trl_set_error("missing tensors of %s", p.c_str());
                return TRL_ERR_WEIGHTS;
            }
            if (K < 0) K = (int)it->second->dims[0];
            if (K != (int)it->second->dims[0]) { trl_set_error("fused convs disagree on K (%s)", p.c_str()); return TRL_ERR_WEIGHTS; }
            tot += (int)it->second->dims[1];
        }
        keep_alive.emplace_back((size_t)K * tot); const size_t iw = keep_alive.size() - 1;
        keep_alive.emplace_back((size_t)tot);      const size_t isc = keep_alive.size() - 1;
        keep_alive.emplace_back((size_t)tot);      const size_t ish = keep_alive.size() - 1;
        int col = 0;
        for (auto& p : parts) {
            const Entry* e = idx[p + ".w"];
            const float* src = (const float*)(b + e->offset);
            const float* ssc = (const float*)(b + idx[p + ".scale"]->offset);
            const float* ssh = (const float*)(b + idx[p + ".shift"]->offset);
            const int co = (int)e->dims[1];
            for (int k = 0; k < K; k++) for (int j = 0; j < co; j++) keep_alive[iw][(size_t)k * tot + col + j] = src[(size_t)k * co + j];
            for (int j = 0; j < co; j++) { keep_alive[isc][col + j] = ssc[j]; keep_alive[ish][col + j] = ssh[j]; }
            col += co;
        }
        add_mat(out + ".w", K, tot, keep_alive[iw].data());
        add_vec(out + ".scale", tot, keep_alive[isc].data());
        add_vec(out + ".shift", tot, keep_alive[ish].data());
        return TRL_OK;
    };
    for (int i = 0; i < 5; i++) {
        const std::string p = "facenet.repeat_1." + std::to_string(i);
        TRL_CHECK(fuse_bconv(p + ".fused", {p + ".branch0", p + ".branch2.0", p + ".branch1.0"}));   // [b0 | b2.0 | b1.0]
    }
    for (int i = 0; i < 10; i++) {
        const std::string p = "facenet.repeat_2." + std::to_string(i);
        TRL_CHECK(fuse_bconv(p + ".fused", {p + ".branch0", p + ".branch1.0"}));
    }
    for (int i = 0; i < 6; i++) {
        const std::string p = i < 5 ? "facenet.repeat_3." + std::to_string(i) : std::string("facenet.block8");
        TRL_CHECK(fuse_bconv(p + ".fused", {p + ".branch0", p + ".branch1.0"}));
    }
    TRL_CHECK(fuse_bconv("facenet.mixed_7a.fused", {"facenet.mixed_7a.branch0.0", "facenet.mixed_7a.branch1.0", "facenet.mixed_7a.branch2.0"}));
    TRL_CHECK(merge_heads("pnet", {"conv4_1", "conv4_2"}));
    TRL_CHECK(merge_heads("rnet", {"dense5_1", "dense5_2"}));
    TRL_CHECK(merge_heads("onet", {"dense6_1", "dense6_2", "dense6_3"}));

    std::vector<char> host(total, 0);
    for (auto& p : items) {
        float* dst = (float*)(host.data() + p.off);
        if (p.mat) { for (int k = 0; k < p.K; k++) memcpy(dst + (size_t)k * p.ld, p.src + (size_t)k * p.Cout, (size_t)p.Cout * 4); }
        else memcpy(dst, p.src, (size_t)p.n * 4);
    }
    if (c->wdev) { TRL_HIP(hipDeviceSynchronize()); TRL_HIP(hipFree(c->wdev)); c->wdev = nullptr; }
    TRL_HIP(hipMalloc((void**)&c->wdev, total));
    TRL_HIP(hipMemcpy(c->wdev, host.data(), total, hipMemcpyHostToDevice));
    c->wbytes = total;
    for (auto& kv : c->W) if (kv.second.pt) (void)hipFree(kv.second.pt);
    c->W.clear(); c->V.clear();
    for (auto& p : items) {
        if (p.mat) { DevW w; w.p = (float*)(c->wdev + p.off); w.K = p.K; w.Cout = p.Cout; w.Kpad = p.Kpad; w.ld = p.ld; c->W[p.name] = w; }
        else { DevV v; v.p = (float*)(c->wdev + p.off); v.n = p.n; c->V[p.name] = v; }
    }

    TRL_CHECK(trl_pnet_prepare(c));
    c->rnet_front_mode = c->onet_front_mode = -1;
    if (c->cfg.embed_precision >= 1) {   // bf16 / fp16 copies of the embedder's conv weights (all but the 3-channel stem and the final linear)
        for (auto& kv : c->W) {
            const std::string& nm = kv.first;
            if (nm.rfind("facenet.", 0) != 0 || nm == "facenet.conv2d_1a.w" || nm == "facenet.last_linear.w") continue;
            TRL_CHECK(trl_make_weight_bf16(&kv.second, nullptr, c->cfg.embed_precision));
        }
        TRL_HIP(hipDeviceSynchronize());
    }
    c->have_weights = true;
    return TRL_OK;
}

// ---- hot path ---------------------------------------------------------------------------------------
// Every attempt raises the one capacity it found too small to what the content needs (level lists -> frame lists -> R-Net batch
// -> O-Net batch, plus the spill workspace), so a call converges in at most a handful; the bound only guards against a bug.
enum { TRL_MAX_ATTEMPTS = 12 };

// A context holds at most one queued call (trl_detect_embed_begin .. _end); anything else that touches its workspaces meanwhile
// would corrupt that call silently.
static int check_idle(trl_ctx* c) {
    if (!c) { trl_set_error("null context"); return TRL_ERR_INVALID; }
    if (c->pend.active) { trl_set_error("the context has a call in flight: trl_detect_embed_end() first"); return TRL_ERR_STATE; }
    return TRL_OK;
}

static int check_call(trl_ctx* c, const void* frames, int n, int H, int W) {
    if (!c) { trl_set_error("null context"); return TRL_ERR_INVALID; }
    TRL_CHECK(check_idle(c));
    if (!c->have_weights) { trl_set_error("trl_load_weights has not been called"); return TRL_ERR_STATE; }
    if (!frames || n <= 0 || n > 65535 || H < 12 || W < 12 || H > 16383 || W > 16383) {   // n: grid.y carries the frame index in several kernels
        trl_set_error("bad frame batch n=%d H=%d W=%d (1..65535 frames of 12..16383 px per side)", n, H, W);
        return TRL_ERR_INVALID;
    }
    return TRL_OK;
}

static void collect_timings(trl_ctx* c) {
    float call_ms = 0.f, pnet_ms = 0.f, pyr_ms = 0.f;
    (void)hipEventElapsedTime(&call_ms, c->ev_call0, c->ev_call1);
    int launches = 0;
    if (c->cfg.pnet_mode == 0 && c->pnet_ev_used >= 2) {
        // pair 0 = pyramid kernel, pair 1 = the fused PNet kernel (the dominant kernel, one launch)
        (void)hipEventElapsedTime(&pyr_ms, c->pnet_ev[0].first, c->pnet_ev[0].second);
        (void)hipEventElapsedTime(&pnet_ms, c->pnet_ev[1].first, c->pnet_ev[1].second);
        launches = 1;
    } else {
        for (int i = 0; i < c->pnet_ev_used; i++) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, c->pnet_ev[i].first, c->pnet_ev[i].second) == hipSuccess) pnet_ms += t;
        }
        launches = c->pnet_ev_used;
    }
    c->last_ms[0] = pnet_ms; c->last_ms[1] = call_ms; c->last_ms[2] = (float)launches; c->last_ms[3] = pyr_ms;
    c->pnet_kernel_ms = 0.f;
    if (c->cfg.pnet_mode == 0 && c->pnet_clk && c->pnet_prof) {   // diagnostic mode only: a blocking copy per call (the last launch's span)
        unsigned long long t = 0;
        int khz = 0;
        if (hipMemcpy(&t, c->pnet_clk + 38, sizeof t, hipMemcpyDeviceToHost) == hipSuccess &&
            hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->cfg.device) == hipSuccess && khz > 0)
            c->pnet_kernel_ms = (float)((double)t / (double)khz);
    }
}

extern "C" {

static int mtcnn_detect_impl(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_boxes, float* d_probs, float* d_points,
                             int32_t* d_counts, void* stream);

int trl_mtcnn_detect(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_boxes, float* d_probs, int32_t* d_counts,
                     void* stream) {
    return mtcnn_detect_impl(c, d_frames, n, H, W, d_boxes, d_probs, nullptr, d_counts, stream);
}

int trl_mtcnn_detect_landmarks(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_boxes, float* d_probs,
                               float* d_points, int32_t* d_counts, void* stream) {
    if (!d_points) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    return mtcnn_detect_impl(c, d_frames, n, H, W, d_boxes, d_probs, d_points, d_counts, stream);
}

static int mtcnn_detect_impl(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_boxes, float* d_probs, float* d_points,
                             int32_t* d_counts, void* stream) {
    TRL_CHECK(check_call(c, d_frames, n, H, W));
    if (!d_boxes || !d_probs || !d_counts) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    TRL_HIP(hipSetDevice(c->cfg.device));
    c->scratch_after_cascade = 0;
    for (int attempt = 0;; attempt++) {
        TRL_HIP(hipEventRecord(c->ev_call0, s));
        TRL_CHECK(trl_cascade_detect(c, d_frames, n, H, W, s, attempt ? c->resume_stage : 0));
        // scratch for the model.py-only outputs
        float* box0 = (float*)c->arena.alloc((size_t)n * 16); float* prob0 = (float*)c->arena.alloc((size_t)n * 4);
        int32_t* rect = (int32_t*)c->arena.alloc((size_t)n * 16); uint8_t* valid = (uint8_t*)c->arena.alloc((size_t)n);
        if (!valid) { trl_set_error("arena exhausted"); return TRL_ERR_STATE; }
        TRL_HIP(hipMemsetAsync(d_boxes, 0, (size_t)n * c->cfg.max_faces * 16, s));
        TRL_HIP(hipMemsetAsync(d_probs, 0, (size_t)n * c->cfg.max_faces * 4, s));
        if (d_points) TRL_HIP(hipMemsetAsync(d_points, 0, (size_t)n * c->cfg.max_faces * 40, s));
        TRL_CHECK(trl_cascade_finish(c, d_frames, n, H, W, d_boxes, d_probs, d_points, d_counts, box0, prob0, rect, valid, nullptr, s));
        TRL_HIP(hipEventRecord(c->ev_call1, s));
        TRL_CHECK(trl_gate_record(c, s));
        TRL_HIP(hipStreamSynchronize(s));             // the call's one host synchronisation
        int retry = 0;
        TRL_CHECK(trl_cascade_check(c, n, &retry));
        c->last_attempts = attempt + 1;
        if (!retry) break;
        if (attempt >= TRL_MAX_ATTEMPTS - 1) { trl_set_error("candidate capacities did not converge"); return TRL_ERR_STATE; }
    }
    collect_timings(c);
    return TRL_OK;
}

int trl_facenet_embed(trl_ctx* c, const float* d_faces, int n, int h, int w, float* d_emb, void* stream) {
    if (!c || !c->have_weights) { trl_set_error("context without weights"); return TRL_ERR_STATE; }
    TRL_CHECK(check_idle(c));
    if (!d_faces || !d_emb || n <= 0 || h < 75 || w < 75) { trl_set_error("bad face batch n=%d %dx%d (min 75x75)", n, h, w); return TRL_ERR_INVALID; }
    TRL_HIP(hipSetDevice(c->cfg.device));
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, (size_t)n * ((size_t)h * w * 110 + 400000) * 4 + (8u << 20)));
    TRL_CHECK(trl_run_facenet(c, d_faces, n, h, w, nullptr, d_emb, (hipStream_t)stream));
    return trl_gate_record(c, (hipStream_t)stream);
}

// model.py:47-58 (detect, largest box, crop + resize) and optionally :59 (embed).  d_faces_out != null: the crops are written there
// (caller-owned [n][S][S][3] f32) and the embedder is NOT run (trl_detect_crop); else they live in scratch and are embedded.
// Split in two so a host thread can keep several contexts busy: detect_embed_enqueue() queues ONE attempt of the call on the
// stream and returns; detect_embed_wait() is the call's one host synchronisation, the capacity check and -- rarely -- the re-run.
static int detect_embed_enqueue(trl_ctx* c) {
    const trl_ctx::Pending& q = c->pend;
    hipStream_t s = (hipStream_t)q.stream;
    const int n = q.n, H = q.H, W = q.W;
    const int S = c->cfg.embed_mode == 0 ? 80 : 160;
    if (!q.attempt) TRL_HIP(hipEventRecord(c->ev_call0, s));
    TRL_CHECK(trl_cascade_detect(c, q.frames, n, H, W, s, q.attempt ? c->resume_stage : 0));
    float* pts0 = nullptr;
    if (c->cfg.embed_mode == 3) {                // the largest face's landmarks steer the aligned crop
        pts0 = (float*)c->arena.alloc((size_t)n * 40);
        if (!pts0) { trl_set_error("arena exhausted"); return TRL_ERR_STATE; }
    }
    TRL_CHECK(trl_cascade_finish(c, q.frames, n, H, W, nullptr, nullptr, nullptr, nullptr, q.box, q.prob, q.rect, q.valid, pts0, s));
    float* faces = q.faces_out;
    if (!faces) {
        c->scratch.reset();                      // stream order: the cascade's kernels are done with it before these run
        faces = (float*)c->scratch.alloc((size_t)n * S * S * 3 * 4);
        if (!faces) { trl_set_error("arena exhausted"); return TRL_ERR_STATE; }
    }
    if (c->cfg.embed_mode == 0) TRL_CHECK(trl_launch_crop_resize80(q.frames, n, H, W, q.rect, q.valid, faces, s));
    else if (c->cfg.embed_mode == 3) TRL_CHECK(trl_launch_crop_aligned(q.frames, n, H, W, pts0, q.valid, S, true, faces, s));
    else TRL_CHECK(trl_launch_crop_area_std(q.frames, n, H, W, q.rect, q.valid, S, c->cfg.embed_mode == 2, faces, s));
    if (!q.faces_out) TRL_CHECK(trl_run_facenet(c, faces, n, S, S, q.valid, q.emb, s));
    TRL_HIP(hipEventRecord(c->ev_call1, s));
    TRL_CHECK(trl_gate_record(c, s));
    return TRL_OK;
}

static int detect_embed_begin(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_box, float* d_prob, int32_t* d_rect,
                              uint8_t* d_valid, float* d_emb, float* d_faces_out, void* stream) {
    TRL_CHECK(check_call(c, d_frames, n, H, W));
    if (c->pend.active) { trl_set_error("the context already has a call in flight: trl_detect_embed_end() first"); return TRL_ERR_STATE; }
    if (!d_box || !d_prob || !d_rect || !d_valid || (!d_emb && !d_faces_out)) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    TRL_HIP(hipSetDevice(c->cfg.device));
    const int S = c->cfg.embed_mode == 0 ? 80 : 160;
    c->scratch_after_cascade = d_faces_out ? 0 : (size_t)n * ((size_t)S * S * 110 + 400000) * 4 + (8u << 20);
    c->pend = trl_ctx::Pending{true, 0, d_frames, n, H, W, d_box, d_prob, d_rect, d_valid, d_emb, d_faces_out, stream};
    const int st = detect_embed_enqueue(c);
    if (st != TRL_OK) c->pend.active = false;
    return st;
}

static int detect_embed_wait(trl_ctx* c) {
    if (!c) { trl_set_error("null context"); return TRL_ERR_INVALID; }
    if (!c->pend.active) { trl_set_error("no call in flight on this context"); return TRL_ERR_STATE; }
    TRL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = (hipStream_t)c->pend.stream;
    int st = TRL_OK;
    for (;;) {
        if ((st = (hipStreamSynchronize(s) == hipSuccess ? TRL_OK : TRL_ERR_HIP)) != TRL_OK) { trl_set_error("hipStreamSynchronize: %s", hipGetErrorString(hipGetLastError())); break; }
        int retry = 0;
        if ((st = trl_cascade_check(c, c->pend.n, &retry)) != TRL_OK) break;
        c->last_attempts = ++c->pend.attempt;
        if (!retry) break;                            // (a retry re-runs the call with larger R-/O-Net batch capacities)
        if (c->pend.attempt >= TRL_MAX_ATTEMPTS) { trl_set_error("candidate capacities did not converge"); st = TRL_ERR_STATE; break; }
        if ((st = detect_embed_enqueue(c)) != TRL_OK) break;
    }
    c->pend.active = false;
    if (st == TRL_OK) collect_timings(c);
    return st;
}

static int detect_embed_impl(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_box, float* d_prob, int32_t* d_rect,
                             uint8_t* d_valid, float* d_emb, float* d_faces_out, void* stream) {
    TRL_CHECK(detect_embed_begin(c, d_frames, n, H, W, d_box, d_prob, d_rect, d_valid, d_emb, d_faces_out, stream));
    return detect_embed_wait(c);
}

int trl_detect_embed_begin(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_box, float* d_prob, int32_t* d_rect,
                           uint8_t* d_valid, float* d_emb, void* stream) {
    if (!d_emb) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    return detect_embed_begin(c, d_frames, n, H, W, d_box, d_prob, d_rect, d_valid, d_emb, nullptr, stream);
}
int trl_detect_crop_begin(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_box, float* d_prob, int32_t* d_rect,
                          uint8_t* d_valid, float* d_faces, void* stream) {
    if (!d_faces) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    return detect_embed_begin(c, d_frames, n, H, W, d_box, d_prob, d_rect, d_valid, nullptr, d_faces, stream);
}
int trl_detect_embed_end(trl_ctx* c) { return detect_embed_wait(c); }

int trl_detect_embed(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_box, float* d_prob, int32_t* d_rect,
                     uint8_t* d_valid, float* d_emb, void* stream) {
    if (!d_emb) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    return detect_embed_impl(c, d_frames, n, H, W, d_box, d_prob, d_rect, d_valid, d_emb, nullptr, stream);
}

int trl_detect_crop(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, float* d_box, float* d_prob, int32_t* d_rect,
                    uint8_t* d_valid, float* d_faces, void* stream) {
    if (!d_faces) { trl_set_error("null output"); return TRL_ERR_INVALID; }
    return detect_embed_impl(c, d_frames, n, H, W, d_box, d_prob, d_rect, d_valid, nullptr, d_faces, stream);
}

int trl_facenet_embed_masked(trl_ctx* c, const float* d_faces, const uint8_t* d_valid, int n, int h, int w, float* d_emb, void* stream) {
    if (!c || !c->have_weights) { trl_set_error("context without weights"); return TRL_ERR_STATE; }
    TRL_CHECK(check_idle(c));
    if (!d_faces || !d_valid || !d_emb || n <= 0 || h < 75 || w < 75) { trl_set_error("bad face batch n=%d %dx%d (min 75x75)", n, h, w); return TRL_ERR_INVALID; }
    TRL_HIP(hipSetDevice(c->cfg.device));
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, (size_t)n * ((size_t)h * w * 110 + 400000) * 4 + (8u << 20)));
    TRL_CHECK(trl_run_facenet(c, d_faces, n, h, w, d_valid, d_emb, (hipStream_t)stream));
    return trl_gate_record(c, (hipStream_t)stream);
}

int trl_drift_score(trl_ctx* c, const float* d_emb, const uint8_t* d_valid, int n, long long frame_count, int fps, float* d_sims,
                    uint8_t* d_flags, int32_t* d_result, void* stream) {
    if (!c || !d_emb || !d_valid || !d_result || n < 0) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    TRL_HIP(hipSetDevice(c->cfg.device));
    float* sims = d_sims;
    if (!sims) {   // the scan needs the similarities even when the caller does not want them
        TRL_CHECK(trl_ensure(c, c->sims_tmp, (size_t)(n > 0 ? n : 1) * sizeof(float)));
        sims = (float*)c->sims_tmp.base;
    }
    return trl_launch_drift(d_emb, d_valid, n, frame_count, fps, sims, d_flags, d_result, (hipStream_t)stream);
}

// trl_drift_score continued across the windows of ONE clip: d_state carries `previous embedding / run / hits` (model.py:60-75)
int trl_drift_update(trl_ctx* c, void* d_state, const float* d_emb, const uint8_t* d_valid, int n, long long frame_count, int fps,
                     float* d_sims, uint8_t* d_flags, int32_t* d_result, void* stream) {
    if (!c || !d_state || !d_result || n < 0 || (n > 0 && (!d_emb || !d_valid)) || ((uintptr_t)d_state & 3)) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    TRL_HIP(hipSetDevice(c->cfg.device));
    float* sims = d_sims;
    if (!sims) {
        TRL_CHECK(trl_ensure(c, c->sims_tmp, (size_t)(n > 0 ? n : 1) * sizeof(float)));
        sims = (float*)c->sims_tmp.base;
    }
    return trl_launch_drift(d_emb, d_valid, n, frame_count, fps, sims, d_flags, d_result, (hipStream_t)stream, d_state);
}

// ---- inspection hooks ----------------------------------------------------------------------------------
int trl_debug_stage_boxes(trl_ctx* c, int stage, int frame, float* h_boxes, int max_rows, int* n_out) {
    TRL_CHECK(check_idle(c));
    if (!c->cb.n1 || frame < 0 || frame >= c->cb.n || stage < 1 || stage > 3 || !n_out || (max_rows > 0 && !h_boxes)) {
        trl_set_error("no cascade state");
        return TRL_ERR_STATE;
    }
    const int32_t* cnt = stage == 1 ? c->cb.n1 : (stage == 2 ? c->cb.n2 : c->cb.n3);
    const float* src = stage == 1 ? c->cb.s1_box : (stage == 2 ? c->cb.s2_box : c->cb.s3_box);
    int32_t k = 0;
    TRL_HIP(hipDeviceSynchronize());
    TRL_HIP(hipMemcpy(&k, cnt + frame, 4, hipMemcpyDeviceToHost));
    *n_out = k;
    const int m = k < max_rows ? k : max_rows;
    if (m > 0) TRL_HIP(hipMemcpy(h_boxes, src + (size_t)frame * c->cb.capF * 5, (size_t)m * 20, hipMemcpyDeviceToHost));
    return TRL_OK;
}

__global__ void k_poison_lds(unsigned word, int nwords) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) lds_words[i] = word;
    __syncthreads();
    if (lds_words[(threadIdx.x * 97) % nwords] != word) __builtin_trap();   // keeps the stores alive
}

// Fills every byte of the activation workspaces with `byte` (0xFF = NaN patterns, 0x7F = huge finite floats): a
// result that depends on workspace contents left by an earlier call or process shows up as a parity failure.
int trl_debug_poison(trl_ctx* c, int byte) {
    TRL_CHECK(check_idle(c));
    TRL_HIP(hipSetDevice(c->cfg.device));
    TRL_HIP(hipDeviceSynchronize());
    c->dbg_poison = byte & 0xFF;                 // sticky: blocks allocated later are filled too
    if (c->scratch.base) TRL_HIP(hipMemset(c->scratch.base, byte, c->scratch.cap));
    if (c->arena.base) TRL_HIP(hipMemset(c->arena.base, byte, c->arena.cap));
    // ... and the LDS of every CU (it keeps the previous kernel's contents): 1024 workgroups of 160 KB, one per CU at a time
    const int lds_bytes = 160 * 1024;
    TRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const unsigned w = (unsigned)(byte & 0xFF) * 0x01010101u;
    k_poison_lds<<<1024, 256, lds_bytes, 0>>>(w, lds_bytes / 4);
    TRL_LAUNCH_CHECK();   // cascade lists: every call re-initialises what it reads
    TRL_HIP(hipDeviceSynchronize());
    return TRL_OK;
}

int trl_debug_level_counts(trl_ctx* c, int frame, int32_t* h_cand, int32_t* h_keep, int* n_levels) {
    TRL_CHECK(check_idle(c));
    if (!c->cb.lvl_cnt || frame < 0 || frame >= c->cb.n || !h_cand || !h_keep || !n_levels) { trl_set_error("no cascade state"); return TRL_ERR_STATE; }
    TRL_HIP(hipDeviceSynchronize());
    const int L = c->cb.L;
    TRL_HIP(hipMemcpy(h_cand, c->cb.lvl_cnt + (size_t)frame * L, (size_t)L * 4, hipMemcpyDeviceToHost));
    TRL_HIP(hipMemcpy(h_keep, c->cb.lvl_keep_cnt + (size_t)frame * L, (size_t)L * 4, hipMemcpyDeviceToHost));
    *n_levels = L;
    return TRL_OK;
}

// Candidate records (generateBoundingBox rows) one (frame, level) of the last call produced, in append order
int trl_debug_level_cands(trl_ctx* c, int frame, int level, void* h_rows, int max_rows, int* n_out) {
    TRL_CHECK(check_idle(c));
    if (!c->cb.lvl_cnt || frame < 0 || frame >= c->cb.n || level < 0 || level >= c->cb.L || !n_out || (max_rows > 0 && !h_rows)) {
        trl_set_error("no cascade state");
        return TRL_ERR_STATE;
    }
    static_assert(sizeof(Cand) == 40, "trl_debug_level_cands row layout");
    TRL_HIP(hipDeviceSynchronize());
    int32_t k = 0;
    TRL_HIP(hipMemcpy(&k, c->cb.lvl_cnt + (size_t)frame * c->cb.L + level, 4, hipMemcpyDeviceToHost));
    if (k > c->cb.lay.capl[level]) k = c->cb.lay.capl[level];
    *n_out = k;
    const int m = k < max_rows ? k : max_rows;
    if (m > 0) TRL_HIP(hipMemcpy(h_rows, c->cb.lvl_rec + (size_t)frame * c->cb.lay.S + c->cb.lay.rec0[level], (size_t)m * sizeof(Cand), hipMemcpyDeviceToHost));
    return TRL_OK;
}

// The per-level NMS picks of one (frame, level) of the last call: indices into that level's candidate records (the rows
// trl_debug_level_cands returns, in the same append order), in pick order (descending score)
int trl_debug_level_keep(trl_ctx* c, int frame, int level, int32_t* h_idx, int max_rows, int* n_out) {
    TRL_CHECK(check_idle(c));
    if (!c->cb.lvl_keep_cnt || frame < 0 || frame >= c->cb.n || level < 0 || level >= c->cb.L || !n_out || (max_rows > 0 && !h_idx)) {
        trl_set_error("no cascade state");
        return TRL_ERR_STATE;
    }
    TRL_HIP(hipDeviceSynchronize());
    int32_t k = 0;
    TRL_HIP(hipMemcpy(&k, c->cb.lvl_keep_cnt + (size_t)frame * c->cb.L + level, 4, hipMemcpyDeviceToHost));
    *n_out = k;
    const int m = k < max_rows ? k : max_rows;
    if (m > 0) TRL_HIP(hipMemcpy(h_idx, c->cb.lvl_keep_idx + (size_t)frame * c->cb.lay.S + c->cb.lay.rec0[level], (size_t)m * 4, hipMemcpyDeviceToHost));
    return TRL_OK;
}

int trl_debug_pyramid_level(trl_ctx* c, const uint8_t* d_frame, int H, int W, int level, float* d_out, int* h, int* w, void* stream) {
    TRL_CHECK(check_call(c, d_frame, 1, H, W));
    hipStream_t s = (hipStream_t)stream;
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, trl_pnet_fused_bytes(c, 1, H, W) + (1u << 20)));
    TRL_CHECK(trl_pyramid_export(c, d_frame, H, W, level, d_out, h, w, s));
    TRL_HIP(hipStreamSynchronize(s));
    return TRL_OK;
}

int trl_debug_pnet_level(trl_ctx* c, const uint8_t* d_frame, int H, int W, int level, float* d_prob, float* d_reg, int* oh, int* ow,
                         void* stream) {
    TRL_CHECK(check_call(c, d_frame, 1, H, W));
    hipStream_t s = (hipStream_t)stream;
    const int L = trl_compute_levels(c, H, W);
    if (level < 0 || level >= L) { trl_set_error("level %d out of range (%d levels)", level, L); return TRL_ERR_INVALID; }
    const LevelGeom& g = c->lv[level];
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, trl_pnet_generic_bytes(1, g.h, g.w) + (size_t)g.h * g.w * 12 + (size_t)g.oh * g.ow * 24 + (4u << 20)));
    float* lvl = (float*)c->scratch.alloc((size_t)g.h * g.w * 12);
    float* heads = (float*)c->scratch.alloc((size_t)g.oh * g.ow * 24);
    TRL_CHECK(trl_launch_area_level(d_frame, 1, H, W, g.h, g.w, lvl, s));
    TRL_CHECK(trl_run_pnet_generic(c, lvl, 1, g.h, g.w, heads, s));
    TRL_CHECK(trl_launch_heads_to_maps(heads, g.oh * g.ow, d_prob, d_reg, s));
    *oh = g.oh; *ow = g.ow;
    TRL_HIP(hipStreamSynchronize(s));
    return TRL_OK;
}

int trl_debug_rnet(trl_ctx* c, const float* d_crops, int n, float* d_out, void* stream) {
    if (!c || !c->have_weights || !d_crops || !d_out || n <= 0) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    TRL_CHECK(check_idle(c));
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, (size_t)n * 100 * 1024 + (4u << 20)));
    return trl_run_rnet(c, d_crops, n, d_out, (hipStream_t)stream);
}
// test hook: the PRODUCTION stage-2 / stage-3 network path on caller-chosen boxes of frame 0 -- the fused front kernel
// (k_mtcnn_front: pad(), crop, area resample, conv1, PReLU, pool) followed by the layer tail -- so its crop paths (small boxes,
// big boxes, boxes clipped by the frame) are checked directly, not only through cascade records.  h_boxes: nb rows of
// x1,y1,x2,y2 (host); d_out: [nb][6] (net = 24) or [nb][16] (net = 48), device.
int trl_debug_front_net(trl_ctx* c, const uint8_t* d_frame, int H, int W, const float* h_boxes, int nb, int net, float* d_out, void* stream) {
    TRL_CHECK(check_call(c, d_frame, 1, H, W));
    if (!h_boxes || !d_out || nb <= 0 || nb > (1 << 20) || (net != 24 && net != 48)) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    TRL_HIP(hipSetDevice(c->cfg.device));
    const int capF = nb;
    Arena& A = c->arena;
    TRL_CHECK(trl_ensure(c, A, (size_t)capF * 32 + (1u << 20)));
    A.reset();
    CascadeBufs& B = c->cb;
    B = CascadeBufs();
    B.n = 1; B.H = H; B.W = W;
    B.cbox = (int32_t*)A.alloc((size_t)capF * 32);
    int32_t* total = (int32_t*)A.alloc(64);
    std::vector<int32_t> hb((size_t)nb * 8, 0);        // the records k_build_map writes: frame 0, pad()'s clamped crop window
    for (int i = 0; i < nb; i++) {
        const float* b = h_boxes + 4 * i;
        const int bx = (int)truncf(b[0]), by = (int)truncf(b[1]), bex = (int)truncf(b[2]), bey = (int)truncf(b[3]);
        const int x = bx < 1 ? 1 : bx, y = by < 1 ? 1 : by, ex = bex > W ? W : bex, ey = bey > H ? H : bey;
        hb[8 * i + 1] = y - 1; hb[8 * i + 2] = x - 1; hb[8 * i + 3] = ey - (y - 1); hb[8 * i + 4] = ex - (x - 1);
    }
    TRL_HIP(hipMemcpyAsync(B.cbox, hb.data(), hb.size() * 4, hipMemcpyHostToDevice, s));
    TRL_HIP(hipMemcpyAsync(total, &nb, 4, hipMemcpyHostToDevice, s));
    TRL_HIP(hipStreamSynchronize(s));                      // the host vectors go out of scope
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, (size_t)nb * 700 * 1024 + (4u << 20)));
    if (net == 24) {
        float* pool1 = (float*)c->scratch.alloc((size_t)nb * 11 * 11 * 28 * 4);
        TRL_CHECK(trl_launch_rnet_front(c, d_frame, H, W, total, 0, nb, pool1, s));
        TRL_CHECK(trl_run_rnet_tail(c, pool1, nb, d_out, s, total, 0));
    } else {
        float* pool1 = (float*)c->scratch.alloc((size_t)nb * 23 * 23 * 32 * 4);
        TRL_CHECK(trl_launch_onet_front(c, d_frame, H, W, total, 0, nb, pool1, s));
        TRL_CHECK(trl_run_onet_tail(c, pool1, nb, d_out, s, total, 0));
    }
    TRL_HIP(hipStreamSynchronize(s));
    B = CascadeBufs();                                     // no cascade state to inspect after this hook
    return TRL_OK;
}

int trl_debug_onet(trl_ctx* c, const float* d_crops, int n, float* d_out, void* stream) {
    if (!c || !c->have_weights || !d_crops || !d_out || n <= 0) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    TRL_CHECK(check_idle(c));
    c->scratch.reset();
    TRL_CHECK(trl_ensure(c, c->scratch, (size_t)n * 640 * 1024 + (4u << 20)));
    return trl_run_onet(c, d_crops, n, d_out, (hipStream_t)stream);
}
int trl_debug_crop_resize(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, const int32_t* d_rect, const uint8_t* d_valid,
                          float* d_faces, void* stream) {
    if (!c || !d_frames || !d_rect || !d_valid || !d_faces || n <= 0) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    return trl_launch_crop_resize80(d_frames, n, H, W, d_rect, d_valid, d_faces, (hipStream_t)stream);
}
// embedding mode 3's crop alone: d_pts [n][10] (x0..x4, y0..y4 per frame) -> f32 [n][S][S][3]
int trl_debug_crop_aligned(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, const float* d_pts, const uint8_t* d_valid, int S,
                           int rgb, float* d_faces, void* stream) {
    if (!c || !d_frames || !d_pts || !d_valid || !d_faces || n <= 0 || H < 1 || W < 1 || S < 1) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    return trl_launch_crop_aligned(d_frames, n, H, W, d_pts, d_valid, S, rgb != 0, d_faces, (hipStream_t)stream);
}
// test hook: set the optimistic R-/O-Net batch capacities (candidates per frame) the next call starts from, and read back how
// many attempts the last call needed (> 1: a capacity was too small and the call was re-run with a larger one)
// test hook: the LDS tiers of the sort + NMS kernels (candidates per list; 0 keeps a value).  Lists longer than `full` take the
// spill tier (global memory): lowering it makes small test inputs exercise that tier.  Results never depend on the tiers.
int trl_debug_nms_tiers(trl_ctx* c, int small_tier, int full_tier) {
    TRL_CHECK(check_idle(c));
    if ((small_tier && (small_tier < 16 || small_tier > 3072 || (small_tier & 3))) || (full_tier && (full_tier < 16 || full_tier > 3072 || (full_tier & 3)))) {
        trl_set_error("tiers must be multiples of 4 in [16, 3072]");
        return TRL_ERR_INVALID;
    }
    if (small_tier) c->nms_small = small_tier;
    if (full_tier) c->nms_full = full_tier;
    return TRL_OK;
}
// What the candidate lists of the last call looked like: h_out8 = {attempts, lists that took the spill tier, spill bytes used,
// spill bytes available, per-frame list capacity, record slots per frame (all levels), largest per-level count, largest per-frame
// stage-1 total}
int trl_debug_list_stats(trl_ctx* c, long long* h_out8) {
    if (!c || !h_out8) { trl_set_error("null argument"); return TRL_ERR_INVALID; }
    const int32_t* f = c->h_pinned + 4;
    unsigned long long used = 0;
    memcpy(&used, f + 8, 8);
    int mx = 0;
    for (int l = 0; l < 32; l++) if (f[16 + l] > mx) mx = f[16 + l];
    h_out8[0] = c->last_attempts; h_out8[1] = f[10]; h_out8[2] = (long long)used; h_out8[3] = (long long)c->cb.spill_cap;
    h_out8[4] = c->cb.capF; h_out8[5] = c->cb.lay.S; h_out8[6] = mx; h_out8[7] = f[6];
    return TRL_OK;
}

// test hooks of the shipped library (it reads no environment variable): "rnet_chunk" / "onet_chunk" = candidates per R-/O-Net
// launch set of this context (>= 16), "no_fnconv" = process-wide: FaceNet's small maps through the generic conv kernels
int trl_debug_option(trl_ctx* c, const char* key, int value) {
    if (!key) { trl_set_error("null key"); return TRL_ERR_INVALID; }
    if (!strcmp(key, "no_fnconv")) { g_trl_no_fnconv = value ? 1 : 0; return TRL_OK; }
    if (!strcmp(key, "pnet_gate")) { g_trl_pnet_gate = value ? 1 : 0; return TRL_OK; }
    TRL_CHECK(check_idle(c));
    if (!strcmp(key, "rnet_chunk") && value >= 16) { c->rnet_chunk = value; return TRL_OK; }
    if (!strcmp(key, "onet_chunk") && value >= 16) { c->onet_chunk = value; return TRL_OK; }
    trl_set_error("unknown option '%s' (or value %d out of range)", key, value);
    return TRL_ERR_INVALID;
}

int trl_debug_batch_capacity(trl_ctx* c, float t2_per_frame, float t3_per_frame, int* last_attempts) {
    if (!c) { trl_set_error("null context"); return TRL_ERR_INVALID; }
    if (t2_per_frame > 0.f) c->t2_per_frame = t2_per_frame;
    if (t3_per_frame > 0.f) c->t3_per_frame = t3_per_frame;
    if (last_attempts) *last_attempts = c->last_attempts;
    return TRL_OK;
}

// R-Net / O-Net candidate totals of the last call (what the front kernels and the tails processed)
int trl_debug_stage_totals(trl_ctx* c, int32_t* h_out2) {
    if (!c || !h_out2) { trl_set_error("null argument"); return TRL_ERR_INVALID; }
    h_out2[0] = c->h_pinned[4 + 4]; h_out2[1] = c->h_pinned[4 + 5];
    return TRL_OK;
}

// test / tuning hook: consecutive tiles a workgroup of the fused PNet launch takes per cursor fetch (0 = automatic).  Runs > 1
// let a tile reuse its left neighbour's halo columns (the carry path); results are identical for every value.
int trl_debug_pnet_run(trl_ctx* c, int run) {
    if (!c || run < 0 || run > 64) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    c->pnet_run = run;
    return TRL_OK;
}

int trl_debug_pnet_span(trl_ctx* c, int reset, double* ms_sum, int32_t* launches) {
    if (!c || !ms_sum || !launches) { trl_set_error("null argument"); return TRL_ERR_INVALID; }
    *ms_sum = 0.0; *launches = 0;
    if (!c->pnet_clk) return TRL_OK;
    TRL_CHECK(check_idle(c));
    TRL_HIP(hipSetDevice(c->cfg.device));
    TRL_HIP(hipDeviceSynchronize());                 // the stamps are written by kernels on the callers' streams
    unsigned long long t[2] = {0, 0};
    int khz = 0;
    TRL_HIP(hipMemcpy(t, c->pnet_clk + 36, sizeof t, hipMemcpyDeviceToHost));
    TRL_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->cfg.device));
    if (khz > 0) *ms_sum = (double)t[0] / (double)khz;
    *launches = (int32_t)t[1];
    if (reset) TRL_HIP(hipMemset(c->pnet_clk + 36, 0, 16));
    return TRL_OK;
}

int trl_debug_pnet_kernel_ms(trl_ctx* c, float* ms) {
    if (!c || !ms) { trl_set_error("null argument"); return TRL_ERR_INVALID; }
    *ms = c->pnet_kernel_ms;
    return TRL_OK;
}

int trl_debug_timings(trl_ctx* c, float* out4) {
    if (!c || !out4) return TRL_ERR_INVALID;
    out4[0] = c->last_ms[0]; out4[1] = c->last_ms[1]; out4[2] = c->last_ms[2]; out4[3] = c->last_ms[3];
    return TRL_OK;
}

}  // extern "C"
