// trl_ingest.hip -- SURVEY.md section 8(f) rank 1: device-side frame ingest.
//
// The reference decodes on the CPU (cv2.VideoCapture.read, server/model.py:43) and analyses every
// max(1,int(fps/7))-th frame (model.py:40,46).  Hardware / library decoders deliver NV12; this kernel turns
// the SAMPLED NV12 frames of a clip straight into the u8 BGR batch trl_detect_embed consumes, without a host
// round trip:   out[j] = BGR(nv12[j * step])   for j < ceil(n_in / step).
// Arithmetic: OpenCV's integer BT.601 limited-range conversion (cvtColor COLOR_YUV2BGR_NV12; RECALLED from
// modules/imgproc/src/color_yuv.simd.hpp: 20-bit fixed point, constants below).  What FFmpeg's swscale
// produces inside VideoCapture depends on its build, so the uint8 BGR tensor stays the parity contract.
// HBM-bound: 1.5 B/pixel read + 3 B/pixel written; one thread = 4x2 pixels (2 Y dwords, 1 UV dword in, 6 dwords out).
#include "trl_ctx.h"

namespace {

constexpr int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;

__device__ __forceinline__ unsigned sat8(int v) { return v < 0 ? 0u : (v > 255 ? 255u : (unsigned)v); }

struct __attribute__((packed, aligned(4))) u32x3 { unsigned x, y, z; };

// PLANAR: the chroma follows the luma as two planes (I420: all U, then all V -- what YUV4MPEG2 files and software decoders
// hold) instead of one interleaved plane (NV12: hardware decoders).  Same arithmetic, two 2-byte loads instead of one dword.
template <bool PLANAR>
__global__ __launch_bounds__(256) void k_nv12_to_bgr(const uint8_t* __restrict__ nv12, int n_out, int step, int H, int W,
                                                     uint8_t* __restrict__ bgr) {
    const int qw = W >> 2, qh = H >> 1;                       // 4x2-pixel quads per row / column
    const long long total = (long long)n_out * qh * qw;
    const size_t frame_in = (size_t)H * W * 3 / 2, frame_out = (size_t)H * W * 3;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int qx = (int)(idx % qw);
        const int qy = (int)((idx / qw) % qh);
        const int j = (int)(idx / ((long long)qw * qh));
        const uint8_t* src = nv12 + (size_t)j * step * frame_in;
        const unsigned y0 = *reinterpret_cast<const unsigned*>(src + (size_t)(2 * qy) * W + 4 * qx);
        const unsigned y1 = *reinterpret_cast<const unsigned*>(src + (size_t)(2 * qy + 1) * W + 4 * qx);
        unsigned uv;                                                                                              // U0 V0 U1 V1
        if (PLANAR) {
            const unsigned u2 = *reinterpret_cast<const unsigned short*>(src + (size_t)H * W + (size_t)qy * (W >> 1) + 2 * qx);
            const unsigned v2 = *reinterpret_cast<const unsigned short*>(src + (size_t)H * W + (size_t)(H >> 1) * (W >> 1) + (size_t)qy * (W >> 1) + 2 * qx);
            uv = (u2 & 0xFF) | ((v2 & 0xFF) << 8) | ((u2 >> 8) << 16) | ((v2 >> 8) << 24);
        } else {
            uv = *reinterpret_cast<const unsigned*>(src + (size_t)H * W + (size_t)qy * W + 4 * qx);
        }
        unsigned o[2][12];
#pragma unroll
        for (int p = 0; p < 2; p++) {                          // two chroma samples, each covers 2x2 pixels
            const int uu = (int)((uv >> (16 * p)) & 0xFF) - 128, vv = (int)((uv >> (16 * p + 8)) & 0xFF) - 128;
            const int ruv = (1 << (SHIFT - 1)) + CVR * vv;
            const int guv = (1 << (SHIFT - 1)) + CVG * vv + CUG * uu;
            const int buv = (1 << (SHIFT - 1)) + CUB * uu;
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int yv = (int)(((r ? y1 : y0) >> (8 * (2 * p + q))) & 0xFF);
                    const int yy = (yv - 16 > 0 ? yv - 16 : 0) * CY;
                    o[r][3 * (2 * p + q) + 0] = sat8((yy + buv) >> SHIFT);
                    o[r][3 * (2 * p + q) + 1] = sat8((yy + guv) >> SHIFT);
                    o[r][3 * (2 * p + q) + 2] = sat8((yy + ruv) >> SHIFT);
                }
        }
        uint8_t* dst = bgr + (size_t)j * frame_out + ((size_t)(2 * qy) * W + 4 * qx) * 3;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            u32x3 v;
            v.x = o[r][0] | (o[r][1] << 8) | (o[r][2] << 16) | (o[r][3] << 24);
            v.y = o[r][4] | (o[r][5] << 8) | (o[r][6] << 16) | (o[r][7] << 24);
            v.z = o[r][8] | (o[r][9] << 8) | (o[r][10] << 16) | (o[r][11] << 24);
            *reinterpret_cast<u32x3*>(dst + (size_t)r * W * 3) = v;
        }
    }
}

}  // namespace

static int ingest_420(trl_ctx* c, const uint8_t* d_in, int n_in, int H, int W, int step, bool planar, uint8_t* d_bgr, int* n_out, void* stream) {
    if (!c || !d_in || !d_bgr || !n_out || n_in < 0 || step < 1) { trl_set_error("bad argument"); return TRL_ERR_INVALID; }
    if ((W & 3) || (H & 1) || W < 4 || H < 2) { trl_set_error("4:2:0 ingest needs W %% 4 == 0 and even H (got %dx%d)", W, H); return TRL_ERR_INVALID; }
    if (((uintptr_t)d_in & 3) || ((uintptr_t)d_bgr & 3)) { trl_set_error("buffers must be 4-byte aligned"); return TRL_ERR_INVALID; }
    const int no = (n_in + step - 1) / step;                   // frames i with i % step == 0 (model.py:46)
    *n_out = no;
    if (no == 0) return TRL_OK;
    TRL_HIP(hipSetDevice(c->cfg.device));
    const long long total = (long long)no * (H >> 1) * (W >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (planar) k_nv12_to_bgr<true><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_in, no, step, H, W, d_bgr);
    else k_nv12_to_bgr<false><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_in, no, step, H, W, d_bgr);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
}

extern "C" int trl_ingest_nv12(trl_ctx* c, const uint8_t* d_nv12, int n_in, int H, int W, int step, uint8_t* d_bgr, int* n_out,
                               void* stream) {
    return ingest_420(c, d_nv12, n_in, H, W, step, false, d_bgr, n_out, stream);
}
extern "C" int trl_ingest_i420(trl_ctx* c, const uint8_t* d_i420, int n_in, int H, int W, int step, uint8_t* d_bgr, int* n_out,
                               void* stream) {
    return ingest_420(c, d_i420, n_in, H, W, step, true, d_bgr, n_out, stream);
}
