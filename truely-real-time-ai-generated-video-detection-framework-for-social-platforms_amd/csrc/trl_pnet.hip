// trl_pnet.hip -- fused PNet kernel (placeholder until the fused kernel lands: reports unavailable)
#include "trl_ctx.h"
int trl_pnet_prepare(trl_ctx* c) { (void)c; return TRL_OK; }
int trl_pnet_fused_level(trl_ctx* c, const uint8_t* d_frames, int n, int H, int W, int l, hipStream_t s) {
    (void)c; (void)d_frames; (void)n; (void)H; (void)W; (void)l; (void)s;
    trl_set_error("fused PNet kernel not built; use pnet_mode=1");
    return TRL_ERR_STATE;
}
