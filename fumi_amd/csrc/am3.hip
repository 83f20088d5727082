// AM3 step -- placeholder until the prototype kernels land (the symbol is part of the ABI).
#include "common.h"
extern "C" int fumi_hip_am3_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int Dt, int Ht, int P, int lamda_fixed, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q, const float* text_s,
        const float* const* w, float* loss, int64_t* preds_q, float* lamda_s, float* correct, float* const* g_w) {
    return FUMI_ENOTSUP;
}
