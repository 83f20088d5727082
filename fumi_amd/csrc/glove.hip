// GloVe / word2vec embedding bag: frozen table gather + pooling over the token axis.
// Replaces WordEmbedding.forward (fumi/models/common.py:23-41):
//   mean: sum over ALL L positions of table[tok] divided by the number of non-PAD tokens (common.py:34-37)
//   max : max over ALL L positions, PAD rows included (common.py:38-39)
// and, in its "select" form, also the per-class first-support-row pick of fumi/models/fumi.py:207-210 (FuMI only needs
// the text of N class rows per episode, not of all S support rows: 5x fewer table rows gathered).
//
// HBM/L2-bound gather: one workgroup (8 waves) per output row.  A wave's token ids are read 64 at a time with one coalesced load and
// handed out by cross-lane shuffles; the table rows of 16 tokens are requested back to back (16 independent 16-byte-per-
// lane gathers in flight) before any is accumulated -- a loop that reads a token id and then its row is two dependent
// memory round trips per token.
#include "common.h"

namespace {

}  // namespace
#include "glove_bag.h"
namespace {

template <bool VEC>
__global__ __launch_bounds__(512) void glove_bag_kernel(GloveArgs ga) {
    extern __shared__ __attribute__((aligned(16))) float part[];
    glove_bag_row<VEC>(ga, blockIdx.x, part);
}

// vec / lds of a request (FUMI_ENOTSUP when a row's partial sums do not fit)
int bag_geometry(const GloveArgs& ga, bool* vec, size_t* lds) {
    *vec = ga.E % 4 == 0 && ((uintptr_t)ga.table & 15) == 0 && ((uintptr_t)ga.out & 15) == 0;
    const int W = *vec ? 4 : 1;
    const int Ep = ((ga.E / W + 63) / 64) * 64 * W;
    *lds = (size_t)(GW * Ep + GW) * sizeof(float);
    return *lds > 64 * 1024 ? FUMI_ENOTSUP : FUMI_OK;
}
int launch_bag_args(hipStream_t st, const GloveArgs& ga) {
    bool vec; size_t lds;
    int rc = bag_geometry(ga, &vec, &lds);
    if (rc) return rc;
    dim3 grid(ga.R), block(64 * GW);
    if (vec) hipLaunchKernelGGL(glove_bag_kernel<true>, grid, block, lds, st, ga);
    else hipLaunchKernelGGL(glove_bag_kernel<false>, grid, block, lds, st, ga);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_bag(fumi_ws_t* ws, hipStream_t st, const int64_t* tok, int R, int L, int64_t pad_id, const float* table, int V,
               int E, int mode, float* out, const int64_t* y_s, int N, int S) {
    GloveArgs ga{tok, R, L, pad_id, table, V, E, mode, out, ws->status, y_s, N, S};
    return launch_bag_args(st, ga);
}

}  // namespace

extern "C" int fumi_hip_glove_bag(fumi_ws_t* ws, fumi_stream_t stream, const int64_t* tok, int R, int L, int64_t pad_id,
        const float* table, int V, int E, int mode, float* out) {
    if (!ws || !tok || !table || !out || R < 1 || L < 1 || V < 1 || E < 1 || mode < 0 || mode > 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_bag(ws, (hipStream_t)stream, tok, R, L, pad_id, table, V, E, mode, out, nullptr, 0, 0);
}

extern "C" int fumi_hip_glove_bag_select(fumi_ws_t* ws, fumi_stream_t stream, const int64_t* tok_s, const int64_t* y_s,
        int B, int N, int S, int L, int64_t pad_id, const float* table, int V, int E, int mode, float* out) {
    if (!ws || !tok_s || !y_s || !table || !out || B < 1 || N < 1 || S < 1 || L < 1 || V < 1 || E < 1 || mode < 0 || mode > 1)
        return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_bag(ws, (hipStream_t)stream, tok_s, B * N, L, pad_id, table, V, E, mode, out, y_s, N, S);
}

// Deferred form of fumi_hip_glove_bag_select: nothing is launched -- the request rides as extra workgroups of the FIRST launch of
// the next fumi_hip_fumi_step / _indexed of this workspace (the pre-split of the layer-0 column operands, xpanel.hip: both are short,
// latency-bound and independent), or is launched on its own at the start of that step when its shapes take another path.
// `out` [B, N, E] must be the cls_text handed to that step.
extern "C" int fumi_hip_glove_bag_select_deferred(fumi_ws_t* ws, const int64_t* tok_s, const int64_t* y_s,
        int B, int N, int S, int L, int64_t pad_id, const float* table, int V, int E, int mode, float* out) {
    if (!ws || !tok_s || !y_s || !table || !out || B < 1 || N < 1 || S < 1 || L < 1 || V < 1 || E < 1 || mode < 0 || mode > 1)
        return FUMI_EINVAL;
    if (!ws->glove) { ws->glove = new GlovePending(); ws->glove->on = 0; }
    GloveArgs ga{tok_s, B * N, L, pad_id, table, V, E, mode, out, ws->status, y_s, N, S};
    bool vec; size_t lds;
    int rc = bag_geometry(ga, &vec, &lds);
    if (rc) return rc;
    ws->glove->a = ga; ws->glove->vec = vec ? 1 : 0; ws->glove->lds = lds; ws->glove->on = 1;
    return FUMI_OK;
}

int glove_flush(fumi_ws* ws, hipStream_t st) {
    if (!ws || !ws->glove || !ws->glove->on) return FUMI_OK;
    ws->glove->on = 0;
    return launch_bag_args(st, ws->glove->a);
}

extern "C" int fumi_hip_glove_flush(fumi_ws_t* ws, fumi_stream_t stream) {
    if (!ws) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return glove_flush(ws, (hipStream_t)stream);
}
