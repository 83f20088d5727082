// GloVe / word2vec embedding bag: frozen table gather + pooling over the token axis.
// Replaces WordEmbedding.forward (fumi/models/common.py:23-41):
//   mean: sum over ALL L positions of table[tok] divided by the number of non-PAD tokens (common.py:34-37)
//   max : max over ALL L positions, PAD rows included (common.py:38-39)
// HBM/L2-bound gather: one wave per output row, lanes stride the embedding dim (float4 when E % 4 == 0), the token id is
// wave-uniform so every table row is read as contiguous 16-byte-per-lane segments.
#include "common.h"

namespace {

template <bool VEC>
__global__ __launch_bounds__(256) void glove_bag_kernel(const int64_t* __restrict__ tok, int R, int L, int64_t pad_id,
                                                        const float* __restrict__ table, int V, int E, int mode,
                                                        float* __restrict__ out, int* status) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const int64_t* t = tok + (long)r * L;
    int cnt = 0;
    constexpr int W = VEC ? 4 : 1;
    const int nchunk = (E / W + 63) / 64;            // chunks of 64 lanes x W floats
    for (int c = 0; c < nchunk; ++c) {
        const int j = (c * 64 + lane) * W;
        const bool ok = j < E;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (mode == 1) acc = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        cnt = 0;
        for (int l = 0; l < L; ++l) {
            int64_t id = t[l];
            if (id != pad_id) ++cnt;
            if (id < 0 || id >= V) { if (lane == 0) atomicOr(status, FUMI_ST_LABEL_RANGE); id = 0; }
            if (ok) {
                const float* row = table + id * (long)E + j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (VEC) v = *(const f32x4*)row; else v[0] = row[0];
                if (mode == 0) acc += v;
                else { acc[0] = fmaxf(acc[0], v[0]); acc[1] = fmaxf(acc[1], v[1]); acc[2] = fmaxf(acc[2], v[2]); acc[3] = fmaxf(acc[3], v[3]); }
            }
        }
        if (ok) {
            float* o = out + (long)r * E + j;
            if (mode == 0) { const float d = (float)cnt; acc[0] /= d; acc[1] /= d; acc[2] /= d; acc[3] /= d; }
            if (VEC) *(f32x4*)o = acc; else o[0] = acc[0];
        }
    }
}

}  // namespace

extern "C" int fumi_hip_glove_bag(fumi_ws_t* ws, fumi_stream_t stream, const int64_t* tok, int R, int L, int64_t pad_id,
        const float* table, int V, int E, int mode, float* out) {
    if (!ws || !tok || !table || !out || R < 1 || L < 1 || V < 1 || E < 1 || mode < 0 || mode > 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = E % 4 == 0 && ((uintptr_t)table & 15) == 0 && ((uintptr_t)out & 15) == 0;
    dim3 grid((R + 3) / 4), block(256);
    if (vec) hipLaunchKernelGGL(glove_bag_kernel<true>, grid, block, 0, st, tok, R, L, pad_id, table, V, E, mode, out, ws->status);
    else hipLaunchKernelGGL(glove_bag_kernel<false>, grid, block, 0, st, tok, R, L, pad_id, table, V, E, mode, out, ws->status);
    LAUNCH_CHECK();
    return FUMI_OK;
}
