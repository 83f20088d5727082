// GloVe / word2vec embedding bag, the device body (glove.hip's kernel and the rider workgroups of xpanel.hip's pre-split launch share it).
// WordEmbedding.forward (fumi/models/common.py:23-41) + the per-class first-support-row pick of fumi/models/fumi.py:207-210.
#pragma once
#include "common.h"

// one 512-thread workgroup per output row: the row's L tokens are split over the 8 waves (8x the gathers in flight per
// row; the whole launch is only B*N = 160 rows at the bench shape, so a wave per row left most CUs idle)
constexpr int GW = 8;        // waves per output row
constexpr int GU = 16;       // row gathers in flight per wave
// one output row `r` by the 512 threads of the calling workgroup; part: [GW][Ep] partial sums / maxima, then [GW] counts (LDS)
template <bool VEC>
__device__ __forceinline__ void glove_bag_row(const GloveArgs& ga, int r, float* part) {
    const int64_t* __restrict__ tok = ga.tok; const float* __restrict__ table = ga.table; float* __restrict__ out = ga.out;
    const int L = ga.L, V = ga.V, E = ga.E, mode = ga.mode, N = ga.N, S = ga.S; const int64_t pad_id = ga.pad_id;
    const int64_t* __restrict__ y_s = ga.y_s; int* status = ga.status;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int W = VEC ? 4 : 1;
    const int nchunk = (E / W + 63) / 64;        // chunks of 64 lanes x W floats
    const int Ep = nchunk * 64 * W;
    float* cnts = part + GW * Ep;
    long src_row = r;
    if (y_s) {                                   // select form: output row r = (episode, class); source = first support row of the class
        const int b = r / N, c = r - b * N;
        const int64_t* ys = y_s + (long)b * S;
        int first = S;
        for (int s0 = 0; s0 < S && first == S; s0 += 64) {
            const int s_ = s0 + lane;
            const unsigned long long m = __ballot(s_ < S && ys[s_] == c);
            if (m) first = s0 + __ffsll((long long)m) - 1;
        }
        if (first == S) {                        // the reference raises IndexError here (fumi.py:209)
            if (threadIdx.x == 0) atomicOr(status, FUMI_ST_CLASS_MISSING);
            for (int j = threadIdx.x; j < E; j += blockDim.x) out[(long)r * E + j] = __builtin_nanf("");
            return;
        }
        src_row = (long)b * S + first;
    }
    const int64_t* t = tok + src_row * L;
    const int lw = (L + GW - 1) / GW;            // tokens per wave
    const int lbeg = wave * lw, lend = min(L, lbeg + lw);
    for (int c = 0; c < nchunk; ++c) {
        const int j = (c * 64 + lane) * W;
        const int jc = j < E ? j : 0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (mode == 1) acc = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int cnt = 0;
        for (int l0 = lbeg; l0 < lend; l0 += 64) {
            const int nl = min(64, lend - l0);
            long my = lane < nl ? t[l0 + lane] : pad_id;
            cnt += __popcll(__ballot(lane < nl && my != pad_id));
            if (my < 0 || my >= V) { if (lane < nl) atomicOr(status, FUMI_ST_LABEL_RANGE); my = 0; }
            const int mylo = (int)my;
            for (int u0 = 0; u0 < nl; u0 += GU) {
                f32x4 v[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int id = __shfl(mylo, min(u0 + u, nl - 1), 64);
                    const float* row = table + (long)id * E + jc;
                    if (VEC) v[u] = *(const f32x4*)row; else { v[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; v[u][0] = row[0]; }
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    if (u0 + u < nl) {
                        if (mode == 0) acc += v[u];
                        else { acc[0] = fmaxf(acc[0], v[u][0]); acc[1] = fmaxf(acc[1], v[u][1]); acc[2] = fmaxf(acc[2], v[u][2]); acc[3] = fmaxf(acc[3], v[u][3]); }
                    }
                }
            }
        }
        float* pp = part + wave * Ep + (c * 64 + lane) * W;
        if (VEC) *(f32x4*)pp = acc; else pp[0] = acc[0];
        if (lane == 0 && c == 0) cnts[wave] = (float)cnt;
    }
    __syncthreads();
    float dn = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < GW; ++w_) dn += cnts[w_];
    for (int j = threadIdx.x; j < E; j += blockDim.x) {
        float a = part[j];
#pragma unroll
        for (int w_ = 1; w_ < GW; ++w_) a = mode == 0 ? a + part[w_ * Ep + j] : fmaxf(a, part[w_ * Ep + j]);
        out[(long)r * E + j] = mode == 0 ? a / dn : a;
    }
}

