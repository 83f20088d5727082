// GPU-resident episode sampler: replaces, for precomputed image embeddings, what
// fumi/dataset/data.py:294-581 + torchmeta's CombinationMetaDataset / ClassSplitter / BatchMetaDataLoader do on the host
// (the reference re-reads all images of a class from HDF5 for every class access, data.py:533-549).
// The whole embedding table lives in HBM (iNat-Anim: ~195k x 2048 fp32 = 1.6 GB of the 288 GB); a meta-batch is
//   sample_episodes:  B x N distinct classes, K + Q distinct images per class (Floyd's subset algorithm + a Fisher-Yates
//                     shuffle driven by a counter-based hash: reproducible from (seed, step), restated bit for bit in
//                     oracle/sampler_ref.py), class-major layout like torchmeta's ConcatTask (labels 0..N-1 in blocks)
//   gather_rows:      out[i,:] = table[idx[i],:]   (HBM-bound row copy; also used for the per-class text rows)
#include "common.h"

namespace {

constexpr int SMAXN = 64;            // classes per episode
constexpr int SMAXM = 256;           // samples per class (K + Q)

__device__ __forceinline__ unsigned smix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// uniform integer in [0, n) from counter (a, b, c); n <= 2^31
__device__ __forceinline__ unsigned srand_below(unsigned key, unsigned a, unsigned b, unsigned c, unsigned n) {
    const unsigned r = smix(smix(smix(key ^ (a * 0x9E3779B9U)) ^ (b * 0x85EBCA6BU)) ^ (c * 0xC2B2AE35U));
    return (unsigned)(((unsigned long long)r * n) >> 32);
}

// Floyd: m distinct values of [0, n) into sel[0..m), then Fisher-Yates so that every ordering is equally likely
__device__ void sample_distinct(unsigned key, unsigned a, unsigned b, int n, int m, int* sel) {
    int cnt = 0;
    for (int j = n - m; j < n; ++j) {
        const int t = (int)srand_below(key, a, b, (unsigned)(2 * j), (unsigned)(j + 1));
        bool seen = false;
        for (int i = 0; i < cnt; ++i) seen |= sel[i] == t;
        sel[cnt++] = seen ? j : t;
    }
    for (int i = m - 1; i > 0; --i) {
        const int t = (int)srand_below(key, a, b, (unsigned)(2 * i + 1), (unsigned)(i + 1));
        const int tmp = sel[i]; sel[i] = sel[t]; sel[t] = tmp;
    }
}

__global__ __launch_bounds__(64) void sample_episodes_kernel(unsigned key, int B, int N, int K, int Q, int C,
                                                             const int64_t* __restrict__ class_ptr,
                                                             const int64_t* __restrict__ class_items,
                                                             int64_t* __restrict__ cls_out, int64_t* __restrict__ img_s,
                                                             int64_t* __restrict__ img_q, int* status) {
    __shared__ int s_cls[SMAXN];
    extern __shared__ int s_rows[];                       // [N][K+Q]
    const int b = blockIdx.x, tid = threadIdx.x, m = K + Q;
    if (tid == 0) sample_distinct(key, (unsigned)b, 0xFFFFu, C, N, s_cls);
    __syncthreads();
    if (tid < N) {
        const int c = s_cls[tid];
        const long p0 = class_ptr[c];
        const int n_c = (int)(class_ptr[c + 1] - p0);
        int* sel = s_rows + tid * m;
        if (n_c < m) {                                    // torchmeta's ClassSplitter raises here; flag it and wrap around
            atomicOr(status, FUMI_ST_CLASS_MISSING);
            for (int i = 0; i < m; ++i) sel[i] = n_c > 0 ? i % n_c : 0;
        } else {
            sample_distinct(key, (unsigned)b, (unsigned)tid, n_c, m, sel);
        }
        cls_out[(long)b * N + tid] = c;
        for (int k = 0; k < K; ++k) img_s[((long)b * N + tid) * K + k] = n_c > 0 ? class_items[p0 + sel[k]] : 0;
        for (int q = 0; q < Q; ++q) img_q[((long)b * N + tid) * Q + q] = n_c > 0 ? class_items[p0 + sel[K + q]] : 0;
    }
}

// torchmeta's task semantics on top of the same hash stream:
//   * Categorical(N): the N class slots of a task get a random permutation of the labels 0..N-1 (labels_out [B,N]);
//   * ClassSplitter(shuffle=True) seeds its per-class permutation with hash(task) + seed, so the support / query split of a
//     given class tuple is the same every time the tuple is drawn: with fixed_split the item key is derived from
//     (seed, the tuple's class ids in slot order) instead of (seed, step, episode).
__global__ __launch_bounds__(64) void sample_episodes_tm_kernel(unsigned key, unsigned seed_key, int fixed_split, int B, int N, int K,
                                                                int Q, int C, const int64_t* __restrict__ class_ptr,
                                                                const int64_t* __restrict__ class_items,
                                                                int64_t* __restrict__ cls_out, int64_t* __restrict__ lab_out,
                                                                int64_t* __restrict__ img_s, int64_t* __restrict__ img_q, int* status) {
    __shared__ int s_cls[SMAXN];
    __shared__ int s_lab[SMAXN];
    __shared__ unsigned s_tkey;
    extern __shared__ int s_rows[];                       // [N][K+Q]
    const int b = blockIdx.x, tid = threadIdx.x, m = K + Q;
    if (tid == 0) {
        sample_distinct(key, (unsigned)b, 0xFFFFu, C, N, s_cls);
        sample_distinct(key, (unsigned)b, 0xFFFEu, N, N, s_lab);            // a uniformly random permutation of 0..N-1
        unsigned tk = seed_key;
        for (int n = 0; n < N; ++n) tk = smix(tk ^ ((unsigned)s_cls[n] * 0x9E3779B9U + (unsigned)n));
        s_tkey = tk;
    }
    __syncthreads();
    if (tid < N) {
        const int c = s_cls[tid];
        const long p0 = class_ptr[c];
        const int n_c = (int)(class_ptr[c + 1] - p0);
        int* sel = s_rows + tid * m;
        if (n_c < m) {
            atomicOr(status, FUMI_ST_CLASS_MISSING);
            for (int i = 0; i < m; ++i) sel[i] = n_c > 0 ? i % n_c : 0;
        } else if (fixed_split) {
            sample_distinct(s_tkey, 0u, (unsigned)tid, n_c, m, sel);
        } else {
            sample_distinct(key, (unsigned)b, (unsigned)tid, n_c, m, sel);
        }
        cls_out[(long)b * N + tid] = c;
        lab_out[(long)b * N + tid] = s_lab[tid];
        for (int k = 0; k < K; ++k) img_s[((long)b * N + tid) * K + k] = n_c > 0 ? class_items[p0 + sel[k]] : 0;
        for (int q = 0; q < Q; ++q) img_q[((long)b * N + tid) * Q + q] = n_c > 0 ? class_items[p0 + sel[K + q]] : 0;
    }
}

// out[i, :] = table[idx[i], :]; rows of `row_f4` float4 (VEC) or `row_f` floats.  One wave per row, all of a row's loads in
// flight before the first store (8 KB rows: 8 x 16 bytes per lane).
template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, long n_rows, int row_f,
                                                          const int64_t* __restrict__ idx, long n_idx,
                                                          float* __restrict__ out, int* status) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long i = (long)blockIdx.x * 4 + wave; i < n_idx; i += (long)gridDim.x * 4) {
        long r = idx[i];
        if (r < 0 || r >= n_rows) { if (lane == 0) atomicOr(status, FUMI_ST_LABEL_RANGE); r = 0; }
        const float* src = table + r * row_f;
        float* dst = out + i * row_f;
        if (VEC) {
            const int n4 = row_f >> 2;
            for (int c0 = 0; c0 < n4; c0 += 64 * 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int c = c0 + u * 64 + lane; v[u] = *(const f32x4*)(src + 4 * (c < n4 ? c : 0)); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int c = c0 + u * 64 + lane; if (c < n4) *(f32x4*)(dst + 4 * c) = v[u]; }
            }
        } else {
            for (int c = lane; c < row_f; c += 64) dst[c] = src[c];
        }
    }
}

__global__ void index_range_check_kernel(const int64_t* __restrict__ idx, long n, long n_rows, int* status) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (idx[i] < 0 || idx[i] >= n_rows) atomicOr(status, FUMI_ST_LABEL_RANGE);
}

}  // namespace

int launch_index_range_check(hipStream_t st, const int64_t* idx, long n, long n_rows, int* status) {
    if (n < 1) return FUMI_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(index_range_check_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, n, n_rows, status);
    LAUNCH_CHECK();
    return FUMI_OK;
}

extern "C" int fumi_hip_sample_episodes(fumi_ws_t* ws, fumi_stream_t stream, uint64_t seed, uint64_t step, int B, int N, int K,
        int Q, int C, const int64_t* class_ptr, const int64_t* class_items, int64_t* classes, int64_t* items_s,
        int64_t* items_q) {
    if (!ws || !class_ptr || !class_items || !classes || !items_s || !items_q) return FUMI_EINVAL;
    if (B < 1 || N < 1 || K < 1 || Q < 0 || C < N) return FUMI_EINVAL;
    if (N > SMAXN || K + Q > SMAXM) return FUMI_ENOTSUP;
    HIP_TRY(hipSetDevice(ws->device));
    // one 32-bit key per (seed, step): the same mixing is restated in oracle/sampler_ref.py
    auto mix = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
    unsigned key = mix((unsigned)(seed & 0xffffffffULL));
    key = mix(key ^ (unsigned)(seed >> 32));
    key = mix(key ^ (unsigned)(step & 0xffffffffULL));
    key = mix(key ^ (unsigned)(step >> 32));
    hipLaunchKernelGGL(sample_episodes_kernel, dim3(B), dim3(64), (size_t)N * (K + Q) * sizeof(int), (hipStream_t)stream, key, B, N,
                       K, Q, C, class_ptr, class_items, classes, items_s, items_q, ws->status);
    LAUNCH_CHECK();
    return FUMI_OK;
}

extern "C" int fumi_hip_sample_episodes_tm(fumi_ws_t* ws, fumi_stream_t stream, uint64_t seed, uint64_t step, int B, int N, int K,
        int Q, int C, const int64_t* class_ptr, const int64_t* class_items, int fixed_split, int64_t* classes, int64_t* labels,
        int64_t* items_s, int64_t* items_q) {
    if (!ws || !class_ptr || !class_items || !classes || !labels || !items_s || !items_q) return FUMI_EINVAL;
    if (B < 1 || N < 1 || K < 1 || Q < 0 || C < N) return FUMI_EINVAL;
    if (N > SMAXN || K + Q > SMAXM) return FUMI_ENOTSUP;
    HIP_TRY(hipSetDevice(ws->device));
    auto mix = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
    unsigned skey = mix((unsigned)(seed & 0xffffffffULL));
    skey = mix(skey ^ (unsigned)(seed >> 32));
    unsigned key = mix(skey ^ (unsigned)(step & 0xffffffffULL));
    key = mix(key ^ (unsigned)(step >> 32));
    hipLaunchKernelGGL(sample_episodes_tm_kernel, dim3(B), dim3(64), (size_t)N * (K + Q) * sizeof(int), (hipStream_t)stream, key,
                       mix(skey ^ 0x5bd1e995U), fixed_split ? 1 : 0, B, N, K, Q, C, class_ptr, class_items, classes, labels, items_s,
                       items_q, ws->status);
    LAUNCH_CHECK();
    return FUMI_OK;
}

extern "C" int fumi_hip_gather_rows(fumi_ws_t* ws, fumi_stream_t stream, const void* table, int64_t n_rows, int64_t row_bytes,
        const int64_t* idx, int64_t n_idx, void* out) {
    if (!ws || !table || !idx || !out || n_rows < 1 || row_bytes < 4 || (row_bytes & 3) || n_idx < 0) return FUMI_EINVAL;
    if (n_idx == 0) return FUMI_OK;
    if (row_bytes / 4 > 0x7fffffff) return FUMI_ENOTSUP;
    HIP_TRY(hipSetDevice(ws->device));
    const int row_f = (int)(row_bytes / 4);
    const bool vec = (row_f % 4 == 0) && ((uintptr_t)table % 16 == 0) && ((uintptr_t)out % 16 == 0);
    long blocks = (n_idx + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (vec) hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                (const float*)table, (long)n_rows, row_f, idx, (long)n_idx, (float*)out, ws->status);
    else hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                            (const float*)table, (long)n_rows, row_f, idx, (long)n_idx, (float*)out, ws->status);
    LAUNCH_CHECK();
    return FUMI_OK;
}

// ---- event-free scalar read-back --------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void publish_scalars_kernel(const float* __restrict__ src, int n, float* dst, unsigned long long seq) {
    const int i = threadIdx.x;
    // write-through (system-scope) stores of the values, all lanes in one instruction; their completion is awaited before
    // lane 0 raises the flag.  No release fence: that would write back every dirty L2 line the optimizer step left behind
    // (~2 us) although nothing but these 64 bytes is read by the host.
    if (i < n) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (i == 0) __hip_atomic_store((unsigned long long*)(dst + 14), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" int fumi_hip_publish_scalars(fumi_ws_t* ws, fumi_stream_t stream, const float* src, int n, void* host_pinned,
        uint64_t seq) {
    if (!ws || !src || !host_pinned || n < 1 || n > 14 || ((uintptr_t)host_pinned & 7)) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    void* dptr = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dptr, host_pinned, 0));          // fails for memory that is not page-locked + mapped
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, n, (float*)dptr,
                       (unsigned long long)seq);
    LAUNCH_CHECK();
    return FUMI_OK;
}

// Deferred form: the request is kept in the workspace and rides on the next fumi_hip_adam_step launch of this workspace (the
// optimizer step that follows a training meta-step); fumi_hip_publish_flush launches it on its own if it is still pending.
extern "C" int fumi_hip_publish_scalars_deferred(fumi_ws_t* ws, const float* src, int n, void* host_pinned, uint64_t seq) {
    if (!ws || !src || !host_pinned || n < 1 || n > 14 || ((uintptr_t)host_pinned & 7) || ws->pub_dst) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    void* dptr = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dptr, host_pinned, 0));
    ws->pub_src = src; ws->pub_n = n; ws->pub_dst = (float*)dptr; ws->pub_seq = seq;
    return FUMI_OK;
}

extern "C" int fumi_hip_publish_flush(fumi_ws_t* ws, fumi_stream_t stream) {
    if (!ws) return FUMI_EINVAL;
    if (!ws->pub_dst) return FUMI_OK;
    HIP_TRY(hipSetDevice(ws->device));
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws->pub_src, ws->pub_n, ws->pub_dst,
                       ws->pub_seq);
    ws->pub_dst = nullptr; ws->pub_src = nullptr;
    LAUNCH_CHECK();
    return FUMI_OK;
}

