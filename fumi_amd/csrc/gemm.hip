// Dense fp32 GEMM family on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 256 FLOP/clk/CU).
//
// Used for every shared-weight contraction of the hot path (ATen mm/addmm + their backward in the reference:
// torchmeta MetaLinear.forward via fumi/models/fumi.py:215, autograd AddmmBackward/MmBackward via :165-176,192):
//   * A0|G  = X [W0;Xs]^T   (layer-0 pre-activations + support Gram matrix, see episode.hip)
//   * gW0   = Abar0^T X     (layer-0 meta-gradient, split over the contraction, slabs summed by reduce_slabs)
//   * the hypernetwork / AM3 encoder linears and their backward
//
// Tiling: BM x BN = 64 x 64 output tile per 256-thread workgroup (4 waves as 2 x 2, one 32x32 accumulator each),
// BK = 32 contraction slab, double-buffered in LDS, next slab prefetched to registers while the MFMAs run.
// LDS images (both conflict-free for the fragment reads, see MI355X_MICROARCH LDS table):
//   k-contiguous operand  -> T[row][BK+4]: a lane reads 4 consecutive k as one ds_read_b128 (row stride 144 B puts
//                            16 consecutive rows on 16 distinct 4-bank slots); MFMA step 4c+j of a slab contracts
//                            k = 8c + 4*(lane>>5) + j  -- any partition of the slab's k works if A and B agree.
//   m/n-contiguous operand-> T[k][BM]    : a lane reads T[k][col0 + (lane&31)] (consecutive lanes, consecutive banks).
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int KC_LD = BK + 4;     // row stride (floats) of a k-contiguous LDS image
constexpr int TILE_F = 64 * KC_LD; // floats reserved per operand image (>= 32*64 for the other layout)

// global -> registers for one operand tile (64 rows/cols x 32 k), 2 float4 per thread.
// VEC (16-byte aligned rows, no float4 straddles a bound): every load is an UNCONDITIONAL float4 from a clamped, always
// valid address and the out-of-range ones are zeroed when the tile is written to LDS.  hipcc turns guarded loads into
// branches separated by vmcnt(0), i.e. one dependent memory round trip per guard; the scalar path below keeps the guards
// and is only taken for unaligned / odd-sized operands.
template <int LAYOUT, bool VEC>
__device__ __forceinline__ void load_tile(f32x4 (&r)[2], bool (&ok)[2], const float* __restrict__ P, long ld, int row0,
                                          int rows, int k0, int kbeg, int kend) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = tid + 256 * j;
        if (LAYOUT == 0) {                       // P(row,k) = P[row*ld + k]
            const int row = row0 + (f >> 3), k = k0 + ((f & 7) << 2);
            if (VEC) {
                ok[j] = row < rows && k + 3 < kend;
                r[j] = *(const f32x4*)(P + (long)(ok[j] ? row : 0) * ld + (ok[j] ? k : kbeg));
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < rows) {
                    const float* p = P + (long)row * ld + k;
                    if (k < kend) v[0] = p[0];
                    if (k + 1 < kend) v[1] = p[1];
                    if (k + 2 < kend) v[2] = p[2];
                    if (k + 3 < kend) v[3] = p[3];
                }
                r[j] = v; ok[j] = true;
            }
        } else {                                 // P(row,k) = P[k*ld + row]
            const int k = k0 + (f >> 4), row = row0 + ((f & 15) << 2);
            if (VEC) {
                ok[j] = k < kend && row + 3 < rows;
                r[j] = *(const f32x4*)(P + (long)(ok[j] ? k : kbeg) * ld + (ok[j] ? row : 0));
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < kend) {
                    const float* p = P + (long)k * ld + row;
                    if (row < rows) v[0] = p[0];
                    if (row + 1 < rows) v[1] = p[1];
                    if (row + 2 < rows) v[2] = p[2];
                    if (row + 3 < rows) v[3] = p[3];
                }
                r[j] = v; ok[j] = true;
            }
        }
    }
}

template <int LAYOUT>
__device__ __forceinline__ void store_tile(const f32x4 (&r)[2], const bool (&ok)[2], float* T) {
    const int tid = threadIdx.x;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = tid + 256 * j;
        const f32x4 v = ok[j] ? r[j] : zero4;
        if (LAYOUT == 0) *(f32x4*)(T + (f >> 3) * KC_LD + ((f & 7) << 2)) = v;
        else             *(f32x4*)(T + (f >> 4) * 64 + ((f & 15) << 2)) = v;
    }
}

template <int AL, int BL, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int zb = blockIdx.z / g.nsplit, zs = blockIdx.z % g.nsplit;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = zs * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const float* A = g.A + (long)zb * g.sA;
    const float* B = g.B + (long)zb * g.sB;
    float* C = g.C + (long)zb * g.sC + (long)zs * g.sCsplit;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // Register staging ring, NST slabs deep: the global loads of slab s+1+NST are issued while slab s is multiplied, so a
    // load has NST slab times (NST * 1024 matrix-pipe cycles) to come back.  The small GEMMs of the hot path (hypernetwork:
    // 160 rows) run a dozen workgroups of ~10 slabs each; with a one-deep prefetch every slab was a full L2 round trip.
    constexpr int NST = 3;
    f32x4 ra[NST][2], rb[NST][2];
    bool oka[NST][2], okb[NST][2];
    const int nslab = (kend - kbeg + BK - 1) / BK;
    if (nslab > 0) {
        load_tile<AL, VEC>(ra[0], oka[0], A, g.lda, m0, g.M, kbeg, kbeg, kend);
        load_tile<BL, VEC>(rb[0], okb[0], B, g.ldb, n0, g.N, kbeg, kbeg, kend);
        store_tile<AL>(ra[0], oka[0], lds[0][0]);
        store_tile<BL>(rb[0], okb[0], lds[0][1]);
    }
    // slab q waits in ring slot q % NST
    if (1 < nslab) { load_tile<AL, VEC>(ra[1], oka[1], A, g.lda, m0, g.M, kbeg + 1 * BK, kbeg, kend); load_tile<BL, VEC>(rb[1], okb[1], B, g.ldb, n0, g.N, kbeg + 1 * BK, kbeg, kend); }
    if (2 < nslab) { load_tile<AL, VEC>(ra[2], oka[2], A, g.lda, m0, g.M, kbeg + 2 * BK, kbeg, kend); load_tile<BL, VEC>(rb[2], okb[2], B, g.ldb, n0, g.N, kbeg + 2 * BK, kbeg, kend); }
    if (3 < nslab) { load_tile<AL, VEC>(ra[0], oka[0], A, g.lda, m0, g.M, kbeg + 3 * BK, kbeg, kend); load_tile<BL, VEC>(rb[0], okb[0], B, g.ldb, n0, g.N, kbeg + 3 * BK, kbeg, kend); }
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    // one slab: multiply LDS buffer s&1, move slab s+1 from its ring slot SL to the other buffer, refill the slot
    auto step = [&](auto slot_c, int s) {
        constexpr int SL = decltype(slot_c)::value;
        const int cur = s & 1;
        const float* TA = lds[cur][0];
        const float* TB = lds[cur][1];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 av, bv;
            if (AL == 0) av = *(const f32x4*)(TA + (wm * 32 + li) * KC_LD + 8 * c + 4 * kh);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = TA[(8 * c + 4 * kh + j) * 64 + wm * 32 + li];
            }
            if (BL == 0) bv = *(const f32x4*)(TB + (wn * 32 + li) * KC_LD + 8 * c + 4 * kh);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = TB[(8 * c + 4 * kh + j) * 64 + wn * 32 + li];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
        }
        if (s + 1 < nslab) {
            store_tile<AL>(ra[SL], oka[SL], lds[cur ^ 1][0]);
            store_tile<BL>(rb[SL], okb[SL], lds[cur ^ 1][1]);
        }
        if (s + 1 + NST < nslab) {
            load_tile<AL, VEC>(ra[SL], oka[SL], A, g.lda, m0, g.M, kbeg + (s + 1 + NST) * BK, kbeg, kend);
            load_tile<BL, VEC>(rb[SL], okb[SL], B, g.ldb, n0, g.N, kbeg + (s + 1 + NST) * BK, kbeg, kend);
        }
        __syncthreads();
    };
    for (int s = 0; s < nslab; s += NST) {
        step(WgInt<1>{}, s);
        if (s + 1 < nslab) step(WgInt<2>{}, s + 1);
        if (s + 2 < nslab) step(WgInt<0>{}, s + 2);
    }

    // epilogue: register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31 of the wave's 32x32 tile
    const int n = n0 + wn * 32 + li;
    if (n < g.N) {
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < g.M) {
                float v = g.alpha * acc[r] + bias;
                if (g.act == 1) {
                    v = v > 0.f ? v : 0.f;
                    if (g.drop_thr) {                       // dropout after the ReLU (AM3 g / h, am3.py:80-88)
                        unsigned x = g.drop_key ^ (unsigned)((long)m * g.N + n);
                        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
                        v = x >= g.drop_thr ? v * g.drop_scale : 0.f;
                    }
                }
                else if (g.act == 2) v = tanhf(v);
                else if (g.act == 3) v = 1.f / (1.f + expf(-v));
                float* c = C + (long)m * g.ldc + n;
                if (g.mask) v = g.mask[(long)m * g.ldc + n] > 0.f ? v : 0.f;
                if (g.accumulate) v += *c;
                *c = v;
            }
        }
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, long stride, long n, float scale,
                                    float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += slabs[k * stride + i];
        out[i] = scale * s;
    }
}

// several slab reductions in ONE launch: segment k: dst[i] = scale * sum_t src[t*stride + i], i < n.  Even slabs and odd
// slabs are summed in two chains (fixed order: bit-reproducible); aligned segments move 16 bytes per thread and slab.
__global__ void reduce_multi_kernel(ReduceSegs sg) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= sg.end[sg.n - 1]) return;
    int k = 0;
    while (k + 1 < sg.n && i >= sg.end[k]) ++k;
    const long u = i - (k ? sg.end[k - 1] : 0);
    const long st = sg.stride[k];
    const int ns = sg.nslab[k];
    if (sg.vec[k]) {
        const float* src = sg.src[k] + 4 * u;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        int t = 0;
        for (; t + 3 < ns; t += 4) {                       // four independent 16-byte loads in flight
            const f32x4 a = *(const f32x4*)(src + t * st), b = *(const f32x4*)(src + (t + 1) * st);
            const f32x4 c = *(const f32x4*)(src + (t + 2) * st), d = *(const f32x4*)(src + (t + 3) * st);
            s0 += a; s1 += b; s0 += c; s1 += d;
        }
        for (; t + 1 < ns; t += 2) { s0 += *(const f32x4*)(src + t * st); s1 += *(const f32x4*)(src + (t + 1) * st); }
        if (t < ns) s0 += *(const f32x4*)(src + t * st);
        *(f32x4*)(sg.dst[k] + 4 * u) = sg.scale * (s0 + s1);
    } else {
        const float* src = sg.src[k] + u;
        float s0 = 0.f, s1 = 0.f;
        int t = 0;
        for (; t + 1 < ns; t += 2) { s0 += src[t * st]; s1 += src[(t + 1) * st]; }
        if (t < ns) s0 += src[t * st];
        sg.dst[k][u] = sg.scale * (s0 + s1);
    }
}

// reduce_multi + the optimizer step + the publication of the statistics (launch_reduce_multi_final).  ap[k] != NULL: segment k's
// outputs are the gradient of a parameter tensor: p / m / v at the same offsets get torch.optim.Adam's update (the expressions of
// adam_kernel, adam.hip, in the same order: bit-identical).  An output that lies in [pub_src, pub_src + pub_n) is also stored to
// the pinned host buffer; the lane whose store is the last one (a device counter) stores the sequence word behind it.
struct AdamSegs { float* p[24]; float* m[24]; float* v[24]; float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd; };
__device__ __forceinline__ void adam1(float g, float& p, float& m, float& v, const AdamSegs& a) {
    adam_update1(g, p, m, v, a.lr_over_bc1, a.inv_sqrt_bc2, a.b1, a.b2, a.eps, a.wd);
}
__global__ void reduce_multi_adam_kernel(ReduceSegs sg, AdamSegs ad, const float* pub_src, int pub_n, float* pub_dst,
                                         unsigned long long pub_seq, int* pub_cnt) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= sg.end[sg.n - 1]) return;
    int k = 0;
    while (k + 1 < sg.n && i >= sg.end[k]) ++k;
    const long u = i - (k ? sg.end[k - 1] : 0);
    const long st = sg.stride[k];
    const int ns = sg.nslab[k];
    if (sg.vec[k]) {
        const float* src = sg.src[k] + 4 * u;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        int t = 0;
        for (; t + 3 < ns; t += 4) {
            const f32x4 a = *(const f32x4*)(src + t * st), b = *(const f32x4*)(src + (t + 1) * st);
            const f32x4 c = *(const f32x4*)(src + (t + 2) * st), d = *(const f32x4*)(src + (t + 3) * st);
            s0 += a; s1 += b; s0 += c; s1 += d;
        }
        for (; t + 1 < ns; t += 2) { s0 += *(const f32x4*)(src + t * st); s1 += *(const f32x4*)(src + (t + 1) * st); }
        if (t < ns) s0 += *(const f32x4*)(src + t * st);
        const f32x4 g = sg.scale * (s0 + s1);
        *(f32x4*)(sg.dst[k] + 4 * u) = g;
        if (ad.p[k]) {
            f32x4 pp = *(f32x4*)(ad.p[k] + 4 * u), mm = *(f32x4*)(ad.m[k] + 4 * u), vv = *(f32x4*)(ad.v[k] + 4 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pp[e], me = mm[e], ve = vv[e];
                adam1(g[e], pe, me, ve, ad);
                pp[e] = pe; mm[e] = me; vv[e] = ve;
            }
            *(f32x4*)(ad.p[k] + 4 * u) = pp; *(f32x4*)(ad.m[k] + 4 * u) = mm; *(f32x4*)(ad.v[k] + 4 * u) = vv;
        }
    } else {
        const float* src = sg.src[k] + u;
        float s0 = 0.f, s1 = 0.f;
        int t = 0;
        for (; t + 1 < ns; t += 2) { s0 += src[t * st]; s1 += src[(t + 1) * st]; }
        if (t < ns) s0 += src[t * st];
        const float g = sg.scale * (s0 + s1);
        float* d = sg.dst[k] + u;
        *d = g;
        if (ad.p[k]) {
            float pp = ad.p[k][u], mm = ad.m[k][u], vv = ad.v[k][u];
            adam1(g, pp, mm, vv, ad);
            ad.p[k][u] = pp; ad.m[k][u] = mm; ad.v[k][u] = vv;
        }
        if (pub_dst && d >= pub_src && d < pub_src + pub_n) {          // (the statistics are single-element segments)
            __hip_atomic_store(pub_dst + (d - pub_src), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (__hip_atomic_fetch_add(pub_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pub_n - 1) {
                __hip_atomic_store(pub_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (ready for the next step)
                __hip_atomic_store((unsigned long long*)(pub_dst + 14), pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void colsum_kernel(const float* __restrict__ X, int M, int N, long ld, float scale, float* __restrict__ out) {
    // one wave-column-group per block: blockDim = (64, 4); each y-slice sums a strided subset of rows
    __shared__ float part[4][64];
    const int n = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (n < N) for (int m = threadIdx.y; m < M; m += 4) s += X[(long)m * ld + n];
    part[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && n < N) out[n] = scale * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// partial column sums: block -> (job, 64-column block, 128-row chunk); wave w of the block takes rows w, w+4, ... of the
// chunk, four independent loads in flight; part[job][chunk][N4] (N4 = N rounded up to 4: aligned slabs for the reduction)
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumJobs jb, float* __restrict__ part) {
    __shared__ float sh[4][64];
    int j = 0;
    while (j + 1 < jb.n && (int)blockIdx.x >= jb.blk_end[j]) ++j;
    const int lb = blockIdx.x - (j ? jb.blk_end[j - 1] : 0);
    const int M = jb.M[j], N = jb.N[j];
    const long ld = jb.ld[j];
    const int ncb = (N + 63) >> 6, cbk = lb % ncb, chunk = lb / ncb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = cbk * 64 + lane, r0 = chunk * 128, r1 = min(M, r0 + 128);
    const float* X = jb.X[j] + (n < N ? n : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int m = r0 + wave; m < r1; m += 16) {
        const float a = X[(long)m * ld];
        const float b = m + 4 < r1 ? X[(long)(m + 4) * ld] : 0.f;
        const float c = m + 8 < r1 ? X[(long)(m + 8) * ld] : 0.f;
        const float d = m + 12 < r1 ? X[(long)(m + 12) * ld] : 0.f;
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    sh[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && n < N) {
        const long N4 = ((long)N + 3) & ~3L;
        part[jb.part_off[j] + chunk * N4 + n] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

int launch_gemm(hipStream_t st, const GemmArgs& g, int AL, int BL) {
    if (g.M <= 0 || g.N <= 0) return FUMI_OK;
    if (g.K < 0 || g.nsplit < 1 || g.nbatch < 1 || !g.A || !g.B || !g.C) return FUMI_EINVAL;
    // float4 path: every float4 16-byte aligned and never straddling a bound (row length / tile extent multiple of 4)
    const bool avec = aligned16(g.A) && g.lda % 4 == 0 && g.sA % 4 == 0 && g.kchunk % 4 == 0 && (AL == 0 ? g.K % 4 == 0 : g.M % 4 == 0);
    const bool bvec = aligned16(g.B) && g.ldb % 4 == 0 && g.sB % 4 == 0 && g.kchunk % 4 == 0 && (BL == 0 ? g.K % 4 == 0 : g.N % 4 == 0);
    const bool vec = avec && bvec;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nbatch * g.nsplit), block(256);
#define GEMM_LAUNCH(a, b)                                                                              \
    if (vec) hipLaunchKernelGGL((gemm_kernel<a, b, true>), grid, block, 0, st, g);                   \
    else hipLaunchKernelGGL((gemm_kernel<a, b, false>), grid, block, 0, st, g);
    if (AL == 0 && BL == 0) { GEMM_LAUNCH(0, 0) }
    else if (AL == 0 && BL == 1) { GEMM_LAUNCH(0, 1) }
    else if (AL == 1 && BL == 0) { GEMM_LAUNCH(1, 0) }
    else { GEMM_LAUNCH(1, 1) }
#undef GEMM_LAUNCH
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_reduce_slabs(hipStream_t st, const float* slabs, int nslab, long stride, long n, float scale, float* out) {
    if (n <= 0) return FUMI_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, slabs, nslab, stride, n, scale, out);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_reduce_multi(hipStream_t st, ReduceSegs& sg) {
    if (sg.n < 1) return FUMI_OK;
    const long tot = sg.end[sg.n - 1];
    if (tot < 1) return FUMI_OK;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sg);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_reduce_multi_final(fumi_ws* ws, hipStream_t st, ReduceSegs& sg) {
    AdamPending* ap = ws ? ws->adam : nullptr;
    if (!ap || !ap->on || sg.n < 1) return launch_reduce_multi(st, sg);
    static const int fuse_env = getenv("FUMI_ADAM_FUSE") ? atoi(getenv("FUMI_ADAM_FUSE")) : 1;
    // every gradient tensor of the pending step must be exactly one segment (whole tensor, from its first element); no other
    // segment may write into a gradient tensor; a pending publication must read single-element segments of this reduction only
    AdamSegs ad; memset(&ad, 0, sizeof(ad));
    bool ok = fuse_env != 0;
    int matched = 0, pub_found = 0;
    for (int k = 0; k < sg.n && ok; ++k) {
        const long cnt = (sg.end[k] - (k ? sg.end[k - 1] : 0)) * (sg.vec[k] ? 4 : 1);
        const float* d = sg.dst[k];
        for (int j = 0; j < ap->n; ++j) {
            if (d == ap->g[j] && cnt == ap->numel[j]) { ad.p[k] = ap->p[j]; ad.m[k] = ap->m[j]; ad.v[k] = ap->v[j]; ++matched; break; }
            if (d + cnt > ap->g[j] && d < ap->g[j] + ap->numel[j]) { ok = false; break; }            // partial cover: not this way
        }
        if (ad.p[k] && sg.vec[k] && ((((uintptr_t)ad.p[k] | (uintptr_t)ad.m[k] | (uintptr_t)ad.v[k]) & 15) != 0)) ok = false;
        if (ws->pub_dst && d >= ws->pub_src && d < ws->pub_src + ws->pub_n) { if (cnt == 1 && !sg.vec[k]) ++pub_found; else ok = false; }
    }
    ok = ok && matched == ap->n && (!ws->pub_dst || pub_found == ws->pub_n);
    if (!ok) {                                             // separate launches, same results
        int rc = launch_reduce_multi(st, sg);
        return rc ? rc : launch_adam_pending(ws, st);
    }
    ad.lr_over_bc1 = ap->lr_over_bc1; ad.inv_sqrt_bc2 = ap->inv_sqrt_bc2; ad.b1 = ap->b1; ad.b2 = ap->b2; ad.eps = ap->eps; ad.wd = ap->wd;
    const long tot = sg.end[sg.n - 1];
    hipLaunchKernelGGL(reduce_multi_adam_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sg, ad, ws->pub_src, ws->pub_n,
                       ws->pub_dst, ws->pub_seq, ws->status + 48);
    ap->on = 0;
    ws->pub_dst = nullptr; ws->pub_src = nullptr;          // (the publication rode along)
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_colsum_multi(hipStream_t st, const ColsumJobs& jobs, float* part, const ReduceSegs* extra) {
    ReduceSegs rs; rs.n = 0; rs.scale = 1.f;
    if (extra) rs = *extra;                            // further slab sums of the caller ride in the same reduction launch
    if (jobs.n < 1) return launch_reduce_multi(st, rs);
    if (rs.n + jobs.n > 24 || rs.scale != 1.f) return FUMI_EINVAL;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(jobs.blk_end[jobs.n - 1]), dim3(256), 0, st, jobs, part);
    LAUNCH_CHECK();
    for (int j = 0; j < jobs.n; ++j) {
        const long N4 = ((long)jobs.N[j] + 3) & ~3L;
        rs.add(part + jobs.part_off[j], (jobs.M[j] + 127) / 128, N4, jobs.N[j], jobs.out[j]);
    }
    return launch_reduce_multi(st, rs);
}

int launch_colsum(hipStream_t st, const float* X, int M, int N, long ld, float scale, float* out) {
    if (N <= 0) return FUMI_OK;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(64, 4), 0, st, X, M, N, ld, scale, out);
    LAUNCH_CHECK();
    return FUMI_OK;
}
