// The two shared GEMM passes over the 2048-wide inputs -- the only kernels of a meta-step that touch X.
//
//   xpanel_fwd_kernel:  for every episode b,  [A0_b | G_b] = [Xs_b ; Xq_b] . [W0 ; Xs_b]^T
//        A0_b [R, h0] = layer-0 pre-activations of all R = S+Qn rows with the SHARED meta-weight W0   (torchmeta
//        MetaLinear.forward via fumi/models/fumi.py:215 at inner step 0), G_b [R, S] = Gram matrix against the support
//        rows (what turns the per-episode fast weight W0 - alpha*D^T Xs of later steps into a rank-S correction).
//   xpanel_bwd_kernel:  gW0 = sum_b Abar0_b^T [Xs_b ; Xq_b]   (autograd's AddmmBackward of layer 0 summed over every
//        use inside the second-order graph, fumi/models/fumi.py:192), contraction over all B*R rows split into slabs.
//
// Kernels in this file, in the order they were written (the launchers at the end pick; DESIGN.md sections 5 and 12):
//   fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32; 157 TFLOP/s peak): xpanel_fwd_generic / xpanel_fwd_kernel (64x64 tiles, 2x2 waves,
//     32- / 64-deep slabs double-buffered in LDS, next slab prefetched into registers), xpanel_bwd_kernel, xpanel_bwd256_kernel;
//   split-bf16 (every fp32 operand split exactly into three bf16 pieces, six piece products on v_mfma_f32_32x32x16_bf16, fp32
//     accuracy at 2.7x the matrix rate): xpanel_fwd_sb_kernel (both operands split per tile), xpanel_presplit_kernel +
//     xpanel_fwd_ps_kernel (the column operand split once per step into planes in MFMA fragment order: the default forward),
//     xpanel_bwd256_sb_kernel (transposing LDS reads: the default backward).
// X is never copied: a "virtual row" r of episode b is read from
// x_s[b, r] (r < S) or x_q[b, r-S]; a virtual column c of the forward pass from W0[c] (c < h0) or x_s[b, c-h0].
//
// XCD-aware placement (forward): workgroup ids that are equal mod 8 share an XCD (round-robin dispatch), so episode
// b is given only ids with id % 8 == b % 8: all column tiles of an episode's row panel, and the W0 slab they all stream,
// hit one 4 MiB L2.  Placement only affects speed.
#include "common.h"
#include "hyper_fwd.h"
#include "hyper_bwd.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;           // row stride of a k-contiguous LDS image: 144 B -> conflict-free ds_read_b128
constexpr int TILE_F = 64 * KC_LD;

struct XPanel {
    const float* x_s; const float* x_q; const float* W0;
    int B, S, Qn, D, h0;
    // zero-copy episodes (sampler.hip): when `table` is set, row r of episode b is table[idx_s[b,r]] / table[idx_q[b,r-S]]
    // instead of x_s[b,r] / x_q[b,r-S] -- the meta-batch is never materialised
    const float* table; const int64_t* idx_s; const int64_t* idx_q; long n_rows;
    int gcols;              // Gram columns computed: S (FuMI / MAML), or 0 when the caller passed G = NULL (AM3's image encoder)
    int ksplit;             // split-bf16 forward only: the contraction is cut into ksplit parts (one workgroup each); part z writes its
    long part_stride;       // partial product to A0 + z * part_stride (narrow outputs: too few tiles to fill the chip otherwise)
};

__device__ __forceinline__ const float* xrow(const XPanel& p, int b, int r) {
    if (p.table) {                                   // uniform branch
        long i = r < p.S ? p.idx_s[(long)b * p.S + r] : p.idx_q[(long)b * p.Qn + (r - p.S)];
        if (i < 0 || i >= p.n_rows) i = 0;           // (flagged by the launcher's range check; never fault)
        return p.table + i * p.D;
    }
    return r < p.S ? p.x_s + ((long)b * p.S + r) * p.D : p.x_q + ((long)b * p.Qn + (r - p.S)) * p.D;
}

// ---- forward -----------------------------------------------------------------------------------------------------
// Generic-shape forward (any D / alignment): simple 32-deep double-buffered loop.  The bench shapes take
// xpanel_fwd_kernel below.
template <bool FAST>
__global__ __launch_bounds__(256) void xpanel_fwd_generic_kernel(XPanel p, float* __restrict__ A0, float* __restrict__ G,
                                                         int tiles_m, int tiles_n, int stagger) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.S + p.Qn, C = p.h0 + p.gcols, K = p.D;
    // id -> (episode, tile) with id % 8 == episode % 8
    const int tiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = xcd + 8 * (j / tiles), t = j % tiles;
    if (b >= p.B) return;
    const int m0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * 64;

    // each thread stages 2 float4 of the A tile and 2 of the B tile per slab: rows (f>>3), k offset (f&7)*4
    const float* arow[2]; const float* brow[2]; bool aok[2], bok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        const int r = m0 + (f >> 3), c = n0 + (f >> 3);
        aok[i] = r < R; bok[i] = c < C;
        arow[i] = xrow(p, b, aok[i] ? r : 0);
        brow[i] = c < p.h0 ? p.W0 + (long)c * K : xrow(p, b, bok[i] ? c - p.h0 : 0);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](f32x4 (&ra)[2], f32x4 (&rb)[2], int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + (((tid + 256 * i) & 7) << 2);
            if (FAST) {
                ra[i] = *(const f32x4*)(arow[i] + k);       // raw; masked when written to LDS (keeps the loads in
                rb[i] = *(const f32x4*)(brow[i] + k);       // flight behind the MFMAs instead of waiting here)
            } else {
                f32x4 va = zero4, vb = zero4;
                for (int e = 0; e < 4; ++e) if (aok[i] && k + e < K) va[e] = arow[i][k + e];
                for (int e = 0; e < 4; ++e) if (bok[i] && k + e < K) vb[e] = brow[i][k + e];
                ra[i] = va; rb[i] = vb;
            }
        }
    };
    auto store = [&](const f32x4 (&ra)[2], const f32x4 (&rb)[2], int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            *(f32x4*)(lds[buf][0] + (f >> 3) * KC_LD + ((f & 7) << 2)) = (!FAST || aok[i]) ? ra[i] : zero4;
            *(f32x4*)(lds[buf][1] + (f >> 3) * KC_LD + ((f & 7) << 2)) = (!FAST || bok[i]) ? rb[i] : zero4;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 ra[2], rb[2];
    const int nslab = (K + BK - 1) / BK;
    // rotate the slab order per workgroup: workgroups that run in lock-step then stream different 128-byte columns of
    // their 8 KiB-strided rows at any moment (spreads the L2/fabric channels); the sum order per tile stays fixed
    const int s0 = stagger ? (int)(((long)blockIdx.x * stagger) % nslab) : 0;
    auto slab_k = [&](int s) { int q = s + s0; if (q >= nslab) q -= nslab; return q * BK; };
    load(ra, rb, slab_k(0));
    store(ra, rb, 0);
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        if (s + 1 < nslab) load(ra, rb, slab_k(s + 1));
        const float* TA = lds[cur][0] + (wm * 32 + li) * KC_LD + 4 * kh;
        const float* TB = lds[cur][1] + (wn * 32 + li) * KC_LD + 4 * kh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 av = *(const f32x4*)(TA + 8 * c);
            const f32x4 bv = *(const f32x4*)(TB + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
        if (s + 1 < nslab) store(ra, rb, cur ^ 1);
        __syncthreads();
    }
    // epilogue: column n -> A0 (n < h0) or G (n - h0 < S); register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5)
    const int n = n0 + wn * 32 + li;
    if (n < C) {
        float* base; long ld; int col;
        if (n < p.h0) { base = A0 + (long)b * R * p.h0; ld = p.h0; col = n; }
        else          { base = G + (long)b * R * p.S;  ld = p.S;  col = n - p.h0; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < R) base[(long)m * ld + col] = acc[r];
        }
    }
}


// ---- forward, fast path (D % 64 == 0, 16-byte aligned rows): the dominant kernel of a meta-step ---------------------
// 64x64 tile, 64-deep slabs in two LDS buffers, software-pipelined by hand so the matrix pipe never waits for memory:
//   * global -> registers for slab s+1 is issued at the top of slab s and written to the other LDS buffer only after
//     24 of the slab's 32 MFMAs have been issued (unconditional float4 loads: hipcc turns guarded per-element loads into
//     branches separated by vmcnt(0), i.e. dependent L2 round trips; rows past the panel are clamped and zeroed at the
//     LDS write instead)
//   * MFMA operand fragments are double-buffered in registers per group of 8 MFMAs (16 k): the ds_read_b128s of group
//     g+1 are in flight while group g runs, so only the first group after the slab barrier exposes LDS latency
// A lone workgroup is then ~90 % matrix-bound, which is what lets two co-resident workgroups share a CU without the
// older one starving the younger (measured: with a memory-latency-bound loop the second workgroup only got the gaps).
constexpr int FBK = 64, FLD = FBK + 4;          // 272-byte rows: 16 consecutive rows hit 16 distinct 4-bank slots
// NST = depth of the register staging ring: the global loads of slab s+NST are issued while slab s is multiplied, so a
// load has NST*2048 matrix-pipe cycles to return (an HBM/Infinity-Cache miss under load is ~2 us = ~4000 cycles).
template <int NST>
__global__ __launch_bounds__(256) void xpanel_fwd_kernel(XPanel p, float* __restrict__ A0, float* __restrict__ G,
                                                         int tiles_m, int tiles_n, unsigned long long* trace) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][64 * FLD];
    unsigned long long t_rt = 0, t_ck = 0;
    if (trace) { t_rt = __builtin_amdgcn_s_memrealtime(); t_ck = __builtin_amdgcn_s_memtime(); }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.S + p.Qn, C = p.h0 + p.gcols, K = p.D;
    const int tiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = xcd + 8 * (j / tiles), t = j % tiles;
    if (b >= p.B) return;
    const int m0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * 64;

    // staging map: float4 f = tid + 256 i  ->  tile row (f >> 4), k offset (f & 15) * 4
    const float* arow[4]; const float* brow[4]; bool aok[4], bok[4];
    const int k4 = (tid & 15) << 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (tid >> 4) + 16 * i;
        const int r = m0 + rr, c = n0 + rr;
        aok[i] = r < R; bok[i] = c < C;
        arow[i] = xrow(p, b, aok[i] ? r : 0) + k4;
        brow[i] = (c < p.h0 ? p.W0 + (long)c * K : xrow(p, b, bok[i] ? c - p.h0 : 0)) + k4;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ga[NST][4], gb[NST][4];
    const int li = lane & 31, kh = lane >> 5;
    const int aoff = (wm * 32 + li) * FLD + 4 * kh, boff = (wn * 32 + li) * FLD + 4 * kh;
    f32x4 fa[2][2], fb[2][2];                   // [register buffer][half of the 16-k group]
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nslab = K / FBK;

#define XP_GLOAD(st, k0)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        ga[st][i] = *(const f32x4*)(arow[i] + (k0)); gb[st][i] = *(const f32x4*)(brow[i] + (k0)); }
#define XP_LSTORE(st, buf)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        const int off = ((tid >> 4) + 16 * i) * FLD + k4;                                                    \
        *(f32x4*)(lds[buf][0] + off) = aok[i] ? ga[st][i] : zero4;                                           \
        *(f32x4*)(lds[buf][1] + off) = bok[i] ? gb[st][i] : zero4; }
#define XP_FREAD(rb, buf, g)                                                                                 \
    { fa[rb][0] = *(const f32x4*)(lds[buf][0] + aoff + 16 * (g));     fb[rb][0] = *(const f32x4*)(lds[buf][1] + boff + 16 * (g)); \
      fa[rb][1] = *(const f32x4*)(lds[buf][0] + aoff + 16 * (g) + 8); fb[rb][1] = *(const f32x4*)(lds[buf][1] + boff + 16 * (g) + 8); }
#define XP_MMA8(rb)                                                                                          \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int e = 0; e < 4; ++e)              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rb][h][e], fb[rb][h][e], acc, 0, 0, 0);
// one slab: stage ST holds slab s+1 (loaded NST slabs ago) and is refilled with slab s+1+NST after it went to LDS
#define XP_SLAB(ST, s)                                                                                       \
    if ((s) < nslab) {                                                                                       \
        const int cur = (s) & 1;                                                                             \
        const bool more = (s) + 1 < nslab;                                                                   \
        XP_FREAD(1, cur, 1) XP_MMA8(0)                                                                       \
        XP_FREAD(0, cur, 2) XP_MMA8(1)                                                                       \
        XP_FREAD(1, cur, 3) XP_MMA8(0)                                                                       \
        if (more) { XP_LSTORE(ST, cur ^ 1) }                                                                 \
        if ((s) + 1 + NST < nslab) { XP_GLOAD(ST, ((s) + 1 + NST) * FBK) }                                   \
        XP_MMA8(1)                                                                                           \
        __syncthreads();                                                                                     \
        if (more) XP_FREAD(0, cur ^ 1, 0)                                                                    \
    }

// steady-state slab (s + 1 + NST < nslab): no conditionals, one scheduling region, with the issue order spelled out --
// the 8 LDS writes of the staged slab and the 8 global loads that refill the stage are dealt one per MFMA into the
// shadows of the dependent MFMA chain (an MFMA occupies the pipe for 64 cycles; whatever sits between two MFMAs in
// program order issues for free), instead of in two clusters during which the matrix pipe drains.
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define XP_SLAB_MAIN(ST, s)                                                                                  \
    {                                                                                                        \
        const int cur = (s) & 1;                                                                             \
        XP_FREAD(1, cur, 1) XP_MMA8(0)                                                                       \
        XP_FREAD(0, cur, 2) XP_MMA8(1)                                                                       \
        XP_FREAD(1, cur, 3) XP_MMA8(0)                                                                       \
        XP_LSTORE(ST, cur ^ 1)                                                                               \
        XP_GLOAD(ST, ((s) + 1 + NST) * FBK)                                                                  \
        XP_MMA8(1)                                                                                           \
        SGB(0x100, 8) SGB(0x008, 8) SGB(0x100, 4)                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { SGB(0x008, 1) SGB(0x002, 4) SGB(0x200, 1) }          \
        SGB(0x100, 4)                                                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { SGB(0x008, 1) SGB(0x002, 2) SGB(0x020, 1) }          \
        SGB(0x008, 8)                                                                                        \
        __syncthreads();                                                                                     \
        XP_FREAD(0, cur ^ 1, 0)                                                                              \
    }

    // prologue: slab 0 straight to LDS, slabs 1..NST into the ring (stage of slab q is (q-1) % NST)
    XP_GLOAD(0, 0)
    XP_LSTORE(0, 0)
    if (1 < nslab) { XP_GLOAD(0, 1 * FBK) }
    if (NST > 1 && 2 < nslab) { XP_GLOAD(1 % NST, 2 * FBK) }
    if (NST > 2 && 3 < nslab) { XP_GLOAD(2 % NST, 3 * FBK) }
    __syncthreads();
    XP_FREAD(0, 0, 0)
    int s = 0;
    for (; s + 2 * NST < nslab; s += NST) {          // every slab of this round has s' + 1 + NST < nslab
        XP_SLAB_MAIN(0, s)
        if (NST > 1) { XP_SLAB_MAIN(1 % NST, s + 1) }
        if (NST > 2) { XP_SLAB_MAIN(2 % NST, s + 2) }
    }
    for (; s < nslab; s += NST) {
        XP_SLAB(0, s)
        if (NST > 1) { XP_SLAB(1 % NST, s + 1) }
        if (NST > 2) { XP_SLAB(2 % NST, s + 2) }
    }
#undef XP_SLAB_MAIN
#undef SGB
#undef XP_GLOAD
#undef XP_LSTORE
#undef XP_FREAD
#undef XP_MMA8
#undef XP_SLAB
    if (trace && tid == 0) {      // dev tracing (tools/trace_xpanel.py): wall ticks (100 MHz), shader cycles, placement
        trace[blockIdx.x * 6 + 0] = t_rt;
        trace[blockIdx.x * 6 + 1] = __builtin_amdgcn_s_memrealtime();
        trace[blockIdx.x * 6 + 2] = t_ck;
        trace[blockIdx.x * 6 + 3] = __builtin_amdgcn_s_memtime();
        trace[blockIdx.x * 6 + 4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        trace[blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    }
    const int n = n0 + wn * 32 + li;
    if (n < C) {
        float* base; long ld; int col;
        if (n < p.h0) { base = A0 + (long)b * R * p.h0; ld = p.h0; col = n; }
        else          { base = G + (long)b * R * p.S;  ld = p.S;  col = n - p.h0; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < R) base[(long)m * ld + col] = acc[r];
        }
    }
}

// ---- forward on the bf16 matrix pipe with fp32-equivalent accuracy ("split-bf16") -------------------------------------
// gfx950 multiplies bf16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_bf16: 32 cycles for 16 k; the fp32 form
// v_mfma_f32_32x32x2_f32: 64 cycles for 2 k).  Every fp32 operand is split EXACTLY into three bf16 pieces when its slab
// goes to LDS,  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round to nearest even; both
// subtractions are exact in fp32, 8 significand bits per piece), and a product is accumulated in fp32 from the six
// piece products of weight >= 2^-16:   a*b ~= ah*bl + al*bh + am*bm + ah*bm + am*bh + ah*bh.
// The dropped terms (am*bl, al*bm, al*bl) are below 2^-24 |a*b|, i.e. below the rounding of the fp32 product itself.
// Measured against fp64 (tools/xpanel_error.py, bench shapes): random-sign data 7e-7 rms like the fp32 MFMA kernel; the Gram
// matrix of all-positive inputs 2.5e-6 rms vs 6e-7 (pieces are truncated, so the dropped tails have one sign).  Six bf16 MFMAs
// replace eight fp32 ones per 16 k at a quarter of the cycles each: 2.67x the matrix rate.  Non-finite inputs give NaN.
// LDS: per slab and operand three [64 rows][32 k] bf16 planes, row stride 80 B: conflict-free ds_read_b128 for the
// 32x32x16 operand map (lane l holds row l&31, k = 8*(l>>5) .. +7).
constexpr int SBK = 32;                    // contraction slab
constexpr int SROW = SBK + 8;              // ushorts per LDS row (80 bytes)
constexpr int SPLANE = 64 * SROW;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two floats -> one dword of two bf16 (low half = a), round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// exact three-way split of a pair: x = h + m + l + (a rounding residue below 2^-25 |x|, either sign: no bias in long sums)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = pk_bf16(s0, s1);
}
__device__ __forceinline__ void split3(const f32x4& v, u32x2& h, u32x2& m, u32x2& l) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(v[0], v[1], h0, m0, l0);
    split_pair(v[2], v[3], h1, m1, l1);
    h = (u32x2){h0, h1}; m = (u32x2){m0, m1}; l = (u32x2){l0, l1};
}

template <int I, int N, class F>
__device__ __forceinline__ void xp_static_for(F&& f) {
    if constexpr (I < N) { f(WgInt<I>{}); xp_static_for<I + 1, N>(f); }
}

typedef unsigned short (*SbLds)[2][3][SPLANE];      // [buffer][operand][piece]

// one 64 x 64 output tile (rows m0.., virtual columns n0..) of episode b, contraction part kz
template <int NST>
__device__ __forceinline__ void xpanel_fwd_sb_tile(const XPanel& p, float* __restrict__ A0, float* __restrict__ G, int b, int m0, int n0,
                                                   int kz, SbLds lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.S + p.Qn, C = p.h0 + p.gcols, K = p.D;
    const int ks = p.ksplit;
    const int kbeg = kz * (K / ks);             // this part's first contraction column

    // staging map: float4 f = tid + 256 i  ->  tile row (f >> 3), k offset (f & 7) * 4
    const float* arow[2]; const float* brow[2]; bool aok[2], bok[2];
    const int k4 = ((tid & 7) << 2) + kbeg;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = (tid >> 3) + 32 * i;
        const int r = m0 + rr, c = n0 + rr;
        aok[i] = r < R; bok[i] = c < C;
        arow[i] = xrow(p, b, aok[i] ? r : 0) + k4;
        brow[i] = (c < p.h0 ? p.W0 + (long)c * K : xrow(p, b, bok[i] ? c - p.h0 : 0)) + k4;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ga[NST][2], gb[NST][2];
    const int li = lane & 31, hh = lane >> 5;
    const int aoff = (wm * 32 + li) * SROW + 8 * hh, boff = (wn * 32 + li) * SROW + 8 * hh;
    // two accumulators, one per 16-deep step of a slab: two independent MFMA chains per wave (a wave owns a single 32x32 block, and
    // with two waves per SIMD a chain of twelve dependent MFMAs per slab left the matrix pipe waiting on its own results)
    f32x16 acc2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc2[0][i] = 0.f; acc2[1][i] = 0.f; }
    const int nslab = K / ks / SBK;

    auto gload = [&](auto sc, int k0) {
        constexpr int ST = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) { ga[ST][i] = *(const f32x4*)(arow[i] + k0); gb[ST][i] = *(const f32x4*)(brow[i] + k0); }
    };
    auto lstore = [&](auto sc, int buf) {
        constexpr int ST = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = ((tid >> 3) + 32 * i) * SROW + ((tid & 7) << 2);
            // no masking: a row past the panel (read from a clamped, valid address) only feeds outputs that are never stored
            u32x2 h, m, l;
            split3(ga[ST][i], h, m, l);
            *(u32x2*)(lds[buf][0][0] + off) = h; *(u32x2*)(lds[buf][0][1] + off) = m; *(u32x2*)(lds[buf][0][2] + off) = l;
            split3(gb[ST][i], h, m, l);
            *(u32x2*)(lds[buf][1][0] + off) = h; *(u32x2*)(lds[buf][1][1] + off) = m; *(u32x2*)(lds[buf][1][2] + off) = l;
        }
    };
    auto mma_slab = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SBK / 16; ++s) {
            bf16x8 ah = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[buf][0][0] + aoff + 16 * s));
            bf16x8 am = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[buf][0][1] + aoff + 16 * s));
            bf16x8 al = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[buf][0][2] + aoff + 16 * s));
            bf16x8 bh = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[buf][1][0] + boff + 16 * s));
            bf16x8 bm = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[buf][1][1] + boff + 16 * s));
            bf16x8 bl = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[buf][1][2] + boff + 16 * s));
            f32x16& acc = acc2[s & 1];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);      // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    };
    // one slab: slab s+1 sits in ring slot SL (loaded NST slabs ago); the slot is refilled with slab s+1+NST
    auto slab = [&](auto sc, int s) {
        const int cur = s & 1;
        if (s + 1 < nslab) lstore(sc, cur ^ 1);
        if (s + 1 + NST < nslab) gload(sc, (s + 1 + NST) * SBK);
        mma_slab(cur);
        __syncthreads();
    };
    // steady state (s + 1 + NST < nslab), issue order spelled out: the slab's 12 LDS fragment reads, then twelve segments
    // of one MFMA plus a twelfth of the work that splits the NEXT slab (8 VALU operations, or 6 byte-permutes and 3 LDS
    // writes) -- an MFMA leaves the vector ALU free for 24 of its 32 cycles.  Left alone, hipcc issues the whole split and
    // then the whole MFMA chain and the matrix pipe idles half of the time.  sched_barrier(0) pins the segments.
    auto slab_main = [&](auto sc, int s) {
        constexpr int ST = decltype(sc)::value;
        const int cur = s & 1, nxt = cur ^ 1;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                fa[st][pl] = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[cur][0][pl] + aoff + 16 * st));
                fb[st][pl] = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds[cur][1][pl] + boff + 16 * st));
            }
        __builtin_amdgcn_sched_barrier(0);
        // piece pairs in accumulation order (smallest first): (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // float4 q of the staged slab: A0, A1, B0, B1
            const f32x4 v = q < 2 ? ga[ST][q] : gb[ST][q - 2];
            const int off = ((tid >> 3) + 32 * (q & 1)) * SROW + ((tid & 7) << 2);
            unsigned hp[2], mp[2], lp[2];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const int u = 3 * q + part;                // MFMA number 0..11: step u & 1, piece pair u >> 1
                acc2[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u & 1][PA[u >> 1]], fb[u & 1][PB[u >> 1]], acc2[u & 1], 0, 0, 0);
                if (part < 2) {
                    split_pair(v[2 * part], v[2 * part + 1], hp[part], mp[part], lp[part]);
                } else {
                    *(u32x2*)(lds[nxt][q >> 1][0] + off) = (u32x2){hp[0], hp[1]};
                    *(u32x2*)(lds[nxt][q >> 1][1] + off) = (u32x2){mp[0], mp[1]};
                    *(u32x2*)(lds[nxt][q >> 1][2] + off) = (u32x2){lp[0], lp[1]};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gload(sc, (s + 1 + NST) * SBK);
        __syncthreads();
    };
    // prologue: slab 0 straight to LDS, slabs 1..NST into the ring (slot of slab q is (q-1) % NST)
    gload(WgInt<0>{}, 0);
    lstore(WgInt<0>{}, 0);
    int s = 0;
    if (nslab > 2 * NST) {
        // The steady-state loop is entered ONLY behind unconditional ring loads: hipcc sizes every s_waitcnt vmcnt(n) of the loop for
        // the worst way into it, and a load it must assume skipped counts as not issued -- with `if (q < nslab) gload(q)` in front,
        // the loop waited for all but its newest load, i.e. the full memory latency every slab instead of a load issued NST slabs ago.
        xp_static_for<0, NST>([&](auto ic) { gload(ic, (decltype(ic)::value + 1) * SBK); });
        __syncthreads();
        for (; s + 2 * NST < nslab; s += NST)            // every slab of this round has s' + 1 + NST < nslab
            xp_static_for<0, NST>([&](auto ic) { slab_main(ic, s + decltype(ic)::value); });
    } else {
        xp_static_for<0, NST>([&](auto ic) { if (decltype(ic)::value + 1 < nslab) gload(ic, (decltype(ic)::value + 1) * SBK); });
        __syncthreads();
    }
    for (; s < nslab; s += NST)
        xp_static_for<0, NST>([&](auto ic) { if (s + decltype(ic)::value < nslab) slab(ic, s + decltype(ic)::value); });
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc2[0][i] + acc2[1][i];
    const int n = n0 + wn * 32 + li;
    if (n < C) {
        float* base; long ld; int col;
        if (n < p.h0) { base = A0 + kz * p.part_stride + (long)b * R * p.h0; ld = p.h0; col = n; }
        else          { base = G + (long)b * R * p.S;  ld = p.S;  col = n - p.h0; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m < R) base[(long)m * ld + col] = acc[r];
        }
    }
}

// RIDER: the first rider.nblk workgroups (a multiple of 8, so the XCD grouping of the rest is unchanged) run the split
// hypernetwork forward (hyper_fwd.h) in this kernel's LDS and leave; they are dispatched first and are done in ~10 us.
template <int NST, bool RIDER>
__global__ __launch_bounds__(256, 2) void xpanel_fwd_sb_kernel(XPanel p, float* __restrict__ A0, float* __restrict__ G,
                                                                int tiles_m, int tiles_n, HyperFwdArgs rider) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][2][3][SPLANE];
    int bid = blockIdx.x;
    if constexpr (RIDER) {
        __shared__ int s_last;
        if (bid < rider.nblk) { hyper_fwd_split_body<HF_RIDER_KS, true>(rider, bid, (float*)&lds[0][0][0][0], &s_last); return; }
        bid -= rider.nblk;
    }
    const int tiles = tiles_m * tiles_n;
    const int xcd = bid & 7, j = bid >> 3;
    const int per = tiles * p.ksplit;
    const int b = xcd + 8 * (j / per), tz = j % per, kz = tz / tiles, t = tz - kz * tiles;
    if (b >= p.B) return;
    xpanel_fwd_sb_tile<NST>(p, A0, G, b, (t / tiles_n) * 64, (t % tiles_n) * 64, kz, lds);
}

// ---- forward with the column operand split ONCE ----------------------------------------------------------------------
// In xpanel_fwd_sb_tile every 64 x 64 tile splits its 64 rows of X AND its 64 rows of the column operand into bf16 pieces, slab
// by slab: the split (ten vector operations and three LDS writes per pair of values) and the LDS traffic of both operands cost
// more than the matrix instructions they feed, and W0 -- the same for all episodes and row tiles -- is split 96 times over.
// Here the column operand (W0, and each episode's support rows for the Gram block) is split once per step by
// xpanel_presplit_kernel into three bf16 planes stored in the ORDER THE MATRIX INSTRUCTION READS ITS B OPERAND:
//     [plane][16-deep step][32-column block][lane][8]   with lane l = column (l & 31), k = 16 step + 8 (l >> 5) .. + 7,
// so a wave fetches a fragment with one fully coalesced 1 KiB global_load_dwordx4 straight into registers (3 MiB for W0 + 12 MiB
// for the support rows of 32 episodes: L2 / MALL resident) -- no LDS, no vector work on the column side.  A workgroup computes
// 32 RB rows x (32 columns per wave); every wave owns one 32-column block -- W0 columns or Gram columns alike -- and RB
// accumulators that share each B fragment.  Only the slab of X is split and staged through LDS.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void xpanel_presplit_kernel(XPanel p, int cbg, unsigned short* __restrict__ Wp, unsigned short* __restrict__ Xp) {
    const int bid = blockIdx.x;
    // one wave = one fragment (32 columns x 16 k) in all three planes: lane l holds column (l & 31), k = 8 (l >> 5) .. + 7 -- the
    // three 1 KiB stores are contiguous; consecutive waves take consecutive 16-deep steps of one column block (they share the
    // 128-byte lines of the rows they read)
    const int K = p.D, nks = K >> 4, CB = p.h0 >> 5;
    const long gw = (long)bid * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, k8 = (lane >> 5) * 8;
    const long blk = gw / nks; const int ks = (int)(gw - blk * nks);
    const float* src; u32x4* dst; long plane;
    if (blk < CB) {
        src = p.W0 + ((long)blk * 32 + c) * K + ks * 16 + k8;
        plane = (long)p.h0 * (K >> 3);
        dst = (u32x4*)Wp + ((long)ks * CB + blk) * 64 + lane;
    } else {
        const long xb = blk - CB;
        if (xb >= (long)p.B * cbg) return;
        const int b = (int)(xb / cbg), g = (int)(xb - (long)b * cbg), col = g * 32 + c;
        src = col < p.S ? xrow(p, b, col) + ks * 16 + k8 : nullptr;
        plane = (long)cbg * 32 * (K >> 3);
        dst = (u32x4*)Xp + (long)b * 3 * plane + ((long)ks * cbg + g) * 64 + lane;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v0 = src ? *(const f32x4*)src : z4, v1 = src ? *(const f32x4*)(src + 4) : z4;
    u32x2 h0_, m0_, l0_, h1_, m1_, l1_;
    split3(v0, h0_, m0_, l0_);
    split3(v1, h1_, m1_, l1_);
    dst[0] = (u32x4){h0_[0], h0_[1], h1_[0], h1_[1]};
    dst[plane] = (u32x4){m0_[0], m0_[1], m1_[0], m1_[1]};
    dst[2 * plane] = (u32x4){l0_[0], l0_[1], l1_[0], l1_[1]};
}

// The pre-split with a pending embedding bag (glove_bag.h) as its first `nglove` workgroups: two short, latency-bound, independent
// launches (6-7 us and 10 us) become one.  512 threads: a bag workgroup handles one output row, a pre-split workgroup 8 fragments.
}  // namespace
#include "glove_bag.h"
namespace {
template <bool VEC>
__global__ __launch_bounds__(512) void xpanel_presplit_glove_kernel(XPanel p, int cbg, unsigned short* __restrict__ Wp, unsigned short* __restrict__ Xp,
                                                                    GloveArgs ga, int nglove) {
    extern __shared__ __attribute__((aligned(16))) float gpart[];
    if ((int)blockIdx.x < nglove) { glove_bag_row<VEC>(ga, blockIdx.x, gpart); return; }
    const int bid = blockIdx.x - nglove;
    const int K = p.D, nks = K >> 4, CB = p.h0 >> 5;
    const long gw = (long)bid * 8 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, k8 = (lane >> 5) * 8;
    const long blk = gw / nks; const int ks = (int)(gw - blk * nks);
    const float* src; u32x4* dst; long plane;
    if (blk < CB) {
        src = p.W0 + ((long)blk * 32 + c) * K + ks * 16 + k8;
        plane = (long)p.h0 * (K >> 3);
        dst = (u32x4*)Wp + ((long)ks * CB + blk) * 64 + lane;
    } else {
        const long xb = blk - CB;
        if (xb >= (long)p.B * cbg) return;
        const int b = (int)(xb / cbg), g = (int)(xb - (long)b * cbg), col = g * 32 + c;
        src = col < p.S ? xrow(p, b, col) + ks * 16 + k8 : nullptr;
        plane = (long)cbg * 32 * (K >> 3);
        dst = (u32x4*)Xp + (long)b * 3 * plane + ((long)ks * cbg + g) * 64 + lane;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v0 = src ? *(const f32x4*)src : z4, v1 = src ? *(const f32x4*)(src + 4) : z4;
    u32x2 h0_, m0_, l0_, h1_, m1_, l1_;
    split3(v0, h0_, m0_, l0_);
    split3(v1, h1_, m1_, l1_);
    dst[0] = (u32x4){h0_[0], h0_[1], h1_[0], h1_[1]};
    dst[plane] = (u32x4){m0_[0], m0_[1], m1_[0], m1_[1]};
    dst[2 * plane] = (u32x4){l0_[0], l0_[1], l1_[0], l1_[1]};
}

// One tile of xpanel_fwd_ps_kernel: 256 threads stage SR x 32 rows of X (from row m0) per slab; every wave owns RB 32-row blocks
// (from tile row `wrow`) of ONE 32-column block whose fragments it reads at `bq` (+ plane per piece, + kstride per 16-deep step).
// W0 tiles: RB = SR = 2, the four waves sit side by side on the same 64 rows.  Gram tiles: RB = 1, SR = 4, the four waves sit
// one above the other on 128 rows of the same column block (equal work per wave either way: 12 RB MFMAs a slab).
template <int NST, int RB, int SR>
__device__ __forceinline__ void xpanel_ps_tile(const XPanel& p, int b, int m0, int wrow, const u32x4* __restrict__ bq, long plane, long kstride,
                                               int kbeg, int nslab, unsigned short* lds, float* __restrict__ out, int ld, int n, bool nok,
                                               unsigned long long* trace) {
    constexpr int PLN = SR * 32 * SROW;                         // ushorts per LDS plane; lds: [buffer][piece][row][k]
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;                  // dev tracing (tests/dev/trace_xpanel_wg.py): 100 MHz wall ticks
    if (trace && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    const int R = p.S + p.Qn;
    // X staging map: float4 f = tid + 256 i -> tile row (tid >> 3) + 32 i, k offset (tid & 7) * 4
    const float* arow[SR];
#pragma unroll
    for (int i = 0; i < SR; ++i) {
        const int r = m0 + (tid >> 3) + 32 * i;
        arow[i] = xrow(p, b, r < R ? r : 0) + (((tid & 7) << 2) + kbeg);       // (rows past the panel: valid address, outputs not stored)
    }
    const int li = lane & 31, hh = lane >> 5;
    const int aoff = (wrow + li) * SROW + 8 * hh;
    f32x4 ga[NST][SR];
    bf16x8 fb[2][2][3];                          // [slot = slab & 1][step][piece]
    f32x16 acc[RB];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][i] = 0.f;

    auto gload = [&](auto sc, int k0) {
        constexpr int ST = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < SR; ++i) ga[ST][i] = *(const f32x4*)(arow[i] + k0);
    };
    auto bload = [&](auto sc, auto stc, int slab_) {                             // the fragments of one 16-deep step of slab `slab_`
        constexpr int SL = decltype(sc)::value & 1, st = decltype(stc)::value;
        const u32x4* q = bq + (long)(2 * slab_ + st) * kstride;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[SL][st][pl] = __builtin_bit_cast(bf16x8, q[pl * plane]);
    };
    auto lstore = [&](auto sc, int buf) {
        constexpr int ST = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < SR; ++i) {
            const int off = ((tid >> 3) + 32 * i) * SROW + ((tid & 7) << 2);
            u32x2 h, m, l;
            split3(ga[ST][i], h, m, l);
            unsigned short* d = lds + buf * 3 * PLN + off;
            *(u32x2*)d = h; *(u32x2*)(d + PLN) = m; *(u32x2*)(d + 2 * PLN) = l;
        }
    };
    // piece pairs in accumulation order (smallest first): (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    auto mma_slab = [&](auto sc, int buf) {
        constexpr int SL = decltype(sc)::value & 1;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 f[RB][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    f[rb][pl] = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds + (buf * 3 + pl) * PLN + aoff + rb * 32 * SROW + 16 * st));
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[rb][PA[q]], fb[SL][st][PB[q]], acc[rb], 0, 0, 0);
        }
    };
    auto slab = [&](auto sc, int s) {                                            // the last slabs: nothing (or not everything) left to fetch
        const int cur = s & 1;
        if (s + 1 < nslab) lstore(sc, cur ^ 1);
        if (s + 1 + NST < nslab) gload(sc, (s + 1 + NST) * SBK);
        mma_slab(sc, cur);
        if (s + 2 < nslab) { bload(sc, WgInt<0>{}, s + 2); bload(sc, WgInt<1>{}, s + 2); }
        __syncthreads();
    };
    // steady state, issue order spelled out (see xpanel_fwd_sb_tile): 3 SR segments of 4 RB / SR MFMAs plus one third of the work
    // that splits a float4 of the NEXT slab of X; when a step's last MFMA is out, the step's B registers are refilled for the
    // slab after next
    auto slab_main = [&](auto sc, int s) {
        constexpr int ST = decltype(sc)::value, SL = ST & 1;
        constexpr int NSEG = 3 * SR, MPS = 12 * RB / NSEG;
        static_assert(MPS * NSEG == 12 * RB, "MFMAs per segment");
        const int cur = s & 1, nxt = cur ^ 1;
        bf16x8 fa[RB][2][3];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    fa[rb][st][pl] = __builtin_bit_cast(bf16x8, *(const f32x4*)(lds + (cur * 3 + pl) * PLN + aoff + rb * 32 * SROW + 16 * st));
        __builtin_amdgcn_sched_barrier(0);
        unsigned hp[2], mp[2], lp[2];
        xp_static_for<0, NSEG>([&](auto gc) {                      // segment g: MFMAs MPS g .. + MPS - 1 of the slab's 12 RB, work item g
            constexpr int g = decltype(gc)::value, q = g / 3, part = g % 3;
#pragma unroll
            for (int u = 0; u < MPS; ++u) {
                const int m = MPS * g + u, st = m / (6 * RB), pr = (m / RB) % 6, rb = m % RB;
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb][st][PA[pr]], fb[SL][st][PB[pr]], acc[rb], 0, 0, 0);
            }
            const f32x4 v = ga[ST][q];
            if constexpr (part < 2) {
                split_pair(v[2 * part], v[2 * part + 1], hp[part], mp[part], lp[part]);
            } else {
                const int off = ((tid >> 3) + 32 * q) * SROW + ((tid & 7) << 2);
                unsigned short* d = lds + nxt * 3 * PLN + off;
                *(u32x2*)d = (u32x2){hp[0], hp[1]};
                *(u32x2*)(d + PLN) = (u32x2){mp[0], mp[1]};
                *(u32x2*)(d + 2 * PLN) = (u32x2){lp[0], lp[1]};
            }
            if constexpr (MPS > 1) {            // inside a segment: one MFMA, then a share of the item's vector operations / LDS writes
#pragma unroll
                for (int u = 0; u < MPS; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (part < 2) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((6 * RB - 1) / MPS == g) { bload(sc, WgInt<0>{}, s + 2); __builtin_amdgcn_sched_barrier(0); }       // step 0's last MFMA is out
            if constexpr ((12 * RB - 1) / MPS == g) { bload(sc, WgInt<1>{}, s + 2); __builtin_amdgcn_sched_barrier(0); }
        });
        gload(sc, (s + 1 + NST) * SBK);
        __syncthreads();
    };
    // prologue.  The loads are issued in the ORDER THE STEADY STATE LEAVES THEM IN (..., B(s), X(s+NST-1), B(s+1), X(s+NST) at the
    // top of slab s) and UNCONDITIONALLY (the launcher guarantees nslab > 2 NST): hipcc sizes every s_waitcnt vmcnt(n) in the loop
    // for the worst way into it, and a load it must assume skipped counts as not issued -- the loop then waits for all but its
    // newest load, i.e. the full memory latency every slab.
    gload(WgInt<0>{}, 0);
    bload(WgInt<0>{}, WgInt<0>{}, 0); bload(WgInt<0>{}, WgInt<1>{}, 0);
    lstore(WgInt<0>{}, 0);
    xp_static_for<0, NST - 1>([&](auto ic) { gload(ic, (decltype(ic)::value + 1) * SBK); });
    bload(WgInt<1>{}, WgInt<0>{}, 1); bload(WgInt<1>{}, WgInt<1>{}, 1);
    gload(WgInt<NST - 1>{}, NST * SBK);
    __syncthreads();
    if (trace && tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (; s + 2 * NST < nslab; s += NST)
        xp_static_for<0, NST>([&](auto ic) { slab_main(ic, s + decltype(ic)::value); });
    for (; s < nslab; s += NST)
        xp_static_for<0, NST>([&](auto ic) { if (s + decltype(ic)::value < nslab) slab(ic, s + decltype(ic)::value); });
    if (trace && tid == 0) {
        t2 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* t = trace + (long)blockIdx.x * 6;
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = RB;
        t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    }
    if (!nok) return;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wrow + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m < R) out[(long)m * ld + n] = acc[rb][r];
        }
}

// 256 threads.  Ids (after the XCD decode) per episode: tiles_m x (h0 / 128) x ksplit W0 tiles of 64 rows x 128 columns, then
// tiles_g x cbg Gram tiles of 128 rows x 32 columns.  All W0 tiles of the launch come before all Gram tiles.
// RIDER: the first rider.nblk workgroups (a multiple of 8, so the XCD grouping of the rest is unchanged) run the split hypernetwork
// forward (hyper_fwd.h) in this kernel's LDS and leave; they are dispatched first and are done in ~10 us.
template <int NST, bool RIDER>
__global__ __launch_bounds__(256, 2) void xpanel_fwd_ps_kernel(XPanel p, const unsigned short* __restrict__ Wp, const unsigned short* __restrict__ Xp,
                                                                float* __restrict__ A0, float* __restrict__ G, int tiles_m, int tiles_g, int cbg,
                                                                HyperFwdArgs rider, unsigned long long* trace) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 3 * 128 * SROW];
    int bid = blockIdx.x;
    if constexpr (RIDER) {
        __shared__ int s_last;
        if (bid < rider.nblk) { hyper_fwd_split_body<HF_RIDER_KS, true>(rider, bid, (float*)lds, &s_last); return; }
        bid -= rider.nblk;
    }
    const int xcd = bid & 7, j = bid >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31;
    const int R = p.S + p.Qn, K = p.D, CB = p.h0 >> 5, tiles_w = CB >> 2;
    const int ks = p.ksplit, tw = tiles_m * tiles_w, heavy_per = tw * ks, nheavy = ((p.B + 7) >> 3) * heavy_per;
    if (j < nheavy) {
        const int b = xcd + 8 * (j / heavy_per), tz = j % heavy_per, kz = tz / tw, t = tz - kz * tw;
        if (b >= p.B) return;
        const int kbeg = kz * (K / ks), cb = (t % tiles_w) * 4 + wave;
        xpanel_ps_tile<NST, 2, 2>(p, b, (t / tiles_w) * 64, 0, (const u32x4*)Wp + ((long)(kbeg >> 4) * CB + cb) * 64 + lane,
                                  (long)p.h0 * (K >> 3), CB * 64L, kbeg, K / ks / SBK, lds,
                                  A0 + kz * p.part_stride + (long)b * R * p.h0, p.h0, cb * 32 + li, true, trace);
    } else {
        const int jj = j - nheavy, per = max(1, tiles_g * cbg);
        const int b = xcd + 8 * (jj / per), t = jj % per, g = t % cbg;
        if (b >= p.B) return;
        const long plane = (long)cbg * 32 * (K >> 3);
        xpanel_ps_tile<NST, 1, 4>(p, b, (t / cbg) * 128, wave * 32, (const u32x4*)Xp + (long)b * 3 * plane + (long)g * 64 + lane,
                                  plane, cbg * 64L, 0, K / SBK, lds, G + (long)b * R * p.S, p.S, g * 32 + li, g * 32 + li < p.S, trace);
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// slab[z][i, j] = sum over global rows g in [z*kchunk, (z+1)*kchunk) of Abar0[g, i] * X(g)[j];  g = b*R + r
template <bool FAST>     // FAST: h0 % 64 == 0, D % 64 == 0, aligned -> unconditional float4 staging loads (see forward)
__global__ __launch_bounds__(256) void xpanel_bwd_kernel(XPanel p, const float* __restrict__ Abar, float* __restrict__ slabs,
                                                         int kchunk) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.S + p.Qn, M = p.h0, Nn = p.D;
    const long Ktot = (long)p.B * R;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const long kbeg = (long)blockIdx.z * kchunk;
    const long kend = min(Ktot, kbeg + kchunk);
    float* C = slabs + (long)blockIdx.z * M * Nn;

    // staging map: float4 f -> contraction row (f>>4), columns (f&15)*4 .. +3 of the tile.  The (episode, row) of each
    // thread's contraction row is tracked incrementally (+32 rows per slab) instead of dividing by R every slab.
    bool okf[2] = {true, true};
    int gb[2], gr[2];                        // episode / row-in-panel of contraction row  kbeg + s*BK + (f>>4)
    const float* xp[2];                      // its X row, looked up one slab ahead (indexed rows: a dependent index load)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long g = kbeg + ((tid + 256 * i) >> 4);
        gb[i] = (int)(g / R); gr[i] = (int)(g - (long)gb[i] * R);
        xp[i] = xrow(p, g < kend ? gb[i] : 0, g < kend ? gr[i] : 0);
    }
    auto load = [&](f32x4 (&ra)[2], f32x4 (&rb)[2], long k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            const long g = k0 + (f >> 4);
            const int c4 = (f & 15) << 2;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const bool ok = g < kend;
            const int b = ok ? gb[i] : 0, r = ok ? gr[i] : 0;
            const float* pa = Abar + ((long)b * R + r) * M + m0 + c4;
            const float* pb = xp[i] + n0 + c4;
            if (FAST) {
                ra[i] = *(const f32x4*)pa;                  // raw; masked when written to LDS
                rb[i] = *(const f32x4*)pb;
                okf[i] = ok;
            } else {
                f32x4 va = zero4, vb = zero4;
                for (int e = 0; e < 4; ++e) if (ok && m0 + c4 + e < M) va[e] = pa[e];
                for (int e = 0; e < 4; ++e) if (ok && n0 + c4 + e < Nn) vb[e] = pb[e];
                ra[i] = va; rb[i] = vb;
            }
            gr[i] += BK;                                    // next slab's row
            while (gr[i] >= R) { gr[i] -= R; ++gb[i]; }
            const bool okn = g + BK < kend;
            xp[i] = xrow(p, okn ? gb[i] : 0, okn ? gr[i] : 0);
        }
    };
    auto store = [&](const f32x4 (&ra)[2], const f32x4 (&rb)[2], int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(lds[buf][0] + (f >> 4) * 64 + ((f & 15) << 2)) = (!FAST || okf[i]) ? ra[i] : zero4;
            *(f32x4*)(lds[buf][1] + (f >> 4) * 64 + ((f & 15) << 2)) = (!FAST || okf[i]) ? rb[i] : zero4;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 ra[2], rb[2];
    const int nslab = (int)((kend - kbeg + BK - 1) / BK);
    if (nslab > 0) { load(ra, rb, kbeg); store(ra, rb, 0); }
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        if (s + 1 < nslab) load(ra, rb, kbeg + (long)(s + 1) * BK);
        const float* TA = lds[cur][0] + wm * 32 + li;
        const float* TB = lds[cur][1] + wn * 32 + li;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = 2 * k2 + kh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(TA[k * 64], TB[k * 64], acc, 0, 0, 0);
        }
        if (s + 1 < nslab) store(ra, rb, cur ^ 1);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + li;
    if (n < Nn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < M) C[(long)m * Nn + n] = acc[r];
        }
    }
}

// ---- backward, 256 x 64 tiles (h0 % 256 == 0, D % 64 == 0, aligned): every X element is fetched by exactly one workgroup ----
// The 64 x 64 kernel reads each X column block once per 64-row group of gW0 (four times at h0 = 256): 97 MB of HBM reads per
// launch for 54.6 MB of operands.  Here a workgroup owns all 256 rows of a 64-column block, so X is streamed once and only
// the small Abar0 panel (L2 / Infinity-Cache resident) is re-read.  8 waves as 4 (M) x 2 (N), two 32x32 accumulators each;
// 32-deep slabs double-buffered in 80 KB of LDS (two workgroups per CU), next slab prefetched to registers.
// RIDER: the first rider.nblk workgroups (a multiple of 8) run the hypernetwork backward (hyper_bwd.h) in this kernel's LDS.
template <bool RIDER>
__global__ __launch_bounds__(512) void xpanel_bwd256_kernel(XPanel p, const float* __restrict__ Abar, float* __restrict__ slabs,
                                                            int kchunk, int nsplit, int tiles_n, int tiles_m, HyperBwdArgs rider) {
    extern __shared__ __attribute__((aligned(16))) float lds256[];       // [2][ A: 32 x 256 | B: 32 x 64 ]
    constexpr int ASZ = 32 * 256, BSZ = 32 * 64, STG = ASZ + BSZ;
    int bid = blockIdx.x;
    if constexpr (RIDER) {
        if (bid < rider.nblk) { hyper_bwd_body(rider, bid, lds256); return; }
        bid -= rider.nblk;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.S + p.Qn, M = p.h0, Nn = p.D;
    const long Ktot = (long)p.B * R;
    // XCD-aware ids (workgroup ids equal mod 8 share an XCD): all tiles of contraction slab z run on XCD z % 8, so the slab's
    // rows of Abar0 (0.4-0.8 MB) stay in that XCD's L2 while its 32 column tiles stream their X pieces
    const int tiles = tiles_n * tiles_m;
    const int xcd = bid & 7, jq = bid >> 3;
    const int z = xcd + 8 * (jq / tiles), tl = jq % tiles;
    if (z >= nsplit) return;
    const int m0 = (tl / tiles_n) * 256, n0 = (tl % tiles_n) * 64;
    const long kbeg = (long)z * kchunk;
    const long kend = min(Ktot, kbeg + kchunk);
    float* C = slabs + (long)z * M * Nn;

    // staging map: A float4 f = tid + 512 i (i < 4) -> contraction row f >> 6, columns (f & 63) * 4;  B float4 tid -> row
    // tid >> 4, columns (tid & 15) * 4.  Abar0 is [B*R, h0] contiguous; the X row of B goes through xrow (looked up a slab ahead).
    const int ac4 = (tid & 63) << 2, bc4 = (tid & 15) << 2;
    int gb, gr;
    {
        const long g = kbeg + (tid >> 4);
        gb = (int)(g / R); gr = (int)(g - (long)gb * R);
    }
    const float* xp = xrow(p, kbeg + (tid >> 4) < kend ? gb : 0, kbeg + (tid >> 4) < kend ? gr : 0);
    f32x4 ra[4], rb; bool oka[4], okb;
    auto load = [&](long k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long g = k0 + (tid >> 6) + 8 * i;
            oka[i] = g < kend;
            ra[i] = *(const f32x4*)(Abar + (oka[i] ? g : 0) * M + m0 + ac4);      // raw; masked when written to LDS
        }
        const long g = k0 + (tid >> 4);
        okb = g < kend;
        rb = *(const f32x4*)(xp + n0 + bc4);
        gr += 32;
        while (gr >= R) { gr -= R; ++gb; }
        const bool okn = g + 32 < kend;
        xp = xrow(p, okn ? gb : 0, okn ? gr : 0);
    };
    auto store = [&](int buf) {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        float* Ab = lds256 + buf * STG; float* Bb = Ab + ASZ;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(Ab + ((tid >> 6) + 8 * i) * 256 + ac4) = oka[i] ? ra[i] : zero4;
        *(f32x4*)(Bb + (tid >> 4) * 64 + bc4) = okb ? rb : zero4;
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int nslab = (int)((kend - kbeg + 31) / 32);
    if (nslab > 0) { load(kbeg); store(0); }
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        if (s + 1 < nslab) load(kbeg + (long)(s + 1) * 32);
        const float* TA = lds256 + cur * STG + wm * 64 + li;
        const float* TB = lds256 + cur * STG + ASZ + wn * 32 + li;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = 2 * k2 + kh;
            const float a0 = TA[k * 256], a1 = TA[k * 256 + 32], b0 = TB[k * 64];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc1, 0, 0, 0);
        }
        if (s + 1 < nslab) store(cur ^ 1);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        C[(long)m * Nn + n] = acc0[r];
        C[(long)(m + 32) * Nn + n] = acc1[r];
    }
}

// ---- backward on the bf16 matrix pipe with fp32-equivalent accuracy (the forward kernel's split) -----------------
// Same 256 x 64 tiles, slabs of the contraction and XCD-aware ids as xpanel_bwd256_kernel, but every fp32 operand is split exactly
// into three bf16 pieces when its slab goes to LDS and the six piece products of weight >= 2^-16 run on v_mfma_f32_32x32x16_bf16
// (a quarter of the cycles of the eight fp32 MFMAs they replace).  The operands are k-major here (a row of Abar0 / of X is one
// contraction index), while a bf16 MFMA lane needs eight CONSECUTIVE k of one output row: the LDS images stay [k][m] (coalesced
// 8-byte writes of four split values) and the fragments are read with gfx950's transposing ds_read_b64_tr_b16 -- per 16-lane
// group a 4 (k) x 16 (m) block, lane i receiving column i's four k values.  Row strides 576 B (A: 256 m) / 192 B (B: 64 n) put
// the four rows a 32-lane half reads into four different quarters of the 64 banks: conflict-free.
//   16-deep slabs (one MFMA k-step), double-buffered: 2 x (3 x 16 x 576 + 3 x 16 x 192) B = 72 KB -> two workgroups per CU.
//   8 waves as 4 (M) x 2 (N), two 32x32 accumulators each; register ring of 2 slabs of global loads.
constexpr int BSK = 16;
constexpr int BRSA = 288;                             // ushorts per LDS row of the A image (576 B)
__host__ __device__ constexpr int bsb_buf_ushorts(int nb, int sk) { return 3 * sk * BRSA + 3 * sk * (nb == 1 ? 96 : 160); }
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* a, int row4_ushorts) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + row4_ushorts));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NB = 32-column blocks per wave in N: 1 -> 256 x 64 tiles (two workgroups per CU), 2 -> 256 x 128 tiles (half the re-reads of the
// Abar0 panel, four accumulators per wave, 92 KB of LDS: one workgroup per CU)
// SK = contraction rows per slab: 16 (one MFMA k-step; 72 KB of LDS at NB = 1: two workgroups per CU) or 32 (two k-steps between
// barriers, 144 KB: one workgroup per CU)
// SWAP (narrow gradients, h0 = 64: AM3's image encoder): the 256-wide side of the tile is a block of X's COLUMNS (rows through xrow)
// and the 64-wide side the plain matrix Abar [K, 64]; the tile is the transpose of a 64 x 256 block of gW and leaves through LDS
// so that its rows are stored contiguously.  (NB = 1, SK = 16, no rider.)
template <bool RIDER, int NST, int NB, int SK, bool SWAP = false>
__global__ __launch_bounds__(512, (NB == 1 && SK == 16) ? 2 : 1) void xpanel_bwd256_sb_kernel(XPanel p, const float* __restrict__ Abar, float* __restrict__ slabs,
                                                                   int kchunk, int nsplit, int tiles_n, int tiles_m, HyperBwdArgs rider) {
    extern __shared__ __attribute__((aligned(16))) float lds256[];
    unsigned short* const L = (unsigned short*)lds256;            // [2][ A h|m|l : 16 x 288 | B h|m|l : 16 x BRSB ]
    constexpr int TN = 64 * NB;                                   // tile columns
    constexpr int BRSB = NB == 1 ? 96 : 160;                      // ushorts per B row: 192 B / 320 B (both = 16 or 48 mod 64 dwords)
    constexpr int BPA = SK * BRSA, BPB = SK * BRSB, BBUF = 3 * BPA + 3 * BPB;
    constexpr int KS = SK / 16, NA = SK / 8;                      // MFMA k-steps per slab; A float4 per thread and slab
    int bid = blockIdx.x;
    if constexpr (RIDER) {
        if (bid < rider.nblk) { hyper_bwd_body(rider, bid, lds256); return; }
        bid -= rider.nblk;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.S + p.Qn, M = SWAP ? p.D : p.h0, Nn = SWAP ? p.h0 : p.D;        // wide / narrow extents of the product
    const long Ktot = (long)p.B * R;
    const int tiles = tiles_n * tiles_m;
    const int xcd = bid & 7, jq = bid >> 3;
    const int z = xcd + 8 * (jq / tiles), tl = jq % tiles;
    if (z >= nsplit) return;
    const int m0 = (tl / tiles_n) * 256, n0 = (tl % tiles_n) * TN;
    const long kbeg = (long)z * kchunk;
    const long kend = min(Ktot, kbeg + kchunk);
    float* C = slabs + (long)z * M * Nn;

    // staging map: A float4 f = tid + 512 i (i < NA) -> contraction row f >> 6, columns (f & 63) * 4; B float4 tid -> row tid >> BL,
    // columns (tid & (2^BL - 1)) * 4 (NB = 1: BL = 4 and waves 4..7 carry no B element; NB = 2: BL = 5, every thread has one)
    constexpr int BL = NB == 1 ? 4 : 5;
    const int ac4 = (tid & 63) << 2, bc4 = (tid & ((1 << BL) - 1)) << 2;
    const int brow = tid >> BL;
    const bool hasb = brow < SK;
    // the virtual rows of X this thread fetches: its B row (brow), or with SWAP its NA A rows; tracked incrementally (+SK per slab)
    constexpr int NXR = SWAP ? NA : 1;
    int gb[NXR], gr[NXR]; const float* xp[NXR];
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
        const int row = SWAP ? (tid >> 6) + 8 * i : brow;
        const long g = kbeg + row;
        gb[i] = (int)(g / R); gr[i] = (int)(g - (long)gb[i] * R);
        const bool ok = (SWAP || hasb) && g < kend;
        xp[i] = xrow(p, ok ? gb[i] : 0, ok ? gr[i] : 0);
    }
    f32x4 ra[NST][NA], rb[NST]; bool oka[NST][NA], okb[NST];
    auto gload = [&](auto sc, long k0) {
        constexpr int ST = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const long g = k0 + (tid >> 6) + 8 * i;
            oka[ST][i] = g < kend;
            if constexpr (SWAP) ra[ST][i] = *(const f32x4*)(xp[i] + m0 + ac4);
            else ra[ST][i] = *(const f32x4*)(Abar + (oka[ST][i] ? g : 0) * M + m0 + ac4);        // raw; masked when written to LDS
        }
        const long g = k0 + brow;
        okb[ST] = hasb && g < kend;
        if constexpr (SWAP) rb[ST] = *(const f32x4*)(Abar + (okb[ST] ? g : 0) * Nn + n0 + bc4);
        else rb[ST] = *(const f32x4*)(xp[0] + n0 + bc4);
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int row = SWAP ? (tid >> 6) + 8 * i : brow;
            gr[i] += SK;
            while (gr[i] >= R) { gr[i] -= R; ++gb[i]; }
            const bool okn = (SWAP || hasb) && k0 + row + SK < kend;
            xp[i] = xrow(p, okn ? gb[i] : 0, okn ? gr[i] : 0);
        }
    };
    auto lstore = [&](auto sc, int buf) {
        constexpr int ST = decltype(sc)::value;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        unsigned short* Ab = L + buf * BBUF; unsigned short* Bb = Ab + 3 * BPA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 h, m, l;
            split3(oka[ST][i] ? ra[ST][i] : zero4, h, m, l);
            const int off = ((tid >> 6) + 8 * i) * BRSA + ac4;
            *(u32x2*)(Ab + off) = h; *(u32x2*)(Ab + BPA + off) = m; *(u32x2*)(Ab + 2 * BPA + off) = l;
        }
        if (hasb) {
            u32x2 h, m, l;
            split3(okb[ST] ? rb[ST] : zero4, h, m, l);
            const int off = brow * BRSB + bc4;
            *(u32x2*)(Bb + off) = h; *(u32x2*)(Bb + BPB + off) = m; *(u32x2*)(Bb + 2 * BPB + off) = l;
        }
    };
    // fragment addresses: 16-lane group g = lane >> 4 reads the 4 x 16 block rows 8 (g >> 1) + {0..3} (second read: + 4), columns
    // 16 (g & 1) ..; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p + 3
    const int grp = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    const int arow = 8 * (grp >> 1) + fq;
    const int aoff = arow * BRSA + wm * 64 + 16 * (grp & 1) + 4 * fp;       // + 32 for the wave's second 32-row block
    const int boff = arow * BRSB + wn * 32 * NB + 16 * (grp & 1) + 4 * fp;      // + 32 for the wave's second 32-column block (NB = 2)
    f32x16 acc[2][NB];                                            // [m block][n block]
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y_ = 0; y_ < NB; ++y_) acc[x][y_][i] = 0.f;
    const int nslab = (int)((kend - kbeg + SK - 1) / SK);
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};     // piece pairs, smallest first: (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
    auto mma_slab = [&](int buf) {
        const unsigned short* Ab = L + buf * BBUF; const unsigned short* Bb = Ab + 3 * BPA;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a[2][3], b[NB][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[0][pl] = tr_frag(Ab + pl * BPA + aoff + 16 * ks * BRSA, 4 * BRSA);
                a[1][pl] = tr_frag(Ab + pl * BPA + aoff + 16 * ks * BRSA + 32, 4 * BRSA);
#pragma unroll
                for (int y_ = 0; y_ < NB; ++y_) b[y_][pl] = tr_frag(Bb + pl * BPB + boff + 16 * ks * BRSB + 32 * y_, 4 * BRSB);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y_ = 0; y_ < NB; ++y_)
                        acc[x][y_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][PA[u]], b[y_][PB[u]], acc[x][y_], 0, 0, 0);
        }
    };
    // prologue: slab 0 straight to LDS, slabs 1..NST into the ring (slot of slab q is (q - 1) % NST)
    if (nslab > 0) { gload(WgInt<0>{}, kbeg); lstore(WgInt<0>{}, 0); }
    auto slab = [&](auto sc, int s_) {
        const int cur = s_ & 1;
        if (s_ + 1 < nslab) lstore(sc, cur ^ 1);
        if (s_ + 1 + NST < nslab) gload(sc, kbeg + (long)(s_ + 1 + NST) * SK);
        mma_slab(cur);
        __syncthreads();
    };
    // steady state (s_ + 1 + NST < nslab), issue order spelled out like the forward kernel's: the slab's 18 fragment reads, then six
    // segments of two MFMAs plus a sixth of the work that splits the NEXT slab (one pair of values; every second segment ends
    // with the three LDS writes of a finished float4).  sched_barrier(0) pins the segments.
    auto slab_main = [&](auto sc, int s_) {
        constexpr int ST = decltype(sc)::value;
        const int cur = s_ & 1;
        const unsigned short* Ab = L + cur * BBUF; const unsigned short* Bb = Ab + 3 * BPA;
        unsigned short* An = L + (cur ^ 1) * BBUF; unsigned short* Bn = An + 3 * BPA;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        unsigned hp[2], mp[2], lp[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a[2][3], b[NB][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[0][pl] = tr_frag(Ab + pl * BPA + aoff + 16 * ks * BRSA, 4 * BRSA);
                a[1][pl] = tr_frag(Ab + pl * BPA + aoff + 16 * ks * BRSA + 32, 4 * BRSA);
#pragma unroll
                for (int y_ = 0; y_ < NB; ++y_) b[y_][pl] = tr_frag(Bb + pl * BPB + boff + 16 * ks * BRSB + 32 * y_, 4 * BRSB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 6; ++u) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y_ = 0; y_ < NB; ++y_)
                        acc[x][y_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][PA[u]], b[y_][PB[u]], acc[x][y_], 0, 0, 0);
                // split pair j of the staged slab: float4 q = j / 2 (A rows i = 0 .. NA - 1, then B), its values 2 part, 2 part + 1
                const int j = 6 * ks + u, q = j >> 1, part = j & 1;
                if (q <= NA) {
                    const f32x4 v = q < NA ? (oka[ST][q < NA ? q : 0] ? ra[ST][q < NA ? q : 0] : zero4) : (okb[ST] ? rb[ST] : zero4);
                    split_pair(v[2 * part], v[2 * part + 1], hp[part], mp[part], lp[part]);
                    if (part == 1) {
                        if (q < NA) {
                            const int off = ((tid >> 6) + 8 * q) * BRSA + ac4;
                            *(u32x2*)(An + off) = (u32x2){hp[0], hp[1]};
                            *(u32x2*)(An + BPA + off) = (u32x2){mp[0], mp[1]};
                            *(u32x2*)(An + 2 * BPA + off) = (u32x2){lp[0], lp[1]};
                        } else if (hasb) {
                            const int off = brow * BRSB + bc4;
                            *(u32x2*)(Bn + off) = (u32x2){hp[0], hp[1]};
                            *(u32x2*)(Bn + BPB + off) = (u32x2){mp[0], mp[1]};
                            *(u32x2*)(Bn + 2 * BPB + off) = (u32x2){lp[0], lp[1]};
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gload(sc, kbeg + (long)(s_ + 1 + NST) * SK);
        __syncthreads();
    };
    int s_ = 0;
    static_assert(NST == 2 || NST == 3, "ring depth");
    if (nslab > 2 * NST) {
        // unconditional ring loads in front of the steady-state loop (see xpanel_fwd_sb_tile: the loop's s_waitcnt vmcnt(n) are sized
        // for the worst way in)
        xp_static_for<0, NST>([&](auto ic) { gload(ic, kbeg + (long)(decltype(ic)::value + 1) * SK); });
        __syncthreads();
        for (; s_ + 2 * NST < nslab; s_ += NST)                  // every slab of this round has s' + 1 + NST < nslab
            xp_static_for<0, NST>([&](auto ic) { slab_main(ic, s_ + decltype(ic)::value); });
    } else {
        xp_static_for<0, NST>([&](auto ic) { if (decltype(ic)::value + 1 < nslab) gload(ic, kbeg + (long)(decltype(ic)::value + 1) * SK); });
        __syncthreads();
    }
    for (; s_ < nslab; s_ += NST)
        xp_static_for<0, NST>([&](auto ic) { if (s_ + decltype(ic)::value < nslab) slab(ic, s_ + decltype(ic)::value); });
    const int li = lane & 31, kh = lane >> 5;
    if constexpr (SWAP) {
        // transpose through LDS (the last slab's barrier has passed: every wave is done reading): T[n][m], rows of 256 + 4 floats
        float* T = lds256;
        constexpr int TLD = 256 + 4;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[(wn * 32 + li) * TLD + wm * 64 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * kh] = acc[x][0][r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = tid + 512 * j, row = f >> 6, c4 = (f & 63) << 2;
            *(f32x4*)(C + (long)(n0 + row) * M + m0 + c4) = *(const f32x4*)(T + row * TLD + c4);
        }
        return;
    }
#pragma unroll
    for (int y_ = 0; y_ < NB; ++y_) {
        const int n = n0 + wn * 32 * NB + 32 * y_ + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            C[(long)m * Nn + n] = acc[0][y_][r];
            C[(long)(m + 32) * Nn + n] = acc[1][y_][r];
        }
    }
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }
unsigned long long* g_trace = nullptr;      // dev tracing only (tools/trace_xpanel.py)

}  // namespace

void set_xpanel_trace(void* p) { g_trace = (unsigned long long*)p; }

// contraction parts of the split-bf16 forward: 1 unless the output is narrow (no Gram block, few column tiles) and the tile count
// would leave most of the chip idle -- AM3's image encoder (P = 64 columns: 96 workgroups for 256 CUs, 46 us for a pass that
// is HBM-bound at 8 us).  The caller then passes `parts` ([ks, B, S+Qn, h0] floats) and the launch ends with their sum.
int xpanel_fwd_ksplit(int B, int S, int Qn, int D, int h0, int with_gram) {
    static const int off = getenv("FUMI_XP_KSPLIT") ? atoi(getenv("FUMI_XP_KSPLIT")) : -1;    // 1: never split, n: force n parts
    if (with_gram || D % SBK) return 1;
    const long wgs = 8L * ((B + 7) / 8) * ((S + Qn + 63) / 64) * ((h0 + 63) / 64);
    int ks = 1;
    if (off > 0) ks = off;
    else while (ks < 8 && wgs * ks < 384 && D % (2 * ks * SBK) == 0 && D / (2 * ks) >= 8 * SBK) ks *= 2;
    while (ks > 1 && (D % (ks * SBK) != 0)) ks >>= 1;
    return ks;
}

constexpr int PS_MIN_SLABS = 2 * 4 + 1;          // xpanel_fwd_ps_kernel's prologue fetches NST + 1 <= 5 slabs unconditionally
static bool xpanel_fwd_ps_ok(int D, int h0) {
    static const int on = getenv("FUMI_XP_PS") ? atoi(getenv("FUMI_XP_PS")) : 1;       // 0: every tile splits its own column operand
    static const int sb = getenv("FUMI_XP_SB") ? atoi(getenv("FUMI_XP_SB")) : 1;
    return on && sb && h0 % 128 == 0 && D % SBK == 0;
}
static size_t ps_w0_bytes(int D, int h0) { return (size_t)3 * h0 * D * sizeof(unsigned short); }
static size_t ps_xs_bytes(int B, int S, int D) { return (size_t)B * 3 * ((S + 31) / 32 * 32) * D * sizeof(unsigned short); }

unsigned short* xpanel_planes(fumi_ws* ws, int B, int S, int D, int h0) {
    if (!ws || !xpanel_fwd_ps_ok(D, h0)) return nullptr;
    const size_t need = ps_w0_bytes(D, h0) + ps_xs_bytes(B, S, D);
    if (ws->w0p_cap < need) {
        if (ws->w0p) (void)hipFree(ws->w0p);          // (hipFree waits for the device: nothing in flight reads the old planes)
        ws->w0p = nullptr; ws->w0p_cap = 0;
        if (hipMalloc((void**)&ws->w0p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // the 64 x 64 kernel still works
        ws->w0p_cap = need;
    }
    return ws->w0p;
}

bool xpanel_fwd_presplits(int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q, const float* W0, bool gram,
                          const XRows* rows, unsigned short* planes) {
    if (rows && rows->table) x_s = x_q = rows->table;
    const bool aligned = al16(x_s) && al16(x_q) && al16(W0);
    // (a split contraction -- narrow outputs with a `parts` buffer, AM3's encoder -- takes the 64 x 64 kernel; callers that pass no
    // `parts` never split)
    return aligned && planes && xpanel_fwd_ps_ok(D, h0) && D / SBK >= PS_MIN_SLABS;
}

int launch_xpanel_fwd(hipStream_t st, int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q,
                      const float* W0, float* A0, float* G, const XRows* rows, const HyperFwdArgs* rider, int* rider_done,
                      float* parts, int* parts_unreduced, unsigned short* planes, GlovePending* glove) {
    if (parts_unreduced) *parts_unreduced = 0;
    if (rider_done) *rider_done = 0;
    if (glove && !glove->on) glove = nullptr;
    float* const A0_final = A0;
    XPanel p{x_s, x_q, W0, B, S, Qn, D, h0, nullptr, nullptr, nullptr, 0, G ? S : 0, 1, 0};
    if (rows && rows->table) { p.table = rows->table; p.idx_s = rows->idx_s; p.idx_q = rows->idx_q; p.n_rows = rows->n_rows; p.x_s = p.x_q = rows->table; }
    const int tiles_m = (S + Qn + 63) / 64, tiles_n = (h0 + p.gcols + 63) / 64;
    const int nper = (B + 7) / 8;
    const bool aligned = al16(p.x_s) && al16(p.x_q) && al16(W0);
    static const int use_sb = getenv("FUMI_XP_SB") ? atoi(getenv("FUMI_XP_SB")) : 1;
    const int ks = xpanel_fwd_ksplit(B, S, Qn, D, h0, G != nullptr);
    if (ks > 1 && parts && aligned && D % SBK == 0 && use_sb) { p.ksplit = ks; p.part_stride = (long)B * (S + Qn) * h0; A0 = parts; }
    const dim3 grid(8 * nper * tiles_m * tiles_n * p.ksplit);
    static const int nst = getenv("FUMI_XP_NST") ? atoi(getenv("FUMI_XP_NST")) : 2;      // staging ring depth (tuning knob)
    // Default: split-bf16 (error against fp64 at the fp32 MFMA kernel's level: DESIGN.md) -- with the column operand split once per
    // step where the shape allows (50 + 6 us at the bench shapes), else per tile (70 us).  FUMI_XP_SB=0: the fp32 MFMA kernel (78 us).
    const bool presplit = aligned && planes && p.ksplit == 1 && xpanel_fwd_ps_ok(D, h0) && D / SBK >= PS_MIN_SLABS;
    if (glove && !presplit) {                                   // (callers check xpanel_fwd_presplits first: not expected)
        glove->on = 0;
        if (glove->vec) hipLaunchKernelGGL(xpanel_presplit_glove_kernel<true>, dim3((unsigned)glove->a.R), dim3(512), glove->lds, st, p, 0, nullptr, nullptr, glove->a, glove->a.R);
        else hipLaunchKernelGGL(xpanel_presplit_glove_kernel<false>, dim3((unsigned)glove->a.R), dim3(512), glove->lds, st, p, 0, nullptr, nullptr, glove->a, glove->a.R);
        glove = nullptr;
    }
    if (presplit) {      // (split contractions: the 64 x 64 kernel)
        // column operands split once (bf16 planes in fragment order), then tiles that only split X: 64 x 128 against W0, 128 x 32
        // against the support rows
        static const int ride = getenv("FUMI_XP_RIDER") ? atoi(getenv("FUMI_XP_RIDER")) : 1;
        const int cbg = (p.gcols + 31) / 32;
        unsigned short* Wp = planes; unsigned short* Xp = planes + ps_w0_bytes(D, h0) / sizeof(unsigned short);
        const long nfrag = ((long)h0 / 32 + (long)B * cbg) * (D / 16);      // one wave per fragment
        if (glove) {                                                        // the pending embedding bag rides in front (its rows first)
            const unsigned ng = (unsigned)glove->a.R, nb = (unsigned)((nfrag + 7) / 8);
            if (glove->vec) hipLaunchKernelGGL(xpanel_presplit_glove_kernel<true>, dim3(ng + nb), dim3(512), glove->lds, st, p, cbg, Wp, Xp, glove->a, (int)ng);
            else hipLaunchKernelGGL(xpanel_presplit_glove_kernel<false>, dim3(ng + nb), dim3(512), glove->lds, st, p, cbg, Wp, Xp, glove->a, (int)ng);
            glove->on = 0; glove = nullptr;
        } else
        hipLaunchKernelGGL(xpanel_presplit_kernel, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, st, p, cbg, Wp, Xp);
        const int tm = (S + Qn + 63) / 64, tg = (S + Qn + 127) / 128;
        const unsigned nwg = 8u * nper * (tm * (h0 / 128) * p.ksplit + tg * cbg);
        HyperFwdArgs none; memset(&none, 0, sizeof(none));
        if (ride && rider && rider_done && rider->nblk > 0 && rider->nblk % 8 == 0 && rider->d.Dt <= HF_RIDER_MAXDT &&
            hyper_fwd_split_lds_bytes(rider->d.ldx) <= sizeof(unsigned short) * 2 * 3 * 128 * SROW) {
            hipLaunchKernelGGL((xpanel_fwd_ps_kernel<2, true>), dim3(nwg + rider->nblk), dim3(256), 0, st, p, Wp, Xp, A0, G, tm, tg, cbg, *rider, g_trace);
            *rider_done = 1;
        } else {
            hipLaunchKernelGGL((xpanel_fwd_ps_kernel<2, false>), dim3(nwg), dim3(256), 0, st, p, Wp, Xp, A0, G, tm, tg, cbg, none, g_trace);
        }
    } else if (aligned && D % SBK == 0 && use_sb) {
        static const int sbn = getenv("FUMI_XP_SBN") ? atoi(getenv("FUMI_XP_SBN")) : 2;          // ring depth (tuning knob)
        static const int ride = getenv("FUMI_XP_RIDER") ? atoi(getenv("FUMI_XP_RIDER")) : 1;     // 0: never carry the hypernetwork forward
        HyperFwdArgs none; memset(&none, 0, sizeof(none));
        if (ride && rider && rider_done && rider->nblk > 0 && rider->nblk % 8 == 0 && rider->d.Dt <= HF_RIDER_MAXDT &&
            hyper_fwd_split_lds_bytes(rider->d.ldx) <= sizeof(unsigned short) * 2 * 2 * 3 * SPLANE && sbn <= 2) {
            hipLaunchKernelGGL((xpanel_fwd_sb_kernel<2, true>), dim3(grid.x + rider->nblk), dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, *rider);
            *rider_done = 1;
        }
        else if (sbn <= 2) hipLaunchKernelGGL((xpanel_fwd_sb_kernel<2, false>), grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, none);
        else hipLaunchKernelGGL((xpanel_fwd_sb_kernel<4, false>), grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, none);
    } else if (aligned && D % FBK == 0) {
        if (nst == 1) hipLaunchKernelGGL(xpanel_fwd_kernel<1>, grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, g_trace);
        else if (nst == 2) hipLaunchKernelGGL(xpanel_fwd_kernel<2>, grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, g_trace);
        else hipLaunchKernelGGL(xpanel_fwd_kernel<3>, grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, g_trace);
    }
    else if (aligned && D % BK == 0) hipLaunchKernelGGL(xpanel_fwd_generic_kernel<true>, grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, 0);
    else hipLaunchKernelGGL(xpanel_fwd_generic_kernel<false>, grid, dim3(256), 0, st, p, A0, G, tiles_m, tiles_n, 0);
    LAUNCH_CHECK();
    if (p.ksplit > 1) {
        if (parts_unreduced) { *parts_unreduced = p.ksplit; return FUMI_OK; }      // the consumer adds the parts where it reads them
        return launch_reduce_slabs(st, parts, p.ksplit, p.part_stride, p.part_stride, 1.f, A0_final);
    }
    return FUMI_OK;
}

static bool xpanel_bwd_wide(int D, int h0) {
    static const int off = getenv("FUMI_XPB_64") ? atoi(getenv("FUMI_XPB_64")) : 0;     // 1: always the 64 x 64 kernel
    return !off && (h0 % 256 == 0) && (D % 64 == 0);
}

// narrow gradients (h0 = 64: AM3's image encoder) on the split-bf16 kernel with the roles of the operands swapped
static bool xpanel_bwd_narrow(int D, int h0) {
    static const int on = getenv("FUMI_XPB_SB") ? atoi(getenv("FUMI_XPB_SB")) : 1;
    return on && h0 == 64 && D % 256 == 0;
}

int xpanel_bwd_nsplit(int B, int S, int Qn, int D, int h0, int* kchunk_out) {
    const long Ktot = (long)B * (S + Qn);
    const bool wide = xpanel_bwd_wide(D, h0);
    if (xpanel_bwd_narrow(D, h0)) {                 // D / 256 tiles: one workgroup per CU
        long ns = (256 + D / 256 - 1) / (D / 256);
        if (ns > 32) ns = 32;
        if (ns < 1) ns = 1;
        long kc = ((Ktot + ns - 1) / ns + BK - 1) / BK * BK;
        *kchunk_out = (int)kc;
        return (int)((Ktot + kc - 1) / kc);
    }
    const long tiles = wide ? (long)(h0 / 256) * (D / 64) : (long)((h0 + 63) / 64) * ((D + 63) / 64);
    static const int target = getenv("FUMI_XPB_WG") ? atoi(getenv("FUMI_XPB_WG")) : 0;
    const long want = target > 0 ? target : (wide ? 512 : 1024);   // workgroups: two 8-wave ones or four 4-wave ones per CU
    long ns = (want + tiles - 1) / tiles;
    if (ns < 1) ns = 1;
    if (ns > 32) ns = 32;
    long kc = ((Ktot + ns - 1) / ns + BK - 1) / BK * BK;
    if (kc < BK) kc = BK;
    *kchunk_out = (int)kc;
    return (int)((Ktot + kc - 1) / kc);
}

// Two-launch form (run_episodes, T >= 2): the contraction over the B Qn query rows needs only the query pass's adjoints, so it runs
// on a second stream BESIDE the reverse sweep (a latency chain on half of the CUs); the B S support rows follow the sweep as a short
// launch of 64-row slabs.  Each part is this same kernel family on a panel with S = 0 (query rows) or Qn = 0 (support rows) and the
// matching half of the split adjoint array.
bool xpanel_bwd_two_part_ok(int D, int h0) {
    static const int bsb = getenv("FUMI_XPB_SB") ? atoi(getenv("FUMI_XPB_SB")) : 1;
    return bsb && xpanel_bwd_wide(D, h0) && D % 128 == 0;
}
void xpanel_bwd_two_part_split(int B, int S, int Qn, int D, int h0, int* nsq, int* kcq, int* nss, int* kcs) {
    *nsq = xpanel_bwd_nsplit(B, 0, Qn, D, h0, kcq);
    long kc = 64;                                       // short slabs: the launch is as long as one slab's chain
    const long K = (long)B * S;
    if ((K + kc - 1) / kc > 32) kc = ((K + 31) / 32 + BK - 1) / BK * BK;
    *kcs = (int)kc; *nss = (int)((K + kc - 1) / kc);
}

int launch_xpanel_bwd(hipStream_t st, int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q,
                      const float* Abar, float* slabs, int kchunk, int nsplit, const XRows* rows, const HyperBwdArgs* rider,
                      int* rider_done) {
    if (rider_done) *rider_done = 0;
    XPanel p{x_s, x_q, nullptr, B, S, Qn, D, h0, nullptr, nullptr, nullptr, 0, S};
    if (rows && rows->table) { p.table = rows->table; p.idx_s = rows->idx_s; p.idx_q = rows->idx_q; p.n_rows = rows->n_rows; p.x_s = p.x_q = rows->table; }
    const bool fast = (D % 64 == 0) && (h0 % 64 == 0) && al16(p.x_s) && al16(p.x_q) && al16(Abar);
    if (fast && xpanel_bwd_narrow(D, h0) && kchunk % BSK == 0) {
        const size_t lds_sb = 2 * (size_t)bsb_buf_ushorts(1, 16) * sizeof(unsigned short);      // (72 KB: also holds the 64 x 260 float transpose)
        HyperBwdArgs none; memset(&none, 0, sizeof(none));
        const int tm = D / 256;
        FUMI_SET_DYN_LDS((xpanel_bwd256_sb_kernel<false, 2, 1, 16, true>), lds_sb);
        hipLaunchKernelGGL((xpanel_bwd256_sb_kernel<false, 2, 1, 16, true>), dim3(8 * ((nsplit + 7) / 8) * tm), dim3(512), lds_sb, st,
                           p, Abar, slabs, kchunk, nsplit, 1, tm, none);
        LAUNCH_CHECK();
        return FUMI_OK;
    }
    if (fast && xpanel_bwd_wide(D, h0)) {
        const size_t lds_bytes = 2 * (32 * 256 + 32 * 64) * sizeof(float);
        const int tn = D / 64, tm = h0 / 256;
        const unsigned nwg = 8 * ((nsplit + 7) / 8) * tn * tm;
        static const int ride = getenv("FUMI_XP_RIDER") ? atoi(getenv("FUMI_XP_RIDER")) : 1;     // 0: never carry the hypernetwork backward
        static const int bsb = getenv("FUMI_XPB_SB") ? atoi(getenv("FUMI_XPB_SB")) : 1;          // 0: the fp32-MFMA kernel
        if (bsb && kchunk % BSK == 0) {
            static const int bnb = getenv("FUMI_XPB_NB") ? atoi(getenv("FUMI_XPB_NB")) : 2;      // 2: 256 x 128 tiles (default), 1: 256 x 64
            static const int bsk = getenv("FUMI_XPB_SK") ? atoi(getenv("FUMI_XPB_SK")) : 16;     // 32: two k-steps per slab (NB = 1)
            const int NB = (bnb == 2 && D % 128 == 0) ? 2 : 1;
            const int SKv = (NB == 1 && bsk == 32 && kchunk % 32 == 0) ? 32 : 16;
            const size_t lds_sb = 2 * (size_t)bsb_buf_ushorts(NB, SKv) * sizeof(unsigned short);
            const unsigned nwg2 = 8 * ((nsplit + 7) / 8) * (D / (64 * NB)) * tm;
            const bool ridden = ride && rider && rider_done && rider->nblk > 0 && rider->nblk % 8 == 0 &&
                                (size_t)hyper_bwd_lds_floats(rider->Dt, rider->H1) * 4 <= lds_sb;
            HyperBwdArgs none; memset(&none, 0, sizeof(none));
#define BSB_LAUNCH(RID_, NB_, SK_)                                                                                      \
            do {                                                                                                        \
                FUMI_SET_DYN_LDS((xpanel_bwd256_sb_kernel<RID_, 2, NB_, SK_>), lds_sb);                                  \
                hipLaunchKernelGGL((xpanel_bwd256_sb_kernel<RID_, 2, NB_, SK_>), dim3(nwg2 + (RID_ ? rider->nblk : 0)), dim3(512), lds_sb, st, \
                                   p, Abar, slabs, kchunk, nsplit, D / (64 * NB_), tm, RID_ ? *rider : none);            \
            } while (0)
            if (ridden) { if (NB == 2) BSB_LAUNCH(true, 2, 16); else if (SKv == 32) BSB_LAUNCH(true, 1, 32); else BSB_LAUNCH(true, 1, 16); *rider_done = 1; }
            else { if (NB == 2) BSB_LAUNCH(false, 2, 16); else if (SKv == 32) BSB_LAUNCH(false, 1, 32); else BSB_LAUNCH(false, 1, 16); }
#undef BSB_LAUNCH
            LAUNCH_CHECK();
            return FUMI_OK;
        }
        if (ride && rider && rider_done && rider->nblk > 0 && rider->nblk % 8 == 0 &&
            (size_t)hyper_bwd_lds_floats(rider->Dt, rider->H1) * 4 <= lds_bytes) {
            FUMI_SET_DYN_LDS(xpanel_bwd256_kernel<true>, lds_bytes);
            hipLaunchKernelGGL(xpanel_bwd256_kernel<true>, dim3(nwg + rider->nblk), dim3(512), lds_bytes, st, p, Abar, slabs,
                               kchunk, nsplit, tn, tm, *rider);
            *rider_done = 1;
        } else {
            HyperBwdArgs none; memset(&none, 0, sizeof(none));
            FUMI_SET_DYN_LDS(xpanel_bwd256_kernel<false>, lds_bytes);
            hipLaunchKernelGGL(xpanel_bwd256_kernel<false>, dim3(nwg), dim3(512), lds_bytes, st, p, Abar, slabs,
                               kchunk, nsplit, tn, tm, none);
        }
        LAUNCH_CHECK();
        return FUMI_OK;
    }
    const dim3 grid((D + 63) / 64, (h0 + 63) / 64, nsplit);
    if (fast) hipLaunchKernelGGL(xpanel_bwd_kernel<true>, grid, dim3(256), 0, st, p, Abar, slabs, kchunk);
    else hipLaunchKernelGGL(xpanel_bwd_kernel<false>, grid, dim3(256), 0, st, p, Abar, slabs, kchunk);
    LAUNCH_CHECK();
    return FUMI_OK;
}
