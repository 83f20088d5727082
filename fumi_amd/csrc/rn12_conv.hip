// Matrix products of the bf16 ResNet-12 encoder (rn12.h): forward / input-gradient convolutions as implicit GEMMs over shifted
// copies of the flattened padded pixel axis, weight gradients as pixel-contracted products, all on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation.  Hand-written for gfx950 (wave64, 160 KiB LDS, ds_read_b64_tr_b16).
#include "rn12.h"
#include <stdlib.h>
#include <type_traits>

unsigned long long* g_rn_trace = nullptr;          // dev tracing only (tests/dev/trace_rn12_conv.py)
void set_rn12_trace(void* p) { g_rn_trace = (unsigned long long*)p; }

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }
// wave-uniform values that reach the kernel through LDS or lane arithmetic live in VGPRs unless told otherwise: these move them to SGPRs
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uni(long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffL)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ const T* uni(const T* p) { return (const T*)uni((long)p); }

// =====================================================================================================================================
// convolution (forward and input-gradient)
//   workgroup = 4 waves, tile = (128 MW) consecutive INTERIOR pixels (image after image, row after row; the zero border of the padded
//   layout is skipped: at 10 x 10 it would be 44 % of the matrix work) x (32 NF) output channels; wave w owns pixels [32 MW w, 32 MW (w+1)).
//   The tile's pixels are not contiguous in the padded map (2 border pixels per row, 2 border rows per image), its INPUT slab is:
//   padded pixels [first - halo, last + halo]; a lane keeps the slab row of each of its pixels, every tap is the same shift for all.
//   for every 64-channel chunk of every source: the input slab (tile + halo on both sides) goes to LDS ONCE as [pixel][64 ch] bf16 with
//   its 16-byte chunks XOR-ed by (pixel >> 1) & 7 (conflict-free ds_read_b128 for every tap shift); per tap the weight tile
//   (4 k-steps x NF fragments x 1 KiB, already in MFMA B-fragment order in memory) is double-buffered through LDS -- the copy is linear,
//   every wave reads each fragment with one ds_read_b128 -- while the matrix pipe works on the previous tap.
//   epilogue (interior pixels only; nothing reads the border of a convolution's output): the products are formed transposed (weights as the
//   A operand), so a lane holds one pixel; one lane swap per packed dword (v_permlane16_swap / 32) gives it eight consecutive channels =
//   a 16-byte store straight from registers.  The batch-norm statistics (sum, sum of squares or of products with `dot`) are those of the
//   STORED values, summed per lane, over the lanes of a pixel group on the DPP path, over the waves through LDS.
// =====================================================================================================================================
constexpr int CV_KC = 64;                              // channels per staged chunk

// padded pixel (within an episode) of the compact interior index c = (image, y, x) row-major.  32-bit arithmetic: an episode has
// fewer than 2^31 pixels (the launcher checks), and a 64-bit division costs ~100 instructions in a prologue that a short tile
// (block 1: 36 k-steps) feels -- 5 600 of its 31 000 cycles in the first trace
__host__ __device__ inline long rn_pix_of(long c, const RnGeom& g) {
    const unsigned hw = (unsigned)(g.H * g.W), cu = (unsigned)c;
    const unsigned img = cu / hw, r = cu - img * hw;
    const unsigned y = r / (unsigned)g.W, x = r - y * (unsigned)g.W;
    return (long)(img * (unsigned)g.Pp + (y + 1) * (unsigned)g.Wp + x + 1);
}
// Large images (>= 4 tiles each): tiles restart at every image -- a tile never straddles the 2 border rows between two images, which
// would put them (and for wide rows a lot of LDS) into its slab; the last tile of an image is partial (1.6 % at 84 x 84 / 256).
inline bool rn_per_image(const RnGeom& g, int mt) { return (long)g.H * g.W >= 4L * mt; }
// rows of the LDS image of a tile's input slab (without the halo): the largest span of mt consecutive interior pixels in padded pixels
inline int rn_slab_span(const RnGeom& g, int mt) {
    const int hw = g.H * g.W;
    int best = 0;
    if (rn_per_image(g, mt)) {
        for (long c0 = 0; c0 < hw; c0 += mt) {
            const long c1 = (c0 + mt < hw ? c0 + mt : hw) - 1;
            const int span = (int)(rn_pix_of(c1, g) - rn_pix_of(c0, g)) + 1;
            best = span > best ? span : best;
        }
        return best;
    }
    for (int k = 0; k < hw; ++k) {                                 // every phase of a tile start within an image (starts are k * mt)
        const long c0 = (long)k * mt;
        const int span = (int)(rn_pix_of(c0 + mt - 1, g) - rn_pix_of(c0, g)) + 1;
        best = span > best ? span : best;
        if (k > 4096) break;
    }
    return best;
}

template <int NF, int MW, int BKS>
struct ConvCfg {
    static constexpr int MT = 128 * MW, NT = 32 * NF;
    static constexpr int BT = BKS * NF * 1024;                                   // bytes of one weight tile (BKS k-steps of one tap)
    static int slab_rows(const RnGeom& g) { return (rn_slab_span(g, MT) + 2 * g.halo + 7) / 8 * 8; }
    static int lds_bytes(const RnGeom& g) {
        const int main_ = slab_rows(g) * 128 + 2 * BT;
        const int epi = 4 * NT * 2 * 4;                                          // the epilogue's statistics: [4 waves][NT][2] floats
        return (main_ > epi ? main_ : epi) + 16;
    }
};

// (Weight tiles: two buffers, the next tile requested while this one is multiplied, s_waitcnt vmcnt(0) + one barrier per tile.  A
// three-buffer ring with requests two tiles ahead and a COUNTED wait -- vmcnt(NBL): the newest tile may still be in flight -- was
// built and measured: the wait for the loads halves (9.9 -> 5.3 % of a wave's time) and the barrier takes it over (3.6 -> 8.8 %):
// 884 vs 889 us per launch, 558.6 vs 557.1 ms per step.  Repeated at the end of round 4, when the shorter k-steps had made the wait for
// the next tile 22 % of a traced wave's time: layer set 2.57 / 2.05 vs 2.57 / 2.08 ms, 17.20-17.35 vs 17.23-17.33 episodes/s -- the partner
// workgroup's products run under that wait either way.)
// S16: the same tile on v_mfma_f32_16x16x32_bf16 (four times as many instructions of half the cycles and twice the depth: equal
// cycles per flop, equal LDS bytes per flop).  Why: under this kernel's load the chip holds its clock well below 2.4 GHz, and the
// clock it holds depends on the MFMA shape (MI355X_MICROARCH.md, DVFS give-back item 7: the 16x16x32 loop delivers 1.12-1.15 x the
// FLOP/s of the 32x32x16 loop at equal cycles per FLOP on random data, operands read from LDS).  Fragment order of the weights in
// memory (rn_wprep_kernel) is then [tap][Cin/32][Cout/16][64 lanes][8]: lane l = column l & 15, k = 8 (l >> 4) + j of a 32-deep step;
// the A fragment of a lane is 16 bytes of pixel row l & 15 at channel 8 (l >> 4) of the step, and the slab's 16-byte chunks are XOR-ed
// with (row & 7) instead of (row >> 1) & 7 (conflict-free for THIS read pattern: tools/lds_swizzle_check.py).
template <int NF, int MW, int BKS, bool S16>
__global__ __launch_bounds__(256, 2) void rn_conv_kernel(RnConvArgs a) {
    typedef ConvCfg<NF, MW, BKS> C;
    auto swz = [](int r) { return S16 ? (r & 7) : ((r >> 1) & 7); };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ RnSrc s_src[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup id -> (pixel tile, column group, episode).  All tiles of one (episode, column group) stream the SAME weight slice
    // (up to 1.8 MB): with a.xcd the ids are dealt so that such a group stays on one XCD (ids i and i + 8 share an XCD) and its
    // slice stays in that XCD's L2; speed only -- any placement is correct
    int tile, cg, b;
    if (a.xcd) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int grp = (j / a.tiles) * 8 + xcd;
        tile = j % a.tiles; cg = grp % a.ncg; b = grp / a.ncg;
        if (b >= a.B) return;
    } else {
        tile = blockIdx.x % a.tiles;
        const int r = blockIdx.x / a.tiles;
        cg = r % a.ncg; b = r / a.ncg;
    }
    const int halo = a.g.halo, Wp = a.g.Wp;
    const long NC = (long)((unsigned)a.npix / (unsigned)a.g.Pp) * (a.g.H * a.g.W);   // interior pixels of the episode
    long c0 = (long)tile * C::MT, cend = NC;
    if (a.tpi) {                                                    // tiles restart at every image
        const int hw = a.g.H * a.g.W;
        const long img = (unsigned)tile / (unsigned)a.tpi;
        c0 = img * hw + (long)(tile - img * a.tpi) * C::MT; cend = (img + 1) * hw;
    }
    const long clast = min(c0 + C::MT, cend) - 1;
    const long pfirst = rn_pix_of(c0, a.g), plast = rn_pix_of(clast, a.g);
    // (what the register epilogue needs of these, moved to scalar registers here: they are live across the whole main loop)
    [[maybe_unused]] const unsigned epi_pbase = (unsigned)uni((int)((unsigned)pfirst * (unsigned)a.Cout)), epi_cmax = (unsigned)uni((int)(clast - c0));
    if (tid == 0) { s_src[0] = a.src[0]; s_src[1] = a.src[1]; s_src[2] = a.src[2]; s_src[3] = a.src[3]; }
    unsigned char* const As = lds;
    unsigned char* const Bs = lds + a.slab_rows * 128;            // (slab_rows: a multiple of 8, the granule of the LDS-direct loads)
    constexpr int NPR = S16 ? 2 * MW : MW;                          // pixel sub-tiles per lane: 16 rows each (S16) or 32
    int prow[NPR];                                                  // this lane's pixels as rows of a halo-less slab
#pragma unroll
    for (int m = 0; m < NPR; ++m) {
        long c = c0 + wave * 32 * MW + (S16 ? m * 16 + (lane & 15) : m * 32 + (lane & 31));
        c = c > clast ? clast : c;                                  // (rows past the end compute a copy of the last pixel; never stored)
        prow[m] = (int)(rn_pix_of(c, a.g) - pfirst);
    }
    f32x16 acc[S16 ? 1 : MW][S16 ? 1 : NF];
    f32x4 acc16[S16 ? 2 * MW : 1][S16 ? 2 * NF : 1];
    if constexpr (!S16) {
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][f][i] = 0.f;
    } else {
#pragma unroll
        for (int m = 0; m < 2 * MW; ++m)
#pragma unroll
            for (int f = 0; f < 2 * NF; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc16[m][f][i] = 0.f;
    }
    __syncthreads();
    const int CF = a.Cout >> 5;
    const int srow = tid >> 3, schunk = tid & 7;
    // dev tracing: cycles per phase summed in registers (a store inside the loop would join the vmcnt queue the waits count)
    const bool traced = a.trace && tid == 0 && blockIdx.x == gridDim.x / 2;
    unsigned long long t_prev = traced ? __builtin_amdgcn_s_memtime() : 0, t_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): start stamps to memory at once -- no registers held across the kernel
    if (traced) { a.trace[10] = t_prev; a.trace[11] = __builtin_amdgcn_s_memrealtime(); }
#define RNSTAMP(k) if (traced) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_sum[k] += t_now - t_prev; t_prev = t_now; }
    for (int s = 0; s < a.nsrc; ++s) {
        RnSrc S;                                                    // (wave-uniform: kept in scalar registers)
        S.in = uni(s_src[s].in); S.in_stride = uni(s_src[s].in_stride); S.frag = uni(s_src[s].frag);
        S.frag_stride = uni(s_src[s].frag_stride); S.Cin = uni(s_src[s].Cin); S.ntaps = uni(s_src[s].ntaps);
        const int hs = S.ntaps == 9 ? halo : 0;
        const long p0 = pfirst;                                     // slab row r = padded pixel p0 - hs + r
        const int rows = (int)(plast - pfirst) + 1 + 2 * hs;
        const int KS = S.Cin >> 4;
        const rbf16* in = S.in + (long)b * S.in_stride;
        const rbf16* frag = S.frag + (long)b * S.frag_stride;
        for (int c0 = 0; c0 < S.Cin; c0 += CV_KC) {
            const int kc = min(CV_KC, S.Cin - c0), nks = kc >> 4;
            __syncthreads();                                       // the previous chunk's products are done with As / Bs
            // ---- input slab -> LDS.  Default: LDS-direct loads (global_load_lds_dwordx4, no registers: every load of the slab is in
            //      flight at once, ONE memory round trip per chunk instead of one per batch of 4).  A wave instruction fills 8 rows
            //      (1 KiB, lane-linear): the chunk swizzle is applied to the SOURCE address.  Pixels outside the episode are clamped
            //      to its first / last pixel -- both are border pixels of the padded layout, i.e. zeros, which is what a halo outside
            //      the map must read as.  (k-steps past the chunk's width are never multiplied: their columns may hold anything.)
            const bool chok = schunk * 8 < kc;
            const int chs = chok ? schunk * 8 : 0;
            if (a.glds) {
                // (32-bit arithmetic off a scalar base: the address math of these loops was 7 % of the kernel as 64-bit VALU code)
                const int pbase = (int)p0 - hs + (lane >> 3), plim = (int)a.npix - 1;
                const char* inb = (const char*)(in + c0);
                for (int r0 = uni(wave * 8); r0 < rows; r0 += 32) {
                    const int r = r0 + (lane >> 3);
                    int p = pbase + r0;
                    p = p < 0 ? 0 : (p > plim ? plim : p);
                    int chunk = (lane & 7) ^ swz(r);
                    chunk = chunk * 8 < kc ? chunk : 0;
                    const unsigned off = ((unsigned)p * (unsigned)S.Cin + (unsigned)(chunk * 8)) * 2u;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(inb + off),
                                                     (__attribute__((address_space(3))) void*)(As + r0 * 128), 16, 0, 0);
                }
            } else {
                for (int r0 = 0; r0 < rows; r0 += 128) {
                    u32x4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = r0 + 32 * i + srow;
                        long p = p0 - hs + r;
                        p = p < 0 ? 0 : (p >= a.npix ? a.npix - 1 : p);
                        v[i] = ld16(in + p * S.Cin + c0 + chs);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = r0 + 32 * i + srow;
                        const long p = p0 - hs + r;
                        const bool ok = chok && p >= 0 && p < a.npix;
                        if (r < rows) *(u32x4*)(As + r * 128 + ((schunk ^ swz(r)) << 4)) = ok ? v[i] : (u32x4){0u, 0u, 0u, 0u};
                    }
                }
            }
            // ---- weight tiles: tile i of the chunk = k-steps [BKS (i % npt), ...) of tap i / npt; NF fragments of 1 KiB per k-step,
            //      contiguous in memory AND in the order the LDS image has them: LDS-direct loads (one wave instruction per KiB block,
            //      no registers, no ds_write pass), double-buffered, requested one tile ahead, one barrier per tile.  Addresses: a
            //      scalar base per tile (advanced by constants) + lane * 16
            // A 16-channel source (the 3-channel image layers) has ONE k-step per tap: a tile per tap would be 4 MFMAs between two barriers
            // and eight weight requests of which six are clamped duplicates.  Its fragments of consecutive taps lie like consecutive
            // k-steps (KS = 1), so the nine taps are walked as nine k-steps of one "tap" -- tiles of BKS taps, each k-step with its own
            // pixel shift (TAPK).
            const bool tapk = !S16 && S.ntaps == 9 && S.Cin == 16;
            const int nksL = tapk ? 9 : nks, ntL = tapk ? 1 : S.ntaps;
            const int npt = (nksL + BKS - 1) / BKS;                // tiles per tap
            constexpr int NBL = (BKS * NF + 3) / 4;                // KiB blocks per wave and tile (4 waves); clamped duplicates past the end
            const int nblk = min(BKS, nksL) * NF;                  // KiB blocks of a full tile (a chunk's last tile may hold fewer: nb below)
            // (S16: a 32-deep step is 2 NF blocks of 16 columns; the same bytes per tap and per tile, another order)
            const char* fragb = S16 ? (const char*)(frag + ((long)(c0 >> 5) * (2 * CF) + cg * 2 * NF) * 512) + lane * 16
                                    : (const char*)(frag + ((long)(c0 >> 4) * CF + cg * NF) * 512) + lane * 16;
            const long tap_stride = (long)KS * CF * 1024, part_stride = (long)BKS * CF * 1024;
            // one KiB block of a weight tile of nb blocks: block wave + 4 j of the tile whose first block is at tbase (clamped duplicate past
            // the end: never a read past the tile)
            auto bissue1 = [&](const char* tbase, int buf, int j, int nb) {
                int blk = uni(wave) + 4 * j;
                blk = blk < nb ? blk : nb - 1;
                constexpr int PB = S16 ? 2 * NF : NF;               // blocks per step
                const int ks = blk / PB, f = blk - ks * PB;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tbase + ((long)ks * (S16 ? 2 * CF : CF) + f) * 1024),
                                                 (__attribute__((address_space(3))) void*)(Bs + buf * C::BT + blk * 1024), 16, 0, 0);
            };
            // (the same with the block's two offsets worked out ONCE per chunk, in scalar registers, for requests whose j is a compile-time
            //  constant -- the 16 x 16 x 32 loop below: computed in place they were ~20 scalar / vector instructions in front of every
            //  request, issued in order between two groups of MFMAs.  Indexed with run-time j the two tables would live in scratch.)
            int b_src[NBL], b_dst[NBL];
#pragma unroll
            for (int j = 0; j < NBL; ++j) {
                int blk = uni(wave) + 4 * j;
                blk = blk < nblk ? blk : nblk - 1;
                constexpr int PB = S16 ? 2 * NF : NF;
                const int ks = blk / PB, f = blk - ks * PB;
                b_src[j] = uni((ks * (S16 ? 2 * CF : CF) + f) * 1024);
                b_dst[j] = uni(blk * 1024);
            }
            auto bissue_j = [&](const char* tbase, int buf, int J) {                // (J: a constant after unrolling at every call site)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tbase + b_src[J]),
                                                 (__attribute__((address_space(3))) void*)(Bs + buf * C::BT + b_dst[J]), 16, 0, 0);
            };
            auto bissue = [&](const char* tbase, int buf) {
#pragma unroll
                for (int j = 0; j < NBL; ++j) bissue1(tbase, buf, j, nblk);
            };
            // the k-steps of one tile; `nxt` != NULL: the NEXT tile's NBL load requests are placed one by one BETWEEN the groups of MFMAs
            // (a wave issues in order: as a block in front of the k-steps they cost 880 cycles of a 2 300-cycle tile during which this
            // wave fed the matrix pipe nothing; behind an MFMA they issue while the pipe works)
            auto compute = [&](int t, int k0, int buf, const char* nxt, int nbuf, int nbn) {
                const int kn = min(BKS, nksL - k0);
                const int toff = (S.ntaps == 9 && !tapk) ? (t / 3 - 1) * Wp + (t % 3 - 1) : 0;
                const unsigned char* Bt = Bs + buf * C::BT;
                int arow[NPR];
#pragma unroll
                for (int m = 0; m < NPR; ++m) arow[m] = prow[m] + hs + toff;
                int slot = nxt ? 0 : NBL;
                if constexpr (!S16) {
                    auto kstep = [&](int ks) {
                        rbf16x8 bf[NF];
#pragma unroll
                        for (int f = 0; f < NF; ++f) bf[f] = *(const rbf16x8*)(Bt + ((ks * NF + f) * 64 + lane) * 16);
                        const int tt = k0 + ks;                     // TAPK: this k-step IS tap tt (channel chunk 0, its own pixel shift)
                        const int tsh = tapk ? (tt / 3 - 1) * Wp + (tt % 3 - 1) : 0;
#pragma unroll
                        for (int m = 0; m < MW; ++m) {
                            const int ch = (tapk ? 0 : tt * 2) + (lane >> 5), ar = arow[m] + tsh;
                            const rbf16x8 af = *(const rbf16x8*)(As + ar * 128 + ((ch ^ swz(ar)) << 4));
#pragma unroll
                            for (int f = 0; f < NF; ++f) acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[f], af, acc[m][f], 0, 0, 0);   // (transposed)
                            if (slot < NBL) { bissue1(nxt, nbuf, slot, nbn); ++slot; }
                        }
                    };
                    // (a runtime loop on purpose: fully unrolled, hipcc hoists the next k-steps' fragment reads and spills -- 576 bytes of
                    //  scratch per lane at NF = 5, MW = 2 and 8.6x the time; at NF = 2 it fits, 200 VGPRs, and is no faster: 950 vs 910 us)
                    for (int ks = 0; ks < kn; ++ks) kstep(ks);
                } else {
                    // 32-deep steps: the 2 MW A fragments of the step stay in registers, the B fragments come in two halves of NF.
                    // (k2 is unrolled behind a run-time guard: every request's index is then a constant after unrolling)
                    constexpr int PER = 2 * (NF >> 1);                 // requests placed per 32-deep step: behind every second fragment
#pragma unroll
                    for (int k2 = 0; k2 < BKS / 2; ++k2) {
                        if (k2 < (kn >> 1)) {
                            rbf16x8 af[2 * MW];
                            const int ch = ((k0 >> 1) + k2) * 4 + (lane >> 4);
#pragma unroll
                            for (int m = 0; m < 2 * MW; ++m) af[m] = *(const rbf16x8*)(As + arow[m] * 128 + ((ch ^ swz(arow[m])) << 4));
                            // fragment-outer order: a weight fragment is dead after its 2 MW products, so the SAME registers take the
                            // second half's fragment at once -- its LDS read runs under the first half's remaining products instead of
                            // in front of the second half (where all five reads sat behind an lgkmcnt(0) in the middle of every tile)
                            rbf16x8 bf[NF];
#pragma unroll
                            for (int f = 0; f < NF; ++f) bf[f] = *(const rbf16x8*)(Bt + (((k2 * 2) * NF + f) * 64 + lane) * 16);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
#pragma unroll
                                for (int f = 0; f < NF; ++f) {
#pragma unroll
                                    for (int m = 0; m < 2 * MW; ++m)
                                        // (transposed: weight fragment as the A operand -- rows = output channels, columns = pixels)
                                        acc16[m][h * NF + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[f], af[m], acc16[m][h * NF + f], 0, 0, 0);
                                    if (h == 0) bf[f] = *(const rbf16x8*)(Bt + (((k2 * 2 + 1) * NF + f) * 64 + lane) * 16);
                                    const int J = k2 * PER + h * (NF >> 1) + (f >> 1);
                                    if ((f & 1) && J < NBL && nxt) bissue_j(nxt, nbuf, J);
                                }
                            }
                        }
                    }
                    if (nxt) {                                          // (a short tile, or more blocks than places: the rest after it)
                        const int done = min(NBL, (kn >> 1) * PER);
#pragma unroll
                        for (int q = 0; q < NBL; ++q) if (q >= done) bissue_j(nxt, nbuf, q);
                    }
                    slot = NBL;
                }
                for (; slot < NBL; ++slot) bissue1(nxt, nbuf, slot, nbn);    // (a short tile: the rest after it)
            };
            RNSTAMP(0)                                              // (prologue / between chunks + slab loads issued)
            // the tile after the one being multiplied, walked in (tap, k-part) order: `ahead` is its address, (at, akp) its tap and
            // k-part; nullptr past the chunk's last tile
            int at = 0, akp = 0;
            const char* ahead = fragb;
            auto advance = [&]() {
                if (++akp == npt) { akp = 0; ++at; }
                ahead = at < ntL ? fragb + at * tap_stride + akp * part_stride : nullptr;
            };
            bissue(fragb, 0);
            advance();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            RNSTAMP(1)                                              // (slab and first weight tile in LDS)
            int gbuf = 0;
            for (int t = 0; t < ntL; ++t) {
                for (int kp = 0; kp < npt; ++kp, gbuf ^= 1) {
                    const char* nxt = ahead;                        // requested into the other buffer while this tile is multiplied
                    const int k0n = kp + 1 < npt ? (kp + 1) * BKS : 0;                     // the next tile's first k-step ...
                    const int nbn = min(BKS, nksL - k0n) * NF;                             // ... and its blocks
                    RNSTAMP(2)
                    compute(t, kp * BKS, gbuf, nxt, gbuf ^ 1, nbn);
                    RNSTAMP(3)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    RNSTAMP(4)
                    __syncthreads();
                    RNSTAMP(6)
                    if (nxt) advance();
                }
            }
        }
    }
    RNSTAMP(0)
    {
        // ---- epilogue, without an LDS image (16 x 16 x 32 form; the 32 x 32 x 16 form of the 3-channel layers likewise with 32-lane swaps): the products were formed TRANSPOSED (weight fragments as the A
        //      operand), so a lane holds one pixel (its column of the tile) and, per 16-channel tile, four consecutive output
        //      channels (4 (lane >> 4) ..); one v_permlane16_swap per packed dword between two neighbouring channel tiles leaves
        //      every lane with EIGHT consecutive channels of its pixel: a 16-byte store straight from registers, the four lanes of a
        //      pixel writing 64 contiguous bytes.  Statistics of the stored values: per lane over its pixels, over the 16 lanes that
        //      share the channels (4 shuffle steps), over the waves through LDS.  (Round 4: 160 ds_write_b16 + two barriers + a read-
        //      back per wave before; forward + input-gradient layer set 5.41 -> 5.19 ms, 1 x 1 layers -13..-17 %.)
        float* red = (float*)lds;                                   // [4 waves][NT][2]
        rbf16* outb = a.out + (long)b * a.out_stride + (long)cg * C::NT;
        const rbf16* dotb = a.dot ? a.dot + (long)b * a.dot_stride + (long)cg * C::NT : nullptr;
        const bool st = a.stats != nullptr;
        if (st) __syncthreads();                                    // (As / Bs are dead in every wave before `red` is written)
        auto pk = [](float x, float y) { return (unsigned)rn_f2bf(x) | ((unsigned)rn_f2bf(y) << 16); };
        // (32-bit element offsets off the episode's base: the launcher checks that the output map has fewer than 2^31 elements)
        const unsigned pbase = epi_pbase, cmax = epi_cmax;
        const unsigned clane = (unsigned)(wave * 32 * MW + (S16 ? (lane & 15) : (lane & 31)));
        auto tile_out = [&](const u32x4& v, int m, int co, float (&s1)[8], float (&s2)[8], auto with_stats) {
            if (clane + (unsigned)(m * (S16 ? 16 : 32)) > cmax) return;        // (rows past the end computed a copy of the last pixel)
            const unsigned off = pbase + (unsigned)prow[m] * (unsigned)a.Cout + (unsigned)co;
            *(u32x4*)(outb + off) = v;
            if constexpr (decltype(with_stats)::value) {
                u32x4 d = v;
                if (dotb) d = ld16(dotb + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = __uint_as_float(v[j] << 16), hi = __uint_as_float(v[j] & 0xffff0000u);
                    const float dl = __uint_as_float(d[j] << 16), dh = __uint_as_float(d[j] & 0xffff0000u);
                    s1[2 * j] += lo; s1[2 * j + 1] += hi;
                    s2[2 * j] += lo * dl; s2[2 * j + 1] += hi * dh;
                }
            }
        };
        // sum over the 16 lanes of a row on the DPP path (quad swaps, then row shifts by 4 and 8 with zero fill): the total lands in
        // lanes 12-15 of the row.  (__shfl_xor compiles to ds_bpermute_b32 here: 320 LDS round trips per wave in this epilogue.)
        auto row16_sum = [](float v) {
            auto dpp = [](float x, auto ctrl) {
                return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
            };
            v += dpp(v, std::integral_constant<int, 0xB1>{});          // quad_perm [1,0,3,2]
            v += dpp(v, std::integral_constant<int, 0x4E>{});          // quad_perm [2,3,0,1]
            v += dpp(v, std::integral_constant<int, 0x114>{});         // row_shr:4
            v += dpp(v, std::integral_constant<int, 0x118>{});         // row_shr:8
            return v;
        };
        auto stats_out = [&](int co, float (&s1)[8], float (&s2)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] = row16_sum(s1[j]); s2[j] = row16_sum(s2[j]);
                if constexpr (!S16) {                               // 32 lanes share the channels: rows 1 / 3 add lane 15 of rows 0 / 2
                    s1[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1[j]), 0x142, 0xa, 0xf, false));
                    s2[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2[j]), 0x142, 0xa, 0xf, false));
                }
            }
            if ((lane & (S16 ? 15 : 31)) == (S16 ? 15 : 31)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { red[((wave * C::NT) + co + j) * 2] = s1[j]; red[((wave * C::NT) + co + j) * 2 + 1] = s2[j]; }
            }
        };
        auto body = [&](auto with_stats) {
            constexpr bool WS = decltype(with_stats)::value;
            if constexpr (!S16) {
                // 32 x 32 tiles: a lane holds channels 8 q + 4 (lane >> 5) .. + 3 of its pixel for q = 0..3; v_permlane32_swap between the
                // quads 2 qp and 2 qp + 1 leaves the lower half of the wave with channels 8 (2 qp) .. + 7, the upper half with the next eight
                const int h = lane >> 5;
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        const int co = f * 32 + 8 * (2 * qp + h);
                        float s1[8], s2[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
                        for (int m = 0; m < MW; ++m) {
                            const f32x16& x = acc[m][f];
                            const int i0 = 8 * qp, i1 = 8 * qp + 4;
                            const u32x2 r0 = __builtin_amdgcn_permlane32_swap(pk(x[i0], x[i0 + 1]), pk(x[i1], x[i1 + 1]), false, false);
                            const u32x2 r1 = __builtin_amdgcn_permlane32_swap(pk(x[i0 + 2], x[i0 + 3]), pk(x[i1 + 2], x[i1 + 3]), false, false);
                            tile_out((u32x4){r0[0], r1[0], r0[1], r1[1]}, m, co, s1, s2, with_stats);
                        }
                        if constexpr (WS) stats_out(co, s1, s2);
                    }
            } else {
                const int g = lane >> 4;
#pragma unroll
                for (int fp = 0; fp < NF; ++fp) {                       // tiles 2 fp, 2 fp + 1: 32 channels
                    const int co = (2 * fp + (g & 1)) * 16 + 8 * (g >> 1);
                    float s1[8], s2[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
                    for (int m = 0; m < 2 * MW; ++m) {
                        const f32x4& x = acc16[m][2 * fp]; const f32x4& y = acc16[m][2 * fp + 1];
                        const u32x2 r0 = __builtin_amdgcn_permlane16_swap(pk(x[0], x[1]), pk(y[0], y[1]), false, false);
                        const u32x2 r1 = __builtin_amdgcn_permlane16_swap(pk(x[2], x[3]), pk(y[2], y[3]), false, false);
                        tile_out((u32x4){r0[0], r1[0], r0[1], r1[1]}, m, co, s1, s2, with_stats);
                    }
                    if constexpr (WS) stats_out(co, s1, s2);
                }
            }
        };
        if (st) body(std::true_type{}); else body(std::false_type{});
        if (st) {
            __syncthreads();
            if (tid < C::NT) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { t1 += red[(w * C::NT + tid) * 2]; t2 += red[(w * C::NT + tid) * 2 + 1]; }
                float* sp = a.stats + (((long)b * a.tiles + tile) * 2) * a.Cout + cg * C::NT + tid;
                sp[0] = t1; sp[a.Cout] = t2;
            }
        }
        if (traced) {
            RNSTAMP(7)
            for (int k = 0; k < 8; ++k) a.trace[k] = t_sum[k];
            a.trace[8] = __builtin_amdgcn_s_memtime(); a.trace[9] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
}

// the 16x16x32 form needs 32-deep steps: every source's channel count a multiple of 32 (FUMI_RN_S16=0: the 32x32x16 form everywhere).
// rn_wprep_kernel lays a layer's fragments out for the form its channel count selects: both sides call this.
bool rn_use_s16(int Cin) {
    static const int on = getenv("FUMI_RN_S16") ? atoi(getenv("FUMI_RN_S16")) : 1;
    return on && (Cin & 31) == 0;
}

template <int NF, int MW, int BKS, bool S16>
int conv_launch2(hipStream_t st, const RnConvArgs& a, int* nt_out) {
    typedef ConvCfg<NF, MW, BKS> C;
    const int lds = C::lds_bytes(a.g);
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    FUMI_SET_DYN_LDS((rn_conv_kernel<NF, MW, BKS, S16>), lds);
    RnConvArgs k = a;
    const long NC = a.npix / a.g.Pp * ((long)a.g.H * a.g.W);
    k.tiles = (int)((NC + C::MT - 1) / C::MT); k.ncg = a.Cout / C::NT; k.slab_rows = C::slab_rows(a.g); k.tpi = 0;
    if (rn_per_image(a.g, C::MT)) {
        k.tpi = (a.g.H * a.g.W + C::MT - 1) / C::MT;
        k.tiles = (int)(a.npix / a.g.Pp) * k.tpi;
    }
    const int groups = k.ncg * a.B;
    static const int xcd_env = getenv("FUMI_RN_XCD") ? atoi(getenv("FUMI_RN_XCD")) : 1;
    k.xcd = xcd_env && groups >= 8 && groups % 8 == 0;
    static const int glds_env = getenv("FUMI_RN_GLDS") ? atoi(getenv("FUMI_RN_GLDS")) : 1;
    k.glds = glds_env; k.trace = g_rn_trace;
    const dim3 grid((unsigned)((long)k.tiles * groups));
    hipLaunchKernelGGL((rn_conv_kernel<NF, MW, BKS, S16>), grid, dim3(256), lds, st, k);
    LAUNCH_CHECK();
    if (nt_out) *nt_out = k.tiles;
    return FUMI_OK;
}
template <int NF, int MW, int BKS>
int conv_launch(hipStream_t st, const RnConvArgs& a, int* nt_out) {
    bool s16 = true;
    for (int s = 0; s < a.nsrc; ++s) s16 = s16 && rn_use_s16(a.src[s].Cin);
    return s16 ? conv_launch2<NF, MW, BKS, true>(st, a, nt_out) : conv_launch2<NF, MW, BKS, false>(st, a, nt_out);
}

// tile shape for a layer: 256-pixel tiles (each wave 64 pixels: every weight fragment feeds two MFMAs, 0.7 KiB of LDS reads per MFMA
// instead of 1.2) whenever two workgroups still fit a CU's LDS -- with 4-k-step weight tiles, else with 2-k-step ones
template <int NF>
int conv_dispatch(hipStream_t st, const RnConvArgs& a, int* nt_out) {
    // dev knobs: FUMI_RN_MW = 1 | 2 forces the pixels per wave (32 | 64), FUMI_RN_BKS = 2 | 4 the k-steps per weight tile (also
    // when that leaves one workgroup per CU)
    static const int force = getenv("FUMI_RN_MW") ? atoi(getenv("FUMI_RN_MW")) : 0;
    static const int fbks = getenv("FUMI_RN_BKS") ? atoi(getenv("FUMI_RN_BKS")) : 0;
    // (the tile shape is chosen from per-episode quantities -- priced for RN_BREF episodes per chunk, the production chunk -- so that an
    // episode's results do not depend on how many episodes share its chunk)
    const long tiles256 = (a.npix + 255) / 256 * (a.Cout / (32 * NF)) * RN_BREF;
    const int two = 80 * 1024 - 256;                                  // two workgroups per CU
    // (Round 4, the 10 x 10 maps on 128-pixel tiles -- three workgroups per CU at 42 KB / 166 registers: 640 -> 640 303 vs 341 us on one
    // stream (972 TFLOP/s), nothing in the two-lane step (16.72-16.83 vs 16.73-16.76 episodes/s); 21 x 21 and above lose (354 vs 308).)
    // (Measured in round 4 and not kept: 96 or 128 pixels per wave -- MW = 3 | 4, accumulators in AGPRs, one workgroup per CU, 0.53 /
    // 0.45 KiB of LDS reads per MFMA instead of 0.7: 551 vs 420 us at 160 -> 160 42 x 42, 424 vs 354 at 320 -> 320 21 x 21, equal at
    // 640 -> 640 10 x 10; MW = 4 with NF = 5 needs 320 accumulators and spills.  NF = 4 on the 640-channel layers is 8-9 % faster on
    // one stream (800 instead of 640 workgroups on 512 slots) and changes nothing in the two-lane step: 15.78 vs 15.78 episodes/s.
    // The epilogue below is written for any MW; only MW = 1 | 2 are instantiated.  Two attempts at unequal s_setprio for the two
    // workgroups that share a CU (by HW_ID wave slot; by a per-CU flag taken with atomicCAS) changed nothing -- and could not: both
    // guarded the s_setprio with a condition hipcc does not know to be wave-uniform, which lowers to an exec mask around an
    // UNCONDITIONAL scalar s_setprio (cdna_hip_programming.md T5: the guard must go through readfirstlane).  Repeated with that
    // guard (a scalar branch around s_setprio 3 in the object), flag holder or the other workgroup high: 15.39-15.42 / 15.37 vs
    // 15.45-15.47 episodes/s -- the issue arbiter already prefers the older of two waves (MI355X_MICROARCH.md, Two waves per SIMD,
    // items 2-4).  In-kernel clock under this kernel (s_memtime / s_memrealtime, tests/dev/trace_rn12_conv.py): 1.98 GHz at
    // 160 -> 160, 2.30-2.33 at the other layers -- the distance to the peak is idle pipe, not a lowered clock.  The 96-pixel form
    // with two fragment sets in registers (reads of k-step ks + 1 under the MFMAs of ks, a full
    // tile as straight-line code): hipcc waits with lgkmcnt(0) around every LDS-direct load and shuffles accumulators between AGPRs
    // and VGPRs -- 1 685 us at 160 -> 160, four times the two-workgroup kernel.)
    if (force != 1 && (tiles256 >= 256 || force == 2)) {
        if (fbks == 4 && ConvCfg<NF, 2, 4>::lds_bytes(a.g) <= 160 * 1024) return conv_launch<NF, 2, 4>(st, a, nt_out);
        if (fbks == 2 && ConvCfg<NF, 2, 2>::lds_bytes(a.g) <= 160 * 1024) return conv_launch<NF, 2, 2>(st, a, nt_out);
        // (NF = 5 with 4-k-step tiles and two register sets of weight loads spills: 2-k-step tiles there)
        if (NF < 5 && ConvCfg<NF, 2, 4>::lds_bytes(a.g) <= two) return conv_launch<NF, 2, 4>(st, a, nt_out);
        if (ConvCfg<NF, 2, 2>::lds_bytes(a.g) <= two) return conv_launch<NF, 2, 2>(st, a, nt_out);
    }
    if (fbks == 2) return conv_launch<NF, 1, 2>(st, a, nt_out);
    return conv_launch<NF, 1, 4>(st, a, nt_out);
}

// NF (32-column blocks per workgroup) for an output width: the widest of {5, 4, 3, 2, 1} that divides it
int conv_nf(int Cout) {
    const int cf = Cout / 32;
    for (int nf = 5; nf >= 1; --nf) if (cf % nf == 0) return nf;
    return 1;
}

// =====================================================================================================================================
// weight gradient:  dW[tap][co][ci] = sum_p dy[p][co] x[p + off_tap][ci]
//   workgroup = 64 (co) x 64 (ci) outputs for all taps; wave w owns the 32 x 32 quadrant (w >> 1, w & 1) with one accumulator per
//   tap.  Both operands are pixel-major ([pixel][channel] rows) while an MFMA lane needs 8 consecutive PIXELS of one channel: the LDS
//   images stay [pixel][64 ch] (128-byte rows, bit 6 of the byte offset XOR-ed with bit 1 of the row: the four rows a 32-lane half
//   reads fall into four different quarters of the banks) and fragments come back through gfx950's transposing ds_read_b64_tr_b16.
//   The contraction runs over 128-pixel stages; the x slab of a stage (stage + halo) serves all 9 taps.
// =====================================================================================================================================
constexpr int WG_PK = 128;

__device__ __forceinline__ rbf16x8 tr_frag(const unsigned char* img, int r0, int cbase, int lane) {
    const int grp = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    const int row = r0 + 8 * (grp >> 1) + fq;
    const int colb = (cbase + 16 * (grp & 1) + 4 * fp) * 2;
    const unsigned char* p0 = img + row * 128 + (colb ^ (((row >> 1) & 1) << 6));
    const unsigned char* p1 = img + (row + 4) * 128 + (colb ^ ((((row + 4) >> 1) & 1) << 6));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(rbf16x8, v);
}

// (One stage buffer, two workgroups per CU covering each other's loads.  Double-buffering the stages inside ONE workgroup per CU --
// next stage's loads in flight under this stage's 72 MFMAs per wave, one barrier per stage -- was built and measured: 194 vs 171 ms
// per 8-episode step.  Contracting over INTERIOR pixels only (dy is zero on the border: 44 % of the padded pixels at 10 x 10) with a
// gathered dy stage, a wider x slab and a 128-entry row table for the x fragments was built as well: 30 % fewer stages at 10 x 10, but
// the index arithmetic in front of the loads and the table-dependent fragment addresses cost as much -- 155.8 vs 156.4 ms with it on
// the 21 x 21 and 10 x 10 layers, 170.5 on all layers.  Round 4, timing probes with wrong results: ONE x fragment per dy row instead of
// three -- what deriving the dx = -1 / +1 fragments from the dx = 0 one with lane shifts would save at best -- 2.67 -> 2.43 ms per layer
// set, one per k-step 2.27: the nine transposing LDS reads per k-step are not what holds this kernel at 0.29 of the peak.  64-pixel
// stages in two buffers at two workgroups per CU (the next stage's loads under this stage's 36 MFMAs): 3.00 vs 2.86 ms per layer set,
// 15.55 vs 15.84 episodes/s; 128-pixel stages in two buffers where both fit beside a second workgroup (21 x 21 and 10 x 10 maps, 2 x 38.5 KB),
// after the fragment ring: 332 / 372 vs 333 / 372 us -- the partner workgroup already covers a stage's loads.)
template <int NTAP>
__global__ __launch_bounds__(256, 2) void rn_wgrad_kernel(RnWgradArgs a, int ci_tiles, int Ci32, int xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup id -> (output tile, pixel slab, episode).  Every tile of one (slab, episode) walks through the SAME dy / x pixels at
    // about the same time: with `xcd` such a group is dealt to one XCD (ids i and i + 8 share one), so those pixels are fetched into
    // that XCD's L2 once instead of once per XCD (L2 hit rate 37 % without).  Speed only: any placement is correct.
    const int ntile = ci_tiles * ((a.Cout + 63) / 64);
    int tileid, split, b;
    if (xcd) {
        const int x8 = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int grp = (j / ntile) * 8 + x8;
        tileid = j % ntile; split = grp % a.nsplit; b = grp / a.nsplit;
        if (b >= a.B) return;
    } else {
        tileid = blockIdx.x % ntile;
        const int r = blockIdx.x / ntile;
        split = r % a.nsplit; b = r / a.nsplit;
    }
    const int co0 = (tileid / ci_tiles) * 64, ci0 = (tileid % ci_tiles) * 64;
    const int hs = NTAP == 9 ? a.g.halo : 0, Wp = a.g.Wp;
    const int xrows = WG_PK + 2 * hs;
    const int xr8 = (xrows + 7) / 8 * 8;
    const int stage_bytes = (WG_PK + xr8) * 128;           // one stage: dy [128][128 B] then x [128 + 2 hs (rounded to 8)][128 B]
    // a wave whose 32 x 32 quadrant lies outside the matrix (Cout or Cin = 160: the last 64-tile is half empty) skips its products
    const bool live = co0 + (wave >> 1) * 32 < a.Cout && ci0 + (wave & 1) * 32 < Ci32;
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nchunks = (int)((a.npix + WG_PK - 1) / WG_PK);
    const int cps = (nchunks + a.nsplit - 1) / a.nsplit;
    const int cbeg = split * cps, cend = min(nchunks, cbeg + cps);
    const int nst = max(cend - cbeg, 0), Q = a.npair * nst;   // stages of this workgroup: (pair, pixel chunk)
    const int lr = lane >> 3, plim = (int)a.npix - 1;
    // LDS-direct loads of stage q (no registers, everything in flight at once); a wave instruction fills 8 rows, the swizzle goes on
    // the source address; 32-bit byte offsets off the episode's base.  Pixels outside the episode are clamped to its first / last
    // pixel: border pixels of the padded layout = zeros.
    auto issue = [&](int q, int buf) {
        const int pr = q >= nst ? 1 : 0;
        const int p0 = (cbeg + q - pr * nst) * WG_PK;
        const char* dyb = (const char*)((pr ? a.dy[1] : a.dy[0]) + (long)b * a.dy_stride + co0);
        const char* xb = (const char*)((pr ? a.x[1] : a.x[0]) + (long)b * a.x_stride + ci0);
        unsigned char* Dy = lds + buf * stage_bytes; unsigned char* Xs = Dy + WG_PK * 128;
        for (int r0 = uni(wave * 8); r0 < WG_PK; r0 += 32) {
            const int r = r0 + lr;
            int p = p0 + r;
            p = p > plim ? plim : p;
            int chunk = (lane & 7) ^ (((r >> 1) & 1) << 2);
            chunk = co0 + chunk * 8 < a.Cout ? chunk : 0;
            const unsigned off = ((unsigned)p * (unsigned)a.Cout + (unsigned)(chunk * 8)) * 2u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dyb + off),
                                             (__attribute__((address_space(3))) void*)(Dy + r0 * 128), 16, 0, 0);
        }
        for (int r0 = uni(wave * 8); r0 < xrows; r0 += 32) {
            const int r = r0 + lr;
            int p = p0 - hs + r;
            p = p < 0 ? 0 : (p > plim ? plim : p);
            int chunk = (lane & 7) ^ (((r >> 1) & 1) << 2);
            chunk = ci0 + chunk * 8 < a.Cin ? chunk : 0;
            const unsigned off = ((unsigned)p * (unsigned)a.Cin + (unsigned)(chunk * 8)) * 2u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off),
                                             (__attribute__((address_space(3))) void*)(Xs + r0 * 128), 16, 0, 0);
        }
    };
    // A lane's two byte offsets into a stage image for the fragment whose first row is r0 (tr_frag's address arithmetic).  A k-step adds
    // 16 rows = 2 048 bytes and never touches bit 1 of the row, which is all the swizzle looks at: per tap the offsets are worked out ONCE
    // and every read of the stage loop is base + constant (ten vector instructions per fragment used to sit between the MFMAs).
    auto frag_off = [&](int r0, int cbase, int& o0, int& o1) {
        const int grp = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
        const int row = r0 + 8 * (grp >> 1) + fq;
        const int colb = (cbase + 16 * (grp & 1) + 4 * fp) * 2;
        o0 = row * 128 + (colb ^ (((row >> 1) & 1) << 6));
        o1 = (row + 4) * 128 + (colb ^ ((((row + 4) >> 1) & 1) << 6));
    };
    int xo0[NTAP], xo1[NTAP], yo0, yo1;
    frag_off(0, (wave >> 1) * 32, yo0, yo1);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) frag_off(hs + (NTAP == 9 ? (t / 3 - 1) * Wp + (t % 3 - 1) : 0), (wave & 1) * 32, xo0[t], xo1[t]);
    auto ld_frag = [&](const unsigned char* img, int o0, int o1, int kb) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + o0 + kb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + o1 + kb));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(rbf16x8, v);
    };
    auto compute = [&](int buf) {
        const unsigned char* Dy = lds + buf * stage_bytes; const unsigned char* Xs = Dy + WG_PK * 128;
        if constexpr (NTAP == 9) {
            // the 72 products of a stage as ONE sequence with a ring of RING x fragments read ahead (hipcc alone keeps one fragment in
            // flight: every MFMA then waits out most of an LDS latency)
            constexpr int KS = WG_PK / 16, NP = KS * NTAP, RING = 4;
            rbf16x8 ring[RING], af[2];
            af[0] = ld_frag(Dy, yo0, yo1, 0);
#pragma unroll
            for (int i = 0; i < RING; ++i) ring[i] = ld_frag(Xs, xo0[i % NTAP], xo1[i % NTAP], (i / NTAP) * 2048);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int ks = i / NTAP, t = i % NTAP;
                if (t == 0 && ks + 1 < KS) af[(ks + 1) & 1] = ld_frag(Dy, yo0, yo1, (ks + 1) * 2048);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1], ring[i % RING], acc[t], 0, 0, 0);
                const int n = i + RING;
                if (n < NP) ring[i % RING] = ld_frag(Xs, xo0[n % NTAP], xo1[n % NTAP], (n / NTAP) * 2048);
                __builtin_amdgcn_sched_barrier(0);                  // (pins this order: left alone the scheduler re-forms its one-ahead pattern)
            }
        } else {
#pragma unroll 2
            for (int ks = 0; ks < WG_PK / 16; ++ks) {
                const rbf16x8 af = ld_frag(Dy, yo0, yo1, ks * 2048);
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const rbf16x8 bf = ld_frag(Xs, xo0[t], xo1[t], ks * 2048);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
    };
    for (int q = 0; q < Q; ++q) {
        __syncthreads();
        issue(q, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (live) compute(0);
    }
    float* part = a.part + (((long)b * a.nsplit + split) * NTAP) * a.Cout * Ci32;
    const int ci = ci0 + (wave & 1) * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + (wave >> 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            if (co < a.Cout && ci < Ci32) part[((long)t * a.Cout + co) * Ci32 + ci] = acc[t][i];
        }
}

// workgroup = 64 consecutive ci x 4 rows; a wave reads 256 contiguous bytes of one slab row per step.  SPLIT4 (long slab lists: the
// 84 x 84 layers leave 128 slabs per episode and only 64 x 64 outputs -- a thread per output walking them one after the other was
// latency-bound, 450 us): the four waves share one (tap, co) row, wave g adds slabs g, g + 4, .. and the four sums are added in
// fixed order through LDS.  Otherwise the waves are four consecutive co rows.  Writes go to torch's OIHW order.
template <bool SPLIT4>
__global__ __launch_bounds__(256) void rn_wgrad_reduce_kernel(int nsplit, int ntaps, int Cout, int Ci32, int Cin_real, const float* part,
                                                              float* G, long gstride) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cib = (Cin_real + 63) / 64;
    const int rows = ntaps * Cout;                                     // (tap, co) rows of a slab
    int row, k0, kstep;
    const int ci = (blockIdx.x % cib) * 64 + lane;
    if (SPLIT4) { row = blockIdx.x / cib; k0 = wv; kstep = 4; }
    else { row = (blockIdx.x / cib) * 4 + wv; k0 = 0; kstep = 1; }
    const bool ok = row < rows && ci < Cin_real;
    const long slab = (long)rows * Ci32;
    float s = 0.f;
    if (ok) {
        const float* p = part + (long)b * nsplit * slab + (long)row * Ci32 + ci;
        float s0 = 0.f, s1 = 0.f;
        int k = k0;
        for (; k + kstep < nsplit; k += 2 * kstep) { s0 += p[(long)k * slab]; s1 += p[(long)(k + kstep) * slab]; }
        if (k < nsplit) s0 += p[(long)k * slab];
        s = s0 + s1;
    }
    if (SPLIT4) {
        red[wv][lane] = s;
        __syncthreads();
        if (wv) return;
        s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
    if (ok) {
        const int tap = row / Cout, co = row - tap * Cout;
        G[(long)b * gstride + ((long)co * Cin_real + ci) * ntaps + tap] = s;
    }
}

// The same sum for 3 x 3 layers with contiguous stores: torch's OIHW order puts the nine taps of one (co, ci) side by side, so a workgroup
// that owns ONE (tap, co) row scatters its 64 results 36 bytes apart (PMC: 174 MB written per launch for ~15 MB of gradient).  Here a
// workgroup owns all nine taps of (co, 64 ci): wave g adds slabs g, g + 4, .. of every tap, the four sums meet in LDS in fixed order, and
// the 576 results leave as one contiguous run.
__global__ __launch_bounds__(256) void rn_wgrad_reduce9_kernel(int nsplit, int Cout, int Ci32, int Cin_real, const float* part, float* G,
                                                               long gstride) {
    __shared__ float red[4][9][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cib = (Cin_real + 63) / 64;
    const int co = blockIdx.x / cib, ci0 = (blockIdx.x % cib) * 64, ci = ci0 + lane;
    const long rowst = (long)Ci32, tapst = (long)Cout * Ci32, slab = 9 * tapst;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.f;
    if (ci < Cin_real) {
        const float* p = part + (long)b * nsplit * slab + (long)co * rowst + ci;
        for (int k = wv; k < nsplit; k += 4) {
#pragma unroll
            for (int t = 0; t < 9; ++t) s[t] += p[(long)k * slab + t * tapst];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[wv][t][lane] = s[t];
    __syncthreads();
    const int nci = min(64, Cin_real - ci0);
    float* out = G + (long)b * gstride + ((long)co * Cin_real + ci0) * 9;
    for (int i = threadIdx.x; i < nci * 9; i += 256) {
        const int c = i / 9, t = i - c * 9;
        out[i] = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
    }
}

// fp32 OIHW master -> bf16 fragment copies.  sf / sb: the 16x16x32 fragment order (32-deep steps, 16-column blocks) for the forward /
// the backward-data copy, else the 32x32x16 order (16-deep steps, 32-column blocks); a 1 KiB block is 64 lanes x 8 elements either way
__global__ __launch_bounds__(256) void rn_wprep_kernel(int Cout, int Cin, int Cin_real, int ntaps, const float* W, long wstride,
                                                       rbf16* fwd, rbf16* bwd, long fstride, int sf, int sb) {
    const int b = blockIdx.y;
    const long nel = (long)ntaps * Cin * Cout;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    const float* w = W + (long)b * wstride;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long blk = i >> 9;
    {   // forward: blk = (tap * KS + ks) * CF + cb -- contraction over ci, columns co
        const int CF = sf ? Cout >> 4 : Cout >> 5, KS = sf ? Cin >> 5 : Cin >> 4;
        const int cb = (int)(blk % CF), ks = (int)((blk / CF) % KS), tap = (int)(blk / ((long)CF * KS));
        const int co = sf ? 16 * cb + (lane & 15) : 32 * cb + (lane & 31);
        const int ci = sf ? 32 * ks + 8 * (lane >> 4) + j : 16 * ks + 8 * (lane >> 5) + j;
        const float v = ci < Cin_real ? w[((long)co * Cin_real + ci) * ntaps + tap] : 0.f;
        fwd[(long)b * fstride + i] = rn_f2bf(v);
    }
    if (bwd) {  // backward data: blk = (tap * KSo + ks) * CFi + cb, contraction over co, columns ci; tap flipped
        const int CFi = sb ? Cin >> 4 : Cin >> 5, KSo = sb ? Cout >> 5 : Cout >> 4;
        const int cb = (int)(blk % CFi), ks = (int)((blk / CFi) % KSo), tap = (int)(blk / ((long)CFi * KSo));
        const int ci = sb ? 16 * cb + (lane & 15) : 32 * cb + (lane & 31);
        const int co = sb ? 32 * ks + 8 * (lane >> 4) + j : 16 * ks + 8 * (lane >> 5) + j;
        bwd[(long)b * fstride + i] = rn_f2bf(w[((long)co * Cin_real + ci) * ntaps + (ntaps - 1 - tap)]);
    }
}

}  // namespace

// upper bound of the statistics slabs a convolution writes per episode: one per pixel tile (128 or 256 interior pixels); with tiles
// that restart at every image (rn_per_image) that is M * ceil(H W / 128), which can exceed ceil(M Pp / 128) (23 x 23: 5 M vs 4.88 M)
int rn_conv_tiles(long npix, const RnGeom& g) {
    const long M = npix / g.Pp, per_image = M * (((long)g.H * g.W + 127) / 128), flat = (npix + 127) / 128;
    return (int)(per_image > flat ? per_image : flat);
}

size_t rn_conv_lds_bytes(const RnGeom& g, int Cout) {
    switch (conv_nf(Cout)) {
        case 5: return ConvCfg<5, 1, 4>::lds_bytes(g);
        case 4: return ConvCfg<4, 1, 4>::lds_bytes(g);
        case 3: return ConvCfg<3, 1, 4>::lds_bytes(g);
        case 2: return ConvCfg<2, 1, 4>::lds_bytes(g);
        default: return ConvCfg<1, 1, 4>::lds_bytes(g);
    }
}

int launch_rn_conv(hipStream_t st, const RnConvArgs& a, int* nt_out) {
    if (a.npix >= (1L << 31) - 4096) return FUMI_ENOTSUP;            // (pixel indices are 32-bit inside the kernels)
    for (int s = 0; s < a.nsrc; ++s)
        if (a.npix * a.src[s].Cin * 2 >= (1L << 32) - 65536) return FUMI_ENOTSUP;      // (byte offsets within an episode's map: 32-bit)
    if (a.B < 1 || a.nsrc < 1 || a.nsrc > 4 || a.Cout < 32 || (a.Cout & 31) || a.npix < 1) return FUMI_EINVAL;
    if (a.npix * a.Cout >= (1L << 31) - 65536) return FUMI_ENOTSUP;                        // (element offsets within an episode's output map: 32-bit)
    for (int s = 0; s < a.nsrc; ++s)
        if (!a.src[s].in || !a.src[s].frag || a.src[s].Cin < 16 || (a.src[s].Cin & 15) || (a.src[s].ntaps != 9 && a.src[s].ntaps != 1))
            return FUMI_EINVAL;
    switch (conv_nf(a.Cout)) {
        case 5: return conv_dispatch<5>(st, a, nt_out);
        case 4: return conv_dispatch<4>(st, a, nt_out);
        case 3: return conv_dispatch<3>(st, a, nt_out);
        case 2: return conv_dispatch<2>(st, a, nt_out);
        default: return conv_dispatch<1>(st, a, nt_out);
    }
}

// slabs of the pixel axis per episode.  A launch is a few hundred to a few thousand LONG workgroups on 512 slots (256 CUs x 2): the
// chip runs them in rounds, and 1080 workgroups cost three rounds where 1008 cost two -- so the count is chosen for the fullest
// last round among the splits that fill the chip one to four times over (at least 4 stages per workgroup).
int rn_wgrad_nsplit(int B_chunk, long npix, int Cin, int Cout) {
    // (priced for RN_BREF episodes per chunk whatever the chunk holds: the split, and with it an episode's summation order, must not
    // depend on the chunk size; a chunk of 2 RN_BREF episodes has twice the workgroups and the same fill of its rounds)
    (void)B_chunk;
    const int B = RN_BREF;
    const int Ci32 = (Cin + 31) / 32 * 32;
    const long tiles = (long)((Cout + 63) / 64) * ((Ci32 + 63) / 64) * B;
    const long chunks = (npix + WG_PK - 1) / WG_PK;
    static const int legacy = getenv("FUMI_RN_WSPLIT") ? atoi(getenv("FUMI_RN_WSPLIT")) : 0;     // dev knob: 1 = ceil(1024 / tiles)
    long nsmax = chunks / 4;
    nsmax = nsmax < 1 ? 1 : (nsmax > 512 ? 512 : nsmax);
    long ns;
    if (legacy) {
        ns = (1024 + tiles - 1) / tiles;
        ns = ns > nsmax ? nsmax : ns;
    } else {
        // (256, not the 512 slots of the chip: with two lanes of chunks in flight the other lane's kernels fill what a launch leaves,
        //  and half the slabs are half the partial-sum traffic -- 17.36-17.42 -> 17.54-17.63 episodes/s; 128: the same)
        static const long slots = getenv("FUMI_RN_WSLOTS") ? atol(getenv("FUMI_RN_WSLOTS")) : 256;
        auto eff = [&](long c) {
            const long w = tiles * c, rounds = (w + slots - 1) / slots;
            const double e = (double)w / (double)(rounds * slots);    // share of the rounds' slots that hold a workgroup
            return rounds == 1 ? 0.9 * e : e;                         // (a single round leaves nothing to cover the launch's ramps)
        };
        long cmax = nsmax;
        while (cmax > 1 && tiles * cmax > 6 * slots) --cmax;          // at most six rounds (every slab is reduced again afterwards)
        double best = -1.0;
        for (long c = 1; c <= cmax; ++c) best = eff(c) > best ? eff(c) : best;
        ns = 1;
        while (ns < cmax && eff(ns) < best - 0.03) ++ns;              // the fewest slabs within 3 % of the best fill
    }
    if (ns < 1) ns = 1;
    if (ns * B >= 8 && B < 8 && 8 % B == 0) ns = (ns + 8 / B - 1) / (8 / B) * (8 / B);     // slabs x episodes in multiples of 8: XCD-grouped ids
    return (int)ns;
}

int launch_rn_wgrad(hipStream_t st, const RnWgradArgs& a) {
    if (a.B < 1 || a.npair < 1 || a.npair > 2 || (a.Cin & 15) || (a.Cout & 31) || a.nsplit < 1 || (a.ntaps != 9 && a.ntaps != 1))
        return FUMI_EINVAL;
    if (a.npix * (a.Cin > a.Cout ? a.Cin : a.Cout) * 2 >= (1L << 32) - 65536) return FUMI_ENOTSUP;   // (32-bit byte offsets inside the kernel)
    const int Ci32 = (a.Cin + 31) / 32 * 32;
    const int ci_tiles = (Ci32 + 63) / 64, co_tiles = (a.Cout + 63) / 64;
    const int hs = a.ntaps == 9 ? a.g.halo : 0;
    const int stage = (WG_PK + (WG_PK + 2 * hs + 7) / 8 * 8) * 128;
    const int lds = stage + 64;
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    const int groups = a.nsplit * a.B;
    static const int xcd_env = getenv("FUMI_RN_XCD") ? atoi(getenv("FUMI_RN_XCD")) : 1;
    const int xcd = xcd_env && groups >= 8 && groups % 8 == 0;
    const dim3 grid((unsigned)((long)co_tiles * ci_tiles * groups));
    if (a.ntaps == 9) {
        FUMI_SET_DYN_LDS(rn_wgrad_kernel<9>, lds);
        hipLaunchKernelGGL(rn_wgrad_kernel<9>, grid, dim3(256), lds, st, a, ci_tiles, Ci32, xcd);
    } else {
        FUMI_SET_DYN_LDS(rn_wgrad_kernel<1>, lds);
        hipLaunchKernelGGL(rn_wgrad_kernel<1>, grid, dim3(256), lds, st, a, ci_tiles, Ci32, xcd);
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_wgrad_reduce(hipStream_t st, int B, int nsplit, int ntaps, int Cout, int Cin, int Cin_real, const float* part,
                           float* G, long gstride) {
    const int Ci32 = (Cin + 31) / 32 * 32;
    const int cib = (Cin_real + 63) / 64, rows = ntaps * Cout;
    if (ntaps == 9)
        hipLaunchKernelGGL(rn_wgrad_reduce9_kernel, dim3((unsigned)(Cout * cib), B), dim3(256), 0, st, nsplit, Cout, Ci32, Cin_real, part, G,
                           gstride);
    else if (nsplit >= 16)
        hipLaunchKernelGGL(rn_wgrad_reduce_kernel<true>, dim3((unsigned)(rows * cib), B), dim3(256), 0, st, nsplit, ntaps, Cout, Ci32,
                           Cin_real, part, G, gstride);
    else
        hipLaunchKernelGGL(rn_wgrad_reduce_kernel<false>, dim3((unsigned)((rows + 3) / 4 * cib), B), dim3(256), 0, st, nsplit, ntaps, Cout,
                           Ci32, Cin_real, part, G, gstride);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_wprep(hipStream_t st, int B, int Cout, int Cin, int Cin_real, int ntaps, const float* W, long wstride,
                    rbf16* fwd, rbf16* bwd, long fstride) {
    if ((Cout & 31) || (Cin & 15) || (bwd && (Cin & 31))) return FUMI_EINVAL;
    const long nel = (long)ntaps * Cin * Cout;
    hipLaunchKernelGGL(rn_wprep_kernel, dim3((unsigned)((nel + 255) / 256), B), dim3(256), 0, st, Cout, Cin, Cin_real, ntaps, W, wstride,
                       fwd, bwd, fstride, rn_use_s16(Cin) ? 1 : 0, rn_use_s16(Cout) ? 1 : 0);
    LAUNCH_CHECK();
    return FUMI_OK;
}
