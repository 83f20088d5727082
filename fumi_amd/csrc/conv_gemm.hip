// 3x3 / pad 1 convolutions of the Conv4 encoder on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32).
// Forward, input-gradient (the same kernel with the flipped, channel-swapped weights) and weight-gradient products, each
// for every episode of a meta-batch in ONE launch with per-episode (fast) weights.  Layouts and the reasoning behind them:
// conv4.h.  The set {forward, backward-data, backward-weight} is closed under differentiation, so the second-order sweep
// (oracle/conv4_manual.py) needs nothing else: its tangent passes are these kernels with two sources.
#include "conv4.h"

namespace {

// consecutive logical ids on one XCD (blocks id and id + 8 share an XCD): the workgroups of an episode then share that
// XCD's L2 for the episode's weights.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Shared epilogue of the two forward kernels.  A wave holds the [32 pixels x 64 channels] block of its tile in two
// accumulators (column = channel = lane & 31 (+32), row = pixel = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).  Border pixels of
// the padded grid are written as 0 (the buffer stays a valid convolution input) and left out of the statistics.
__device__ __forceinline__ void conv_epilogue(const f32x16& acc0, const f32x16& acc1, float* lds, const CvGeom& g, long npix,
                                              long ep0, long g0, int b, int t, int tiles, float* out, float* stats,
                                              const float* dot, const f32x16* dreg0 = nullptr, const f32x16* dreg1 = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long gw = g0 + wave * 32;
    unsigned mask;
    {
        const int gp = (int)gw + r;                              // (pixels per episode < 2^22: launch_* check it)
        int im, p, y, x;
        cv_divmod(gp, g.Pp, 1.0f / (float)g.Pp, im, p);
        cv_divmod(p, g.Wp, 1.0f / (float)g.Wp, y, x);
        const bool in = gp < npix && x >= 1 && x <= g.W && y >= 1 && y <= g.H;
        mask = (unsigned)__ballot(in);
    }
    float* o = out + (ep0 + gw) * 64;
    const float* d = dot ? dot + (ep0 + gw) * 64 : nullptr;
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool in = (mask >> row) & 1u;
        const bool st = gw + row < npix;
        const float v0 = in ? acc0[i] : 0.f, v1 = in ? acc1[i] : 0.f;
        if (st && out) { o[row * 64 + r] = v0; o[row * 64 + 32 + r] = v1; }
        if (stats) {
            float e0 = v0, e1 = v1;
            if (dreg0) { e0 = (*dreg0)[i]; e1 = (*dreg1)[i]; }
            else if (d) { const int rr = st ? row : 0; e0 = d[rr * 64 + r]; e1 = d[rr * 64 + 32 + r]; }
            s1a += v0; s2a += v0 * e0; s1b += v1; s2b += v1 * e1;
        }
    }
    if (stats) {
        s1a += __shfl_xor(s1a, 32); s2a += __shfl_xor(s2a, 32); s1b += __shfl_xor(s1b, 32); s2b += __shfl_xor(s2b, 32);
        __syncthreads();                                   // every wave is done with the staged patch
        if (h == 0) {
            float* red = lds + wave * 128;                 // [wave][k][64]
            red[r] = s1a; red[32 + r] = s1b; red[64 + r] = s2a; red[96 + r] = s2b;
        }
        __syncthreads();
        if (tid < 128) {
            const float s = (lds[tid] + lds[128 + tid]) + (lds[256 + tid] + lds[384 + tid]);
            stats[((long)b * tiles + t) * 128 + tid] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels.  One workgroup = 128 consecutive padded pixels of one episode x all 64 output channels; wave w owns
// pixels 32w .. 32w+31.  The input slab (tile + halo) is staged once per source into LDS with the 16-byte chunk index
// XOR-ed with (pixel & 15): the A-fragment read (ds_read_b128 of channels 8j + 4 (lane>>5) .. +3 of pixel lane&31 + shift)
// is conflict-free for every tap without padding.  Weights: two coalesced 1 KiB loads per 8 MFMAs, straight from L1 / L2.
// LDS 56 KiB at 42 x 42 -> two workgroups per CU: one stages while the other multiplies.
// ------------------------------------------------------------------------------------------------------------
template <int NSRC>
__global__ __launch_bounds__(256, 2) void conv64_kernel(Conv64Args a, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / tiles, t = L - b * tiles;
    const long ep0 = (long)b * a.npix, g0 = (long)t * CV_TILE;
    const int halo = a.g.halo, Wp = a.g.Wp;
    const int nchunk = (CV_TILE + 2 * halo) * 16;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        if (s) __syncthreads();
        const float* src = a.in[s] + ep0 * 64;
        const long lo = g0 - halo;
        for (int i0 = tid; i0 < nchunk; i0 += 4 * 256) {      // 4 independent 16-byte loads in flight per thread
            f32x4 v[4]; int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                const int pi = i >> 4, q = i & 15;
                const long gp = lo + pi;
                const bool ok = i < nchunk && gp >= 0 && gp < a.npix;
                const f32x4 t4 = *(const f32x4*)(src + (ok ? gp * 64 + q * 4 : 0));
                v[u] = ok ? t4 : z4;
                dst[u] = i < nchunk ? pi * 64 + ((q ^ (pi & 15)) << 2) : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (dst[u] >= 0) *(f32x4*)(lds + dst[u]) = v[u];
        }
        __syncthreads();
        const float* fr = a.frag[s] + (long)b * a.frag_stride[s] + lane * 4;
        const int pbase = halo + wave * 32 + r;
        // 72 steps (tap, j) of 8 MFMAs.  hipcc left to itself issues each weight load right before its use and waits for
        // it (vmcnt(0) every 4 MFMAs: 55 % of the matrix rate); here the weight fragments of the next CV_DB steps are always
        // in flight in a register ring and the activation fragment of the next step is read from LDS one step ahead.
        constexpr int NIT = 72, CV_DB = 4;
        auto ldB = [&](int it, int ct) { return *(const f32x4*)(fr + (it * 2 + ct) * 256); };
        auto ldA = [&](int it) {
            const int tap = it >> 3, j = it & 7;
            const int pi = pbase + (tap / 3 - 1) * Wp + (tap % 3 - 1);
            return *(const f32x4*)(lds + pi * 64 + (((2 * j + h) ^ (pi & 15)) << 2));
        };
        f32x4 Bq[CV_DB][2];
#pragma unroll
        for (int d = 0; d < CV_DB; ++d) { Bq[d][0] = ldB(d, 0); Bq[d][1] = ldB(d, 1); }
        f32x4 Acur = ldA(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 Anext = Acur;
            if (it + 1 < NIT) Anext = ldA(it + 1);
            const f32x4 B0 = Bq[it % CV_DB][0], B1 = Bq[it % CV_DB][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc0 = mfma32(Acur[i], B0[i], acc0); acc1 = mfma32(Acur[i], B1[i], acc1); }
            if (it + CV_DB < NIT) { Bq[it % CV_DB][0] = ldB(it + CV_DB, 0); Bq[it % CV_DB][1] = ldB(it + CV_DB, 1); }
            Acur = Anext;
            __builtin_amdgcn_sched_barrier(0);           // (the scheduler otherwise sinks the prefetches back to their uses)
        }
    }
    conv_epilogue(acc0, acc1, lds, a.g, a.npix, ep0, g0, b, t, tiles, a.out, a.stats, a.dot);
}

// ------------------------------------------------------------------------------------------------------------
// first block: Cin <= 4 image planes (dense NCHW, unpadded) -> 64 channels.  K = 9 Cin (27): the product is small, the
// kernel is bound by writing its output.  The padded image slab of the tile is built in LDS plane by plane; the MFMA's
// k index is kappa = c * 9 + tap, two per instruction (lane half h takes kappa = 2m + h).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv1_kernel(Conv1Args a, long npix, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / tiles, t = L - b * tiles;
    const long ep0 = (long)b * npix, g0 = (long)t * CV_TILE;
    const int halo = a.g.halo, Wp = a.g.Wp, Pp = a.g.Pp, H = a.g.H, W = a.g.W;
    const int npatch = CV_TILE + 2 * halo;
    const float* img = a.img + (long)b * a.M * a.Cin * H * W;
    const float rPp = 1.0f / (float)Pp, rWp = 1.0f / (float)Wp;
    for (int i = tid; i < a.Cin * npatch; i += 256) {
        const int c = i >= 2 * npatch ? 2 : i >= npatch ? 1 : 0, pi = i - c * npatch;
        const int gp = (int)g0 - halo + pi;
        float v = 0.f;
        if (gp >= 0 && gp < npix) {
            int im, p, y, x;
            cv_divmod(gp, Pp, rPp, im, p);
            cv_divmod(p, Wp, rWp, y, x);
            if (x >= 1 && x <= W && y >= 1 && y <= H) v = img[((long)im * a.Cin + c) * H * W + (y - 1) * W + (x - 1)];
        }
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int nk = (a.Cin * 9 + 1) >> 1;
    const float* fr = a.frag + (long)b * a.frag_stride + lane;
    const float* fd = a.frag_dot ? a.frag_dot + (long)b * a.frag_dot_stride + lane : nullptr;
    f32x16 dc0, dc1;                                             // the dot operand conv(img, frag_dot), when asked for
#pragma unroll
    for (int i = 0; i < 16; ++i) { dc0[i] = 0.f; dc1[i] = 0.f; }
    const int pbase = halo + wave * 32 + r;
    for (int m = 0; m < nk; ++m) {
        const int kap = 2 * m + h;
        const int c = kap / 9, tap = kap - c * 9;
        const bool kok = kap < a.Cin * 9;
        const int off = kok ? c * npatch + (tap / 3 - 1) * Wp + (tap % 3 - 1) : 0;
        const float av = lds[off + pbase];                       // (the weight of an invalid kappa is 0)
        const float b0 = fr[(m * 2 + 0) * 64], b1 = fr[(m * 2 + 1) * 64];
        acc0 = mfma32(av, b0, acc0); acc1 = mfma32(av, b1, acc1);
        if (fd) { dc0 = mfma32(av, fd[(m * 2 + 0) * 64], dc0); dc1 = mfma32(av, fd[(m * 2 + 1) * 64], dc1); }
    }
    __syncthreads();
    conv_epilogue(acc0, acc1, lds, a.g, npix, ep0, g0, b, t, tiles, a.out, a.stats, a.dot, fd ? &dc0 : nullptr, fd ? &dc1 : nullptr);
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient, 64 x 64 channels: dW[tap][co][ci] = sum_pix dy[pix][co] x[pix + off_tap][ci].  Workgroup (episode b,
// slab s of the pixel axis): the four waves own the four 32 x 32 quadrants of the [co][ci] matrix for ALL 9 taps (9
// accumulators = 144 registers), the contraction runs over the slab's pixels in steps of 64 staged in LDS (dy tile + x
// tile with halo, both read with conflict-free ds_read_b32: consecutive lanes = consecutive channels).
// ------------------------------------------------------------------------------------------------------------
constexpr int WG_PT = 64;
template <int NSRC>
__global__ __launch_bounds__(256, 2) void wgrad64_kernel(Wgrad64Args a, long chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave >> 1, it = wave & 1;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / a.nsplit, sp = L - b * a.nsplit;
    const long ep0 = (long)b * a.npix;
    const long p_beg = (long)sp * chunk, p_end = min(a.npix, p_beg + chunk);
    const int halo = a.g.halo, Wp = a.g.Wp;
    const int nxp = WG_PT + 2 * halo;
    float* dyl = lds;                      // [WG_PT][64]
    float* xl = lds + WG_PT * 64;          // [nxp][64]
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (long p0 = p_beg; p0 < p_end; p0 += WG_PT) {
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            __syncthreads();
            const float* xs = a.x[s] + ep0 * 64;
            const float* ds = a.dy[s] + ep0 * 64;
            const int nch = (WG_PT + nxp) * 16;                  // 16-byte chunks: dy tile first, then the x slab
            for (int i0 = tid; i0 < nch; i0 += 4 * 256) {
                f32x4 v[4]; int dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 256;
                    const bool isx = i >= WG_PT * 16;
                    const int k = isx ? i - WG_PT * 16 : i;
                    const int pi = k >> 4, q = k & 15;
                    const long gp = isx ? p0 - halo + pi : p0 + pi;
                    const bool ok = i < nch && gp >= 0 && gp < (isx ? a.npix : p_end);
                    const float* base = isx ? xs : ds;
                    const f32x4 t4 = *(const f32x4*)(base + (ok ? gp * 64 + q * 4 : 0));
                    v[u] = ok ? t4 : z4;
                    dst[u] = i < nch ? (isx ? WG_PT * 64 : 0) + pi * 64 + q * 4 : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (dst[u] >= 0) *(f32x4*)(lds + dst[u]) = v[u];
            }
            __syncthreads();
            const float* ap = dyl + h * 64 + ct * 32 + r;
            const float* bp = xl + (halo + h) * 64 + it * 32 + r;
#pragma unroll 4
            for (int m = 0; m < WG_PT / 2; ++m) {
                const float av = ap[m * 128];
                const float* bq = bp + m * 128;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float bv = bq[((tap / 3 - 1) * Wp + (tap % 3 - 1)) * 64];
                    acc[tap] = mfma32(av, bv, acc[tap]);
                }
            }
        }
    }
    float* o = a.part + ((long)b * a.nsplit + sp) * (9 * 64 * 64) + (ct * 32) * 64 + it * 32 + r;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            o[tap * 4096 + row * 64] = acc[tap][i];
        }
}

// first block's weight gradient: dW1[co][kappa] = sum_pix dy[pix][co] img_c[pix + off_tap], kappa = c * 9 + tap < 32.
// Waves: (column tile of co) x (half of the staged pixels); partial slabs [b][split][64][32].
constexpr int W1_PT = 256;
__global__ __launch_bounds__(256, 2) void wgrad1_kernel(Wgrad1Args a, long npix, long chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 1, ph = wave >> 1;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / a.nsplit, sp = L - b * a.nsplit;
    const long ep0 = (long)b * npix;
    const long p_beg = (long)sp * chunk, p_end = min(npix, p_beg + chunk);
    const int halo = a.g.halo, Wp = a.g.Wp, Pp = a.g.Pp, H = a.g.H, W = a.g.W;
    const int npatch = W1_PT + 2 * halo;
    float* dyl = lds;                      // [W1_PT][64]
    float* pl = lds + W1_PT * 64;          // [Cin][npatch]
    const float* img = a.img + (long)b * a.M * a.Cin * H * W;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // this lane's B column: kappa = r
    const int kc = r / 9, ktap = r - kc * 9;
    const bool kok = r < a.Cin * 9;
    const int koff = kok ? kc * npatch + halo + (ktap / 3 - 1) * Wp + (ktap % 3 - 1) : 0;
    for (long p0 = p_beg; p0 < p_end; p0 += W1_PT) {
        __syncthreads();
        const float* ds = a.dy + ep0 * 64;
        for (int i0 = tid; i0 < W1_PT * 16; i0 += 4 * 256) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                const long gp = p0 + (i >> 4);
                const bool ok = gp < p_end;
                const f32x4 t4 = *(const f32x4*)(ds + (ok ? gp * 64 + (i & 15) * 4 : 0));
                v[u] = ok ? t4 : z4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) *(f32x4*)(dyl + (i0 + u * 256) * 4) = v[u];
        }
        for (int i = tid; i < a.Cin * npatch; i += 256) {
            const int c = i >= 2 * npatch ? 2 : i >= npatch ? 1 : 0, pi = i - c * npatch;
            const int gp = (int)p0 - halo + pi;
            float v = 0.f;
            if (gp >= 0 && gp < npix) {
                int im, p, y, x;
                cv_divmod(gp, Pp, 1.0f / (float)Pp, im, p);
                cv_divmod(p, Wp, 1.0f / (float)Wp, y, x);
                if (x >= 1 && x <= W && y >= 1 && y <= H) v = img[((long)im * a.Cin + c) * H * W + (y - 1) * W + (x - 1)];
            }
            pl[i] = v;
        }
        __syncthreads();
        const float* ap = dyl + (ph * (W1_PT / 2) + h) * 64 + ct * 32 + r;
        const float* bp = pl + koff + ph * (W1_PT / 2) + h;
#pragma unroll 8
        for (int m = 0; m < W1_PT / 4; ++m) {
            const float av = ap[m * 128];
            const float bv = kok ? bp[m * 2] : 0.f;
            acc = mfma32(av, bv, acc);
        }
    }
    __syncthreads();
    float* red = lds;                                              // [2 halves][64 co][32]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        red[(ph * 64 + ct * 32 + row) * 32 + r] = acc[i];
    }
    __syncthreads();
    float* o = a.part + ((long)b * a.nsplit + sp) * 2048;
    for (int i = tid; i < 2048; i += 256) o[i] = red[i] + red[2048 + i];
}

// ------------------------------------------------------------------------------------------------------------
// weight bookkeeping
// ------------------------------------------------------------------------------------------------------------
__global__ void wfrag64_kernel(int n, const float* toi, long toi_stride, float* ff, float* fb, long frag_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * CV_WFRAG) return;
    const int w = (int)(i / CV_WFRAG), e = (int)(i - (long)w * CV_WFRAG);
    const int ii = e & 3, lane = (e >> 2) & 63, ctl = (e >> 8) & 1, j = (e >> 9) & 7, tap = e >> 12;
    const int col = 32 * ctl + (lane & 31), k = 8 * j + 4 * (lane >> 5) + ii;
    const float* W = toi + (long)w * toi_stride;
    if (ff) ff[(long)w * frag_stride + e] = W[(tap * 64 + col) * 64 + k];                 // out = col (co), in = k (ci)
    if (fb) fb[(long)w * frag_stride + e] = W[((8 - tap) * 64 + k) * 64 + col];           // out = col (ci), in = k (co), flipped
}

__global__ void wfrag1_kernel(int n, int Cin, const float* w1, float* frag) {
    const int nf = cv_frag1_floats(Cin);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * nf) return;
    const int w = (int)(i / nf), e = (int)(i - (long)w * nf);
    const int lane = e & 63, ctl = (e >> 6) & 1, m = e >> 7;
    const int kap = 2 * m + (lane >> 5), co = 32 * ctl + (lane & 31);
    frag[i] = kap < Cin * 9 ? w1[((long)w * 64 + co) * 32 + kap] : 0.f;
}

// OIHW [64][64][3][3] -> TOI [9][64][64]
__global__ void oihw_to_toi_kernel(int n, const float* src, float* dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * 36864) return;
    const int w = (int)(i / 36864), e = (int)(i - (long)w * 36864);
    const int ci = e & 63, co = (e >> 6) & 63, tap = e >> 12;
    dst[i] = src[(long)w * 36864 + (co * 64 + ci) * 9 + tap];
}
__global__ void toi_to_oihw_kernel(int n, const float* src, float* dst, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * 36864) return;
    const int w = (int)(i / 36864), e = (int)(i - (long)w * 36864);
    const int tap = e % 9, ci = (e / 9) & 63, co = e / 576;
    dst[i] = scale * src[(long)w * 36864 + (tap * 64 + co) * 64 + ci];
}

}  // namespace

int launch_conv64(hipStream_t st, const Conv64Args& a) {
    if (a.B < 1 || a.npix < 1 || a.nsrc < 1 || a.nsrc > 2) return FUMI_EINVAL;
    if (a.npix >= (1L << 22)) return FUMI_ENOTSUP;            // (per-episode pixel indices are 22-bit in the kernels)
    const int tiles = cv_tiles(a.npix);
    const size_t lds = (size_t)(CV_TILE + 2 * a.g.halo) * 256;
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    if (a.nsrc == 1) {
        FUMI_SET_DYN_LDS(conv64_kernel<1>, lds);
        hipLaunchKernelGGL(conv64_kernel<1>, dim3(a.B * tiles), dim3(256), lds, st, a, tiles);
    } else {
        FUMI_SET_DYN_LDS(conv64_kernel<2>, lds);
        hipLaunchKernelGGL(conv64_kernel<2>, dim3(a.B * tiles), dim3(256), lds, st, a, tiles);
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_conv1(hipStream_t st, const Conv1Args& a) {
    if (a.B < 1 || a.M < 1 || a.Cin < 1 || a.Cin > 3) return FUMI_EINVAL;
    const long npix = (long)a.M * a.g.Pp;
    if (npix >= (1L << 22)) return FUMI_ENOTSUP;
    const int tiles = cv_tiles(npix);
    size_t lds = (size_t)a.Cin * (CV_TILE + 2 * a.g.halo) * 4;
    if (lds < 2048) lds = 2048;
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    FUMI_SET_DYN_LDS(conv1_kernel, lds);
    hipLaunchKernelGGL(conv1_kernel, dim3(a.B * tiles), dim3(256), lds, st, a, npix, tiles);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int cv_wgrad_nsplit(int B, long npix) {
    int ns = (768 + B - 1) / B;                               // ~3 workgroups per CU in flight
    const long maxs = (npix + 4 * WG_PT - 1) / (4 * WG_PT);   // at least 4 staged steps per slab
    if (ns > maxs) ns = (int)maxs;
    return ns < 1 ? 1 : ns;
}

int launch_wgrad64(hipStream_t st, const Wgrad64Args& a) {
    if (a.B < 1 || a.npix < 1 || a.nsrc < 1 || a.nsrc > 2 || a.nsplit < 1) return FUMI_EINVAL;
    long chunk = (a.npix + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + WG_PT - 1) / WG_PT * WG_PT;
    const size_t lds = (size_t)(WG_PT + WG_PT + 2 * a.g.halo) * 256;
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    if (a.nsrc == 1) {
        FUMI_SET_DYN_LDS(wgrad64_kernel<1>, lds);
        hipLaunchKernelGGL(wgrad64_kernel<1>, dim3(a.B * a.nsplit), dim3(256), lds, st, a, chunk);
    } else {
        FUMI_SET_DYN_LDS(wgrad64_kernel<2>, lds);
        hipLaunchKernelGGL(wgrad64_kernel<2>, dim3(a.B * a.nsplit), dim3(256), lds, st, a, chunk);
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_wgrad1(hipStream_t st, const Wgrad1Args& a) {
    if (a.B < 1 || a.M < 1 || a.Cin < 1 || a.Cin > 3 || a.nsplit < 1) return FUMI_EINVAL;      // kappa = 9 Cin <= 32 columns
    const long npix = (long)a.M * a.g.Pp;
    if (npix >= (1L << 22)) return FUMI_ENOTSUP;
    long chunk = (npix + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + W1_PT - 1) / W1_PT * W1_PT;
    size_t lds = (size_t)W1_PT * 256 + (size_t)a.Cin * (W1_PT + 2 * a.g.halo) * 4;
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    FUMI_SET_DYN_LDS(wgrad1_kernel, lds);
    hipLaunchKernelGGL(wgrad1_kernel, dim3(a.B * a.nsplit), dim3(256), lds, st, a, npix, chunk);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_wfrag64(hipStream_t st, int n, const float* toi, long toi_stride, float* ff, float* fb, long frag_stride) {
    const long tot = (long)n * CV_WFRAG;
    hipLaunchKernelGGL(wfrag64_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, toi, toi_stride, ff, fb, frag_stride);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_wfrag1(hipStream_t st, int n, int Cin, const float* w1, float* frag) {
    const long tot = (long)n * cv_frag1_floats(Cin);
    hipLaunchKernelGGL(wfrag1_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, Cin, w1, frag);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_oihw_to_toi(hipStream_t st, int n, const float* oihw, float* toi) {
    const long tot = (long)n * 36864;
    hipLaunchKernelGGL(oihw_to_toi_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, oihw, toi);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_toi_to_oihw(hipStream_t st, int n, const float* toi, float* oihw, float scale) {
    const long tot = (long)n * 36864;
    hipLaunchKernelGGL(toi_to_oihw_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, toi, oihw, scale);
    LAUNCH_CHECK();
    return FUMI_OK;
}
