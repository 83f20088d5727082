// Matrix products of the bf16 ResNet-12 encoder (rn12.h): forward / input-gradient convolutions as implicit GEMMs over shifted
// copies of the flattened padded pixel axis, weight gradients as pixel-contracted products, all on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation.  Hand-written for gfx950 (wave64, 160 KiB LDS, ds_read_b64_tr_b16).
#include "rn12.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }

// =====================================================================================================================================
// convolution (forward and input-gradient)
//   workgroup = 4 waves, tile = (128 MW) consecutive padded pixels x (32 NF) output channels; wave w owns pixels [32 MW w, 32 MW (w+1))
//   for every 64-channel chunk of every source: the input slab (tile + halo on both sides) goes to LDS ONCE as [pixel][64 ch] bf16 with
//   its 16-byte chunks XOR-ed by (pixel >> 1) & 7 (conflict-free ds_read_b128 for every tap shift); per tap the weight tile
//   (4 k-steps x NF fragments x 1 KiB, already in MFMA B-fragment order in memory) is double-buffered through LDS -- the copy is linear,
//   every wave reads each fragment with one ds_read_b128 -- while the matrix pipe works on the previous tap.
//   epilogue: the fp32 tile is rounded to bf16 into LDS, then rows leave with 16-byte stores (border pixels as 0) and the batch-norm
//   statistics (sum, sum of squares or of products with `dot`) of the STORED values are taken on the way.
// =====================================================================================================================================
constexpr int CV_KC = 64;                              // channels per staged chunk

template <int NF, int MW>
struct ConvCfg {
    static constexpr int MT = 128 * MW, NT = 32 * NF, NTP = NT + 8;            // NTP: padded row of the epilogue image (bf16)
    static constexpr int BT = 4 * NF * 1024;                                     // bytes of one tap's weight tile (64 channels)
    static constexpr int NCH = NT / 8, NRG = 256 / NCH;                          // epilogue: 16-byte chunks per row, row groups
    static __host__ __device__ int a_bytes(int halo) { return (MT + 2 * halo) * 128; }
    static __host__ __device__ int lds_bytes(int halo) {
        const int main_ = a_bytes(halo) + 2 * BT;
        const int epi = MT * NTP * 2 + NRG * NT * 2 * 4;
        return (main_ > epi ? main_ : epi) + 16;
    }
};

template <int NF, int MW>
__global__ __launch_bounds__(256) void rn_conv_kernel(RnConvArgs a) {
    typedef ConvCfg<NF, MW> C;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ RnSrc s_src[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, cg = blockIdx.y;
    const long p0 = (long)blockIdx.x * C::MT;
    const int halo = a.g.halo, Wp = a.g.Wp;
    if (tid == 0) { s_src[0] = a.src[0]; s_src[1] = a.src[1]; s_src[2] = a.src[2]; s_src[3] = a.src[3]; }
    unsigned char* const As = lds;
    unsigned char* const Bs = lds + C::a_bytes(halo);
    f32x16 acc[MW][NF];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][f][i] = 0.f;
    __syncthreads();
    const int CF = a.Cout >> 5;
    const int srow = tid >> 3, schunk = tid & 7;
    for (int s = 0; s < a.nsrc; ++s) {
        const RnSrc S = s_src[s];
        const int hs = S.ntaps == 9 ? halo : 0;
        const int rows = C::MT + 2 * hs;
        const int KS = S.Cin >> 4;
        const rbf16* in = S.in + (long)b * S.in_stride;
        const rbf16* frag = S.frag + (long)b * S.frag_stride;
        for (int c0 = 0; c0 < S.Cin; c0 += CV_KC) {
            const int kc = min(CV_KC, S.Cin - c0), nks = kc >> 4;
            __syncthreads();                                       // the previous chunk's products are done with As / Bs
            // ---- input slab -> LDS (unconditional loads from clamped addresses, masked at the LDS write)
            const bool chok = schunk * 8 < kc;
            const int chs = chok ? schunk * 8 : 0;
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
                    if (r < rows) *(u32x4*)(As + r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4)) = ok ? v[i] : (u32x4){0u, 0u, 0u, 0u};
                }
            }
            // ---- weight tiles: tap t's tile = nks x NF fragments of 1 KiB, contiguous per k-step
            const int units = nks * NF * 64;                       // 16-byte units of a tile
            u32x4 breg[NF];
            auto bload = [&](int tap) {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    int u = tid + 256 * i;
                    u = u < units ? u : units - 1;
                    const int ks = u / (NF * 64), rem = u - ks * (NF * 64);
                    breg[i] = ld16(frag + (((long)tap * KS + (c0 >> 4) + ks) * CF + cg * NF) * 512 + rem * 8);
                }
            };
            auto bstore = [&](int buf) {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int u = tid + 256 * i;
                    if (u < units) *(u32x4*)(Bs + buf * C::BT + u * 16) = breg[i];
                }
            };
            bload(0);
            bstore(0);
            __syncthreads();
            for (int t = 0; t < S.ntaps; ++t) {
                if (t + 1 < S.ntaps) bload(t + 1);
                const int toff = S.ntaps == 9 ? (t / 3 - 1) * Wp + (t % 3 - 1) : 0;
                const unsigned char* Bt = Bs + (t & 1) * C::BT;
                int arow[MW];
#pragma unroll
                for (int m = 0; m < MW; ++m) arow[m] = wave * 32 * MW + m * 32 + (lane & 31) + hs + toff;
                for (int ks = 0; ks < nks; ++ks) {
                    rbf16x8 bf[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) bf[f] = *(const rbf16x8*)(Bt + ((ks * NF + f) * 64 + lane) * 16);
#pragma unroll
                    for (int m = 0; m < MW; ++m) {
                        const int ch = ks * 2 + (lane >> 5);
                        const rbf16x8 af = *(const rbf16x8*)(As + arow[m] * 128 + ((ch ^ ((arow[m] >> 1) & 7)) << 4));
#pragma unroll
                        for (int f = 0; f < NF; ++f) acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[f], acc[m][f], 0, 0, 0);
                    }
                }
                if (t + 1 < S.ntaps) bstore((t + 1) & 1);
                __syncthreads();
            }
        }
    }
    // ---- epilogue: fp32 tile -> bf16 image in LDS (rows of NTP elements: the two lane halves land 16 banks apart)
    rbf16* Ot = (rbf16*)lds;
    float* red = (float*)(lds + C::MT * C::NTP * 2);
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wave * 32 * MW + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                Ot[row * C::NTP + f * 32 + (lane & 31)] = rn_f2bf(acc[m][f][i]);
            }
    __syncthreads();
    const int ch = tid % C::NCH, rg = tid / C::NCH;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (rg < C::NRG) {
        const unsigned Pp = a.g.Pp;
        unsigned q = (unsigned)((p0 + rg) % Pp);
        int y = q / (unsigned)Wp, x = q - y * Wp;
        rbf16* out = a.out + (long)b * a.out_stride + (long)cg * C::NT + ch * 8;
        const rbf16* dot = a.dot ? a.dot + (long)b * a.dot_stride + (long)cg * C::NT + ch * 8 : nullptr;
        for (int row = rg; row < C::MT; row += C::NRG) {
            const long p = p0 + row;
            if (p < a.npix) {
                const bool interior = y >= 1 && y <= a.g.H && x >= 1 && x <= a.g.W;
                u32x4 v = *(const u32x4*)(Ot + row * C::NTP + ch * 8);
                if (!interior) v = (u32x4){0u, 0u, 0u, 0u};
                *(u32x4*)(out + p * a.Cout) = v;
                if (a.stats && interior) {
                    u32x4 d = v;
                    if (dot) d = ld16(dot + p * a.Cout);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = __uint_as_float(v[j] << 16), hi = __uint_as_float(v[j] & 0xffff0000u);
                        const float dl = __uint_as_float(d[j] << 16), dh = __uint_as_float(d[j] & 0xffff0000u);
                        s1[2 * j] += lo; s1[2 * j + 1] += hi;
                        s2[2 * j] += lo * dl; s2[2 * j + 1] += hi * dh;
                    }
                }
            }
            x += C::NRG;
            while (x >= Wp) { x -= Wp; ++y; }
            while (y >= a.g.Hp) y -= a.g.Hp;
        }
    }
    if (a.stats) {
        if (rg < C::NRG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(rg * C::NT + ch * 8 + j) * 2] = s1[j];
                red[(rg * C::NT + ch * 8 + j) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < C::NT) {
            float t1 = 0.f, t2 = 0.f;
            for (int g = 0; g < C::NRG; ++g) { t1 += red[(g * C::NT + tid) * 2]; t2 += red[(g * C::NT + tid) * 2 + 1]; }
            float* st = a.stats + (((long)b * gridDim.x + blockIdx.x) * 2) * a.Cout + cg * C::NT + tid;
            st[0] = t1; st[a.Cout] = t2;
        }
    }
}

template <int NF, int MW>
int conv_launch(hipStream_t st, const RnConvArgs& a) {
    typedef ConvCfg<NF, MW> C;
    const int lds = C::lds_bytes(a.g.halo);
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    FUMI_SET_DYN_LDS((rn_conv_kernel<NF, MW>), lds);
    const dim3 grid((unsigned)((a.npix + C::MT - 1) / C::MT), a.Cout / C::NT, a.B);
    hipLaunchKernelGGL((rn_conv_kernel<NF, MW>), grid, dim3(256), lds, st, a);
    LAUNCH_CHECK();
    return FUMI_OK;
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

template <int NTAP>
__global__ __launch_bounds__(256) void rn_wgrad_kernel(RnWgradArgs a, int ci_tiles, int Ci32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, split = blockIdx.y;
    const int co0 = (blockIdx.x / ci_tiles) * 64, ci0 = (blockIdx.x % ci_tiles) * 64;
    const int hs = NTAP == 9 ? a.g.halo : 0, Wp = a.g.Wp;
    unsigned char* const Dy = lds;                         // [128][128 B]
    unsigned char* const Xs = lds + WG_PK * 128;           // [128 + 2 hs][128 B]
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const long nchunks = (a.npix + WG_PK - 1) / WG_PK;
    const long cps = (nchunks + a.nsplit - 1) / a.nsplit;
    const long cbeg = (long)split * cps, cend = min(nchunks, cbeg + cps);
    const int srow = tid >> 3, sch = tid & 7;
    const bool dyok = co0 + sch * 8 < a.Cout, xok = ci0 + sch * 8 < a.Cin;
    const int dych = dyok ? co0 + sch * 8 : 0, xch = xok ? ci0 + sch * 8 : 0;
    const int xrows = WG_PK + 2 * hs;
    for (int pr = 0; pr < a.npair; ++pr) {
        const rbf16* x = (pr ? a.x[1] : a.x[0]) + (long)b * a.x_stride;
        const rbf16* dy = (pr ? a.dy[1] : a.dy[0]) + (long)b * a.dy_stride;
        for (long c = cbeg; c < cend; ++c) {
            const long p0 = c * WG_PK;
            __syncthreads();
            {   // dy stage: 128 rows x 8 chunks = 4 units per thread
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long p = p0 + 32 * i + srow;
                    p = p >= a.npix ? a.npix - 1 : p;
                    v[i] = ld16(dy + p * a.Cout + dych);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 32 * i + srow;
                    const bool ok = dyok && p0 + r < a.npix;
                    *(u32x4*)(Dy + r * 128 + ((sch * 16) ^ (((r >> 1) & 1) << 6))) = ok ? v[i] : (u32x4){0u, 0u, 0u, 0u};
                }
            }
            for (int r0 = 0; r0 < xrows; r0 += 128) {
                u32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long p = p0 - hs + r0 + 32 * i + srow;
                    p = p < 0 ? 0 : (p >= a.npix ? a.npix - 1 : p);
                    v[i] = ld16(x + p * a.Cin + xch);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + 32 * i + srow;
                    const long p = p0 - hs + r;
                    const bool ok = xok && p >= 0 && p < a.npix;
                    if (r < xrows) *(u32x4*)(Xs + r * 128 + ((sch * 16) ^ (((r >> 1) & 1) << 6))) = ok ? v[i] : (u32x4){0u, 0u, 0u, 0u};
                }
            }
            __syncthreads();
#pragma unroll 2
            for (int ks = 0; ks < WG_PK / 16; ++ks) {
                const rbf16x8 af = tr_frag(Dy, ks * 16, (wave >> 1) * 32, lane);
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const int toff = NTAP == 9 ? (t / 3 - 1) * Wp + (t % 3 - 1) : 0;
                    const rbf16x8 bf = tr_frag(Xs, ks * 16 + hs + toff, (wave & 1) * 32, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
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

__global__ __launch_bounds__(256) void rn_wgrad_reduce_kernel(int nsplit, int ntaps, int Cout, int Ci32, int Cin_real, const float* part,
                                                              float* G, long gstride) {
    const int b = blockIdx.y;
    const long n = (long)Cout * Cin_real * ntaps;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tap = (int)(i % ntaps);
    const long r = i / ntaps;
    const int ci = (int)(r % Cin_real), co = (int)(r / Cin_real);
    const long slab = (long)ntaps * Cout * Ci32;
    const float* p = part + (long)b * nsplit * slab + ((long)tap * Cout + co) * Ci32 + ci;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += p[(long)k * slab];
    G[(long)b * gstride + i] = s;
}

// fp32 OIHW master -> bf16 fragment copies
__global__ __launch_bounds__(256) void rn_wprep_kernel(int Cout, int Cin, int Cin_real, int ntaps, const float* W, long wstride,
                                                       rbf16* fwd, rbf16* bwd, long fstride) {
    const int b = blockIdx.y;
    const long nel = (long)ntaps * Cin * Cout;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    const float* w = W + (long)b * wstride;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long blk = i >> 9;
    {   // forward: blk = (tap * KS + ks) * CF + cb
        const int CF = Cout >> 5, KS = Cin >> 4;
        const int cb = (int)(blk % CF), ks = (int)((blk / CF) % KS), tap = (int)(blk / ((long)CF * KS));
        const int co = 32 * cb + (lane & 31), ci = 16 * ks + 8 * (lane >> 5) + j;
        const float v = ci < Cin_real ? w[((long)co * Cin_real + ci) * ntaps + tap] : 0.f;
        fwd[(long)b * fstride + i] = rn_f2bf(v);
    }
    if (bwd) {  // backward data: blk = (tap * KSo + ks) * CFi + cb, contraction over co, columns ci; tap flipped
        const int CFi = Cin >> 5, KSo = Cout >> 4;
        const int cb = (int)(blk % CFi), ks = (int)((blk / CFi) % KSo), tap = (int)(blk / ((long)CFi * KSo));
        const int ci = 32 * cb + (lane & 31), co = 16 * ks + 8 * (lane >> 5) + j;
        bwd[(long)b * fstride + i] = rn_f2bf(w[((long)co * Cin_real + ci) * ntaps + (ntaps - 1 - tap)]);
    }
}

}  // namespace

int rn_conv_tiles(long npix, int Cout) {
    (void)Cout;
    return (int)((npix + 127) / 128);
}

size_t rn_conv_lds_bytes(const RnGeom& g, int Cout) {
    switch (conv_nf(Cout)) {
        case 5: return ConvCfg<5, 1>::lds_bytes(g.halo);
        case 4: return ConvCfg<4, 1>::lds_bytes(g.halo);
        case 3: return ConvCfg<3, 1>::lds_bytes(g.halo);
        case 2: return ConvCfg<2, 1>::lds_bytes(g.halo);
        default: return ConvCfg<1, 1>::lds_bytes(g.halo);
    }
}

int launch_rn_conv(hipStream_t st, const RnConvArgs& a) {
    if (a.B < 1 || a.nsrc < 1 || a.nsrc > 4 || a.Cout < 32 || (a.Cout & 31) || a.npix < 1) return FUMI_EINVAL;
    for (int s = 0; s < a.nsrc; ++s)
        if (!a.src[s].in || !a.src[s].frag || a.src[s].Cin < 16 || (a.src[s].Cin & 15) || (a.src[s].ntaps != 9 && a.src[s].ntaps != 1))
            return FUMI_EINVAL;
    switch (conv_nf(a.Cout)) {
        case 5: return conv_launch<5, 1>(st, a);
        case 4: return conv_launch<4, 1>(st, a);
        case 3: return conv_launch<3, 1>(st, a);
        case 2: return conv_launch<2, 1>(st, a);
        default: return conv_launch<1, 1>(st, a);
    }
}

// slabs of the pixel axis per episode: enough workgroups to fill the chip a few times over, at least 4 stages each
int rn_wgrad_nsplit(int B, long npix, int Cin, int Cout) {
    const int Ci32 = (Cin + 31) / 32 * 32;
    const long tiles = (long)((Cout + 63) / 64) * ((Ci32 + 63) / 64) * B;
    const long chunks = (npix + WG_PK - 1) / WG_PK;
    long ns = (1024 + tiles - 1) / tiles;
    if (ns > chunks / 4) ns = chunks / 4;
    if (ns < 1) ns = 1;
    if (ns > 512) ns = 512;
    return (int)ns;
}

int launch_rn_wgrad(hipStream_t st, const RnWgradArgs& a) {
    if (a.B < 1 || a.npair < 1 || a.npair > 2 || (a.Cin & 15) || (a.Cout & 31) || a.nsplit < 1 || (a.ntaps != 9 && a.ntaps != 1))
        return FUMI_EINVAL;
    const int Ci32 = (a.Cin + 31) / 32 * 32;
    const int ci_tiles = (Ci32 + 63) / 64, co_tiles = (a.Cout + 63) / 64;
    const int hs = a.ntaps == 9 ? a.g.halo : 0;
    const int lds = WG_PK * 128 + (WG_PK + 2 * hs) * 128 + 1024;       // (+ slack: transposing reads of the last k-step stay in bounds)
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    const dim3 grid(co_tiles * ci_tiles, a.nsplit, a.B);
    if (a.ntaps == 9) {
        FUMI_SET_DYN_LDS(rn_wgrad_kernel<9>, lds);
        hipLaunchKernelGGL(rn_wgrad_kernel<9>, grid, dim3(256), lds, st, a, ci_tiles, Ci32);
    } else {
        FUMI_SET_DYN_LDS(rn_wgrad_kernel<1>, lds);
        hipLaunchKernelGGL(rn_wgrad_kernel<1>, grid, dim3(256), lds, st, a, ci_tiles, Ci32);
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_wgrad_reduce(hipStream_t st, int B, int nsplit, int ntaps, int Cout, int Cin, int Cin_real, const float* part,
                           float* G, long gstride) {
    const int Ci32 = (Cin + 31) / 32 * 32;
    const long n = (long)Cout * Cin_real * ntaps;
    hipLaunchKernelGGL(rn_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, nsplit, ntaps, Cout, Ci32, Cin_real,
                       part, G, gstride);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_wprep(hipStream_t st, int B, int Cout, int Cin, int Cin_real, int ntaps, const float* W, long wstride,
                    rbf16* fwd, rbf16* bwd, long fstride) {
    if ((Cout & 31) || (Cin & 15) || (bwd && (Cin & 31))) return FUMI_EINVAL;
    const long nel = (long)ntaps * Cin * Cout;
    hipLaunchKernelGGL(rn_wprep_kernel, dim3((unsigned)((nel + 255) / 256), B), dim3(256), 0, st, Cout, Cin, Cin_real, ntaps, W, wstride,
                       fwd, bwd, fstride);
    LAUNCH_CHECK();
    return FUMI_OK;
}
