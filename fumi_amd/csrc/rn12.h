// ResNet-12 image encoder in bf16 at the im_net seam (fumi/models/fumi.py:89-100 is the seam; BASELINE.json configs[4]: FuMI 20-way
// 5-shot, ResNet-12 backbone bf16, 5 inner steps, second-order outer gradients).  The reference has no such encoder ("parity
// unpinned"); the algebra is oracle/resnet12_manual.py.  Shared declarations of rn12_conv.hip (matrix products on
// v_mfma_f32_32x32x16_bf16), rn12_ew.hip (batch-norm / LeakyReLU / residual join / pooling passes) and rn12.hip (the meta-step).
//
// Data layout in HBM ("padded channels-last", as conv4.h, in bf16): an activation at resolution H x W with C channels is
//     [episode][image][(H+2) * (W+2) padded pixels][C] bf16, border pixels = 0,
// so a 3x3 / pad 1 convolution is a sum of 9 SHIFTED copies of the flattened pixel axis, out[p] = sum_tap in[p + off_tap] W_tap.
// C is a multiple of 16 (the 3 image channels are padded to 16): one MFMA k-step is 16 channels of one tap.
//
// Weights: fp32 masters per episode in torch's OIHW layout inside one parameter slab; before every pass that multiplies with them
// they are rounded to bf16 into "fragment order": the B operand of v_mfma_f32_32x32x16_bf16 has lane l hold B[k = 8 (l >> 5) + j]
// [col = l & 31], j = 0..7 = 16 bytes, so one (tap, 16-channel k-step, 32-column block) is ONE 1 KiB block in lane order:
//     fwd frag  [tap][Cin/16][Cout/32][64 lanes][8]  = W[co = 32 cb + (l & 31)][ci = 16 ks + 8 (l >> 5) + j][tap]
//     bwd frag  [tap][Cout/16][Cin/32][64 lanes][8]  = W[co = 16 ks + 8 (l >> 5) + j][ci = 32 cb + (l & 31)][ntaps - 1 - tap]
// (the input-gradient product is the same kernel run on the flipped, channel-swapped copy).
#pragma once
#include "common.h"

typedef unsigned short rbf16;                         // storage type of every map
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));

constexpr float RN_EPS = 1e-5f;
constexpr float RN_SLOPE = 0.1f;
constexpr int RN_MAXBLK = 4;
constexpr int RN_NCONV = 4;                           // convolutions per block: c1, c2, c3 (3x3) and the 1x1 shortcut cs
constexpr int RN_BREF = 4;                            // episodes per chunk the launch geometry is priced for (configs[4]: two lanes of 4)

struct RnGeom { int H, W, Hp, Wp, Pp, halo; };
static inline RnGeom rn_geom(int H, int W) { RnGeom g; g.H = H; g.W = W; g.Hp = H + 2; g.Wp = W + 2; g.Pp = g.Hp * g.Wp; g.halo = g.Wp + 1; return g; }

#ifdef __HIPCC__
__device__ __forceinline__ float rn_bf2f(rbf16 v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ rbf16 rn_f2bf(float f) { return __builtin_bit_cast(rbf16, (__bf16)f); }      // v_cvt_pk_bf16_f32: RNE, NaN kept
#endif

// ---- rn12_conv.hip ---------------------------------------------------------------------------------------------------------------
// out[b][p][Cout] = sum over sources s of conv_{ntaps}(in_s[b], frag_s[b])   (1 <= nsrc <= 4: tangent passes add products)
struct RnSrc {
    const rbf16* in; long in_stride;                  // [B][npix][Cin]; elements between episodes
    const rbf16* frag; long frag_stride;              // fragment-order weights of the episode (stride 0: shared)
    int Cin, ntaps;                                   // Cin multiple of 16; 9 or 1
};
struct RnConvArgs {
    int B, nsrc, Cout; long npix; RnGeom g;           // npix = images * Pp pixels per episode
    RnSrc src[4];
    rbf16* out; long out_stride;
    float* stats;                                     // != NULL: [B][tiles][2][Cout] partial sums over interior pixels of (out, out * dot)
    const rbf16* dot; long dot_stride;                // NULL: out * out
    unsigned long long* trace;                        // dev: cycle stamps of one workgroup's wave 0 (fumi_hip_set_trace_buffer(2, ..))
    // set by the launcher: pixel tiles / column groups per episode, XCD-grouped ids, rows of the input slab's LDS image, LDS-direct
    // slab loads, tiles per image (tpi > 0: the tiles restart at every image)
    int tiles, ncg, xcd, slab_rows, glds, tpi;
};
int rn_conv_tiles(long npix, const RnGeom& g);          // upper bound of the statistics slabs per episode (sizing)
size_t rn_conv_lds_bytes(const RnGeom& g, int Cout);
// nt_out (optional): the number of statistics slabs per episode the launch wrote (its pixel tiles: 128 or 256 pixels each)
int launch_rn_conv(hipStream_t st, const RnConvArgs& a, int* nt_out = nullptr);

// dW[b][tap][co][ci] = sum over pairs s of sum_p dy_s[b][p][co] x_s[b][p + off_tap][ci]; pixels cut into nsplit slabs per episode
struct RnWgradArgs {
    int B, npair, Cin, Cout, ntaps, nsplit; long npix; RnGeom g;
    const rbf16* x[2]; const rbf16* dy[2]; long x_stride, dy_stride;
    float* part;                                      // [B][nsplit][ntaps][Cout][Ci32], Ci32 = Cin rounded up to 32
};
int rn_wgrad_nsplit(int B, long npix, int Cin, int Cout);
int launch_rn_wgrad(hipStream_t st, const RnWgradArgs& a);
// G[b][(co * Cin_real + ci) * ntaps + tap] = sum_split part   (torch OIHW), b-stride gstride
int launch_rn_wgrad_reduce(hipStream_t st, int B, int nsplit, int ntaps, int Cout, int Cin, int Cin_real, const float* part,
                           float* G, long gstride);
// fp32 OIHW master -> bf16 fragment copies (fwd always, bwd when bwd != NULL)
int launch_rn_wprep(hipStream_t st, int B, int Cout, int Cin, int Cin_real, int ntaps, const float* W, long wstride,
                    rbf16* fwd, rbf16* bwd, long fstride);

// ---- rn12_ew.hip -------------------------------------------------------------------------------------------------------------------
// per (episode, channel) coefficient table of one BN and pass: [B][RCF_N][C]
enum { RCF_MU = 0, RCF_R, RCF_A, RCF_C0,        // forward:  xh = (u - mu) r,  v = A u + C0   (A = g r, C0 = beta - mu A)
       RCF_D1, RCF_D2,                           // backward: du = A (dv - D1 - xh D2)
       RCF_TB, RCF_TC,                           // tangent forward: v' = A u' + TB xh + TC
       RCF_M1, RCF_M2,                           //                  xh' = r (u' - M1 - xh M2)
       RCF_K0, RCF_DD1, RCF_E12,                 // tangent backward: du' = K0 (dv - D1 - xh D2) + A (dv' - DD1 - xh' D2 - xh E12)
       RCF_N };
enum { RCM_FWD = 0, RCM_BWD = 1, RCM_TFWD = 2, RCM_TBWD = 3 };
struct RnCoefArgs {
    int B, C, mode, nt, K; int k0, k1, k2;       // partial sums [B][nt][K][C]; the slices this BN reads (k2 unused unless TBWD)
    float n;                                     // interior pixels per (episode, channel) = images * H * W
    const float* part; float* coef;
    const float* g; const float* beta; long pstride;        // BN weight / bias (per episode)
    const float* gd; const float* betad; long dstride;      // tangent direction
    float* dg; float* dbeta; long gstride;       // BWD / TBWD: gradients (or their tangents) of BN weight / bias
};
// scratch (optional, rn_coef_scratch_floats): long lists of partials are first added in groups of 64 by many workgroups
size_t rn_coef_scratch_floats(int B, int nt, int K, int C);
int launch_rn_coef(hipStream_t st, const RnCoefArgs& a, float* scratch);

struct RnMap { int B, M, C; RnGeom g; };         // M images per episode
// a = lrelu(A u + C0)            | tangent: a' = lrelu'(v) (A u' + TB xh + TC)
int launch_rn_act(hipStream_t st, const RnMap& m, const rbf16* u, const rbf16* ud, const float* coef, rbf16* out);
// partial sums of (dv, dv xh) | tangent (dv', dv' xh, dv xh'), dv = da lrelu'(v): part [B][nt][K][C]
int rn_red_nt(const RnMap& m);
int launch_rn_bwd_reduce(hipStream_t st, const RnMap& m, const rbf16* u, const rbf16* ud, const rbf16* da, const rbf16* dad,
                         const float* coef, float* part, int tangent);
int launch_rn_bwd_apply(hipStream_t st, const RnMap& m, const rbf16* u, const rbf16* ud, const rbf16* da, const rbf16* dad,
                        const float* coef, rbf16* du, int tangent);
// residual join + LeakyReLU + max-pool 2:  o = maxpool(lrelu(BN3(u3) + BNs(us)))   (tangent: o' at the arg-max)
struct RnJoin { RnMap m; RnGeom gn; int Ho, Wo; const rbf16* u3; const rbf16* us; const float* coef3; const float* coefs;
                const rbf16* u3d; const rbf16* usd; };
int launch_rn_join_fwd(hipStream_t st, const RnJoin& j, rbf16* o, int tangent);
// sums (ds, ds xh3, ds xhs) | tangent (ds', ds' xh3, ds xh3', ds' xhs, ds xhs'): part [B][nt][K = 3 | 5][C]
int launch_rn_join_reduce(hipStream_t st, const RnJoin& j, const rbf16* dout, const rbf16* doutd, float* part, int tangent);
int launch_rn_join_apply(hipStream_t st, const RnJoin& j, const rbf16* dout, const rbf16* doutd, rbf16* du3, rbf16* dus, int tangent);
// global average pool of the last block's output map and its adjoint
int launch_rn_avgpool(hipStream_t st, int BM, int C, const RnGeom& g, const rbf16* o, float* f);
int launch_rn_avgpool_bwd(hipStream_t st, int BM, int C, const RnGeom& g, const float* df, rbf16* dout);
// images fp32 [BM][Cin][H][W] -> bf16 padded channels-last with 16 channels
int launch_rn_img_prep(hipStream_t st, long BM, int Cin, const RnGeom& g, const float* img, rbf16* out);
