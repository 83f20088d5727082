// Conv4 image encoder at the im_net seam (fumi/models/fumi.py:89-100 is the seam; the reference has no convolutional
// encoder: BASELINE.json's configs are worded with one).  Shared declarations of the three
// translation units: conv_gemm.hip (convolutions on the fp32 MFMA), conv_ew.hip (batch-norm / ReLU / max-pool passes,
// head, updates), conv4.hip (the meta-step and the C ABI).
//
// Data layout in HBM ("padded channels-last"): an activation of a block at resolution H x W is
//     [image][(H+2) * (W+2) padded pixels][64 channels] fp32, border pixels = 0,
// images of an episode consecutive, episodes consecutive.  A 3x3 / pad 1 convolution is then a sum of 9 SHIFTED copies of
// the flattened pixel axis: out[p] = sum_tap in[p + (ky-1)*(W+2) + (kx-1)] W_tap, for every interior p, with no bounds
// logic in the inner loop -- a tile of consecutive pixels needs one contiguous slab of input (tile + 2*(W+3) halo pixels),
// staged with fully coalesced 16-byte loads.  Border outputs are computed and discarded (written as 0): 9 % of the
// matrix work at 42 x 42.
//
// Weights ("fragment order"): the MFMA B operand of v_mfma_f32_32x32x2_f32 is one float per lane (k = lane>>5, column =
// lane&31).  Four consecutive MFMAs of a wave contract the channel pairs (8j+i, 8j+4+i), i = 0..3, so a lane needs
// W[co][8j + 4*(lane>>5) + 0..3] = 16 bytes: stored lane after lane, every (tap, j, column tile) is ONE 1 KiB block that a
// wave fetches with a single fully coalesced global_load_dwordx4 straight into the operand registers -- weights never pass
// through LDS (they are shared by every workgroup of an episode and stay in L1 / L2).
#pragma once
#include "common.h"

constexpr int CV_C = 64;                 // channels of every block
constexpr int CV_MAXBLK = 4;
constexpr int CV_TILE = 128;             // output pixels per workgroup of the conv kernels (4 waves x 32)
constexpr int CV_WFRAG = 9 * 8 * 2 * 256;   // floats of one [64 x 64 x 3 x 3] weight in fragment order (= 36864)
constexpr float CV_EPS = 1e-5f;

struct CvGeom {                          // one block's convolution resolution
    int H, W, Hp, Wp, Pp, halo;          // Hp = H+2, Wp = W+2, Pp = Hp*Wp, halo = Wp+1
};
static inline CvGeom cv_geom(int H, int W) { CvGeom g; g.H = H; g.W = W; g.Hp = H + 2; g.Wp = W + 2; g.Pp = g.Hp * g.Wp; g.halo = g.Wp + 1; return g; }
#ifdef __HIPCC__
// p / d and p % d for 0 <= p < 2^22, 1 <= d: one multiply by the reciprocal and a one-step correction (an integer division is
// ~40 instructions, a 64-bit one several times that; these sit in per-element staging loops)
__device__ __forceinline__ void cv_divmod(int p, int d, float rd, int& q, int& r) {
    q = (int)((float)p * rd);
    r = p - q * d;
    if (r < 0) { --q; r += d; } else if (r >= d) { ++q; r -= d; }
}
#endif
static inline int cv_tiles(long npix) { return (int)((npix + CV_TILE - 1) / CV_TILE); }

// ---- conv_gemm.hip -----------------------------------------------------------------------------------------------
// out[b][pix][64] = sum_s conv3x3(in_s[b], W_s[b]) over nsrc (1 or 2) sources, padded channels-last, per-episode weights in
// fragment order (frag stride 0 = one weight shared by all episodes).  npix = pixels per episode (images * Pp).
// stats != NULL: per-tile partial sums [B][tiles][2][64] of (out, out * dot) over interior pixels (dot == NULL: out * out).
struct Conv64Args {
    int B, nsrc; long npix; CvGeom g;
    const float* in[2]; const float* frag[2]; long frag_stride[2];
    float* out; float* stats; const float* dot;
};
int launch_conv64(hipStream_t st, const Conv64Args& a);
// first block: images [B][M][Cin][H][W] (dense NCHW, Cin <= 4) -> out padded channels-last; weights frag1 (cv_frag1_floats)
struct Conv1Args {
    int B, M, Cin; CvGeom g;
    const float* img; const float* frag; long frag_stride;
    float* out; float* stats; const float* dot;      // out == NULL: statistics only (the fused block-1 path stores no map)
    const float* frag_dot; long frag_dot_stride;     // != NULL: the dot operand is conv(img, frag_dot), computed in the kernel
};
int launch_conv1(hipStream_t st, const Conv1Args& a);
__host__ __device__ static inline int cv_frag1_floats(int Cin) { return ((Cin * 9 + 1) / 2) * 2 * 64; }
// dW[b][tap][co][ci] (+)= sum_pix dy_s[b][pix][co] * x_s[b][pix + off_tap][ci] over nsrc source pairs; the pixel range of an
// episode is cut into nsplit slabs (partial sums [b][split][9][64][64], summed by wsum)
struct Wgrad64Args {
    int B, nsrc, nsplit; long npix; CvGeom g;
    const float* x[2]; const float* dy[2];
    float* part;           // [B][nsplit][9*64*64]
};
int launch_wgrad64(hipStream_t st, const Wgrad64Args& a);
// first block: dW1[b][co][Cin*9] partial slabs [B][nsplit][64][32]
struct Wgrad1Args {
    int B, M, Cin, nsplit; CvGeom g;
    const float* img; const float* dy; float* part;   // [B][nsplit][64*32]
};
int launch_wgrad1(hipStream_t st, const Wgrad1Args& a);
int cv_wgrad_nsplit(int B, long npix);

// weight bookkeeping: canonical per-episode weights are [tap][co][ci] ("TOI", 9*64*64) / [co][Cin*9 padded to 32] for block 1
// frag_fwd / frag_bwd (<- the flipped, channel-swapped kernel of the input-gradient convolution) from TOI weights
int launch_wfrag64(hipStream_t st, int n, const float* toi, long toi_stride, float* frag_fwd, float* frag_bwd, long frag_stride);
int launch_wfrag1(hipStream_t st, int n, int Cin, const float* w1 /*[n][64][32]*/, float* frag /*[n][cv_frag1_floats]*/);
// OIHW (torch) <-> TOI / block-1 canonical, n weights
int launch_oihw_to_toi(hipStream_t st, int n, const float* oihw, float* toi);
int launch_toi_to_oihw(hipStream_t st, int n, const float* toi, float* oihw, float scale);

// ---- conv_ew.hip ---------------------------------------------------------------------------------------------------
// per (episode, channel) coefficient table of one block and pass: [B][CF_N fields][64 channels]
enum { CF_MU = 0, CF_R, CF_A, CF_C0,         // forward:  xh = (u - mu) r,  v = A u + C0   (A = g r, C0 = beta - mu A)
       CF_D1, CF_D2, CF_GR,                  // backward: du = GR (dv - D1 - xh D2),  GR = g r
       CF_TA, CF_TB, CF_TC,                  // tangent forward: v' = TA u' + TB xh + TC
       CF_M1, CF_M2,                         //                  xh' = r (u' - M1 - xh M2)
       CF_K0, CF_DD1, CF_E12, CF_pad };      // tangent backward: du' = K0 (dv - D1 - xh D2) + GR (dv' - DD1 - xh' D2 - xh E12)
constexpr int CF_N = 16;

struct EwGeom { int B, M; CvGeom g; int Ho, Wo; CvGeom gn; int last; };   // gn: next block's geometry (padded output grid);
                                                                           // last: output is the feature matrix [B*M][64*Ho*Wo]

// Coefficient tables are [B][CF_N][64] floats (a thread reads the four channels it owns with one 16-byte load per field).
// finalize modes: sums of K partial slabs [B][nt][K][64] -> coefficients (+ the BN weight / bias gradients of the pass)
enum { CFM_FWD = 0, CFM_BWD = 1, CFM_TFWD = 2, CFM_TBWD = 3 };
struct CoefArgs {
    int B, mode, nt, K; float n;                 // n = pixels per (episode, channel) = M * H * W
    const float* part;                           // [B][nt][K][64]
    float* coef;                                 // [B][CF_N][64]
    const float* g; const float* beta; long pstride;      // BN weight / bias of the pass's parameter slot (per episode)
    const float* gd; const float* betad; long dstride;    // tangent direction (CFM_TFWD / CFM_TBWD)
    float* dg; float* dbeta; long gstride;       // CFM_BWD / CFM_TBWD: gradients (or their tangents) of BN weight / bias
};
size_t coef_scratch_doubles(int B, int nt);
int launch_coef(hipStream_t st, const CoefArgs& a, double* scratch /* coef_scratch_doubles(B, nt) */);

// max-pool(ReLU(BN(u))) -> next block's padded input (or the feature matrix); TAN: also the tangent x' from u'
struct PoolFwdArgs { EwGeom e; const float* u; const float* ud; const float* coef; float* x; float* xd; };
int launch_pool_fwd(hipStream_t st, const PoolFwdArgs& a, int tangent);
// partial sums over pooled windows: backward (sum dv, sum dv xh) / tangent backward (sum dv', sum dv' xh, sum dv xh')
struct BwdRedArgs { EwGeom e; const float* u; const float* ud; const float* dxo; const float* dxod; const float* coef; float* part; int nt; };
int ew_bwd_red_nt(const EwGeom& e);
int launch_bwd_reduce(hipStream_t st, const BwdRedArgs& a, int tangent);
// du (or du') over the whole padded grid (border = 0)
struct BwdApplyArgs { EwGeom e; const float* u; const float* ud; const float* dxo; const float* dxod; const float* coef; float* du; };
int launch_bwd_apply(hipStream_t st, const BwdApplyArgs& a, int tangent);

// head: logits / soft-max CE / arg-max / dz of M rows per episode; then d head and d features
struct HeadArgs {
    int B, M, N, F; float scale;                 // dz = (p - y) * scale
    const float* f; const float* head;           // [B*M][F], [B][N][F+1]
    const int64_t* y;                            // [B][M]
    const float* fd; const float* headd;         // tangent inputs (NULL: plain pass)
    float* z;                                    // [B][M][N] logits (plain) -- may be NULL in the tangent pass
    float* p; float* dz;                         // [B][M][N] soft-max (plain: written, tangent: read); dz (plain) / dz' (tangent) written
    int64_t* preds; float* preds_f; float* loss_b; float* acc_b; int* status;    // plain pass outputs (may be NULL)
    float* row_loss; float* row_hit;             // [B][M] scratch, needed when loss_b / acc_b are asked for
};
int launch_head_logits(hipStream_t st, const HeadArgs& a);
// dh[b][n][F+1] = sum_s dz_s^T f_s | colsum dz_0 ;  df[b][m][F] = sum_s dz_s head_s   (s over 1 or 2 source pairs)
struct HeadGradArgs {
    int B, M, N, F, nsrc;
    const float* dz[2]; const float* f[2]; const float* head[2];    // pair s: dh += dz[s]^T f[s];  df += dz[s] head[s][:, :F]
    float* dh; float* df;                         // df may be NULL
};
int launch_head_grad(hipStream_t st, const HeadGradArgs& a);

// out[i] = a[i] + s * b[i]
int launch_axpy(hipStream_t st, long n, const float* a, float s, const float* b, float* out);
// out[b][i] = scale * sum_s part[b][s][i] written at out + b * ostride  (i < n)
int launch_reduce_batched(hipStream_t st, int B, int ns, long n, const float* part, float scale, float* out, long ostride);
// first-block weights: OIHW [64][Cin][3][3] <-> canonical [64][32]
int launch_w1_to_canon(hipStream_t st, int n, int Cin, const float* oihw, long istride, float* canon, long ostride);
int launch_w1_from_canon(hipStream_t st, int Cin, const float* canon, float* oihw, float scale);
// dst[b][i] = src[i] for b < B (meta-parameters -> per-episode slot 0)
int launch_broadcast(hipStream_t st, int B, long n, const float* src, float* dst, long dstride);
int launch_pad_cl(hipStream_t st, long M, const CvGeom& g, const float* src, float* dst);
int launch_unpad_cl(hipStream_t st, long M, const CvGeom& g, const float* src, float* dst);
int launch_ce(hipStream_t st, int M, int N, const float* z, const int64_t* y, float* loss, float* dz, int64_t* preds, int* status);
int launch_proto(hipStream_t st, int B, int S, int N, int P, const float* x, const int64_t* y, float* out, int* status);

// ---- conv_first.hip: block 1 without its full-resolution maps (u = conv(image, W1) recomputed band by band) ----------------
struct C1Args {
    int B, M, Cin; CvGeom g; CvGeom gn; int Ho, Wo;
    const float* img;
    const float* frag; long frag_stride;        // W1 in fragment order, per episode
    const float* fragd; long fragd_stride;      // tangent direction W1' (tangent passes)
    const float* coef;
    float* x; float* xd;                        // pool: pooled output (plain) / its tangent, padded at gn
    const float* dxo; const float* dxod;        // reduce / wgrad: gradient w.r.t. the pooled output (and its tangent)
    float* part;                                // reduce: [B][nt][K][64] partial sums, nt = c1_chunks(...)
    float* wpart;                               // wgrad: [B][nt][64][32] partial dW1
};
int c1_chunks(int B, int M, const CvGeom& g, int* chunk_out);
int launch_c1(hipStream_t st, const C1Args& a, int mode /*0 pool, 1 reduce, 2 wgrad, 3 batch statistics*/, int tangent);
