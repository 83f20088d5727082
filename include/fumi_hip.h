/*
 * fumi_hip.h -- C ABI of the MI355X (gfx950) episodic meta-training engine.
 *
 * The reference (s-a-malik/fumi) has no FFI: its seam is a Python API.  This ABI is what replaces the
 * arithmetic the reference obtains from torch autograd + torchmeta *inside* these Python entry points
 * (citations into /root/reference):
 *
 *   fumi_hip_fumi_step   <- FUMI.evaluate                 fumi/models/fumi.py:115-196
 *                           (+ get_hyper_params :198-212, im_forward :214-218, hyper_net :70-86,104-113)
 *   fumi_hip_maml_step   <- maml.evaluate                 fumi/models/maml.py:134-193 (PureImageNetwork :15-33)
 *   fumi_hip_am3_step    <- AM3.evaluate / forward        fumi/models/am3.py:90-126,128-212
 *                           (+ get_prototypes / prototypical_loss / get_preds  fumi/utils/utils.py:302-402)
 *   fumi_hip_glove_bag   <- WordEmbedding.forward         fumi/models/common.py:23-41
 *   fumi_hip_class_text_select <- the per-class "first support row" loop  fumi/models/fumi.py:207-210
 *   fumi_hip_linear_*    <- torchmeta MetaLinear.forward / autograd AddmmBackward (requirements.txt:10)
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer borrowed for the call (owned by the caller, e.g. a torch tensor);
 *     row-major, contiguous; fp32 values, int64 labels/tokens (the dtypes the reference's tensors have).
 *   - `theta`, `phi`, `g_theta`, `g_phi`, `hid` are HOST arrays (of device pointers / ints).
 *   - `stream` is a hipStream_t (NULL = the null stream).  All work is enqueued asynchronously on it; no call
 *     synchronises the device except workspace growth and fumi_hip_read_status.
 *   - return value: 0 on success, a negative FUMI_E* code otherwise; fumi_hip_strerror() names it.
 *   - one workspace per (process, device); a workspace is not re-entrant.
 *   - gradients are written (not accumulated) as  grad_scale * SUM over the call's episodes of d(loss_b)/d(param);
 *     pass grad_scale = 1/B for the reference's mean loss on one GPU, 1/B_global before an all-reduce(sum).
 */
#ifndef FUMI_HIP_H
#define FUMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fumi_ws fumi_ws_t;
typedef void* fumi_stream_t;          /* hipStream_t */

#define FUMI_OK            0
#define FUMI_EINVAL      (-1)         /* bad argument (shape, NULL pointer, unsupported size) */
#define FUMI_ENOMEM      (-2)         /* workspace allocation failed */
#define FUMI_EHIP        (-3)         /* a HIP runtime call or kernel launch failed (see fumi_hip_last_hip_error) */
#define FUMI_ENOTSUP     (-4)         /* valid request this build does not implement */

/* device-side status bits (fumi_hip_read_status) -- the reference raises IndexError in both situations */
#define FUMI_ST_LABEL_RANGE   1       /* a label outside [0, N) */
#define FUMI_ST_CLASS_MISSING 2       /* a class without a support sample (fumi.py:209 would raise) */
#define FUMI_ST_SYNC_TIMEOUT  4       /* a workgroup gave up waiting for the sibling workgroups of its episode (split inner loop /
                                       * split reverse sweep): the step's results are not to be trusted; the host raises RuntimeError */

#define FUMI_MAX_HIDDEN 8

int          fumi_hip_version(void);
const char*  fumi_hip_strerror(int code);
const char*  fumi_hip_last_hip_error(void);

/* A workspace owns all device memory the library allocates for its device: the scratch slab (grown on demand, bump-allocated per
 * call), a few small counters / the status word, and -- since round 2 -- the bf16 planes of the forward pass's column operand
 * (W0 and the support rows split once per step, ~16 MB at the reference sizes; rebuilt by every step, nothing to keep coherent). */
int   fumi_hip_workspace_create(int device, size_t bytes_hint, fumi_ws_t** out);
void  fumi_hip_workspace_destroy(fumi_ws_t* ws);
size_t fumi_hip_workspace_bytes(const fumi_ws_t* ws);
/* Copies the device status word to the host (synchronises `stream`) and clears it. */
int   fumi_hip_read_status(fumi_ws_t* ws, fumi_stream_t stream, int* status_out);
/* Polls a workgroup spends waiting for the sibling workgroups of its episode before it sets FUMI_ST_SYNC_TIMEOUT (process-wide;
 * default 1 << 22, about a second).  Tests set 0 to see the bit; returns the previous value. */
int   fumi_hip_set_spin_limit(int polls);
/* Development hooks: a device buffer of uint64 wall-clock stamps written by block 0 of the per-episode kernels (which = 0) or by
 * every workgroup of the forward X-panel kernel (which = 1), or cycle stamps of wave 0 of the middle workgroup of a ResNet-12
 * convolution launch (which = 2); NULL switches the stamps off (the default). */
int   fumi_hip_set_trace_buffer(int which, void* device_u64);

/* ---- in-library phase timing (HIP events recorded on the caller's stream around each phase) ---------------------
 * Used by bench.py to time the dominant kernel live inside the timed region.  Off by default (no events recorded). */
#define FUMI_PH_SELECT     0   /* class text select                                   */
#define FUMI_PH_HYPER_FWD  1   /* hypernetwork forward (2 GEMMs)                       */
#define FUMI_PH_GEMM_A0S   2   /* AM3: image encoder on the support rows               */
#define FUMI_PH_GEMM_A0Q   3   /* AM3: image encoder on the query rows                 */
#define FUMI_PH_XPANEL_FWD 4   /* [A0|G] = [Xs;Xq][W0;Xs]^T  <- dominant kernel of the FuMI/MAML step */
#define FUMI_PH_ADAPT      5   /* inner loop                                           */
#define FUMI_PH_QUERY      6   /* query forward/backward                               */
#define FUMI_PH_REVERSE    7   /* second-order reverse sweep                           */
#define FUMI_PH_REDUCE     8   /* sums over episodes                                   */
#define FUMI_PH_XPANEL_BWD 9   /* gW0 = Abar0^T X (split-K slabs + slab reduce)        */
#define FUMI_PH_HYPER_BWD 10   /* hypernetwork backward                                */
#define FUMI_PH_AM3       11   /* AM3 fused head kernels                               */
#define FUMI_PH_CONV_GEMM 12   /* Conv4: the 64 -> 64 channel products (forward, input-gradient, weight-gradient) on the MFMA */
#define FUMI_PH_CONV_FIRST 13  /* Conv4: block 1 (Cin <= 3 -> 64: K = 27, bound by writing / reading its 84x84x64 maps)   */
#define FUMI_PH_CONV_EW   14   /* Conv4: batch-norm / ReLU / max-pool passes, head, updates                               */
#define FUMI_PH_RN_CONV   15   /* ResNet-12 (bf16): forward / input-gradient convolutions on v_mfma_f32_32x32x16_bf16       */
#define FUMI_PH_RN_WGRAD  16   /* ResNet-12: weight-gradient products                                                     */
#define FUMI_PH_RN_EW     17   /* ResNet-12: batch-norm / LeakyReLU / residual join / pooling passes, head, updates       */
#define FUMI_PH_COUNT     18
/* phase_mask: bit p set = record a HIP event pair around phase p (FUMI_PH_*) on the caller's stream; -1 = every phase,
 * 0 = off.  An event pair costs a few microseconds of stream time, so time only what is needed.  Also clears the records. */
int          fumi_hip_set_profiling(fumi_ws_t* ws, int phase_mask);
/* Time only every `every`-th occurrence of a selected phase (default 1): an event record is a ~6 us bubble on the stream. */
int          fumi_hip_set_profiling_every(fumi_ws_t* ws, int every);
/* Synchronises the device; total elapsed ms and number of records of `phase` since profiling was switched on. */
int          fumi_hip_get_profile(fumi_ws_t* ws, int phase, double* total_ms, int* count);
const char*  fumi_hip_phase_name(int phase);

/* ---- FuMI meta-step (replaces fumi/models/fumi.py:146-192 for B episodes) ------------------------------------
 * theta   : 2*n_hidden pointers  W_i [hid[i], hid[i-1]] (hid[-1] = D), b_i [hid[i]]   (im_net.linear{i}.*)
 * phi     : 4 pointers           A0 [Ht,Dt], a0 [Ht], A1 [H+1,Ht], a1 [H+1]           (hyper_net.0.*, hyper_net.2.*)
 * cls_text: [B,N,Dt] per-class text encodings, or NULL to select them from text_s [B,S,Dt] (first support row of
 *           each class, fumi.py:207-210)
 * outputs : logits_q [B,Qn,N], preds_q [B,Qn] (first arg-max), preds_q_f32 [B,Qn] or NULL (the same indices as floats:
 *           what the reference's float `test_preds` tensor holds, fumi.py:180-183), loss_b [B] (query CE), acc_b [B];
 *           stats [2] (optional, may be NULL) = grad_scale * (sum_b loss_b, sum_b acc_b): with grad_scale = 1/B the
 *           meta-batch mean loss / accuracy that fumi.py:187-188 computes, ready for the same all-reduce as the gradients;
 *           g_theta/g_phi (same shapes as theta/phi) only when need_grad != 0.
 * second-order outer gradient always (fumi.py:176 hard-codes first_order=False).
 * dropout_p > 0 applies the reference's train-mode Dropout after every ReLU of im_net (fumi.py:93-99) with a fresh mask per
 * forward call (each inner step and the query pass), drawn from a counter-based hash of (seed, episode, call, layer,
 * element): statistically the reference's nn.Dropout, not its RNG stream.  Pass 0 for task != "train". */
int fumi_hip_fumi_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi);

/* ---- MAML meta-step (replaces fumi/models/maml.py:156-191) ---------------------------------------------------
 * params: 2*n_hidden + 2 pointers: hidden layers as above, then lin_final W [N,H], b [N].  n_hidden = 0 is the reference's
 * hidden_dims=None: lin_final W [N,D] alone (hid may be NULL). */
int fumi_hip_maml_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid,
        int T, float alpha, int first_order, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* const* params,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_params);

/* ---- AM3 step (replaces am3.py:160-200 + utils.py:302-402, dropout 0) -----------------------------------------
 * w: 10 pointers  Wi [P,D], bi [P], G0 [Ht,Dt], g0 [Ht], G1 [P,Ht], g1 [P], H0 [Ht,P], h0 [Ht], H1 [1,Ht], h1 [1]
 * lamda_fixed: -1 = learned, 0 or 1 = fixed (am3.py:174-179).
 * dropout_p > 0: train-mode Dropout after the ReLU of g and of h (am3.py:82,88), counter-based masks from `seed`.
 * outputs: loss [1] (mean over B*Qn), preds_q [B,Qn] (first arg-min of the distances), lamda_s [B,S],
 *          correct [1] (number of correct query predictions, float); g_w (10 pointers) when need_grad;
 *          stats [3 + N*N] or NULL: [loss, correct count, grad_scale * sum_b mean_s lamda, confusion counts (rows = target
 *          class, columns = predicted class)] -- with grad_scale = 1 / global meta-batch a sum over ranks gives the global
 *          mean loss / lamda and the global counts, and fumi_hip_am3_metrics turns those into the reference's metrics. */
int fumi_hip_am3_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int Dt, int Ht, int P, int lamda_fixed, int need_grad, float grad_scale,
        float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q, const float* text_s,
        const float* const* w,
        float* loss, int64_t* preds_q, float* lamda_s, float* correct,
        float* const* g_w, float* stats);
/* The same step with the adjoints of the image rows as extra outputs (need_grad): dx_s [B,S,D], dx_q [B,Qn,D] = imbar Wi.  An
 * image encoder in front of the step (the Conv4 backbone at the image_encoder seam, am3.py:41-46) continues from them. */
int fumi_hip_am3_step_dx(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int Dt, int Ht, int P, int lamda_fixed, int need_grad, float grad_scale,
        float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q, const float* text_s,
        const float* const* w,
        float* loss, int64_t* preds_q, float* lamda_s, float* correct,
        float* const* g_w, float* stats, float* dx_s, float* dx_q);
/* out6 = [loss, accuracy, macro F1, macro precision, macro recall, mean lamda] from `stats` (fumi_hip_am3_step): what
 * AM3.evaluate returns per meta-batch (am3.py:203-212 via sklearn on the host, utils.py:319-326), without leaving the device.
 * N <= 64. */
int fumi_hip_am3_metrics(fumi_ws_t* ws, fumi_stream_t stream, int N, const float* stats, float* out6);

/* ---- finer-grained ops (unit parity tests; building blocks of the steps) --------------------------------------- */
/* out[r,:] = mean (mode 0: sum / #non-PAD tokens) or max (mode 1: over ALL L positions) of table[tok[r,l],:]. */
int fumi_hip_glove_bag(fumi_ws_t* ws, fumi_stream_t stream, const int64_t* tok, int R, int L, int64_t pad_id,
        const float* table, int V, int E, int mode, float* out);
/* Fused form for FuMI: out[b,n,:] = bag(tok_s[b, first s with y_s[b,s]==n, :])  -> [B,N,E]  (common.py:23-41 applied to
 * the N class rows that fumi.py:207-210 selects). */
int fumi_hip_glove_bag_select(fumi_ws_t* ws, fumi_stream_t stream, const int64_t* tok_s, const int64_t* y_s,
        int B, int N, int S, int L, int64_t pad_id, const float* table, int V, int E, int mode, float* out);
/* Deferred form: nothing is launched -- the request rides as extra workgroups of the first launch of the next fumi_hip_fumi_step /
 * _indexed of this workspace (or is launched at its start when that step's shapes take another path); `out` [B,N,E] must be the
 * cls_text handed to that step.  fumi_hip_glove_flush launches a pending request on its own. */
int fumi_hip_glove_bag_select_deferred(fumi_ws_t* ws, const int64_t* tok_s, const int64_t* y_s,
        int B, int N, int S, int L, int64_t pad_id, const float* table, int V, int E, int mode, float* out);
int fumi_hip_glove_flush(fumi_ws_t* ws, fumi_stream_t stream);
/* out[b,n,:] = text_s[b, first s with y_s[b,s]==n, :] */
int fumi_hip_class_text_select(fumi_ws_t* ws, fumi_stream_t stream, int B, int N, int S, int Dt,
        const float* text_s, const int64_t* y_s, float* out);
/* The two shared passes over the wide inputs (csrc/xpanel.hip), exported for unit parity tests and tuning:
 *   fwd: A0[B,S+Qn,h0] = [Xs;Xq] W0^T, G[B,S+Qn,S] = [Xs;Xq] Xs^T per episode (support rows first)
 *   bwd: gW0[h0,D] = scale * sum_b Abar[b]^T [Xs_b;Xq_b]   (Abar [B,S+Qn,h0]) */
int fumi_hip_xpanel_fwd(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int D, int h0,
        const float* x_s, const float* x_q, const float* W0, float* A0, float* G);
int fumi_hip_xpanel_bwd(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int D, int h0,
        const float* x_s, const float* x_q, const float* Abar, float scale, float* gW0);
/* One fused launch of torch.optim.Adam's update (coupled L2 weight decay, bias correction; fumi/utils/utils.py:280-283) for
 * up to 32 tensors.  Pointer/size arrays are HOST arrays; `step` is the 1-based step count. */
int fumi_hip_adam_step(fumi_ws_t* ws, fumi_stream_t stream, int n_tensors, float* const* params,
        const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long* numel,
        float lr, float beta1, float beta2, float eps, float weight_decay, int step);
/* Deferred form: nothing is launched -- the update (same arithmetic, same order: bit-identical parameters) is folded into the
 * LAST launch of the next fumi_hip_fumi_step / _indexed of this workspace, whose final reduction produces every gradient element:
 * `optimizer.step()` (fumi/models/fumi.py:193) costs no launch of its own.  Single GPU only (with several ranks the all-reduce
 * lies between gradient and update).  fumi_hip_adam_flush launches a still pending step as the ordinary kernel (*launched = 1)
 * or reports that a meta-step had folded it (*launched = 0). */
int fumi_hip_adam_step_deferred(fumi_ws_t* ws, int n_tensors, float* const* params,
        const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long* numel,
        float lr, float beta1, float beta2, float eps, float weight_decay, int step);
int fumi_hip_adam_flush(fumi_ws_t* ws, fumi_stream_t stream, int* launched);
/* y[M,N] = act(x[M,K] W[N,K]^T + b[N]);  act: 0 none, 1 relu, 2 tanh.  b may be NULL. */
int fumi_hip_linear_fwd(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, int K,
        const float* x, const float* W, const float* b, int act, float* y);
/* dx[M,K] = dy[M,N] W[N,K] */
int fumi_hip_linear_bwd_data(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, int K,
        const float* dy, const float* W, float* dx);
/* dW[N,K] = dy[M,N]^T x[M,K];  db[N] = colsum(dy) (db may be NULL) */
int fumi_hip_linear_bwd_weight(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, int K,
        const float* dy, const float* x, float* dW, float* db);


/* ---- Conv4 image encoder at the im_net seam ---------------------------------------------------------------------------
 * The reference adapts an MLP over pre-computed embeddings; `im_net` is "any module with forward(x, params) and
 * meta_named_parameters()" (fumi/models/fumi.py:89-100, `--im_encoder resnet` is a TODO at fumi/models/am3.py:41-46).
 * BASELINE.json words its configurations with a Conv4 encoder on 84x84 images: nblk blocks of conv3x3(64, pad 1, no bias) .
 * BatchNorm2d with BATCH statistics (training and evaluation alike) . ReLU . MaxPool2d(2), flattened like PyTorch's
 * x.view(M, -1) of NCHW.  PARITY UNPINNED for the convolutional part (nothing to import); oracle = oracle/conv4_ref.py.
 * Everything after the feature vector is the reference's algorithm (fumi.py:146-192 / maml.py:156-191).
 *   x_s [B,S,Cin,H,W], x_q [B,Qn,Cin,H,W] fp32 NCHW (Cin <= 3);  theta: 3*nblk pointers  W_l [64,Cin|64,3,3], BN weight [64],
 *   BN bias [64];  F = fumi_hip_conv4_feature_dim(nblk, H, W) = 64 * (H >> nblk) * (W >> nblk).
 *   fumi: phi = A0 [Ht,Dt], a0, A1 [F+1,Ht], a1 [F+1];  maml: params = theta then lin_final W [N,F], b [N].
 * Outputs / gradient conventions as fumi_hip_fumi_step / fumi_hip_maml_step.  Second order needs T <= 8 taped inner steps. */
int fumi_hip_conv4_feature_dim(int nblk, int H, int W);
/* Process-wide switches of the Conv4 path.  key 0: 1 (default) = block 1 recomputed band by band from the image, its 64-channel
 * full-resolution maps never stored (csrc/conv_first.hip); 0 = every block through the plain passes (what the probe tests read).
 * key 1: lanes -- parts of the meta-batch on concurrent streams of the workspace -- of the Conv4 meta-steps and encoder calls:
 * 0 (default) = the library's choice (up to 3 from 8 episodes; 2 for the encoder calls), 1 = one stream, n <= 4 = at most n. */
int fumi_hip_conv4_set_option(int key, int value);
int fumi_hip_fumi_conv4_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi);
int fumi_hip_maml_conv4_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk,
        int T, float alpha, int first_order, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* const* params,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_params);
/* Forward only: feats [G*M, F] = Conv4(x [G,M,Cin,H,W]); batch statistics per group of M images (one support or query set). */
int fumi_hip_conv4_features(fumi_ws_t* ws, fumi_stream_t stream, int G, int M, int Cin, int H, int W, int nblk,
        const float* x, const float* const* theta, float* feats);
/* Conv4 as the image encoder in front of a step that lays out its own workspace (AM3 at the seam am3.py:41-46; "parity
 * unpinned" like the rest of the Conv4 rows).  encode: feats_s [B,S,F], feats_q [B,Qn,F]; every episode's support set and query
 * set is one batch-statistics group.  keep_tape = 1 leaves the activations laid out in `ws`; encode_bwd (same shapes, same `ws`,
 * no other call on `ws` in between -- give the encoder its own workspace) walks them backwards from the feature adjoints:
 * g_theta (3 nblk pointers, torch layouts) = scale * sum over episodes of d<dfeats, feats>/dtheta.  FUMI_EINVAL without a tape. */
int fumi_hip_conv4_encode(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int Cin, int H, int W, int nblk,
        const float* x_s, const float* x_q, const float* const* theta, float* feats_s, float* feats_q, int keep_tape);
int fumi_hip_conv4_encode_bwd(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int Cin, int H, int W, int nblk,
        const float* x_s, const float* x_q, const float* dfeats_s, const float* dfeats_q, float scale, float* const* g_theta);
/* Test hook: copies one intermediate tensor of the LAST conv4 step of this process out of the workspace (layouts:
 * fumi_amd/csrc/conv4.hip, fumi_hip_conv4_probe).  *n_out = its size in floats; at most max_floats are copied. */
int fumi_hip_conv4_probe(fumi_ws_t* ws, fumi_stream_t stream, int pass, int kind, int block, float* out, size_t max_floats,
        size_t* n_out);
/* ---- ResNet-12 image encoder in bf16 at the im_net seam (fumi/models/fumi.py:89-100; BASELINE.json configs[4]) ----------------------
 * Four residual blocks  a1 = lrelu(BN(conv3x3(x))), a2 = lrelu(BN(conv3x3(a1))), out = maxpool2(lrelu(BN(conv3x3(a2)) + BN(conv1x1(x))))
 * with `channels[l]` output channels (multiples of 32), LeakyReLU slope 0.1, batch statistics, no conv bias; features = global
 * average pool of the last block ([rows, channels[nblk-1]]).  Images x_s [B,S,Cin,H,W], x_q [B,Qn,Cin,H,W] fp32; theta = 12 tensors
 * per block: W1 [C,Cin|C_prev,3,3], g1, b1, W2 [C,C,3,3], g2, b2, W3, g3, b3, Ws [C,Cin|C_prev,1,1], gs, bs (fp32 masters; every
 * matrix product runs in bf16 with fp32 accumulation, maps are stored in bf16).  Second-order outer gradient (fumi.py:176);
 * the reference has no such encoder: "parity unpinned", oracle = oracle/resnet12_manual.py / resnet12_ref.py.
 * `chunk`: episodes processed per pass over the tape (0 = derived from the workspace budget, fumi_hip_resnet12_set_budget /
 * FUMI_RN12_BUDGET_GB, default 200 GB): the meta-gradient is the sum over chunks. */
int fumi_hip_resnet12_set_budget(double gigabytes);
/* Test hooks of the ResNet-12 steps (no reference counterpart: the reference keeps every intermediate in autograd's graph).
 * set_option key 0: probe mode on / off -- one lane; a step whose meta-batch is a single chunk keeps its buffer table, the gradient
 * of every inner step and the direction of the last Hessian-vector product;  key 1: the reverse sweep stops after inner step `value`
 * (0 = whole sweep), so that the tangent maps of that step can be read.
 * fumi_hip_rn12_probe copies one stored intermediate of the last such step (tests/test_resnet12_probe.py compares every stage of the
 * sweep with oracle/resnet12_manual.py on the engine's own upstream maps).  pass 0..T-1: tape of support step `pass`; T: query pass;
 * T+1: tangent maps of the last Hessian-vector product; -1: parameter space.  kind, pass >= 0: 0 u[block][idx], 1 a[block][idx],
 * 2 out[block], 3 du[block][idx], 4 da[block][idx], 5 dout[block] (bf16 padded channels-last [B][M (H+2)(W+2)][C]), 6 coef[block][idx]
 * fp32 [B][13][C] (mu, r, A, C0, D1, D2, TB, TC, M1, M2, K0, DD1, E12), 7 f, 8 df [B][M][F], 9 z, 10 p, 11 dz [B][M][N] fp32.
 * kind, pass -1: 0 parameter slot idx [B][PSZ], 1 head slot idx, 2 G / 3 dh of inner step idx, 4 bar, 5 bar_h, 6 HV, 7 HV_h, 8 V, 9 V_h,
 * 10 / 11 prepared support / query images (bf16, 16 channels).  *n_bytes: size; *is_bf16: element type; copies min(size, max_bytes). */
int fumi_hip_resnet12_set_option(int key, int value);
int fumi_hip_rn12_probe(fumi_ws_t* ws, fumi_stream_t stream, int pass, int kind, int block, int idx, void* out, size_t max_bytes,
        size_t* n_bytes, int* is_bf16);
int fumi_hip_fumi_resnet12_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, const int* channels, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, int chunk,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi);
int fumi_hip_maml_resnet12_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, const int* channels,
        int T, float alpha, int first_order, int need_grad, float grad_scale, int chunk,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* const* params,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_params);
/* Forward only: feats [G*M, channels[nblk-1]] fp32; batch statistics per group of M images. */
int fumi_hip_resnet12_features(fumi_ws_t* ws, fumi_stream_t stream, int G, int M, int Cin, int H, int W, int nblk, const int* channels,
        const float* x, const float* const* theta, float* feats);
/* The matrix kernels on raw maps (unit parity tests): x, y, dy are bf16 "padded channels-last" [B][M (H+2)(W+2)][C] with zero
 * borders, Wt / dW fp32 [B][Cout][Cin][k][k] (k = 3: ntaps 9, pad 1; k = 1: ntaps 1).  transpose != 0: the input-gradient
 * product (x has Cout channels, y has Cin).  stats (optional) [B][2][C_y]: per-channel sum and sum of squares of the stored y. */
int fumi_hip_rn12_conv(fumi_ws_t* ws, fumi_stream_t stream, int B, int M, int H, int W, int Cin, int Cout, int ntaps, int transpose,
        const void* x, const float* Wt, void* y, float* stats);
int fumi_hip_rn12_wgrad(fumi_ws_t* ws, fumi_stream_t stream, int B, int M, int H, int W, int Cin, int Cout, int ntaps,
        const void* x, const void* dy, float* dW);
/* The three 3x3 / pad 1 / stride 1 convolution products on 64 -> 64 channels (the set is closed under differentiation: the
 * second-order sweep uses nothing else).  Dense channels-last tensors x, y, dy [M,H,W,64]; weights W, dW [64,64,3,3] (OIHW). */
int fumi_hip_conv3x3_fwd(fumi_ws_t* ws, fumi_stream_t stream, int M, int H, int W, const float* x, const float* Wt, float* y);
int fumi_hip_conv3x3_bwd_data(fumi_ws_t* ws, fumi_stream_t stream, int M, int H, int W, const float* dy, const float* Wt, float* dx);
int fumi_hip_conv3x3_bwd_weight(fumi_ws_t* ws, fumi_stream_t stream, int M, int H, int W, const float* x, const float* dy, float* dW);
/* torchmeta gradient_update_parameters / the reference's in-place `hyper_params -= step_size * grad` (fumi.py:165-176):
 * out[i] = p[i] - step_size * g[i], n floats (out may alias p). */
int fumi_hip_sgd_axpy(fumi_ws_t* ws, fumi_stream_t stream, long n, const float* p, float step_size, const float* g, float* out);
/* F.cross_entropy forward + backward of M rows of N logits (fumi.py:162,182): loss [1] = mean NLL, dz [M,N] = (softmax - onehot)/M,
 * preds [M] = first arg-max (fumi.py:180).  A label outside [0,N) sets FUMI_ST_LABEL_RANGE. */
int fumi_hip_ce_fwd_bwd(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, const float* z, const int64_t* y, float* loss, float* dz,
        int64_t* preds);
/* get_prototypes (fumi/utils/utils.py:331-376): out[b,n,:] = sum_{s: y[b,s]==n} x[b,s,:] / max(count, 1)  -> [B,N,P] */
int fumi_hip_proto_reduce(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int N, int P, const float* x, const int64_t* y, float* out);


/* ---- beside the episodic path -------------------------------------------------------------------------------
 * CLIP baseline, fumi/models/clip.py.  w: 8 pointers text_fc W [P,Dt], b [P], text_fc2 W [P,P], b, image_fc W [P,D], b,
 * image_fc2 W [P,P], b.  sim [nt,ni] = cosine similarity of every (text row, image row) pair (clip.py:27-41).  loss != NULL
 * (needs nt == ni): the symmetric cross-entropy against the diagonal, (CE(sim) + CE(sim^T)) / 2 (clip.py:101-105); need_grad:
 * its gradient w.r.t. the 8 tensors (written, not accumulated). */
int fumi_hip_clip_step(fumi_ws_t* ws, fumi_stream_t stream, int nt, int ni, int Dt, int D, int P,
        const float* text, const float* image, const float* const* w, int need_grad,
        float* sim, float* loss, float* const* g_w);
/* bi-LSTM text encoders RNN / RnnHid (fumi/models/common.py:44-161), forward only: tokens [R,L] int64 -> embedding rows of
 * table [V,E] -> single-layer bidirectional LSTM over the non-PAD prefix of every row (pack_padded_sequence semantics) ->
 * out [R,2H] = each direction's final hidden state (use_cell = 0: RNN) or cell state (use_cell = 1: RnnHid).
 * w: 8 pointers in nn.LSTM's layout (gate order i,f,g,o): weight_ih_l0 [4H,E], weight_hh_l0 [4H,H], bias_ih_l0, bias_hh_l0, then
 * the four *_reverse tensors. */
int fumi_hip_lstm_bidir(fumi_ws_t* ws, fumi_stream_t stream, int R, int L, int E, int H,
        const int64_t* tokens, int64_t pad_id, const float* table, int64_t V, const float* const* w, int use_cell, float* out);
/* The same encoders under --fine_tune (fumi/models/fumi.py:65-67 leaves text_encoder.parameters() trainable; the word table is
 * nn.Embedding.from_pretrained, frozen either way, common.py:60-63).  The training-mode forward writes, besides out, the tape the
 * backward needs into caller memory of fumi_hip_lstm_tape_floats(R, L, E, H) floats; the backward is back-propagation through time
 * over the same packed prefixes: d_out [R,2H] (adjoint of out) -> g_w, 8 tensors shaped like w (written, not accumulated;
 * bias_ih and bias_hh receive the same sums).  Rows whose d_out is zero contribute nothing. */
int64_t fumi_hip_lstm_tape_floats(int R, int L, int E, int H);
int fumi_hip_lstm_bidir_train(fumi_ws_t* ws, fumi_stream_t stream, int R, int L, int E, int H,
        const int64_t* tokens, int64_t pad_id, const float* table, int64_t V, const float* const* w, int use_cell, float* out,
        float* tape);
int fumi_hip_lstm_bidir_bwd(fumi_ws_t* ws, fumi_stream_t stream, int R, int L, int E, int H,
        const int64_t* tokens, int64_t pad_id, const float* const* w, int use_cell, const float* tape, const float* d_out,
        float* const* g_w);
/* Arms the NEXT meta-step with need_grad != 0 on this workspace to also write the adjoint of its text input, what autograd hands
 * a trainable text encoder in the reference: fumi_hip_fumi_step / _indexed / fumi_hip_fumi_conv4_step / fumi_hip_fumi_resnet12_step
 * write g_text [B*N, Dt] = d(grad_scale * sum_b loss_b) / d(class text rows) (get_hyper_params, fumi.py:196-215);
 * fumi_hip_am3_step / _dx write g_text [B*S, Dt] = d loss / d text_s (every support row feeds its class prototype, am3.py:113-126).
 * One-shot: the step clears it.  NULL disarms. */
int fumi_hip_want_text_grad(fumi_ws_t* ws, float* g_text);

/* FuMI meta-step on ZERO-COPY episodes: identical to fumi_hip_fumi_step except that the image rows are not handed over as
 * x_s [B,S,D] / x_q [B,Qn,D] but addressed in an HBM-resident table [n_rows, D] through idx_s [B,S] / idx_q [B,Qn] (what
 * fumi_hip_sample_episodes produces): the two X-panel kernels read the rows where they lie, the 2*B*(S+Qn)*D*4 bytes of a
 * gathered meta-batch are never written or re-read.  Bit-identical results to the gathered call.  An index outside
 * [0, n_rows) sets FUMI_ST_LABEL_RANGE and is read as row 0. */
int fumi_hip_fumi_step_indexed(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, float dropout_p, uint64_t seed,
        const float* table, int64_t n_rows, const int64_t* idx_s, const int64_t* y_s, const int64_t* idx_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi);

/* ---- GPU-resident episode sampler (replaces fumi/dataset/data.py:294-581 + the torchmeta loader for
 * precomputed embeddings held in HBM) ---------------------------------------------------------------------------------
 * sample_episodes: for every episode b < B: N distinct classes of [0, C) and, per class, K + Q distinct members of its
 *   item list class_items[class_ptr[c] .. class_ptr[c+1]) (CSR, device arrays), all uniformly at random from the
 *   counter-based stream (seed, step) -- reproducible, restated in oracle/sampler_ref.py.  Outputs (device, int64):
 *   classes [B,N], items_s [B,N,K], items_q [B,N,Q] (class-major like torchmeta's ConcatTask: label of slot n is n).
 *   A class with fewer than K + Q items sets FUMI_ST_CLASS_MISSING (torchmeta's ClassSplitter raises) and wraps around.
 * gather_rows: out[i, :] = table[idx[i], :] for rows of row_bytes bytes (a multiple of 4): the image rows of a meta-batch,
 *   or the per-class text rows / token rows.  An index outside [0, n_rows) sets FUMI_ST_LABEL_RANGE and reads row 0. */
int fumi_hip_sample_episodes(fumi_ws_t* ws, fumi_stream_t stream, uint64_t seed, uint64_t step, int B, int N, int K, int Q,
        int C, const int64_t* class_ptr, const int64_t* class_items, int64_t* classes, int64_t* items_s, int64_t* items_q);
/* The same with torchmeta's task semantics: labels [B,N] = a random permutation of 0..N-1 per task
 * (torchmeta.transforms.Categorical relabels the class slots); fixed_split != 0: the K + Q members of every class are a
 * function of (seed, the task's class tuple) only -- ClassSplitter(shuffle=True) seeds its permutation with hash(task) + seed, so
 * a class tuple drawn again has the same support / query split.  Same (seed, step) stream for the class draw as above. */
int fumi_hip_sample_episodes_tm(fumi_ws_t* ws, fumi_stream_t stream, uint64_t seed, uint64_t step, int B, int N, int K, int Q,
        int C, const int64_t* class_ptr, const int64_t* class_items, int fixed_split, int64_t* classes, int64_t* labels,
        int64_t* items_s, int64_t* items_q);
int fumi_hip_gather_rows(fumi_ws_t* ws, fumi_stream_t stream, const void* table, int64_t n_rows, int64_t row_bytes,
        const int64_t* idx, int64_t n_idx, void* out);

/* ---- event-free read-back of a step's scalars (replaces outer_loss.detach().cpu().numpy(), fumi/models/fumi.py:195) ----
 * One single-wave launch on `stream` stores src[0..n) (device, fp32, n <= 14) into host_pinned[0..n) and then `seq` into the
 * 64-bit word at byte offset 56 of host_pinned, all with system-scope stores: the host polls that word instead of waiting on
 * an event (an async copy + event record idles the stream for ~10 us per step).  host_pinned: 64 bytes of page-locked,
 * device-accessible host memory (hipHostMalloc / a pinned torch tensor), 8-byte aligned. */
int fumi_hip_publish_scalars(fumi_ws_t* ws, fumi_stream_t stream, const float* src, int n, void* host_pinned, uint64_t seq);
/* Deferred form: the same stores ride on the next fumi_hip_adam_step launch of this workspace (the optimizer step that
 * follows a training meta-step: one launch less); fumi_hip_publish_flush issues them on their own if none came.  At most one
 * publication may be pending per workspace; src must stay valid and unchanged until it has been issued. */
int fumi_hip_publish_scalars_deferred(fumi_ws_t* ws, const float* src, int n, void* host_pinned, uint64_t seq);
int fumi_hip_publish_flush(fumi_ws_t* ws, fumi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FUMI_HIP_H */
