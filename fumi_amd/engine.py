"""The compute engine behind the model classes.  There is exactly one implementation, the HIP library
(``fumi_amd/hip.py`` -> ``lib/libfumi_hip.so``); it raises when the library is missing or a tensor is not on a GPU.
The indirection exists so host-side plumbing (loops, flags, checkpoints, sharding) can be unit-tested: tests may
install a checker engine with ``set_engine`` -- product code never does."""
import torch

from . import hip

_ENGINE = None


class HipEngine:
    name = "hip-gfx950"
    folds_optimizer_step = True        # fumi_step consumes a deferred Adam step / publication (fumi_hip_adam_step_deferred)

    def _ws(self, t):
        return hip.Workspace.get(t.device if isinstance(t, torch.Tensor) and t.is_cuda else hip._dev(t))

    def fumi_step(self, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, need_grad, grad_scale,
                  g_theta=None, g_phi=None, cls_text=None, stats=None, dropout_p=0.0, seed=0):
        """text_s [B,S,Dt] (class rows selected on the device) or cls_text [B,N,Dt] (already selected).
        stats [2] (optional) receives grad_scale * (sum loss_b, sum acc_b)."""
        if cls_text is not None:
            return hip.fumi_step(self._ws(x_s), x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head, cls_text=cls_text,
                                 need_grad=need_grad, grad_scale=grad_scale, g_theta=g_theta, g_phi=g_phi, stats=stats,
                                 dropout_p=dropout_p, seed=seed)
        return hip.fumi_step_select(self._ws(x_s), n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head,
                                    need_grad=need_grad, grad_scale=grad_scale, g_theta=g_theta, g_phi=g_phi, stats=stats,
                                    dropout_p=dropout_p, seed=seed)

    def glove_bag_select(self, tokens_s, y_s, n_way, table, pad_id, mode, defer=False):
        """defer: the bag rides in the first launch of the ``fumi_step`` that must follow with the result as ``cls_text``"""
        return hip.glove_bag_select(self._ws(tokens_s), tokens_s, y_s, n_way, table, pad_id, mode, defer=defer)

    def maml_step(self, x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad, grad_scale, g_params=None,
                  stats=None):
        return hip.maml_step(self._ws(x_s), x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad=need_grad,
                             grad_scale=grad_scale, g_params=g_params, stats=stats)

    def am3_step(self, x_s, y_s, x_q, y_q, text_s, w, n_way, lamda_fixed, need_grad, grad_scale, g_w=None,
                 dropout_p=0.0, seed=0, stats=None, want_dx=False):
        return hip.am3_step(self._ws(x_s), x_s, y_s, x_q, y_q, text_s, w, n_way, lamda_fixed, need_grad=need_grad,
                            grad_scale=grad_scale, g_w=g_w, dropout_p=dropout_p, seed=seed, stats=stats, want_dx=want_dx)

    def conv4_encode(self, x_s, x_q, theta, keep_tape=False):
        """Conv4 in front of a step with its own workspace (AM3): the tape lives in the device's "encoder" workspace."""
        return hip.conv4_encode(hip.Workspace.get(x_s.device, "encoder"), x_s, x_q, theta, keep_tape=keep_tape)

    def conv4_encode_bwd(self, x_s, x_q, dfeats_s, dfeats_q, theta_like, scale=1.0, g_theta=None):
        return hip.conv4_encode_bwd(hip.Workspace.get(x_s.device, "encoder"), x_s, x_q, dfeats_s, dfeats_q, theta_like,
                                    scale=scale, g_theta=g_theta)

    def fumi_conv4_step(self, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, need_grad, grad_scale,
                        g_theta=None, g_phi=None, cls_text=None, stats=None):
        """FuMI with the Conv4 encoder at the im_net seam: x_s [B,S,C,H,W], x_q [B,Qn,C,H,W]."""
        return hip.fumi_conv4_step(self._ws(x_s), n_way, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head, cls_text=cls_text,
                                   text_s=text_s, need_grad=need_grad, grad_scale=grad_scale, g_theta=g_theta, g_phi=g_phi,
                                   stats=stats)

    def maml_conv4_step(self, x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad, grad_scale, g_params=None,
                        stats=None):
        return hip.maml_conv4_step(self._ws(x_s), x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad=need_grad,
                                   grad_scale=grad_scale, g_params=g_params, stats=stats)

    def fumi_resnet12_step(self, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, need_grad, grad_scale,
                           g_theta=None, g_phi=None, cls_text=None, stats=None):
        """FuMI with the bf16 ResNet-12 encoder at the im_net seam (BASELINE.json configs[4])."""
        return hip.fumi_resnet12_step(self._ws(x_s), n_way, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head, cls_text=cls_text,
                                      text_s=text_s, need_grad=need_grad, grad_scale=grad_scale, g_theta=g_theta, g_phi=g_phi,
                                      stats=stats)

    def maml_resnet12_step(self, x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad, grad_scale, g_params=None,
                           stats=None):
        return hip.maml_resnet12_step(self._ws(x_s), x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad=need_grad,
                                      grad_scale=grad_scale, g_params=g_params, stats=stats)

    def resnet12_features(self, x, theta):
        return hip.resnet12_features(self._ws(x), x, theta)

    def conv4_features(self, x, theta):
        return hip.conv4_features(self._ws(x), x, theta)

    def clip_step(self, text, image, w, need_loss=True, need_grad=True, g_w=None):
        return hip.clip_step(self._ws(text), text, image, w, need_loss=need_loss, need_grad=need_grad, g_w=g_w)

    def lstm_bidir(self, tokens, table, lstm_w, pad_id, use_cell):
        return hip.lstm_bidir(self._ws(tokens), tokens, table, lstm_w, pad_id, use_cell)

    def lstm_bidir_train(self, tokens, table, lstm_w, pad_id, use_cell):
        """(out, tape): the forward of a trainable bi-LSTM encoder (--fine_tune with RNN / RNNhid, fumi/models/fumi.py:65-67)."""
        return hip.lstm_bidir_train(self._ws(tokens), tokens, table, lstm_w, pad_id, use_cell)

    def lstm_bidir_bwd(self, tokens, table, lstm_w, pad_id, use_cell, tape, d_out):
        return hip.lstm_bidir_bwd(self._ws(tokens), tokens, table, lstm_w, pad_id, use_cell, tape, d_out)

    def class_rows_select(self, rows_s, y_s, n_way):
        """[B,N,...] the first support row of every class (fumi.py:207-210) of float32 OR int64 rows [B,S,W]: a pure row copy, so
        token rows go through the float kernel as pairs of 32-bit words."""
        if rows_s.dtype == torch.int64:
            out = hip.class_text_select(self._ws(rows_s), rows_s.contiguous().view(torch.float32), y_s, n_way)
            return out.view(torch.int64)
        return hip.class_text_select(self._ws(rows_s), rows_s.contiguous(), y_s, n_way)

    def want_text_grad(self, device, g_cls_text):
        """The next fumi_step(need_grad=True) on `device` also writes d loss / d class text rows into g_cls_text [B,N,Dt]."""
        hip.want_text_grad(hip.Workspace.get(torch.device(device)), g_cls_text)

    def am3_metrics(self, n_way, stats):
        return hip.am3_metrics(self._ws(stats), n_way, stats)

    def glove_bag(self, tokens, table, pad_id, mode):
        return hip.glove_bag(self._ws(tokens), tokens, table, pad_id, mode)

    def linear(self, x, W, b=None, act=0):
        return hip.linear_fwd(self._ws(x), x.contiguous(), W.contiguous(), b, act)

    def sgd_axpy(self, p, step_size, g):
        return hip.sgd_axpy(self._ws(p), p, step_size, g)

    def check(self, device):
        """Synchronising validity check of the last calls (labels in range, every class has a support sample)."""
        device = torch.device(device)
        if device.type == "cuda":
            hip.raise_on_status(hip.Workspace.get(device).read_status())


def check_status(device):
    """Raise what the reference raises for an invalid episode.  The kernels record a label outside [0, N) or a class without
    a support sample in the workspace's status word instead of faulting; the reference's per-class loop raises IndexError there
    (fumi/models/fumi.py:209).  Reading the word synchronises the stream, so the loops call this where they wait for the
    device anyway: at the end of every validation / test loop (which also covers the training steps before it) and before a
    run ends."""
    eng = get_engine()
    if hasattr(eng, "check"):
        eng.check(device)


def get_engine():
    global _ENGINE
    if _ENGINE is None:
        hip.lib()                      # fail loudly here if the shared object has not been built
        _ENGINE = HipEngine()
    return _ENGINE


def set_engine(engine):
    """Test hook (tests/ only)."""
    global _ENGINE
    old, _ENGINE = _ENGINE, engine
    return old
