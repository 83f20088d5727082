"""A checker engine for the CPU suite: same interface as fumi_amd.engine.HipEngine, arithmetic from the oracle
(oracle/fumi_ref.py).  Lives under tests/ and is installed with fumi_amd.engine.set_engine only by tests: it lets the
host-side plumbing (evaluate, loops, sharding, CLI, checkpoints) run in the GPU-less container."""
import torch
import torch.nn.functional as F

from oracle import fumi_ref as R


class OracleEngine:
    name = "oracle-cpu (tests only)"
    runs_on_host = True                 # (a capability the product engine does not have: fumi_amd.main.check_supported)

    def glove_bag_select(self, tokens_s, y_s, n_way, table, pad_id, mode):
        rows = torch.stack([R.class_text_select(tokens_s[b], y_s[b], n_way) for b in range(tokens_s.shape[0])])
        return R.word_embedding_pool(rows, table, pad_id, mode)

    def fumi_step(self, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, need_grad, grad_scale,
                  g_theta=None, g_phi=None, cls_text=None, stats=None, dropout_p=0.0, seed=0):
        B = x_s.shape[0]
        want, extra = self._take_text_grad(need_grad), None
        if cls_text is not None:                # expand the per-class rows back to per-sample rows for the oracle
            if want is not None:
                cls_text = cls_text.detach().clone().requires_grad_(True)
                extra = [cls_text]
            text_s = torch.gather(cls_text, 1, y_s[..., None].expand(-1, -1, cls_text.shape[-1]))
        th = [t.detach().clone().requires_grad_(True) for t in theta]
        ph = [t.detach().clone().requires_grad_(True) for t in phi]
        drop = None
        if dropout_p > 0:
            from helpers import dropout_mask
            drop = lambda b, call, layer, rows, width: dropout_mask(seed, dropout_p, b, call, layer, rows, width)
        out = R.fumi_meta_step(th, ph, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head, need_grad=need_grad, dropout=drop,
                               extra=extra)
        if need_grad:
            for dst, g in zip(list(g_theta) + list(g_phi), out["g_theta"] + out["g_phi"]):
                dst.copy_(g * (B * grad_scale))               # oracle returns mean-loss grads = (1/B) sum_b
            if want is not None:
                want.copy_((out["g_extra"][0] * (B * grad_scale)).reshape(want.shape))
        if stats is not None:
            stats.copy_(torch.stack([out["loss_b"].sum(), out["acc_b"].sum()]) * grad_scale)
        return dict(logits=out["logits"], preds=out["preds"], preds_f=out["preds"].float(), loss_b=out["loss_b"], acc_b=out["acc_b"])

    def maml_step(self, x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad, grad_scale, g_params=None, stats=None):
        B = x_s.shape[0]
        p = [t.detach().clone().requires_grad_(True) for t in params]
        out = R.maml_meta_step(p, x_s, y_s, x_q, y_q, T, alpha, first_order, need_grad=need_grad)
        if need_grad:
            for dst, g in zip(g_params, out["g_params"]):
                dst.copy_(g * (B * grad_scale))
        if stats is not None:
            stats.copy_(torch.stack([out["loss_b"].sum(), out["acc_b"].sum()]) * grad_scale)
        return dict(logits=out["logits"], preds=out["preds"], preds_f=out["preds"].float(), loss_b=out["loss_b"], acc_b=out["acc_b"])

    def am3_step(self, x_s, y_s, x_q, y_q, text_s, w, n_way, lamda_fixed, need_grad, grad_scale, g_w=None, dropout_p=0.0, seed=0,
                 want_dx=False):
        from fumi_amd.hip import AM3_KEYS
        B, Qn = x_q.shape[0], x_q.shape[1]
        wd = {k: t.detach().clone().requires_grad_(True) for k, t in zip(AM3_KEYS, w)}
        if want_dx and need_grad:           # adjoints of the image rows: one more autograd pass over the same loss
            xs, xq = x_s.detach().clone().requires_grad_(True), x_q.detach().clone().requires_grad_(True)
            masks_ = None
            if dropout_p > 0:
                from helpers import dropout_mask_flat
                Rs_, Ht_ = x_s.shape[0] * x_s.shape[1], w[2].shape[0]
                masks_ = (dropout_mask_flat(seed, dropout_p, 1, Rs_, Ht_), dropout_mask_flat(seed, dropout_p, 2, Rs_, Ht_))
            wl = {k: t.detach() for k, t in zip(AM3_KEYS, w)}
            im_s = torch.nn.functional.linear(xs, wl["Wi"], wl["bi"]); im_q = torch.nn.functional.linear(xq, wl["Wi"], wl["bi"])
            t1 = torch.relu(torch.nn.functional.linear(text_s, wl["G0"], wl["g0"]))
            if masks_ is not None:
                t1 = t1 * masks_[0].view_as(t1)
            tx = torch.nn.functional.linear(t1, wl["G1"], wl["g1"])
            l1 = torch.relu(torch.nn.functional.linear(tx, wl["H0"], wl["h0"]))
            if masks_ is not None:
                l1 = l1 * masks_[1].view_as(l1)
            lam = torch.sigmoid(torch.nn.functional.linear(l1, wl["H1"], wl["h1"]))
            if lamda_fixed == 0:
                lam = torch.zeros_like(lam)
            elif lamda_fixed == 1:
                lam = torch.ones_like(lam)
            loss_x = R.prototypical_loss(R.get_prototypes(im_s, tx, lam, y_s, n_way), im_q, y_q)
            dxs, dxq = torch.autograd.grad(loss_x, [xs, xq])
            self._dx = (dxs * (B * grad_scale), dxq * (B * grad_scale))
        masks = None
        if dropout_p > 0:
            from helpers import dropout_mask_flat
            Rs, Ht = x_s.shape[0] * x_s.shape[1], w[2].shape[0]
            masks = (dropout_mask_flat(seed, dropout_p, 1, Rs, Ht), dropout_mask_flat(seed, dropout_p, 2, Rs, Ht))
        want, extra = self._take_text_grad(need_grad), None
        if want is not None:
            text_s = text_s.detach().clone().requires_grad_(True)
            extra = [text_s]
        out = R.am3_step(wd, text_s, x_s, y_s, x_q, y_q, n_way, lamda_fixed, need_grad=need_grad, masks=masks, extra=extra)
        if need_grad:
            for dst, k in zip(g_w, AM3_KEYS):
                dst.copy_(out["grads"][k] * (B * grad_scale))
            if want is not None:
                want.copy_((out["g_extra"][0] * (B * grad_scale)).reshape(want.shape))
        correct = out["preds"].eq(y_q).float().sum().reshape(1)
        dx = getattr(self, "_dx", (None, None)) if (want_dx and need_grad) else (None, None)
        return dict(loss=(out["loss"] * (B * grad_scale)).reshape(1), preds=out["preds"], lamda_s=out["lamda_s"],
                    correct=correct, dx_s=dx[0], dx_q=dx[1])

    def conv4_encode(self, x_s, x_q, theta, keep_tape=False):
        from oracle import conv4_ref as C
        with torch.no_grad():
            f_s = torch.stack([C.conv4_features(x_s[b], theta) for b in range(x_s.shape[0])])
            f_q = torch.stack([C.conv4_features(x_q[b], theta) for b in range(x_q.shape[0])])
        self._tape = keep_tape
        return f_s, f_q

    def conv4_encode_bwd(self, x_s, x_q, dfeats_s, dfeats_q, theta_like, scale=1.0, g_theta=None):
        from oracle import conv4_ref as C
        assert getattr(self, "_tape", False), "conv4_encode_bwd without a kept tape"
        self._tape = False
        th = [t.detach().clone().requires_grad_(True) for t in theta_like]
        tot = 0.0
        for b in range(x_s.shape[0]):
            tot = tot + (C.conv4_features(x_s[b], th) * dfeats_s[b]).sum() + (C.conv4_features(x_q[b], th) * dfeats_q[b]).sum()
        gs = torch.autograd.grad(tot, th)
        if g_theta is None:
            g_theta = [torch.empty_like(t) for t in theta_like]
        for dst, g in zip(g_theta, gs):
            dst.copy_(g * scale)
        return g_theta

    def fumi_conv4_step(self, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, need_grad, grad_scale,
                        g_theta=None, g_phi=None, cls_text=None, stats=None):
        from oracle import conv4_ref as C
        B = x_s.shape[0]
        want, extra = self._take_text_grad(need_grad), None
        if cls_text is not None:
            if want is not None:
                cls_text = cls_text.detach().clone().requires_grad_(True)
                extra = [cls_text]
            text_s = torch.gather(cls_text, 1, y_s[..., None].expand(-1, -1, cls_text.shape[-1]))
        th = [t.detach().clone().requires_grad_(True) for t in theta]
        ph = [t.detach().clone().requires_grad_(True) for t in phi]
        out = C.fumi_conv4_meta_step(th, ph, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head, need_grad=need_grad, extra=extra)
        if need_grad:
            for dst, g in zip(list(g_theta) + list(g_phi), out["g_theta"] + out["g_phi"]):
                dst.copy_(g * (B * grad_scale))
            if want is not None:
                want.copy_((out["g_extra"][0] * (B * grad_scale)).reshape(want.shape))
        if stats is not None:
            stats.copy_(torch.stack([out["loss_b"].sum(), out["acc_b"].sum()]) * grad_scale)
        return dict(logits=out["logits"], preds=out["preds"], preds_f=out["preds"].float(), loss_b=out["loss_b"], acc_b=out["acc_b"])

    def maml_conv4_step(self, x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad, grad_scale, g_params=None, stats=None):
        from oracle import conv4_ref as C
        B = x_s.shape[0]
        p = [t.detach().clone().requires_grad_(True) for t in params]
        out = C.maml_conv4_meta_step(p, x_s, y_s, x_q, y_q, T, alpha, first_order, need_grad=need_grad)
        if need_grad:
            for dst, g in zip(g_params, out["g_params"]):
                dst.copy_(g * (B * grad_scale))
        if stats is not None:
            stats.copy_(torch.stack([out["loss_b"].sum(), out["acc_b"].sum()]) * grad_scale)
        return dict(logits=out["logits"], preds=out["preds"], preds_f=out["preds"].float(), loss_b=out["loss_b"], acc_b=out["acc_b"])

    def conv4_features(self, x, theta):
        from oracle import conv4_ref as C
        return torch.stack([C.conv4_features(x[g], theta) for g in range(x.shape[0])])

    def fumi_resnet12_step(self, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, need_grad, grad_scale,
                           g_theta=None, g_phi=None, cls_text=None, stats=None):
        from oracle import resnet12_ref as C
        B = x_s.shape[0]
        want, extra = self._take_text_grad(need_grad), None
        if cls_text is not None:
            if want is not None:
                cls_text = cls_text.detach().clone().requires_grad_(True)
                extra = [cls_text]
            text_s = torch.gather(cls_text, 1, y_s[..., None].expand(-1, -1, cls_text.shape[-1]))
        th = [t.detach().clone().requires_grad_(True) for t in theta]
        ph = [t.detach().clone().requires_grad_(True) for t in phi]
        out = C.fumi_meta_step(th, ph, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head, need_grad=need_grad, extra=extra)
        if need_grad:
            for dst, g in zip(list(g_theta) + list(g_phi), out["g_theta"] + out["g_phi"]):
                dst.copy_(g * (B * grad_scale))
            if want is not None:
                want.copy_((out["g_extra"][0] * (B * grad_scale)).reshape(want.shape))
        if stats is not None:
            stats.copy_(torch.stack([out["loss_b"].sum(), out["acc_b"].sum()]) * grad_scale)
        return dict(logits=out["logits"], preds=out["preds"], preds_f=out["preds"].float(), loss_b=out["loss_b"], acc_b=out["acc_b"])

    def maml_resnet12_step(self, x_s, y_s, x_q, y_q, params, T, alpha, first_order, need_grad, grad_scale, g_params=None, stats=None):
        from oracle import resnet12_ref as C
        B = x_s.shape[0]
        p = [t.detach().clone().requires_grad_(True) for t in params]
        out = C.maml_meta_step(p, x_s, y_s, x_q, y_q, T, alpha, first_order, need_grad=need_grad)
        if need_grad:
            for dst, g in zip(g_params, out["g_params"]):
                dst.copy_(g * (B * grad_scale))
        if stats is not None:
            stats.copy_(torch.stack([out["loss_b"].sum(), out["acc_b"].sum()]) * grad_scale)
        return dict(logits=out["logits"], preds=out["preds"], preds_f=out["preds"].float(), loss_b=out["loss_b"], acc_b=out["acc_b"])

    def resnet12_features(self, x, theta):
        from oracle import resnet12_ref as C
        return torch.stack([C.features(x[g], theta) for g in range(x.shape[0])])

    def clip_step(self, text, image, w, need_loss=True, need_grad=True, g_w=None):
        ww = [t.detach().clone().requires_grad_(True) for t in w]
        if not need_loss:
            return dict(sim=R.clip_forward([t.detach() for t in ww], text, image), loss=None, grads=None)
        out = R.clip_step(ww, text, image, need_grad=need_grad)
        return dict(sim=out["sim"], loss=out["loss"].reshape(1), grads=out.get("grads"))

    def lstm_bidir(self, tokens, table, lstm_w, pad_id, use_cell):
        lead = tokens.shape[:-1]
        out = R.lstm_encode(tokens.reshape(1, -1, tokens.shape[-1]), table, lstm_w, pad_id, use_cell)
        return out.reshape(*lead, -1)

    def lstm_bidir_train(self, tokens, table, lstm_w, pad_id, use_cell):
        return self.lstm_bidir(tokens, table, lstm_w, pad_id, use_cell), None         # (the checker re-runs the forward under autograd)

    def lstm_bidir_bwd(self, tokens, table, lstm_w, pad_id, use_cell, tape, d_out):
        w = [t.detach().clone().requires_grad_(True) for t in lstm_w]
        out = R.lstm_encode(tokens.reshape(1, -1, tokens.shape[-1]), table, w, pad_id, use_cell)
        return list(torch.autograd.grad((out.reshape(d_out.shape) * d_out).sum(), w))

    def class_rows_select(self, rows_s, y_s, n_way):
        return torch.stack([R.class_text_select(rows_s[b], y_s[b], n_way) for b in range(rows_s.shape[0])])

    def want_text_grad(self, device, g_text):
        self._text_grad = g_text

    def _take_text_grad(self, need_grad):
        """The armed text-adjoint buffer, consumed by the next step with need_grad (fumi_hip_want_text_grad's contract)."""
        if not need_grad:
            return None
        want, self._text_grad = getattr(self, "_text_grad", None), None
        return want

    def glove_bag(self, tokens, table, pad_id, mode):
        return R.word_embedding_pool(tokens, table, pad_id, mode)

    def linear(self, x, W, b=None, act=0):
        y = F.linear(x, W, b)
        return [y, torch.relu(y), torch.tanh(y), torch.sigmoid(y)][act]

    def sgd_axpy(self, p, step_size, g):
        return p - step_size * g

    def check(self, device):
        pass
