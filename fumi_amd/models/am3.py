"""AM3: prototypical network with a text-gated convex combination -- host-side mirror of fumi/models/am3.py.

Same constructor keywords, attributes (``image_encoder``, ``text_encoder``, ``g``, ``h``), ``state_dict`` keys
(``image_encoder.*``, ``g.{0,3}.*``, ``h.{0,3}.*``) and ``evaluate`` return tuples (am3.py:16-212).  The arithmetic of a
step (encoders, per-class prototypes, squared distances, CE, arg-min, backward) is one call of the MI355X engine
(csrc/am3.hip); precision / recall / F1 are computed from the integer predictions on the host like the reference does
with sklearn (utils.py:319-326)."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import dist as fdist
from .. import engine as _engine
from .. import lazy
from ..flatgrad import FlatGrads, ParamWatch
from ..utils import utils as utils
from ..utils.average_meter import AverageMeter
from ..utils.wandb_compat import wandb
from .common import RNN, RnnHid, WordEmbedding, _BiLstmEncoder


class AM3(nn.Module):
    def __init__(self, im_encoder, im_emb_dim, text_encoder, text_emb_dim=300, text_hid_dim=300, prototype_dim=512,
                 dropout=0.7, fine_tune=False, dictionary=None, pooling_strat="mean", lamda_fixed=None, image_size=84,
                 image_channels=3):
        super().__init__()
        self.im_emb_dim = im_emb_dim
        self.text_encoder_type = text_encoder
        self.text_emb_dim = text_emb_dim
        self.text_hid_dim = text_hid_dim
        self.prototype_dim = prototype_dim
        self.dropout = dropout
        self.fine_tune = fine_tune
        self.dictionary = dictionary
        self.pooling_strat = pooling_strat
        self.lamda_fixed = lamda_fixed

        self.conv = None
        if im_encoder in ("precomputed", "resnet"):            # "resnet" is the same Linear in the reference (am3.py:44-46)
            self.image_encoder = nn.Linear(im_emb_dim, prototype_dim)
        elif im_encoder == "conv4":
            # this engine's extension at the image_encoder seam (am3.py:41-46; BASELINE configs[3] as worded): the Conv4 backbone
            # on raw images, then the reference's Linear into the prototype space.  Every episode's support set and query set
            # is one batch-statistics group (like the MAML / FuMI Conv4 path); "parity unpinned" (oracle/conv4_ref.py).
            from .conv4 import Conv4
            self.conv = Conv4(image_channels, 64, 4, image_size)
            self.im_emb_dim = self.conv.feature_dim
            self.image_encoder = nn.Linear(self.conv.feature_dim, prototype_dim)
        else:
            raise NameError(f"{im_encoder} not allowed as image encoder")
        if text_encoder in ("BERT", "precomputed"):
            self.text_encoder = nn.Identity()
        elif text_encoder in ("w2v", "glove"):
            self.text_encoder = WordEmbedding(text_encoder, pooling_strat, dictionary)
            self.text_emb_dim = self.text_encoder.embedding_dim
        elif text_encoder == "RNN":
            self.text_encoder = RNN("glove", pooling_strat, dictionary, self.text_emb_dim)
        elif text_encoder == "RNNhid":
            self.text_encoder = RnnHid("glove", pooling_strat, dictionary, self.text_emb_dim)
        elif text_encoder == "rand":
            self.text_encoder = nn.Linear(self.text_emb_dim, self.text_emb_dim)
        else:
            raise NameError(f"{text_encoder} not allowed as text encoder")
        if not fine_tune:
            print("Not fine tuning embeddings")
            for p in self.text_encoder.parameters():
                p.requires_grad = False
        self.g = nn.Sequential(nn.Linear(self.text_emb_dim, text_hid_dim), nn.ReLU(), nn.Dropout(p=dropout),
                               nn.Linear(text_hid_dim, prototype_dim))
        self.h = nn.Sequential(nn.Linear(prototype_dim, text_hid_dim), nn.ReLU(), nn.Dropout(p=dropout),
                               nn.Linear(text_hid_dim, 1))
        self._flat = None
        self._pcache = None

    def _w(self):
        return [self.image_encoder.weight, self.image_encoder.bias, self.g[0].weight, self.g[0].bias, self.g[3].weight,
                self.g[3].bias, self.h[0].weight, self.h[0].bias, self.h[3].weight, self.h[3].bias]

    def _flat_grads(self, num_ways=0):
        params = self._w() + (self.conv.theta() if self.conv is not None else [])
        extra = 3 + num_ways * num_ways                 # [loss, correct, lamda | confusion counts] ride in the all-reduce
        if self._flat is None or not self._flat.matches(params) or self._flat.tail.numel() != extra:
            self._flat = FlatGrads(params, extra=extra)
        return self._flat

    def _step_params(self, need_grad, num_ways):
        """(detached weights, detached backbone tensors | None, flat gradient buffer | None) as objects that stay the SAME from
        step to step (hip.py validates a list once); rebuilt when a parameter object was replaced or moved (``_apply``)."""
        c = self._pcache
        if c is None or not c[0].valid():
            w, th = self._w(), (self.conv.theta() if self.conv is not None else [])
            c = self._pcache = (ParamWatch(self, w + th), None, [p.detach() for p in w],
                                [p.detach() for p in th] if self.conv is not None else None)
            self._flat = None
        fg = None
        if need_grad:
            fg = self._flat
            if fg is None or fg.tail.numel() != 3 + num_ways * num_ways:
                fg = self._flat_grads(num_ways)
        return c[2], c[3], fg

    def _apply(self, fn, recurse=True):
        self._pcache = None
        return super()._apply(fn, recurse)

    def _encode_text(self, text):
        if self.text_encoder_type in ("BERT", "precomputed"):
            return text.to(torch.float32).contiguous()
        if self.text_encoder_type == "rand":
            # am3.py:118-121: the text PROTOTYPES are drawn uniformly in [-1, 1), g is not applied
            return 2 * torch.rand(*text.shape[:2], self.prototype_dim, device=text.device) - 1
        return self.text_encoder(text)

    def _rand_g(self, device):
        """text_encoder='rand' hands prototype-space rows to a step that applies g: an exact identity in g's own form,
        x = relu(x) - relu(-x)  (G0 = [I; -I; 0], G1 = [I, -I, 0], zero biases; every sum has one non-zero term)."""
        c = getattr(self, "_rand_g_cache", None)
        if c is None or c[0].device != device:
            P, Ht = self.prototype_dim, self.g[0].weight.shape[0]
            if Ht < 2 * P:
                raise NotImplementedError("text_encoder='rand' needs text_hid_dim >= 2 * prototype_dim on this engine")
            G0 = torch.zeros(Ht, P, device=device); G1 = torch.zeros(P, Ht, device=device)
            eye = torch.eye(P, device=device)
            G0[:P], G0[P:2 * P], G1[:, :P], G1[:, P:2 * P] = eye, -eye, eye, -eye
            c = self._rand_g_cache = [G0, torch.zeros(Ht, device=device), G1, torch.zeros(P, device=device)]
            self._rand_g_scratch = [torch.empty_like(t) for t in c]
        return c

    def forward(self, inputs, im_only=False):
        """Inference helper (am3.py:90-126): prototype-space embeddings through the engine's linear op."""
        eng = _engine.get_engine()
        idx, text, im = inputs
        w = [p.detach() for p in self._w()]
        if self.conv is not None:
            im = self.conv(im)              # [..., M, C, H, W] -> [..., M, F], batch statistics per image set
        lead = im.shape[:-1]
        im_emb = eng.linear(im.reshape(-1, im.shape[-1]).contiguous(), w[0], w[1]).reshape(*lead, -1)
        if im_only:
            return im_emb
        enc = self._encode_text(text)
        if self.text_encoder_type == "rand":
            t = enc.reshape(-1, enc.shape[-1]).contiguous()
        else:
            t = eng.linear(eng.linear(enc.reshape(-1, enc.shape[-1]).contiguous(), w[2], w[3], act=1), w[4], w[5])
        lam = eng.linear(eng.linear(t, w[6], w[7], act=1), w[8], w[9], act=3)
        return im_emb, t.reshape(*lead, -1), lam.reshape(*lead, 1)

    def evaluate(self, batch, optimizer, scheduler, num_ways, device, task="train"):
        """One meta-batch (am3.py:128-212): 6-tuple for train/val, 11-tuple for test."""
        train = task == "train"
        if train:
            if not self.training:                # (nn.Module.train() walks every submodule: skip it when nothing changes)
                self.train()
        elif self.training:
            self.eval()
        drop_p = float(self.dropout) if (train and self.dropout > 0) else 0.0       # nn.Dropout of g / h, train mode only
        drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) + 7919 * fdist.world()[0] if drop_p > 0 else 0
        (s_idx, s_text, s_im), s_y = batch['train']
        (q_idx, _, q_im), q_y = batch['test']
        B, Qn = q_im.shape[0], q_im.shape[1]
        lo, hi = fdist.shard(B)
        whole = lo == 0 and hi == B
        device = torch.device(device) if not isinstance(device, torch.device) else device

        def to(t):                               # this rank's episodes on the device (a tensor that is already there is used as it is)
            if whole and t.device == device and t.is_contiguous():
                return t
            return t[lo:hi].to(device).contiguous()
        x_s, x_q, y_s, y_q = to(s_im), to(q_im), to(s_y), to(q_y)
        if x_s.dtype != torch.float32 or x_q.dtype != torch.float32:
            x_s, x_q = x_s.float(), x_q.float()
        need_grad = train and torch.is_grad_enabled()
        # a trainable bi-LSTM (--fine_tune with RNN / RNNhid, am3.py:61-76): taped forward here, the step below hands back the
        # adjoint of every support row's text encoding, the LSTM's backward follows the step
        lstm_ft = need_grad and isinstance(self.text_encoder, _BiLstmEncoder) and self.text_encoder.trainable()
        if lstm_ft:
            tok = to(s_text)
            text, lstm_tape = self.text_encoder.forward_train(tok)
        else:
            text = self._encode_text(to(s_text))
        w_det, th_det, fg = self._step_params(need_grad, num_ways)
        eng = _engine.get_engine()
        # train / val on the GPU: the step also leaves [loss, correct, mean lamda, confusion counts] in the buffer's tail, one
        # all-reduce covers gradients and statistics, and accuracy / macro P / R / F1 come from one small kernel -- the reference
        # moves the predictions to the host and calls sklearn every meta-batch (utils.py:319-326): a blocking copy per step
        on_device = task != "test" and x_s.is_cuda and num_ways <= 64 and hasattr(eng, "am3_metrics")
        tail = fg.tail if need_grad else torch.empty(3 + num_ways * num_ways, device=x_s.device, dtype=torch.float32)
        img_s = img_q = theta = None
        if self.conv is not None:           # raw images -> Conv4 features; the tape stays in the encoder's own workspace
            img_s, img_q = x_s, x_q
            theta = th_det
            x_s, x_q = eng.conv4_encode(img_s, img_q, theta, keep_tape=need_grad)
        g_w = fg.split(10)[0] if need_grad else None
        rand_text = self.text_encoder_type == "rand"
        if rand_text:
            if drop_p > 0:
                raise NotImplementedError("text_encoder='rand' with dropout > 0 in training: the step's dropout would also "
                                          "hit the identity that stands in for g (the reference applies it to h only)")
            w_det = w_det[:2] + self._rand_g(x_s.device) + w_det[6:]
            if need_grad:
                g_w = list(g_w[:2]) + self._rand_g_scratch + list(g_w[6:])
        if lstm_ft:
            g_text = torch.empty_like(text)
            eng.want_text_grad(x_s.device, g_text)
        out = eng.am3_step(x_s, y_s, x_q, y_q, text, w_det, num_ways,
                           self.lamda_fixed, need_grad=need_grad, grad_scale=1.0 / B,
                           g_w=g_w, dropout_p=drop_p, seed=drop_seed,
                           **({"stats": tail} if on_device else {}),
                           **({"want_dx": True} if (self.conv is not None and need_grad) else {}))
        if self.conv is not None and need_grad:     # ... and backwards from the adjoints of the features (already scaled by 1/B)
            eng.conv4_encode_bwd(img_s, img_q, out["dx_s"], out["dx_q"], theta, scale=1.0, g_theta=fg.split(10)[1])
        if not on_device:
            torch.stack([out["loss"].reshape(()), out["correct"].reshape(()) / (B * Qn),
                         out["lamda_s"].sum() / (B * out["lamda_s"].shape[1])], out=tail[:3])
        fdist.all_reduce_sum_(fg.flat if need_grad else tail)
        m6 = lazy.scalars(eng.am3_metrics(num_ways, tail), 6, defer=need_grad) if on_device else None
        if need_grad:
            optimizer.zero_grad()
            fg.attach()
            if self.lamda_fixed in (0, 1):
                # h is not part of the graph when lamda is overridden (am3.py:174-177): the reference leaves its .grad
                # None, so the optimizer (and its weight decay) skips those tensors
                for p in self.h.parameters():
                    p.grad = None
            if rand_text:                        # g is not part of the graph (am3.py:118-121): its .grad stays None
                for p in self.g.parameters():
                    p.grad = None
            if lstm_ft:                          # the encoder's .grad (this rank's episodes, then summed like every other gradient)
                for g in self.text_encoder.backward(tok, lstm_tape, g_text):
                    fdist.all_reduce_sum_(g)
            getattr(optimizer, "step_fused", optimizer.step)()
            if scheduler:
                scheduler.step()
            lazy.flush(x_s.device)
        if on_device:
            return m6
        preds = fdist.all_gather_rows(out["preds"])
        lam_s = fdist.all_gather_rows(out["lamda_s"])
        stats = tail.detach().cpu().numpy()
        preds_np = preds.detach().cpu().numpy()
        targets_np = q_y.detach().cpu().numpy()
        acc, f1, prec, rec = utils.macro_metrics(targets_np, preds_np)
        if task == "test":
            return (stats[0], acc, f1, prec, rec, stats[2], preds_np, q_y.to(device), q_idx.detach().cpu().numpy(),
                    s_idx.detach().cpu().numpy(), lam_s.detach().cpu().numpy())
        return stats[0], acc, f1, prec, rec, stats[2]


def training_run(args, model, optimizer, train_loader, val_loader, max_test_batches):
    """am3.py:215-305 (validates at batch 0 too, unlike FuMI/MAML; reloads the best checkpoint at the end)."""
    best_loss, best_acc = test_loop(args, model, val_loader, max_test_batches)[:2]
    print(f"\ninitial loss: {best_loss}, acc: {best_acc}")
    best_batch_idx = 0
    opt, scheduler = optimizer if type(optimizer) == tuple else (optimizer, None)
    try:
        for batch_idx, batch in enumerate(train_loader):
            tl, ta, tf1, tp, tr, tlam = model.evaluate(batch=batch, optimizer=opt, scheduler=scheduler,
                                                       num_ways=args.num_ways, device=args.device, task="train")
            wandb.log({"train/acc": ta, "train/f1": tf1, "train/prec": tp, "train/rec": tr, "train/loss": tl,
                       "train/avg_lamda": tlam, "num_episodes": (batch_idx + 1) * args.batch_size}, step=batch_idx)
            if batch_idx % args.eval_freq == 0:
                r = test_loop(args, model, val_loader, max_test_batches)
                val_loss, val_acc, val_f1, val_prec, val_rec, val_lamda = r[:6]
                is_best = val_loss < best_loss
                if is_best:
                    best_loss, best_batch_idx = val_loss, batch_idx
                wandb.log({"val/acc": val_acc, "val/f1": val_f1, "val/prec": val_prec, "val/rec": val_rec,
                           "val/loss": val_loss, "val/avg_lamda": val_lamda}, step=batch_idx)
                utils.save_checkpoint({"batch_idx": batch_idx, "state_dict": model.state_dict(), "best_loss": best_loss,
                                       "optimizer": opt.state_dict(), "args": vars(args)}, is_best)
                print(f"\nBatch {batch_idx + 1}/{args.epochs}: \ntrain/loss: {tl}, train/acc: {ta}, train/avg_lamda: {tlam}"
                      f"\nval/loss: {val_loss}, val/acc: {val_acc}, val/avg_lamda: {val_lamda}")
            if (batch_idx > args.epochs - 1) or (args.patience > 0 and batch_idx - best_batch_idx > args.patience):
                break
    except KeyboardInterrupt:
        pass
    best_file = os.path.join(wandb.run.dir, "best.pth.tar")
    if os.path.exists(best_file):
        model, _ = utils.load_checkpoint(model, opt, args.device, best_file)
    return model


def test_loop(args, model, test_dataloader, max_num_batches):
    """am3.py:308-367: 11-tuple of averages + flattened per-query / per-support records (max_num_batches + 1 batches)."""
    m_acc, m_f1, m_prec, m_rec, m_loss, m_lam = (AverageMeter() for _ in range(6))
    test_preds, test_trues, query_idx, support_idx, support_lamdas = [], [], [], [], []
    for batch_idx, batch in enumerate(test_dataloader):
        with torch.no_grad():
            (loss, acc, f1, prec, rec, lamda, preds, trues, query, support, support_lamda) = model.evaluate(
                batch=batch, optimizer=None, scheduler=None, num_ways=args.num_ways, device=args.device, task="test")
        m_acc.update(acc); m_f1.update(f1); m_prec.update(prec); m_rec.update(rec); m_loss.update(loss); m_lam.update(lamda)
        test_preds += preds.tolist()
        test_trues += trues.tolist()
        query_idx += query.tolist()
        support_idx += support.tolist()
        support_lamdas += support_lamda.tolist()
        if batch_idx > max_num_batches - 1:
            break
    _engine.check_status(args.device)
    return (m_loss.avg, m_acc.avg, m_f1.avg, m_prec.avg, m_rec.avg, m_lam.avg, test_preds, test_trues, query_idx,
            support_idx, support_lamdas)
