"""FuMI: text-conditioned hypernetwork head + MAML inner loop -- host-side mirror of fumi/models/fumi.py.

Same class name, constructor keywords, attributes (``im_net``, ``hyper_net``, ``text_encoder``), ``state_dict`` keys
and method signatures as the reference (fumi/models/fumi.py:18-218); same ``training_run`` / ``test_loop``
(:220-326).  What differs is where the arithmetic happens: ``evaluate`` does not build an autograd graph, it hands
the parameter tensors of the whole meta-batch to the MI355X engine (one C-ABI call, csrc/api.hip) which returns the
query logits / predictions / losses and the second-order meta-gradients, then performs the (optionally sharded)
gradient reduction and the optimizer step.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import dist as fdist
from .. import lazy
from .. import engine as _engine
from ..flatgrad import FlatGrads, ParamWatch
from ..meta import MetaLinear, MetaSequential
from ..utils import utils as utils
from ..utils.average_meter import AverageMeter
from ..utils.hypernet_init import hyper_weight_layer_init
from ..utils.wandb_compat import wandb
from .common import RNN, RnnHid, WordEmbedding, _BiLstmEncoder

try:
    from tqdm import tqdm
except Exception:          # pragma: no cover
    def tqdm(x=None, **k):
        return x


class FUMI(nn.Module):
    def __init__(self, n_way=5, im_emb_dim=2048, im_hid_dim=[64], text_encoder="BERT", text_emb_dim=300,
                 text_hid_dim=1024, dropout_rate=0.0, dictionary=None, pooling_strat="mean", init_all_layers=False,
                 norm_hypernet=True, fine_tune=False, init_bias=False, im_encoder="precomputed", image_size=84,
                 image_channels=3):
        super().__init__()
        # im_encoder (additive): "precomputed" / "resnet" = the reference's MLP over embeddings (fumi.py:89-100);
        # "conv4" / "resnet12" = the Conv4 / bf16 ResNet-12 encoder on raw images at the same seam (fumi_amd/models/conv4.py,
        # resnet12.py), im_hid_dim is then unused
        if im_encoder not in ("precomputed", "resnet", "conv4", "resnet12"):
            raise NameError(f"{im_encoder} not allowed as image encoder")
        self.im_encoder = im_encoder
        self.n_way = n_way
        self.im_emb_dim = im_emb_dim
        self.im_hid_dim = list(im_hid_dim)
        self.text_encoder_type = text_encoder
        self.text_emb_dim = text_emb_dim
        self.text_hid_dim = text_hid_dim
        self.dropout_rate = dropout_rate
        self.dictionary = dictionary
        self.pooling_strat = pooling_strat
        self.norm_hypernet = norm_hypernet
        self.fine_tune = fine_tune
        self.init_bias = init_bias
        self.init_all_layers = init_all_layers

        # text encoder (fumi.py:47-63)
        if text_encoder in ("BERT", "precomputed"):
            self.text_encoder = nn.Identity()
        elif text_encoder in ("w2v", "glove"):
            self.text_encoder = WordEmbedding(text_encoder, pooling_strat, dictionary)
            self.text_emb_dim = self.text_encoder.embedding_dim
        elif text_encoder == "rand":
            self.text_encoder = nn.Linear(self.text_emb_dim, self.text_emb_dim)
        elif text_encoder == "RNN":
            self.text_encoder = RNN("glove", pooling_strat, dictionary, self.text_emb_dim)
        elif text_encoder == "RNNhid":
            self.text_encoder = RnnHid("glove", pooling_strat, dictionary, self.text_emb_dim)
        else:
            raise NameError(f"{text_encoder} not allowed as text encoder")
        if not fine_tune:
            for p in self.text_encoder.parameters():
                p.requires_grad = False
        if init_all_layers:
            raise NotImplementedError("Entire model hypernet initialisation removed")
        if len(self.im_hid_dim) < 1:
            raise IndexError("im_hid_dim needs at least one hidden layer (the reference indexes im_hid_dim[-1])")
        conv = None
        if im_encoder == "conv4":
            from .conv4 import Conv4
            conv = Conv4(image_channels, 64, 4, image_size)
            self.im_hid_dim = [conv.feature_dim]          # the head the hypernetwork emits is [N, feature_dim + 1]
        elif im_encoder == "resnet12":
            from .resnet12 import ResNet12
            conv = ResNet12(image_channels, image_size=image_size)
            self.im_hid_dim = [conv.feature_dim]

        # hypernetwork: Linear . ReLU . Linear(H+1) [. Tanh]  (fumi.py:70-86,104-107)
        head = nn.Linear(self.text_hid_dim, self.im_hid_dim[-1] + 1)
        if init_bias:
            head = hyper_weight_layer_init('relu', 'normc', self.text_hid_dim, self.im_hid_dim[-1] + 1, 1, False,
                                           adjust_weights=False, adjust_bias=True, use_film=False)(head)
        layers = [nn.Linear(self.text_emb_dim, self.text_hid_dim), nn.ReLU(), head]
        if norm_hypernet:
            layers.append(nn.Tanh())
        self.hyper_net = nn.Sequential(*layers)

        # adapted image network: (MetaLinear, ReLU[, Dropout])*  (fumi.py:89-100)
        im = OrderedDict()
        d = im_emb_dim
        for i, h in enumerate(self.im_hid_dim):
            im[f'linear{i}'] = MetaLinear(d, h)
            im[f'relu{i}'] = nn.ReLU()
            if dropout_rate > 0:
                im[f'dropout{i}'] = nn.Dropout(dropout_rate)
            d = h
        self.im_net = conv if conv is not None else MetaSequential(im)
        self._flat = None
        self._pcache = None

    # ---- parameter views handed to the engine --------------------------------------------------------------------
    def _theta(self):
        if self.im_encoder in ("conv4", "resnet12"):
            return self.im_net.theta()
        out = []
        for i in range(len(self.im_hid_dim)):
            lin = getattr(self.im_net, f'linear{i}')
            out += [lin.weight, lin.bias]
        return out

    def _phi(self):
        return [self.hyper_net[0].weight, self.hyper_net[0].bias, self.hyper_net[2].weight, self.hyper_net[2].bias]

    def _flat_grads(self):
        params = self._theta() + self._phi()
        if self._flat is None or not self._flat.matches(params):
            self._flat = FlatGrads(params, extra=2)
        return self._flat

    def _step_params(self, train):
        """(theta, phi, flat gradient buffer) as objects that stay the SAME from step to step: detached views of the parameters
        (they share storage, so optimizer updates show through) and the FlatGrads.  Rebuilt when a parameter object was
        replaced or moved (``_apply``: .to() / .cuda() / .float())."""
        c = self._pcache
        if c is None or not c[0].valid():
            theta, phi = self._theta(), self._phi()
            c = self._pcache = (ParamWatch(self, theta + phi), [p.detach() for p in theta], [p.detach() for p in phi])
            self._flat = None
        fg = None
        if train:
            fg = self._flat
            if fg is None:
                fg = self._flat_grads()
        return c[1], c[2], fg

    def _apply(self, fn, recurse=True):
        self._pcache = None
        return super()._apply(fn, recurse)

    # ---- reference surface ----------------------------------------------------------------------------------------
    def forward(self, text_embed):
        """Hyper-network forward (text -> [.., H+1] head rows), inference helper on the engine's linear op."""
        eng = _engine.get_engine()
        lead = text_embed.shape[:-1]
        x = text_embed.reshape(-1, text_embed.shape[-1]).contiguous()
        A0, a0, A1, a1 = [p.detach() for p in self._phi()]
        u = eng.linear(x, A0, a0, act=1)
        h = eng.linear(u, A1, a1, act=2 if self.norm_hypernet else 0)
        return h.reshape(*lead, h.shape[-1])

    def _encode_text(self, text, device):
        """text_encoder applied to a whole meta-batch [B,S,*] (fumi.py:199-204 per task)."""
        if self.text_encoder_type == "rand":
            B, S = text.shape[:2]
            return (2 * torch.rand(B, S, self.text_emb_dim) - 1).to(device)          # CPU RNG like the reference
        if self.text_encoder_type in ("BERT", "precomputed"):
            return text.to(torch.float32).contiguous()
        return self.text_encoder(text)

    def get_hyper_params(self, text, targets, device, attn_mask=None):
        """Per-class head rows [N, H+1] for ONE task (fumi.py:198-212)."""
        from .. import hip
        enc = self._encode_text(text.unsqueeze(0), device)
        c = hip.class_text_select(hip.Workspace.get(enc.device), enc.contiguous(), targets.unsqueeze(0).contiguous(), self.n_way)
        return self(c[0])

    def im_forward(self, im_embeds, im_params, hyper_params):
        """logits [rows, N] = im_net(x; params) @ h[:, :-1].T + h[:, -1]  (fumi.py:214-218); inference helper."""
        eng = _engine.get_engine()
        if self.im_encoder in ("conv4", "resnet12"):
            x = self.im_net(im_embeds, params=im_params)
            h = hyper_params.detach()
            return eng.linear(x.reshape(-1, x.shape[-1]).contiguous(), h[:, :-1].contiguous(), h[:, -1].contiguous(), act=0)
        x = im_embeds.contiguous()
        names = [f'linear{i}' for i in range(len(self.im_hid_dim))]
        for n in names:
            x = eng.linear(x, im_params[n + '.weight'].detach().contiguous(), im_params[n + '.bias'].detach().contiguous(), act=1)
        h = hyper_params.detach()
        return eng.linear(x, h[:, :-1].contiguous(), h[:, -1].contiguous(), act=0)

    def evaluate(self, args, batch, optimizer, task="train"):
        """One meta-batch (fumi.py:115-196).  Returns (loss np scalar, acc np scalar, preds float [B,Qn], targets)."""
        train = task == "train"
        if train:
            if not self.training:                # (nn.Module.train() walks every submodule: skip it when nothing changes)
                self.train()
        elif self.training:
            self.eval()
        # train-mode Dropout after each ReLU of im_net (fumi.py:93-99; CLI default --dropout 0.25): masks are drawn inside
        # the engine from a counter-based hash of a per-step seed taken from torch's CPU generator (so torch.manual_seed
        # reproduces a run); eval mode uses none, like nn.Dropout
        drop_p = float(self.dropout_rate) if (train and self.dropout_rate > 0) else 0.0
        drop_seed = 0
        if drop_p > 0:
            drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) + 7919 * fdist.world()[0]
        dev = args.device
        (_, s_text, s_im), s_y = batch['train']
        (_, q_text, q_im), q_y = batch['test']
        B = s_im.shape[0]
        lo, hi = fdist.shard(B)
        whole = lo == 0 and hi == B

        def to(t):                               # this rank's episodes on the device (a tensor that is already there is used as it is)
            if whole and t.device == dev and t.is_contiguous():
                return t
            return t[lo:hi].to(dev).contiguous()
        x_s, x_q, y_s, y_q = to(s_im), to(q_im), to(s_y), to(q_y)
        if x_s.dtype != torch.float32 or x_q.dtype != torch.float32:
            x_s, x_q = x_s.float(), x_q.float()
        T = args.num_train_adapt_steps if train else args.num_test_adapt_steps
        eng = _engine.get_engine()
        text_s = cls_text = None
        lstm_ft = train and isinstance(self.text_encoder, _BiLstmEncoder) and self.text_encoder.trainable()
        if isinstance(self.text_encoder, WordEmbedding):
            # only the N class rows of an episode are ever used (fumi.py:207-210): select, gather and pool them in one
            # kernel instead of pooling all S support rows first
            if self.pooling_strat not in ("mean", "max"):
                raise NameError(f"{self.pooling_strat} pooling strat not defined")
            # (with the MLP encoder the bag rides in the first launch of the step below instead of being a launch of its own)
            ride = {"defer": True} if (self.im_encoder not in ("conv4", "resnet12") and getattr(eng, "folds_optimizer_step", False)) else {}
            cls_text = eng.glove_bag_select(to(s_text), y_s, self.n_way, self.text_encoder.embed.weight.detach(),
                                            self.text_encoder.padding_token, self.pooling_strat, **ride)
        elif lstm_ft:
            # trainable bi-LSTM (--fine_tune, fumi.py:65-67): only the class rows carry a text adjoint (fumi.py:207-210), so only
            # they are encoded with a tape; the meta-step below returns d loss / d cls_text for the LSTM's backward
            tok_cls = eng.class_rows_select(to(s_text), y_s, self.n_way)
            cls_text, lstm_tape = self.text_encoder.forward_train(tok_cls)
            g_cls_text = torch.empty_like(cls_text)
            eng.want_text_grad(x_s.device, g_cls_text)
        else:
            text_s = self._encode_text(to(s_text), dev)

        theta, phi, fg = self._step_params(train)
        nth = len(theta)
        g_th, g_ph = fg.split(nth) if train else (None, None)
        # [.. grads .. | sum loss / B | sum acc / B], written by the engine -> one all-reduce(sum) -> global means everywhere
        tail = fg.tail if train else torch.empty(2, device=x_s.device, dtype=torch.float32)
        # One process: optimizer.step() (fumi.py:193) needs no launch of its own -- the step's last launch (the final reduction that
        # produces every gradient element) applies Adam's update right behind each element and publishes the two statistics; the
        # update and the publication are registered with the workspace BEFORE the step (csrc/gemm.hip: launch_reduce_multi_final).
        fold = (train and not lstm_ft and self.im_encoder not in ("conv4", "resnet12") and fdist.world()[1] == 1 and x_s.is_cuda
                and getattr(eng, "folds_optimizer_step", False) and hasattr(optimizer, "defer_step") and not lazy.SYNC
                and not lazy.USE_EVENT and optimizer.defer_step(x_s.device))
        if fold:
            loss, acc = lazy.scalars(tail, 2, defer=True)
        if self.im_encoder in ("conv4", "resnet12"):
            step = eng.fumi_conv4_step if self.im_encoder == "conv4" else eng.fumi_resnet12_step
            out = step(self.n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, args.step_size, self.norm_hypernet,
                       need_grad=train, grad_scale=1.0 / B, g_theta=g_th, g_phi=g_ph, cls_text=cls_text, stats=tail)
        else:
            out = eng.fumi_step(self.n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, args.step_size, self.norm_hypernet,
                                need_grad=train, grad_scale=1.0 / B,
                                g_theta=g_th, g_phi=g_ph, cls_text=cls_text, stats=tail, dropout_p=drop_p, seed=drop_seed)
        fdist.all_reduce_sum_(fg.flat if train else tail)
        if lstm_ft:                              # the encoder's .grad (this rank's episodes, then summed like every other gradient)
            for g in self.text_encoder.backward(tok_cls, lstm_tape, g_cls_text):
                fdist.all_reduce_sum_(g)
        # read back asynchronously (fumi.py:195 blocks here); in training the two stores ride on the optimizer's launch
        if not fold:
            loss, acc = lazy.scalars(tail, 2, defer=train)
        if train:
            fg.attach()                          # .grad of every parameter IS a view of the buffer the engine just filled
            if fold:
                optimizer.finish_deferred(x_s.device)              # (launches the update only if the step could not fold it)
            else:
                getattr(optimizer, "step_fused", optimizer.step)()     # (nothing is left for zero_grad() to clear, fumi.py:190-193)
            lazy.flush(x_s.device)
        preds = out["preds_f"]                                   # float, like the reference's test_preds (fumi.py:180-183)
        if fdist.world()[1] > 1 and not train:
            preds = fdist.all_gather_rows(preds)
        test_preds = preds if preds.shape[0] == B else None
        return loss, acc, test_preds, q_y.to(dev)


def training_run(args, model, optimizer, train_loader, val_loader, max_test_batches):
    """FuMI training loop (fumi.py:220-299): initial validation, per-batch logging, periodic validation +
    checkpoint, early stopping, best checkpoint reloaded at the end."""
    best_loss, best_acc, _, _ = test_loop(args, model, val_loader, max_test_batches)
    print(f"\ninitial loss: {best_loss}, acc: {best_acc}")
    best_batch_idx = 0
    opt, scheduler = optimizer if type(optimizer) == tuple else (optimizer, None)
    try:
        for batch_idx, batch in enumerate(train_loader):
            train_loss, train_acc, _, _ = model.evaluate(args=args, batch=batch, optimizer=opt, task="train")
            wandb.log({"train/acc": train_acc, "train/loss": train_loss,
                       "num_episodes": (batch_idx + 1) * args.batch_size}, step=batch_idx)
            if batch_idx % args.eval_freq == 0 and batch_idx != 0:
                val_loss, val_acc, _, _ = test_loop(args, model, val_loader, max_test_batches)
                is_best = val_loss < best_loss
                if is_best:
                    best_loss, best_batch_idx = val_loss, batch_idx
                wandb.log({"val/acc": val_acc, "val/loss": val_loss}, step=batch_idx)
                utils.save_checkpoint({"batch_idx": batch_idx, "state_dict": model.state_dict(), "best_loss": best_loss,
                                       "optimizer": opt.state_dict(), "args": vars(args)}, is_best)
                print(f"\nBatch {batch_idx + 1}/{args.epochs}: \ntrain/loss: {train_loss}, train/acc: {train_acc}"
                      f"\nval/loss: {val_loss}, val/acc: {val_acc}")
            # the reference's off-by-one is kept: epochs+1 batches are processed (fumi.py:288)
            if (batch_idx > args.epochs - 1) or (args.patience > 0 and batch_idx - best_batch_idx > args.patience):
                break
    except KeyboardInterrupt:
        pass
    best_file = os.path.join(wandb.run.dir, "best.pth.tar")
    if os.path.exists(best_file):
        model, _ = utils.load_checkpoint(model, opt, args.device, best_file)
    return model


def test_loop(args, model, test_loader, max_num_batches):
    """Validation / test loop (fumi.py:302-326).  Like the reference it consumes max_num_batches + 1 batches (the
    break is tested after the batch is processed)."""
    avg_test_acc, avg_test_loss = AverageMeter(), AverageMeter()
    test_preds, test_targets = [], []
    for batch_idx, batch in enumerate(test_loader):
        test_loss, test_acc, preds, target = model.evaluate(args=args, batch=batch, optimizer=None, task="test")
        avg_test_acc.update(test_acc)
        avg_test_loss.update(test_loss)
        test_preds.append(preds)
        test_targets.append(target)
        if batch_idx > max_num_batches - 1:
            break
    _engine.check_status(args.device)            # IndexError for an episode the reference would have refused (fumi.py:209)
    return avg_test_loss.avg, avg_test_acc.avg, test_preds, test_targets


def get_accuracy(logits, targets):
    """fumi.py:329-331"""
    _, predictions = torch.max(logits, dim=-1)
    return torch.mean(predictions.eq(targets).float())
