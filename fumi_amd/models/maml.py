"""Pure-image MAML baseline -- host-side mirror of fumi/models/maml.py.

``PureImageNetwork`` keeps the reference's constructor and ``state_dict`` keys (``net.lin_{i}.*``,
``net.lin_final.*``; maml.py:15-33); the free function ``evaluate`` keeps its signature and return value
(maml.py:134-193) but runs the whole meta-batch in one call of the MI355X engine (csrc/api.hip:fumi_hip_maml_step)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import dist as fdist
from .. import lazy
from .. import engine as _engine
from ..flatgrad import FlatGrads
from ..meta import MetaLinear, MetaModule, MetaSequential
from ..utils import utils as utils
from ..utils.average_meter import AverageMeter
from ..utils.wandb_compat import wandb


class PureImageNetwork(MetaModule):
    def __init__(self, im_embed_dim=2048, n_way=5, hidden_dims=None, im_encoder="precomputed", image_size=84,
                 image_channels=3):
        super().__init__()
        self.im_embed_dim = im_embed_dim
        self.n_way = n_way
        self.hidden_dims = list(hidden_dims) if hidden_dims is not None else None
        self.im_encoder = im_encoder              # additive: "conv4" = Conv4 on raw images in front of lin_final
        layers, d = OrderedDict(), im_embed_dim
        if im_encoder == "conv4":
            from .conv4 import Conv4
            self.hidden_dims = None
            layers['features'] = Conv4(image_channels, 64, 4, image_size)
            d = layers['features'].feature_dim
        elif im_encoder == "resnet12":
            from .resnet12 import ResNet12
            self.hidden_dims = None
            layers['features'] = ResNet12(image_channels, image_size=image_size)
            d = layers['features'].feature_dim
        for i, h in enumerate(self.hidden_dims or []):
            layers[f'lin_{i}'] = MetaLinear(d, h)
            layers[f'relu_{i}'] = nn.ReLU()
            d = h
        layers['lin_final'] = MetaLinear(d, n_way)
        self.net = MetaSequential(layers)
        self._flat = None

    def forward(self, inputs, params=None):
        return self.net(inputs, params=self.get_subdict(params, 'net'))

    def _params(self):
        if self.im_encoder in ("conv4", "resnet12"):
            return self.net.features.theta() + [self.net.lin_final.weight, self.net.lin_final.bias]
        out = []
        for i in range(len(self.hidden_dims or [])):
            lin = getattr(self.net, f'lin_{i}')
            out += [lin.weight, lin.bias]
        return out + [self.net.lin_final.weight, self.net.lin_final.bias]

    def _flat_grads(self):
        params = self._params()
        if self._flat is None or not self._flat.matches(params):
            self._flat = FlatGrads(params, extra=2)
        return self._flat


def evaluate(args, model, batch, optimizer, task="train"):
    """One meta-batch (maml.py:134-193): returns (loss np scalar, acc np scalar).  ``model.train()`` always, like the
    reference (:143); gradients and the optimizer step only for task == 'train' (:188-191)."""
    model.train()
    model.zero_grad()
    train = task == "train"
    dev = args.device
    s_im, s_y = batch['train'][0][2], batch['train'][1]
    q_im, q_y = batch['test'][0][2], batch['test'][1]
    B = s_im.shape[0]
    lo, hi = fdist.shard(B)
    to = lambda t: t[lo:hi].to(dev).contiguous()
    x_s, x_q, y_s, y_q = to(s_im).float(), to(q_im).float(), to(s_y), to(q_y)
    T = args.num_train_adapt_steps if train else args.num_test_adapt_steps
    fg = model._flat_grads() if train else None
    tail = fg.tail if train else torch.empty(2, device=x_s.device, dtype=torch.float32)
    enc = getattr(model, "im_encoder", "")
    eng = _engine.get_engine()
    step = eng.maml_conv4_step if enc == "conv4" else eng.maml_resnet12_step if enc == "resnet12" else eng.maml_step
    step(x_s, y_s, x_q, y_q, [p.detach() for p in model._params()], T, args.step_size,
         bool(args.first_order), need_grad=train, grad_scale=1.0 / B, g_params=fg.views if train else None, stats=tail)
    fdist.all_reduce_sum_(fg.flat if train else tail)
    if train:
        optimizer.zero_grad()
        fg.attach()
        optimizer.step()
    return lazy.scalars(tail, 2)


def training_run(args, model, optimizer, train_loader, val_loader, max_test_batches):
    """MAML training loop (maml.py:36-107); unlike FuMI's it does not reload the best checkpoint at the end."""
    best_loss, best_acc = test_loop(args, model, val_loader, max_test_batches)
    print(f"\ninitial loss: {best_loss}, acc: {best_acc}")
    best_batch_idx = 0
    try:
        for batch_idx, batch in enumerate(train_loader):
            train_loss, train_acc = evaluate(args=args, model=model, batch=batch, optimizer=optimizer, task="train")
            wandb.log({"train/acc": train_acc, "train/loss": train_loss,
                       "num_episodes": (batch_idx + 1) * args.batch_size}, step=batch_idx)
            if batch_idx % args.eval_freq == 0 and batch_idx != 0:
                val_loss, val_acc = test_loop(args, model, val_loader, max_test_batches)
                is_best = val_loss < best_loss
                if is_best:
                    best_loss, best_batch_idx = val_loss, batch_idx
                wandb.log({"val/acc": val_acc, "val/loss": val_loss}, step=batch_idx)
                utils.save_checkpoint({"batch_idx": batch_idx, "state_dict": model.state_dict(), "best_loss": best_loss,
                                       "optimizer": optimizer.state_dict(), "args": vars(args)}, is_best)
                print(f"\nBatch {batch_idx + 1}/{args.epochs}: \ntrain/loss: {train_loss}, train/acc: {train_acc}"
                      f"\nval/loss: {val_loss}, val/acc: {val_acc}")
            if (batch_idx > args.epochs - 1) or (args.patience > 0 and batch_idx - best_batch_idx > args.patience):
                break
    except KeyboardInterrupt:
        pass
    return model


def test_loop(args, model, test_loader, max_num_batches):
    """maml.py:110-131 (max_num_batches + 1 batches, like the reference)."""
    avg_test_acc, avg_test_loss = AverageMeter(), AverageMeter()
    for batch_idx, batch in enumerate(test_loader):
        test_loss, test_acc = evaluate(args=args, model=model, batch=batch, optimizer=None, task="test")
        avg_test_acc.update(test_acc)
        avg_test_loss.update(test_loss)
        if batch_idx > max_num_batches - 1:
            break
    _engine.check_status(args.device)            # labels outside [0, n_way) are an IndexError in the reference's cross_entropy
    return avg_test_loss.avg, avg_test_acc.avg


def get_accuracy(logits, targets):
    _, predictions = torch.max(logits, dim=-1)
    return torch.mean(predictions.eq(targets).float())
