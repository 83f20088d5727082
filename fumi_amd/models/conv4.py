"""Conv4 image encoder for the ``im_net`` seam (fumi/models/fumi.py:89-100: "any module with forward(x, params) and
meta_named_parameters()"; ``--im_encoder resnet`` is a ``# TODO`` in the reference, fumi/models/am3.py:41-46).

The reference adapts an MLP over pre-computed ResNet embeddings; BASELINE.json words its configurations with the standard
few-shot Conv4 on 84 x 84 images instead: four blocks of conv3x3(64, pad 1) . BatchNorm2d . ReLU . MaxPool2d(2), the layout of
torchmeta's MAML example (``MetaConv2d`` + ``MetaBatchNorm2d(momentum=1., track_running_stats=False)``: batch statistics in
training and evaluation).  Convolutions carry no bias: batch-statistic normalisation removes any per-channel constant, so a
bias would have no effect and an exactly zero gradient.

Like ``MetaLinear`` in ``fumi_amd/meta.py`` these modules hold parameters and names (``state_dict`` keys
``block{i}.conv.weight``, ``block{i}.norm.weight``, ``block{i}.norm.bias``); the inner-loop arithmetic runs in the HIP engine
(csrc/conv4.hip), ``forward`` is an inference helper on the engine's feature kernel."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import engine as _engine
from ..meta import MetaModule, MetaSequential


class MetaConv2d(nn.Conv2d, MetaModule):
    pass


class MetaBatchNorm2d(nn.BatchNorm2d, MetaModule):
    pass


class Conv4(MetaModule):
    def __init__(self, in_channels=3, hidden=64, n_blocks=4, image_size=84):
        super().__init__()
        if hidden != 64:
            raise NotImplementedError("the gfx950 convolution kernels are built for 64 channels per block")
        if not 1 <= in_channels <= 3 or not 1 <= n_blocks <= 4:
            raise ValueError("Conv4: 1-3 input channels and 1-4 blocks")
        self.in_channels, self.hidden, self.n_blocks, self.image_size = in_channels, hidden, n_blocks, image_size
        size, c = image_size, in_channels
        for i in range(n_blocks):
            if size < 2:
                raise ValueError(f"{image_size} x {image_size} images are too small for {n_blocks} blocks")
            self.add_module(f"block{i}", MetaSequential(OrderedDict(
                conv=MetaConv2d(c, hidden, 3, padding=1, bias=False),
                norm=MetaBatchNorm2d(hidden, momentum=1.0, track_running_stats=False),
                relu=nn.ReLU(), pool=nn.MaxPool2d(2))))
            size, c = size // 2, hidden
        self.feature_dim = hidden * size * size

    def theta(self):
        """[W_0, g_0, b_0, W_1, ...]: the order the engine takes them in."""
        out = []
        for i in range(self.n_blocks):
            blk = getattr(self, f"block{i}")
            out += [blk.conv.weight, blk.norm.weight, blk.norm.bias]
        return out

    def theta_names(self, prefix=""):
        return [f"{prefix}block{i}.{k}" for i in range(self.n_blocks) for k in ("conv.weight", "norm.weight", "norm.bias")]

    def forward(self, x, params=None):
        """Features [..., M, feature_dim] of image sets x [..., M, C, H, W]: every leading index is one set whose batch
        statistics are taken over its M images (a support or a query set).  params: OrderedDict keyed like theta_names()."""
        th = self.theta() if params is None else [params[k] for k in self.theta_names()]
        lead = x.shape[:-4]
        xs = x.reshape(-1, *x.shape[-4:]).contiguous().float()
        f = _engine.get_engine().conv4_features(xs, [t.detach().contiguous() for t in th])
        return f.reshape(*lead, x.shape[-4], self.feature_dim)
