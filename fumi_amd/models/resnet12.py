"""ResNet-12 image encoder for the ``im_net`` seam (fumi/models/fumi.py:89-100: "any module with forward(x, params) and
meta_named_parameters()"; ``--im_encoder resnet`` is a ``# TODO`` in the reference, fumi/models/am3.py:41-46).

BASELINE.json configs[4] words a configuration the reference never implements: FuMI 20-way 5-shot with a ResNet-12 backbone in
bf16, 5 inner steps, second-order outer gradients.  This module is the few-shot literature's ResNet-12 (TADAM / MetaOptNet form
without DropBlock): four residual blocks of channels (64, 160, 320, 640),

    a1 = lrelu(BN(conv3x3(x))),  a2 = lrelu(BN(conv3x3(a1))),  out = maxpool2(lrelu(BN(conv3x3(a2)) + BN(conv1x1(x)))),

LeakyReLU slope 0.1, batch statistics in training and evaluation (torchmeta's ``MetaBatchNorm2d(track_running_stats=False)``),
no conv bias, global average pool -> [rows, 640].  "Parity unpinned": the oracle is oracle/resnet12_ref.py / resnet12_manual.py.

Like ``Conv4`` these modules hold fp32 master parameters and names (``state_dict`` keys ``block{i}.conv{1,2,3}.weight``,
``block{i}.bn{1,2,3}.weight|bias``, ``block{i}.shortcut.weight``, ``block{i}.bns.weight|bias``); the inner-loop arithmetic runs in
the HIP engine in bf16 with fp32 accumulation (csrc/rn12*.hip), ``forward`` is an inference helper on the engine's feature pass."""
import torch
import torch.nn as nn

from .. import engine as _engine
from ..meta import MetaModule
from .conv4 import MetaBatchNorm2d, MetaConv2d

CHANNELS = (64, 160, 320, 640)


class ResBlock(MetaModule):
    def __init__(self, cin, c):
        super().__init__()
        bn = lambda: MetaBatchNorm2d(c, momentum=1.0, track_running_stats=False)
        self.conv1, self.bn1 = MetaConv2d(cin, c, 3, padding=1, bias=False), bn()
        self.conv2, self.bn2 = MetaConv2d(c, c, 3, padding=1, bias=False), bn()
        self.conv3, self.bn3 = MetaConv2d(c, c, 3, padding=1, bias=False), bn()
        self.shortcut, self.bns = MetaConv2d(cin, c, 1, bias=False), bn()

    def theta(self):
        return [self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
                self.conv3.weight, self.bn3.weight, self.bn3.bias, self.shortcut.weight, self.bns.weight, self.bns.bias]

    NAMES = ("conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias",
             "conv3.weight", "bn3.weight", "bn3.bias", "shortcut.weight", "bns.weight", "bns.bias")


class ResNet12(MetaModule):
    def __init__(self, in_channels=3, channels=CHANNELS, image_size=84):
        super().__init__()
        channels = tuple(int(c) for c in channels)
        if not 1 <= in_channels <= 8 or not 1 <= len(channels) <= 4 or any(c % 32 for c in channels):
            raise ValueError("ResNet12: 1-8 input channels, 1-4 blocks, channel counts in multiples of 32")
        if image_size >> len(channels) < 1:
            raise ValueError(f"{image_size} x {image_size} images are too small for {len(channels)} blocks")
        self.in_channels, self.channels, self.image_size = in_channels, channels, image_size
        self.n_blocks = len(channels)
        c = in_channels
        for i, co in enumerate(channels):
            self.add_module(f"block{i}", ResBlock(c, co))
            c = co
        self.feature_dim = channels[-1]

    def theta(self):
        """12 tensors per block, the order the engine takes them in."""
        out = []
        for i in range(self.n_blocks):
            out += getattr(self, f"block{i}").theta()
        return out

    def theta_names(self, prefix=""):
        return [f"{prefix}block{i}.{k}" for i in range(self.n_blocks) for k in ResBlock.NAMES]

    def forward(self, x, params=None):
        """Features [..., M, feature_dim] of image sets x [..., M, C, H, W]: every leading index is one set whose batch statistics
        are taken over its M images (a support or a query set).  params: OrderedDict keyed like theta_names()."""
        th = self.theta() if params is None else [params[k] for k in self.theta_names()]
        lead = x.shape[:-4]
        xs = x.reshape(-1, *x.shape[-4:]).contiguous().float()
        f = _engine.get_engine().resnet12_features(xs, [t.detach().contiguous() for t in th])
        return f.reshape(*lead, x.shape[-4], self.feature_dim)
