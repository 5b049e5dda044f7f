"""``backbone: _3d_densenet{121,169,201,264}`` -- MONAI's N-dimensional DenseNet (``monai.networks.nets.densenet``), which the reference imports under
those names (ref:vilmedic/blocks/vision/visual_encoder.py:8-13) and builds with ``eval(backbone)(pretrained=..., **kwargs)`` (kwargs: ``spatial_dims``,
``in_channels``, ``out_channels``, ...; :71).

PARITY UNPINNED against MONAI itself: MONAI is not installed in this image (no network), so the module tree below restates MONAI 1.x's published
architecture -- torchvision's DenseNet with Conv / BatchNorm / pooling of ``spatial_dims`` dimensions, each dense layer's six modules inside a
``layers`` Sequential, the head as ``class_layers`` (relu, adaptive average pool, flatten, ``out``) -- and what IS checked is the arithmetic: with
``spatial_dims=2`` the network equals the torchvision-named DenseNet of blocks/vision/cnn.py (itself pinned against torchvision's names and
outputs) under the obvious key mapping (tests/test_host_cpu.py).  No shipped YAML of the reference uses these backbones; they run as plain
torch modules (MIOpen), outside the hand-written path.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
_BN = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}
_MAXPOOL = {1: nn.MaxPool1d, 2: nn.MaxPool2d, 3: nn.MaxPool3d}
_AVGPOOL = {1: nn.AvgPool1d, 2: nn.AvgPool2d, 3: nn.AvgPool3d}
_ADAPT = {1: nn.AdaptiveAvgPool1d, 2: nn.AdaptiveAvgPool2d, 3: nn.AdaptiveAvgPool3d}
_DROPOUT = {1: nn.Dropout, 2: nn.Dropout2d, 3: nn.Dropout3d}


class _DenseLayer(nn.Module):
    def __init__(self, dims, cin, growth, bn_size, dropout_prob):
        super().__init__()
        mid = bn_size * growth
        self.layers = nn.Sequential(OrderedDict([
            ("norm1", _BN[dims](cin)), ("relu1", nn.ReLU(inplace=True)), ("conv1", _CONV[dims](cin, mid, kernel_size=1, bias=False)),
            ("norm2", _BN[dims](mid)), ("relu2", nn.ReLU(inplace=True)), ("conv2", _CONV[dims](mid, growth, kernel_size=3, padding=1, bias=False))]))
        if dropout_prob > 0:
            self.layers.add_module("dropout", _DROPOUT[dims](dropout_prob))

    def forward(self, x):
        return torch.cat([x, self.layers(x)], 1)


class DenseNetND(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, init_features=64, growth_rate=32, block_config=(6, 12, 24, 16), pretrained=False,
                 progress=True, bn_size=4, dropout_prob=0.0, **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError(f"3-D DenseNet: unsupported arguments {sorted(kwargs)} (act / norm variants of MONAI are not restated)")
        d = int(spatial_dims)
        self.features = nn.Sequential(OrderedDict([
            ("conv0", _CONV[d](in_channels, init_features, kernel_size=7, stride=2, padding=3, bias=False)), ("norm0", _BN[d](init_features)),
            ("relu0", nn.ReLU(inplace=True)), ("pool0", _MAXPOOL[d](kernel_size=3, stride=2, padding=1))]))
        nf = init_features
        for i, n in enumerate(block_config):
            block = nn.Sequential()
            for j in range(n):
                block.add_module(f"denselayer{j + 1}", _DenseLayer(d, nf + j * growth_rate, growth_rate, bn_size, dropout_prob))
            self.features.add_module(f"denseblock{i + 1}", block)
            nf += n * growth_rate
            if i == len(block_config) - 1:
                self.features.add_module("norm5", _BN[d](nf))
            else:
                self.features.add_module(f"transition{i + 1}", nn.Sequential(OrderedDict([
                    ("norm", _BN[d](nf)), ("relu", nn.ReLU(inplace=True)), ("conv", _CONV[d](nf, nf // 2, kernel_size=1, bias=False)),
                    ("pool", _AVGPOOL[d](kernel_size=2, stride=2))])))
                nf //= 2
        self.class_layers = nn.Sequential(OrderedDict([
            ("relu", nn.ReLU(inplace=True)), ("pool", _ADAPT[d](1)), ("flatten", nn.Flatten(1)), ("out", nn.Linear(nf, out_channels))]))
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
                nn.init.kaiming_normal_(torch.as_tensor(m.weight))
            elif isinstance(m, nn.Linear):
                nn.init.constant_(torch.as_tensor(m.bias), 0)

    def forward(self, x):
        return self.class_layers(self.features(x))


_BLOCKS = {"121": (64, 32, (6, 12, 24, 16)), "169": (64, 32, (6, 12, 32, 32)), "201": (64, 32, (6, 12, 48, 32)), "264": (64, 32, (6, 12, 64, 48))}


def build(backbone, output_layer, pretrained, **kwargs):
    """ref:visual_encoder.py:71-83 for the ``_3d_densenetNNN`` names: the network, cut after ``output_layer`` (``features`` / ``class_layers``)"""
    key = backbone.lower().replace("_3d_densenet", "")
    if key not in _BLOCKS:
        raise ValueError(f"unknown 3-D backbone {backbone!r}; available: {['_3d_densenet' + k for k in _BLOCKS]}")
    init, growth, blocks = _BLOCKS[key]
    kwargs = dict(kwargs)
    kwargs.setdefault("init_features", init), kwargs.setdefault("growth_rate", growth), kwargs.setdefault("block_config", blocks)
    network = DenseNetND(pretrained=False, **kwargs)          # (pretrained weights would need a download)
    if output_layer is not None and output_layer != "classifier":
        names = [n for n, _ in network.named_children()]
        assert output_layer in names, "{} not in {}".format(output_layer, names)
        sub = []
        for n, c in network.named_children():
            sub.append(c)
            if n == output_layer:
                break
        network = nn.Sequential(*sub)
    return network
