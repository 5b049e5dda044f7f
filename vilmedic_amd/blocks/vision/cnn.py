"""CNN backbones for VisualEncoder.  torchvision is not installed in this image, so the architectures the shipped
YAMLs name (resnet18/34/50/101, densenet121/169) are declared here with torchvision's module / parameter names
(``conv1, bn1, layer1.0.conv1 ...``, ``features.denseblock1.denselayer1.norm1 ...``) so reference checkpoints load;
``hfresnet`` is the HuggingFace-style ResNet (HF names, ResNetConfig kwargs).
Their convolutions run through MIOpen via PyTorch-ROCm (CNN stems are NOT hand-written kernels); every BatchNorm is a
``micro_bn.MicroBatchNorm2d`` (same parameters and state-dict keys), which on channels-last activations runs the hand-written
BatchNorm (+ residual add + ReLU) kernel of csrc/batchnorm.hip -- ``bn_act`` below is how the blocks hand it the add and the ReLU.
ref: vilmedic/blocks/vision/visual_encoder.py:71-81 (eval(backbone)(pretrained=...) truncated at output_layer)."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .micro_bn import bn_act, dense_block_forward, dense_block_ok, use_micro_batch_norm


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = bn_act(self.bn1, self.conv1(x), self.relu)
        return bn_act(self.bn2, self.conv2(out), self.relu, residual=idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = bn_act(self.bn1, self.conv1(x), self.relu)
        out = bn_act(self.bn2, self.conv2(out), self.relu)
        return bn_act(self.bn3, self.conv3(out), self.relu, residual=idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0])
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, block, planes, n, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(bn_act(self.bn1, self.conv1(x), self.relu))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class _DenseLayer(nn.Module):
    def __init__(self, inp, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(inp)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(inp, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, 1, 1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1) if isinstance(feats, (list, tuple)) else feats
        x = self.conv1(bn_act(self.norm1, x, self.relu1))
        return self.conv2(bn_act(self.norm2, x, self.relu2))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n, inp, bn_size, growth):
        super().__init__()
        for i in range(n):
            self.add_module(f"denselayer{i + 1}", _DenseLayer(inp + i * growth, growth, bn_size))

    def forward(self, x):
        # [r6] one feature buffer for the block (blocks/vision/micro_bn.py: each layer's norm1 reads the first c channels of it, the backward pass
        # accumulates onto one gradient buffer): MVQA's DenseNet-169 step ran 82 torch.cat copies of the growing tensor and as many gradient adds
        layers = list(self.values())
        if dense_block_ok(layers, x):
            return dense_block_forward(layers, x)
        # fallback, running concatenation: every layer consumes ONE tensor (the block's features so far) and the next one is cat(that, its 32 new
        # channels) -- the same bytes copied per layer as torchvision's cat(list) form, the same values, but in the backward pass each running
        # tensor has exactly two consumers (its layer and the next cat): one gradient add per layer instead of one per (layer, earlier feature) pair
        for layer in layers:
            x = torch.cat([x, layer(x)], 1)
        return x


class _Transition(nn.Sequential):
    """norm -> relu -> 1x1 conv -> 2x2 average pool (torchvision's _Transition: same child names); BatchNorm + ReLU as one kernel"""

    def forward(self, x):
        return self.pool(self.conv(bn_act(self.norm, x, self.relu)))


class DenseNet(nn.Module):
    def __init__(self, growth=32, blocks=(6, 12, 24, 16), init_feat=64, bn_size=4, num_classes=1000):
        super().__init__()
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv2d(3, init_feat, 7, 2, 3, bias=False)), ("norm0", nn.BatchNorm2d(init_feat)),
            ("relu0", nn.ReLU(inplace=True)), ("pool0", nn.MaxPool2d(3, 2, 1))]))
        nf = init_feat
        for i, n in enumerate(blocks):
            self.features.add_module(f"denseblock{i + 1}", _DenseBlock(n, nf, bn_size, growth))
            nf += n * growth
            if i != len(blocks) - 1:
                self.features.add_module(f"transition{i + 1}", _Transition(OrderedDict([
                    ("norm", nn.BatchNorm2d(nf)), ("relu", nn.ReLU(inplace=True)),
                    ("conv", nn.Conv2d(nf, nf // 2, 1, bias=False)), ("pool", nn.AvgPool2d(2, 2))])))
                nf //= 2
        self.features.add_module("norm5", nn.BatchNorm2d(nf))
        self.classifier = nn.Linear(nf, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        out = F.relu(self.features(x), inplace=True)
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(out, (1, 1)), 1))


_FACTORY = {
    "resnet18": lambda: ResNet(BasicBlock, [2, 2, 2, 2]), "resnet34": lambda: ResNet(BasicBlock, [3, 4, 6, 3]),
    "resnet50": lambda: ResNet(Bottleneck, [3, 4, 6, 3]), "resnet101": lambda: ResNet(Bottleneck, [3, 4, 23, 3]),
    "densenet121": lambda: DenseNet(32, (6, 12, 24, 16), 64), "densenet169": lambda: DenseNet(32, (6, 12, 32, 32), 64),
    "densenet201": lambda: DenseNet(32, (6, 12, 48, 32), 64),
}


# ----------------------------------------------------------------------------- HuggingFace-style ResNet (``backbone: hfresnet``)
_ACT = {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU, "swish": nn.SiLU, "tanh": nn.Tanh}


class _HFConvLayer(nn.Module):
    """convolution (no bias, ``same``-style padding k // 2) -> BatchNorm -> activation, HF parameter names"""

    def __init__(self, cin, cout, kernel_size=3, stride=1, activation="relu"):
        super().__init__()
        self.convolution = nn.Conv2d(cin, cout, kernel_size, stride, kernel_size // 2, bias=False)
        self.normalization = nn.BatchNorm2d(cout)
        self.activation = _ACT[activation]() if activation is not None else nn.Identity()

    def forward(self, x):
        if isinstance(self.activation, (nn.ReLU, nn.Identity)):
            return bn_act(self.normalization, self.convolution(x), self.activation if isinstance(self.activation, nn.ReLU) else None)
        return self.activation(self.normalization(self.convolution(x)))


class _HFShortCut(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.convolution = nn.Conv2d(cin, cout, 1, stride, bias=False)
        self.normalization = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.normalization(self.convolution(x))


class _HFResLayer(nn.Module):
    """basic (two 3x3) or bottleneck (1x1 -> 3x3 -> 1x1, reduction 4) residual layer; the stride sits on the first 3x3 unless
    ``downsample_in_bottleneck`` puts it on the first 1x1"""

    def __init__(self, cin, cout, stride, activation, bottleneck, downsample_in_bottleneck):
        super().__init__()
        self.shortcut = _HFShortCut(cin, cout, stride) if (cin != cout or stride != 1) else nn.Identity()
        if bottleneck:
            mid = cout // 4
            self.layer = nn.Sequential(
                _HFConvLayer(cin, mid, 1, stride if downsample_in_bottleneck else 1, activation),
                _HFConvLayer(mid, mid, 3, 1 if downsample_in_bottleneck else stride, activation),
                _HFConvLayer(mid, cout, 1, 1, None))
        else:
            self.layer = nn.Sequential(_HFConvLayer(cin, cout, 3, stride, activation), _HFConvLayer(cout, cout, 3, 1, None))
        self.activation = _ACT[activation]()

    def forward(self, x):
        last = self.layer[-1]
        if isinstance(self.activation, nn.ReLU) and isinstance(last.activation, nn.Identity):       # relu(bn(conv(h)) + shortcut) in the BatchNorm kernel
            h = x
            for m in list(self.layer)[:-1]:
                h = m(h)
            return bn_act(last.normalization, last.convolution(h), self.activation, residual=self.shortcut(x))
        return self.activation(self.layer(x) + self.shortcut(x))


class _HFStage(nn.Module):
    def __init__(self, cin, cout, stride, depth, **kw):
        super().__init__()
        self.layers = nn.Sequential(_HFResLayer(cin, cout, stride, **kw), *[_HFResLayer(cout, cout, 1, **kw) for _ in range(depth - 1)])

    def forward(self, x):
        return self.layers(x)


class _HFEmbeddings(nn.Module):
    def __init__(self, num_channels, embedding_size, activation):
        super().__init__()
        self.embedder = _HFConvLayer(num_channels, embedding_size, 7, 2, activation)
        self.pooler = nn.MaxPool2d(3, 2, 1)

    def forward(self, x):
        return self.pooler(self.embedder(x))


class _HFEncoder(nn.Module):
    def __init__(self, embedding_size, hidden_sizes, depths, downsample_in_first_stage, **kw):
        super().__init__()
        sizes = [embedding_size] + list(hidden_sizes)
        self.stages = nn.ModuleList([_HFStage(sizes[i], sizes[i + 1], 2 if (i > 0 or downsample_in_first_stage) else 1, depths[i], **kw)
                                     for i in range(len(hidden_sizes))])

    def forward(self, x):
        for stage in self.stages:
            x = stage(x)
        return x


class HFResNetModel(nn.Module):
    """``VisualEncoder(backbone='hfresnet', **ResNetConfig kwargs)`` (ref: visual_encoder.py:63-65): the architecture and the
    state-dict names of HuggingFace's ResNetModel (``embedder.embedder.convolution.weight``,
    ``encoder.stages.i.layers.j.{shortcut,layer.k}.{convolution,normalization}.*``), returning the last feature map -- what the
    reference reads as ``.last_hidden_state`` (visual_encoder.py:188-190).  Convolutions run through MIOpen like every CNN
    backbone here.  Checked against the installed transformers ResNetModel in tests/test_host_cpu.py."""

    def __init__(self, num_channels=3, embedding_size=64, hidden_sizes=(256, 512, 1024, 2048), depths=(3, 4, 6, 3),
                 layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False, **kwargs):
        super().__init__()
        if layer_type not in ("basic", "bottleneck"):
            raise ValueError(f"layer_type={layer_type} is not one of basic, bottleneck")
        self.embedder = _HFEmbeddings(num_channels, embedding_size, hidden_act)
        self.encoder = _HFEncoder(embedding_size, hidden_sizes, depths, downsample_in_first_stage, activation=hidden_act,
                                  bottleneck=layer_type == "bottleneck", downsample_in_bottleneck=downsample_in_bottleneck)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, pixel_values):
        return self.encoder(self.embedder(pixel_values))


def build(backbone, output_layer, pretrained, **kwargs):
    return use_micro_batch_norm(_build(backbone, output_layer, pretrained, **kwargs))


def _build(backbone, output_layer, pretrained, **kwargs):
    if "hfresnet" in backbone.lower():
        kwargs.pop("return_dict", None)
        return HFResNetModel(**kwargs)
    if "densenet" in backbone and output_layer == "avgpool":          # visual_encoder.py:48-53
        sub = _build(backbone, "features", pretrained, **kwargs)
        sub.add_module("relu", nn.ReLU(inplace=True))
        sub.add_module("avgpool", nn.AdaptiveAvgPool2d((1, 1)))
        sub.add_module("flatten", nn.Flatten(1))
        return sub
    if backbone not in _FACTORY:
        raise ValueError(f"unknown backbone {backbone!r}; available: {sorted(_FACTORY)} + 'vit'")
    network = _FACTORY[backbone]()   # `pretrained` weights would need a download: random init (no network here)
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
