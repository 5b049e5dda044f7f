"""VisualEncoder -- same constructor kwargs, ``forward`` / ``encode`` contract and state-dict prefixes
(``model.*``, ``visual_projection.*``) as ref:vilmedic/blocks/vision/visual_encoder.py:86-237.

ViT backbones run on the hand-written HIP path (vilmedic_amd.nn.ViTModel).  CNN backbones are torch modules
executed by MIOpen through PyTorch-ROCm (not in the north-star kernel list)."""
import json

import os

import torch
import torch.nn as nn

from ... import ops
from ...arena import arena_of
from ...nn import VIT_DEFAULTS, ViTModel, make_config, Affine
from . import cnn as _cnn


def get_network(backbone, output_layer, pretrained, **kwargs):
    """ref: visual_encoder.py:43-83."""
    if "vit" in backbone.lower():
        kwargs = {k: v for k, v in kwargs.items() if k not in ("attn_implementation", "return_dict")}
        return ViTModel(make_config(VIT_DEFAULTS, kwargs))
    if "deit" in backbone.lower():                # visual_encoder.py:59-61: DeiTModel(DeiTConfig(**kwargs), add_pooling_layer=False)
        kwargs = {k: v for k, v in kwargs.items() if k not in ("attn_implementation", "return_dict")}
        return ViTModel(make_config(VIT_DEFAULTS, kwargs), distillation=True)
    if "hfpoolformer" in backbone.lower():        # visual_encoder.py:67-69: HFPoolFormerModel(PoolFormerConfig(**kwargs))
        from .poolformer import HFPoolFormerModel
        return HFPoolFormerModel(**kwargs)
    if "3d" in backbone.lower():                  # MONAI's N-d DenseNets (visual_encoder.py:8-13,71): restated, parity unpinned (blocks/vision/densenet3d.py)
        from .densenet3d import build as _build3d
        return _build3d(backbone, output_layer, pretrained, **kwargs)
    return _cnn.build(backbone, output_layer, pretrained, **kwargs)


CNN_AMP = os.environ.get("VM_CNN_AMP", "") == "bf16"
# the CNN towers take their images channels-last (NHWC): MIOpen's convolutions then run their NHWC kernels and every BatchNorm (+ residual add
# + ReLU) runs on the hand-written kernel of csrc/batchnorm.hip (blocks/vision/micro_bn.py) -- in fp32 by default, in bf16 under VM_CNN_AMP.
# VM_CNN_LAYOUT=nchw restores the NCHW / torch-BatchNorm path (A/B switch)
CNN_NHWC = os.environ.get("VM_CNN_LAYOUT", "nhwc").lower() != "nchw"

class VisualEncoder(nn.Module):
    def __init__(self, backbone, permute, dropout_out=0.0, freeze=False, output_layer=None, pretrained=True,
                 slice_encode=None, slice_dim=None, visual_projection=None, **kwargs):
        super().__init__()
        self.backbone = backbone
        self.output_layer = output_layer
        self.permute = permute
        self.freeze = freeze
        self.pretrained = pretrained
        self.model = get_network(self.backbone, self.output_layer, self.pretrained, **kwargs)
        self.dropout_out = nn.Dropout(p=dropout_out)
        self.is3D = "3d" in backbone
        self.slice_encode = slice_encode
        self.slice_dim = slice_dim
        if self.slice_encode and self.output_layer == "features":
            raise Exception("If encoding per slices, output of forward pass should be a vector. Try avgpool or features as output_layer parameter.")
        if self.slice_encode and self.slice_dim is None:
            raise Exception("slice_encode is True but slice_dim is None, please specify slice_dim")
        if visual_projection:
            vp = dict(visual_projection)
            self.visual_projection = Affine(vp["out_features"], vp["in_features"], std=(1.0 / vp["in_features"]) ** 0.5)
        else:
            self.visual_projection = nn.Identity()
        assert permute in ["batch_first", "spatial_first", "no_permute"]
        if freeze:  # the reference touches a stale attribute here; intended behaviour: freeze the backbone
            for _, param in self.model.named_parameters():
                param.requires_grad = False

    # ------------------------------------------------------------------ encode (visual_encoder.py:130-178)
    def encode(self, images, images_mask=None, **kwargs):
        images = images.cuda()
        images_mask = images_mask.cuda() if images_mask is not None else images_mask
        if images.dim() == 4:
            features = self(images)
            return self._mask_and_project(features)
        assert images.dim() == 5, "wrong images shape"
        if self.is3D:                              # visual_encoder.py:144-157: per-slice encoding (stacked vectors) or the full volume
            if self.slice_encode:
                features = torch.stack([self(images.narrow(self.slice_dim, i, 1).squeeze(self.slice_dim)) for i in range(images.size(self.slice_dim))], dim=1)
            else:
                features = self(images)
            return self._mask_and_project(features)
        B, N = images.shape[:2]
        features = self(images.reshape(B * N, *images.shape[2:]))
        if features.dim() <= 2:
            raise Exception("The input size is too small for this model. The spatial dim has been shrunk to 1.")
        features = features.view(B, N, features.shape[-2], features.shape[-1])
        if images_mask is not None:
            features = features * images_mask.unsqueeze(-1).unsqueeze(-1).to(features.dtype)
        features = features.reshape(B, N * features.shape[2], features.shape[3])
        return self._mask_and_project(features)

    def _mask_and_project(self, features):
        if features.dtype != torch.bfloat16:
            features = features.to(torch.bfloat16)
        features = features.contiguous()
        shape = features.shape
        mask = ops.feature_mask(features.view(-1, shape[-1])).view(shape[:-1]).bool()
        if isinstance(self.visual_projection, nn.Identity):
            return features, mask
        arena = arena_of(self)
        arena.refresh()
        vp = self.visual_projection
        out = ops.linear(features, arena.shadow(vp.weight), vp.bias, wgrad_buf=arena.grad(vp.weight),
                         bgrad_buf=arena.grad(vp.bias), anchor=vp.weight)
        return out, mask

    # ------------------------------------------------------------------ forward (visual_encoder.py:180-208)
    def forward(self, images, **kwargs):
        arena_of(self)   # root the arena here so it also covers visual_projection (sub-module forwards reuse it)
        if isinstance(self.model, ViTModel):
            out = self.model(images)
            return self._dropout_out(out)
        # CNN backbones are torch modules on MIOpen.  VM_CNN_AMP=bf16: channels-last bf16 convolutions under autocast with
        # fp32 master weights and fp32 BatchNorm statistics -- the counterpart of the reference's use_amp (fp16 autocast) training mode
        if images.dim() == 4 and (CNN_AMP or CNN_NHWC):
            images = images.contiguous(memory_format=torch.channels_last)
        if CNN_AMP:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = self.model(images)
            out = out.float()
        else:
            out = self.model(images)
        out = self._dropout_out(out)
        if self.permute == "no_permute":
            pass
        elif self.permute == "batch_first":
            out = out.view(*out.size()[:2], -1).permute(0, 2, 1)
            if out.shape[1] == 1:
                out = out.squeeze(1)
        elif self.permute == "spatial_first":
            out = out.view(*out.size()[:2], -1).permute(2, 0, 1)
        else:
            raise NotImplementedError()
        return out

    def _dropout_out(self, x):
        if self.training and self.dropout_out.p > 0:
            from ...nn import DropoutFn
            if x.dtype == torch.bfloat16:
                return DropoutFn.apply(x, self.dropout_out.p)
            return self.dropout_out(x)
        return x

    def train(self, mode: bool = True):
        if self.freeze:
            mode = False
        self.training = mode
        for module in self.children():
            module.train(mode)
        return self

    def __repr__(self):
        is_vit = isinstance(self.model, ViTModel)
        repr_dict = {
            "type": type(self.model).__name__ if is_vit else None,
            "config": str(dict(self.model.config)) if is_vit else None,
            "dropout_out": self.dropout_out.p,
            "freeze": self.freeze,
            "output_layer": str(self.output_layer) if self.output_layer is not None else None,
            "pretrained": self.pretrained if not is_vit else None,
            "visual_projection": str(self.visual_projection),
        }
        repr_dict = {k: v for k, v in repr_dict.items() if v is not None}
        return f"{self.backbone}:\n{json.dumps(repr_dict, indent=2)}"
