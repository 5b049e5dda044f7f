"""``backbone: hfpoolformer`` -- the architecture and the state-dict names of HuggingFace's PoolFormerModel, returning the last feature map (what the
reference reads as ``.last_hidden_state``: ref:vilmedic/blocks/vision/visual_encoder.py:67-69,192-194).

MetaFormer with a pooling token mixer: four stages of [strided patch-embedding convolution, ``depth`` blocks]; a block is
    x = x + s1 * (avgpool3x3(gn(x)) - gn(x))          (GroupNorm with ONE group; the pool excludes its padding from the count)
    x = x + s2 * conv1x1(act(conv1x1(gn(x))))          (hidden width mlp_ratio * channels)
with per-channel layer scales s1, s2 (``use_layer_scale``) and stochastic depth growing linearly over the blocks.  Like every CNN backbone
here the convolutions run through MIOpen (outside the north-star kernel list); no shipped YAML of the reference uses this
backbone.  Checked against the installed transformers PoolFormerModel in tests/test_host_cpu.py (names, feature map, input gradient).
"""
import torch
import torch.nn as nn

_ACT = {"gelu": nn.GELU, "relu": nn.ReLU, "silu": nn.SiLU, "swish": nn.SiLU, "tanh": nn.Tanh}


class _DropPath(nn.Module):
    """per-sample stochastic depth: a whole residual branch is dropped with probability p, the kept ones rescaled"""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        gate = torch.floor(torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device) + keep)
        return x.div(keep) * gate


class _Embeddings(nn.Module):
    def __init__(self, cin, cout, patch, stride, padding):
        super().__init__()
        self.projection = nn.Conv2d(cin, cout, kernel_size=patch, stride=stride, padding=padding)

    def forward(self, x):
        return self.projection(x)


class _Pooling(nn.Module):
    def __init__(self, pool_size):
        super().__init__()
        self.pool = nn.AvgPool2d(pool_size, stride=1, padding=pool_size // 2, count_include_pad=False)

    def forward(self, x):
        return self.pool(x) - x


class _Output(nn.Module):
    def __init__(self, channels, hidden, act, drop):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, hidden, 1)
        self.conv2 = nn.Conv2d(hidden, channels, 1)
        self.drop = _DropPath(drop)
        self.act_fn = _ACT[act]() if isinstance(act, str) else act

    def forward(self, x):
        return self.drop(self.conv2(self.drop(self.act_fn(self.conv1(x)))))


class _Layer(nn.Module):
    def __init__(self, channels, pool_size, hidden, act, drop_path, use_layer_scale, layer_scale_init_value):
        super().__init__()
        self.pooling = _Pooling(pool_size)
        self.output = _Output(channels, hidden, act, drop_path)
        self.before_norm = nn.GroupNorm(1, channels)
        self.after_norm = nn.GroupNorm(1, channels)
        self.drop_path = _DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.use_layer_scale = bool(use_layer_scale)
        if self.use_layer_scale:
            self.layer_scale_1 = nn.Parameter(layer_scale_init_value * torch.ones(channels))
            self.layer_scale_2 = nn.Parameter(layer_scale_init_value * torch.ones(channels))

    def forward(self, x):
        mix = self.pooling(self.before_norm(x))
        if self.use_layer_scale:
            mix = self.layer_scale_1[:, None, None] * mix
        x = x + self.drop_path(mix)
        mlp = self.output(self.after_norm(x))
        if self.use_layer_scale:
            mlp = self.layer_scale_2[:, None, None] * mlp
        return x + self.drop_path(mlp)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        n, depths, widths = c["num_encoder_blocks"], c["depths"], c["hidden_sizes"]
        total = sum(depths)
        rates = [c["drop_path_rate"] * i / max(total - 1, 1) for i in range(total)]          # linspace(0, drop_path_rate, total)
        self.patch_embeddings = nn.ModuleList(
            _Embeddings(c["num_channels"] if i == 0 else widths[i - 1], widths[i], c["patch_sizes"][i], c["strides"][i], c["padding"][i]) for i in range(n))
        blocks, cur = [], 0
        for i in range(n):
            blocks.append(nn.ModuleList(
                _Layer(widths[i], c["pool_size"], int(widths[i] * c["mlp_ratio"]), c["hidden_act"], rates[cur + j], c["use_layer_scale"],
                       c["layer_scale_init_value"]) for j in range(depths[i])))
            cur += depths[i]
        self.block = nn.ModuleList(blocks)

    def forward(self, x):
        for embed, layers in zip(self.patch_embeddings, self.block):
            x = embed(x)
            for layer in layers:
                x = layer(x)
        return x


_DEFAULTS = dict(num_channels=3, patch_size=16, stride=16, pool_size=3, mlp_ratio=4.0, depths=[2, 2, 6, 2], hidden_sizes=[64, 128, 320, 512],
                 patch_sizes=[7, 3, 3, 3], strides=[4, 2, 2, 2], padding=[2, 1, 1, 1], num_encoder_blocks=4, drop_path_rate=0.0, hidden_act="gelu",
                 use_layer_scale=True, layer_scale_init_value=1e-5, initializer_range=0.02)


class HFPoolFormerModel(nn.Module):
    """``VisualEncoder(backbone='hfpoolformer', **PoolFormerConfig kwargs)``: ``encoder.patch_embeddings.i.projection.*``,
    ``encoder.block.i.j.{before_norm,after_norm,output.conv1,output.conv2}.*``, ``encoder.block.i.j.layer_scale_{1,2}``"""

    def __init__(self, **kwargs):
        super().__init__()
        kwargs.pop("return_dict", None)
        unknown = sorted(set(kwargs) - set(_DEFAULTS))
        if unknown:
            raise TypeError(f"hfpoolformer: unknown PoolFormerConfig keys {unknown}")
        self.config = {**_DEFAULTS, **kwargs}
        self.encoder = _Encoder(self.config)
        std = self.config["initializer_range"]
        for m in self.modules():                      # HF's _init_weights: normal(0, initializer_range) convolutions, unit GroupNorms
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0.0, std=std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, pixel_values):
        return self.encoder(pixel_values)
