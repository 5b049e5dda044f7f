"""Transformer building blocks of the hot path, executed by libvmhip kernels.

Module trees and parameter names follow HF transformers 4.55.3 (the reference's pin, setup.py:29) so that
``state_dict()`` keys equal those of the modules the reference instantiates:
  ViTModel                     hf:models/vit/modeling_vit.py        (reached from ref:blocks/vision/visual_encoder.py:56-58)
  BertGenerationDecoder/Encoder hf:models/bert_generation/modeling_bert_generation.py (ref:blocks/huggingface/decoder/decoder_model.py:23-26)
  BertEncoder / BertPooler     hf:models/bert/modeling_bert.py      (ref:models/mvqa/MVQA.py:28-30)
Arithmetic is restated in oracle/torch_ref.py (the CPU checker); nothing here falls back to it.
"""
import math

import os

import torch
import torch.nn as nn

from . import ops
from .arena import arena_of


class Config(dict):
    """dict with attribute access (stands in for HF PretrainedConfig objects; YAML sub-trees map onto it)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return dict(self)


VIT_DEFAULTS = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                    initializer_range=0.02, layer_norm_eps=1e-12, image_size=224, patch_size=16, num_channels=3,
                    qkv_bias=True)
BERT_GEN_DEFAULTS = dict(vocab_size=50358, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                         intermediate_size=4096, hidden_act="gelu", hidden_dropout_prob=0.1,
                         attention_probs_dropout_prob=0.1, max_position_embeddings=512, initializer_range=0.02,
                         layer_norm_eps=1e-12, pad_token_id=0, bos_token_id=2, eos_token_id=1,
                         position_embedding_type="absolute", use_cache=True, is_decoder=False,
                         add_cross_attention=False)


def make_config(defaults, kwargs):
    cfg = Config(defaults)
    cfg.update({k: v for k, v in dict(kwargs).items()})
    if cfg.hidden_act not in ("gelu",):
        raise ValueError(f"hidden_act={cfg.hidden_act!r} unsupported by the HIP path (erf-GELU only)")
    dh = cfg.hidden_size // cfg.num_attention_heads
    if cfg.hidden_size % cfg.num_attention_heads or dh % 8 or dh > 128:
        raise ValueError("the HIP attention kernels need head_dim to be a multiple of 8, at most 128 (32 / 64 / 96 / 128 run natively, other "
                         f"widths zero-padded per head: ops._attention_padded_heads); got hidden_size={cfg.hidden_size}, heads={cfg.num_attention_heads}")
    return cfg


# ----------------------------------------------------------------------------- leaf parameter holders
class Affine(nn.Module):
    """weight+bias holder named like nn.Linear / nn.LayerNorm in the HF tree (compute happens in fused ops)."""

    def __init__(self, *wshape, bias=True, init="normal", std=0.02):
        super().__init__()
        w = torch.empty(*wshape)
        if init == "normal":
            w.normal_(0.0, std)
        elif init == "ones":
            w.fill_(1.0)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(wshape[0])) if bias else None


class _Holder(nn.Module):
    pass


def _linear(mod, arena, x, aff, **kw):
    return ops.linear(x, arena.shadow(aff.weight), aff.bias, wgrad_buf=arena.grad(aff.weight), bgrad_buf=arena.grad(aff.bias),
                      anchor=aff.weight, **kw)


KV_ALL = os.environ.get("VM_CROSS_KV_ALL", "1") != "0"     # 0: one K|V projection GEMM per cross-attention layer (A/B switch)
_LN_FORK = os.environ.get("VM_LN_FORK", "1") != "0"      # 0: leave the residual-fork gradient sums to autograd (A/B switch)


def _ln(arena, x, aff, eps, fork=None):
    if fork is not None and not _LN_FORK:
        y = ops.layer_norm(x, aff.weight, aff.bias, eps, arena.grad(aff.weight), arena.grad(aff.bias))
        return y, (x if fork == "in" else y)
    return ops.layer_norm(x, aff.weight, aff.bias, eps, arena.grad(aff.weight), arena.grad(aff.bias), fork)


# ----------------------------------------------------------------------------- ViT
class ViTSelfAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d, s = cfg.hidden_size, cfg.initializer_range
        self.query, self.key, self.value = Affine(d, d, std=s), Affine(d, d, std=s), Affine(d, d, std=s)

    def arena_groups(self):
        return [[self.query.weight, self.key.weight, self.value.weight], [self.query.bias, self.key.bias, self.value.bias]]


class ViTLayer(nn.Module):
    """pre-LN block (hf:models/vit/modeling_vit.py:255-290)."""

    def __init__(self, cfg):
        super().__init__()
        d, ff, s = cfg.hidden_size, cfg.intermediate_size, cfg.initializer_range
        self.cfg = cfg
        self.attention = _Holder()
        self.attention.attention = ViTSelfAttention(cfg)
        self.attention.output = _Holder()
        self.attention.output.dense = Affine(d, d, std=s)
        self.intermediate = _Holder()
        self.intermediate.dense = Affine(ff, d, std=s)
        self.output = _Holder()
        self.output.dense = Affine(d, ff, std=s)
        self.layernorm_before = Affine(d, init="ones")
        self.layernorm_after = Affine(d, init="ones")

    def forward(self, x, arena):
        cfg = self.cfg
        drop = cfg.hidden_dropout_prob if self.training else 0.0
        adrop = cfg.attention_probs_dropout_prob if self.training else 0.0
        sa = self.attention.attention
        wl, bl = [sa.query.weight, sa.key.weight, sa.value.weight], [sa.query.bias, sa.key.bias, sa.value.bias]
        # fork="in": the residual reads an alias of x returned by the LN op, so the two gradients of x meet inside the
        # LN backward kernel instead of in a separate elementwise add
        h, xr = _ln(arena, x, self.layernorm_before, cfg.layer_norm_eps, fork="in")
        qkv = ops.linear(h, arena.shadow_group(wl), arena.f32_group(bl), wgrad_buf=arena.grad_group(wl),
                         bgrad_buf=arena.grad_group(bl), anchor=sa.query.weight)
        ctx = ops.self_attention(qkv, None, cfg.num_attention_heads, False, adrop)
        x = _linear(self, arena, ctx, self.attention.output.dense, residual=xr, dropout_p=drop)
        h, xr = _ln(arena, x, self.layernorm_after, cfg.layer_norm_eps, fork="in")
        i, o = self.intermediate.dense, self.output.dense
        return ops.mlp(h, arena.shadow(i.weight), i.bias, arena.shadow(o.weight), o.bias, residual=xr, dropout_p=drop,
                       grads=(arena.grad(i.weight), arena.grad(i.bias), arena.grad(o.weight), arena.grad(o.bias)), anchor=i.weight)


class _ViTEmbeddings(nn.Module):
    def arena_groups(self):
        """DeiT: [CLS] and the distillation token adjacent in the arena -- one [2, D] operand of the assemble kernel"""
        if hasattr(self, "distillation_token"):
            return [[self.cls_token, self.distillation_token]]
        return []


class ViTModel(nn.Module):
    """HF ViTModel(add_pooling_layer=False): forward(images fp32 [B,C,H,W]) -> last_hidden_state bf16 [B,1+n,D].
    ``distillation=True``: HF DeiTModel (hf:models/deit/modeling_deit.py) -- the same pre-LN stack with a second special token
    (``embeddings.distillation_token``) and ``n + 2`` position embeddings (ref:vilmedic/blocks/vision/visual_encoder.py:59-61)."""

    def __init__(self, cfg, distillation=False):
        super().__init__()
        self.config = cfg
        d, p, c, s = cfg.hidden_size, cfg.patch_size, cfg.num_channels, cfg.initializer_range
        n = (cfg.image_size // p) ** 2
        self.n_special = 2 if distillation else 1
        self.embeddings = _ViTEmbeddings()
        if distillation:            # hf DeiTEmbeddings: zeros (not trunc_normal) for the three embedding parameters
            self.embeddings.cls_token = nn.Parameter(torch.zeros(1, 1, d))
            self.embeddings.distillation_token = nn.Parameter(torch.zeros(1, 1, d))
            self.embeddings.position_embeddings = nn.Parameter(torch.zeros(1, n + 2, d))
        else:
            self.embeddings.cls_token = nn.Parameter(torch.nn.init.trunc_normal_(torch.empty(1, 1, d), std=s))
            self.embeddings.position_embeddings = nn.Parameter(torch.nn.init.trunc_normal_(torch.empty(1, n + 1, d), std=s))
        self.embeddings.patch_embeddings = _Holder()
        self.embeddings.patch_embeddings.projection = Affine(d, c, p, p, std=s)
        self.encoder = _Holder()
        self.encoder.layer = nn.ModuleList([ViTLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.layernorm = Affine(d, init="ones")

    def forward(self, images):
        cfg = self.config
        arena = arena_of(self)
        arena.refresh()
        if images.shape[-1] != cfg.image_size or images.shape[-2] != cfg.image_size:
            raise ValueError(f"Input image size ({images.shape[-2]}*{images.shape[-1]}) doesn't match model "
                             f"({cfg.image_size}*{cfg.image_size}).")
        e = self.embeddings
        proj = e.patch_embeddings.projection
        images = images.contiguous().float()
        special = [e.cls_token, e.distillation_token] if self.n_special == 2 else [e.cls_token]
        x = ops.patch_embed(proj.weight, images, arena.shadow(proj.weight).view(cfg.hidden_size, -1), proj.bias,
                            arena.f32_group(special).view(-1), e.position_embeddings.view(-1, cfg.hidden_size), cfg.patch_size,
                            grads=(arena.grad(proj.weight), arena.grad(proj.bias), _flat(arena.grad_group(special)),
                                   _flat2(arena.grad(e.position_embeddings), cfg.hidden_size)), ns=self.n_special)
        for i, layer in enumerate(self.encoder.layer):
            if i:
                x = ops.backward_mark(x, ("enc_layer", i))      # data parallel: the gradients of layers >= i can be reduced from here on
            x = layer(x, arena)
        return _ln(arena, x, self.layernorm, cfg.layer_norm_eps)


def _flat(t):
    return t.view(-1) if t is not None else None


def _flat2(t, d):
    return t.view(-1, d) if t is not None else None


# ----------------------------------------------------------------------------- BERT blocks
class BertSelfAttentionParams(nn.Module):
    def __init__(self, cfg, kv_in=None, cross=False):
        super().__init__()
        d, s = cfg.hidden_size, cfg.initializer_range
        kv_in = kv_in or d
        self.query, self.key, self.value = Affine(d, d, std=s), Affine(d, kv_in, std=s), Affine(d, kv_in, std=s)
        self.fuse_q = kv_in == d and not cross
        self.cross = cross

    def arena_groups(self):
        if self.cross:        # the stack glues the K|V parameters of ALL its cross-attention layers (BertStack.arena_groups)
            return []
        if self.fuse_q:
            return [[self.query.weight, self.key.weight, self.value.weight], [self.query.bias, self.key.bias, self.value.bias]]
        return [[self.key.weight, self.value.weight], [self.key.bias, self.value.bias]]


class BertAttentionBlock(nn.Module):
    """HF Bert(Generation)Attention = .self (q,k,v) + .output (dense, LayerNorm)   (post-LN)."""

    def __init__(self, cfg, cross=False, kv_in=None):
        super().__init__()
        d, s = cfg.hidden_size, cfg.initializer_range
        self.cross = cross
        self.self = BertSelfAttentionParams(cfg, kv_in if cross else None, cross=cross)
        self.output = _Holder()
        self.output.dense = Affine(d, d, std=s)
        self.output.LayerNorm = Affine(d, init="ones")


class BertLayer(nn.Module):
    """hf:models/bert_generation/modeling_bert_generation.py:294-358 (decoder layer when add_cross_attention)."""

    def __init__(self, cfg, cross=False, enc_dim=None):
        super().__init__()
        d, ff, s = cfg.hidden_size, cfg.intermediate_size, cfg.initializer_range
        self.cfg = cfg
        self.attention = BertAttentionBlock(cfg)
        if cross:
            self.crossattention = BertAttentionBlock(cfg, cross=True, kv_in=enc_dim)
        self.intermediate = _Holder()
        self.intermediate.dense = Affine(ff, d, std=s)
        self.output = _Holder()
        self.output.dense = Affine(d, ff, std=s)
        self.output.LayerNorm = Affine(d, init="ones")

    # -- pieces shared by training forward and the cached decode step
    def _self_qkv(self, x, arena):
        sa = self.attention.self
        wl, bl = [sa.query.weight, sa.key.weight, sa.value.weight], [sa.query.bias, sa.key.bias, sa.value.bias]
        return ops.linear(x, arena.shadow_group(wl), arena.f32_group(bl), wgrad_buf=arena.grad_group(wl),
                          bgrad_buf=arena.grad_group(bl), anchor=sa.query.weight)

    def cross_kv(self, enc, arena):
        ca = self.crossattention.self
        wl, bl = [ca.key.weight, ca.value.weight], [ca.key.bias, ca.value.bias]
        return ops.linear(enc, arena.shadow_group(wl), arena.f32_group(bl), wgrad_buf=arena.grad_group(wl),
                          bgrad_buf=arena.grad_group(bl), anchor=ca.key.weight)

    # Every post-LN output feeds the next sub-layer AND that sub-layer's residual.  The LN op returns the pair
    # (y, alias of y) -- fork="out" -- so both gradients arrive at its backward and are summed inside the kernel.
    def _attn_out(self, blk, ctx, residual, arena, drop):
        s = _linear(self, arena, ctx, blk.output.dense, residual=residual, dropout_p=drop)
        return _ln(arena, s, blk.output.LayerNorm, self.cfg.layer_norm_eps, fork="out")

    def _ffn(self, x, xr, arena, drop):
        i, o = self.intermediate.dense, self.output.dense
        s = ops.mlp(x, arena.shadow(i.weight), i.bias, arena.shadow(o.weight), o.bias, residual=xr, dropout_p=drop,
                    grads=(arena.grad(i.weight), arena.grad(i.bias), arena.grad(o.weight), arena.grad(o.bias)), anchor=i.weight)
        return _ln(arena, s, self.output.LayerNorm, self.cfg.layer_norm_eps, fork="out")

    def forward(self, x, arena, self_mask, causal, enc=None, enc_mask=None, xr=None, kv=None, dkv_slot=None):
        """-> (y, alias of y); ``xr`` is the alias of x a previous layer returned (None: x itself); ``kv`` / ``dkv_slot``:
        this layer's view of the all-layer cross K|V projection and of its gradient buffer (BertStack.cross_kv_all)"""
        cfg = self.cfg
        drop = cfg.hidden_dropout_prob if self.training else 0.0
        adrop = cfg.attention_probs_dropout_prob if self.training else 0.0
        H = cfg.num_attention_heads
        ctx = ops.self_attention(self._self_qkv(x, arena), self_mask, H, causal, adrop)
        x, xr = self._attn_out(self.attention, ctx, x if xr is None else xr, arena, drop)
        if enc is not None:
            q = _linear(self, arena, x, self.crossattention.self.query)
            ctx = ops.cross_attention(q, kv if kv is not None else self.cross_kv(enc, arena), enc_mask, H, adrop, dkv_out=dkv_slot)
            x, xr = self._attn_out(self.crossattention, ctx, xr, arena, drop)
        return self._ffn(x, xr, arena, drop)


class BertStack(nn.Module):
    """``encoder.layer.{i}`` list (HF BertEncoder)."""

    def __init__(self, cfg, cross=False, enc_dim=None):
        super().__init__()
        self.config = cfg
        self.cross = bool(cross)
        self.layer = nn.ModuleList([BertLayer(cfg, cross, enc_dim) for _ in range(cfg.num_hidden_layers)])

    def arena_groups(self):
        """K|V weights (and biases) of every cross-attention layer adjacent in the arena: [k0 v0 k1 v1 ...]"""
        if not self.cross:
            return []
        ca = [l.crossattention.self for l in self.layer]
        return [[p for c in ca for p in (c.key.weight, c.value.weight)], [p for c in ca for p in (c.key.bias, c.value.bias)]]

    def cross_kv_all(self, enc, arena):
        """one GEMM for the K|V projections of all layers -> (per-layer K|V views, per-layer gradient slots)"""
        ca = [l.crossattention.self for l in self.layer]
        wl = [p for c in ca for p in (c.key.weight, c.value.weight)]
        bl = [p for c in ca for p in (c.key.bias, c.value.bias)]
        return ops.cross_kv_all(enc, arena.shadow_group(wl), arena.f32_group(bl), len(ca), wgrad_buf=arena.grad_group(wl),
                                bgrad_buf=arena.grad_group(bl), anchor=wl[0])

    def forward(self, x, arena, self_mask=None, causal=False, enc=None, enc_mask=None):
        xr = None
        kvs, slots = self.cross_kv_all(enc, arena) if (enc is not None and self.cross and KV_ALL) else (None, None)
        for i, layer in enumerate(self.layer):
            x, xr = layer(x, arena, self_mask, causal, enc, enc_mask, xr=xr, kv=kvs[i] if kvs else None,
                          dkv_slot=slots[i] if slots else None)
        return x


class BertEmbeddings(nn.Module):
    """word + absolute position embeddings -> LayerNorm -> dropout (hf:...bert_generation.py:394-426)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d, s = cfg.hidden_size, cfg.initializer_range
        w = torch.empty(cfg.vocab_size, d).normal_(0.0, s)
        if cfg.pad_token_id is not None and 0 <= cfg.pad_token_id < cfg.vocab_size:
            w[cfg.pad_token_id].zero_()
        self.word_embeddings = _Holder()
        self.word_embeddings.weight = nn.Parameter(w)
        self.position_embeddings = _Holder()
        self.position_embeddings.weight = nn.Parameter(torch.empty(cfg.max_position_embeddings, d).normal_(0.0, s))
        self.LayerNorm = Affine(d, init="ones")

    def arena_pads(self):
        # zero rows after the table so the tied LM-head dgrad can contract over the vocabulary padded to 8
        vp = (self.cfg.vocab_size + 7) // 8 * 8
        return {self.word_embeddings.weight: (vp - self.cfg.vocab_size) * self.cfg.hidden_size}

    def forward(self, ids, arena, past_len=0):
        cfg = self.cfg
        we, pe = self.word_embeddings.weight, self.position_embeddings.weight
        if ids.shape[1] + past_len > cfg.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        x = ops.embedding(we, ids, we, pe, past_len=past_len, padding_idx=cfg.pad_token_id,
                          g_word=arena.grad(we), g_pos=arena.grad(pe))
        x = _ln(arena, x, self.LayerNorm, cfg.layer_norm_eps)
        drop = cfg.hidden_dropout_prob if self.training else 0.0
        if drop > 0:
            x = DropoutFn.apply(x, drop)
        return x


BERT_DEFAULTS = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                     hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
                     type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0, position_embedding_type="absolute",
                     use_cache=True, is_decoder=False, add_cross_attention=False)
ROBERTA_DEFAULTS = dict(BERT_DEFAULTS, vocab_size=50265, pad_token_id=1, bos_token_id=0, eos_token_id=2)


class BertFullEmbeddings(BertEmbeddings):
    """word + token-type + position embeddings -> LayerNorm -> dropout: hf BertEmbeddings (hf:models/bert/modeling_bert.py) and, with
    ``roberta=True``, RobertaEmbeddings (hf:models/roberta/modeling_roberta.py:55-155): position ids = cumsum(ids != pad) * (ids != pad) + pad
    and a position table whose pad row gets no gradient.  The reference never passes ``token_type_ids`` (ref:.../encoder_model.py:44-56,
    decoder_model.py:42-47), so the type embedding is the constant row 0."""

    def __init__(self, cfg, roberta=False):
        super().__init__(cfg)
        self.roberta = bool(roberta)
        d, s = cfg.hidden_size, cfg.initializer_range
        self.token_type_embeddings = _Holder()
        self.token_type_embeddings.weight = nn.Parameter(torch.empty(cfg.type_vocab_size, d).normal_(0.0, s))
        if self.roberta and cfg.pad_token_id is not None and 0 <= cfg.pad_token_id < cfg.max_position_embeddings:
            with torch.no_grad():
                self.position_embeddings.weight[cfg.pad_token_id].zero_()

    def position_ids(self, ids, past_len=0):
        """RoBERTa only (create_position_ids_from_input_ids); BERT positions are the column index + past_len (kernel default)"""
        if not self.roberta:
            return None
        pad = self.cfg.pad_token_id
        mask = ids.ne(pad).to(torch.int64)
        return ((torch.cumsum(mask, dim=1) + past_len) * mask + pad).contiguous()

    def forward(self, ids, arena, past_len=0):
        cfg = self.cfg
        we, pe, te = self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight
        off = (cfg.pad_token_id + 1) if self.roberta else 0
        if ids.shape[1] + past_len + off > cfg.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        g_type = arena.grad(te)
        x = ops.embedding_ex(we, ids.contiguous(), self.position_ids(ids, past_len), we, pe, te, past_len=past_len, padding_idx=cfg.pad_token_id,
                             pos_offset=past_len + off, pos_pad=cfg.pad_token_id if self.roberta else -1,
                             g_word=arena.grad(we), g_pos=arena.grad(pe), g_type=g_type.view(-1) if g_type is not None else None)
        x = _ln(arena, x, self.LayerNorm, cfg.layer_norm_eps)
        drop = cfg.hidden_dropout_prob if self.training else 0.0
        if drop > 0:
            x = DropoutFn.apply(x, drop)
        return x


class DropoutFn(ops.Fn):
    @staticmethod
    def forward(ctx, x, p):
        seed = ops.next_seed()
        ctx.meta = (p, seed)
        return ops.dropout_apply(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        return ops.dropout_apply(dy.contiguous(), p, seed), None


class BertPooler(nn.Module):
    """tanh(W h[:,0] + b)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = Affine(cfg.hidden_size, cfg.hidden_size, std=cfg.initializer_range)

    def forward(self, hidden, arena=None):
        arena = arena or arena_of(self)
        first = hidden[:, 0].contiguous()
        return torch.tanh(_linear(self, arena, first, self.dense).float())


def to_key_mask(mask):
    """[B,S] bool / {0,1} int -> contiguous uint8 (1 = attend)."""
    if mask is None:
        return None
    return (mask != 0).to(torch.uint8).contiguous()
