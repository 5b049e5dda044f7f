"""RRG_HF -- ref:vilmedic/models/rrg/RRG_HF.py:19-177: the RRG math wired the way HF's VisionEncoderDecoderModel wires it.

Same constructor contract (``vision`` / ``decoder`` dicts with ``proto_model``, ``proto_config``, ``proto_config_args``),
same forward (single 4-D image batch -> encoder_attention_mask=None, RRG_HF.py:170; 5-D multi-image batch -> patch mask
from ``images_mask``, :143), same parameter names as ``VisionEncoderDecoderModel`` (``model.encoder.*`` incl. the ViT
pooler the HF class creates by default, ``model.decoder.bert.*`` / ``lm_head.*``, ``model.enc_to_dec_proj.*`` when the
hidden sizes differ, :137-140), returning ``vars(decoder_outputs)``.  Built on the HIP path: vilmedic_amd.nn.ViTModel +
BertGenerationDecoder.  Class lookup by HF mapping name is restricted to the architectures of the hot path
(``vit`` / ``deit`` encoders -- both shipped RRG_HF YAMLs name ``deit``, ref:config/RRG/baseline-HF.yml:22 -- and the
``bert-generation`` decoder).  Pretrained arguments -- ``encoderdecoder=<name>`` (RRG_HF.py:24-25), ``vision`` / ``decoder`` given as strings
(:48-49, :86-87) -- load from a local checkpoint directory (or a hub name already in the local HF cache) through
``blocks/huggingface/pretrained.py``; a name that is not on disk raises (the package never downloads)."""
import torch
import torch.nn as nn

from ... import ops
from ...arena import arena_of
from ...blocks.huggingface.decoder.bert_generation import BertGenerationDecoder, decoder_config
from ...blocks.huggingface.encoder_decoder.vision_evaluation import evaluation
from ...nn import VIT_DEFAULTS, Affine, BertPooler, ViTModel, make_config, Config
from ..utils import get_n_params


class _ViTWithPooler(ViTModel):
    """HF ``ViTModel(config)`` / ``DeiTModel(config)`` (add_pooling_layer=True): the pooler exists in the state dict but the
    encoder-decoder path only reads ``last_hidden_state``."""

    def __init__(self, cfg, distillation=False):
        super().__init__(cfg, distillation=distillation)
        self.pooler = BertPooler(cfg)


class _VisionEncoderDecoder(nn.Module):
    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.config = Config(decoder_start_token_id=None, pad_token_id=None, vocab_size=decoder.config.vocab_size)
        e, d = encoder.config.hidden_size, decoder.config.hidden_size
        if e != d and decoder.config.get("cross_attention_hidden_size") is None:
            self.enc_to_dec_proj = Affine(d, e, std=(1.0 / e) ** 0.5)


class RRG_HF(nn.Module):
    def __init__(self, encoderdecoder=None, decoder=None, vision=None, dl=None, **kwargs):
        super().__init__()
        assert (encoderdecoder is None) ^ (decoder is None or vision is None), \
            "Either proto should be provided, or both decoder and vision should be provided."
        from ...blocks.huggingface import pretrained
        if encoderdecoder is not None:                               # RRG_HF.py:24-25: VisionEncoderDecoderModel.from_pretrained
            self.model = pretrained.vision_encoder_decoder(str(encoderdecoder), _ViTWithPooler, _VisionEncoderDecoder)
            cfg = self.model.config
            if cfg.decoder_start_token_id is None:
                cfg.decoder_start_token_id = self.model.decoder.config.bos_token_id
            if cfg.pad_token_id is None:
                cfg.pad_token_id = self.model.decoder.config.pad_token_id
            assert self.model.decoder.config.is_decoder and self.model.decoder.config.add_cross_attention
            self.eval_func = evaluation
            return
        if isinstance(vision, str):                                  # RRG_HF.py:48-49: AutoModel.from_pretrained(vision)
            encoder = pretrained.auto_vision_model(vision, _ViTWithPooler)
        else:
            vision = dict(vision)
            assert "proto_model" in vision and "proto_config" in vision
            pm, pc = vision.pop("proto_model"), vision.pop("proto_config")
            if pm != pc or pm not in ("vit", "deit"):
                raise NotImplementedError("RRG_HF on the HIP path supports vision proto_model / proto_config 'vit' and 'deit'")
            enc_args = dict(vision.pop("proto_config_args", None) or {})
            encoder = _ViTWithPooler(make_config(VIT_DEFAULTS, enc_args), distillation=(pm == "deit"))
        if isinstance(decoder, str):                                 # RRG_HF.py:86-87: AutoModelForCausalLM.from_pretrained(decoder, add_cross_attention=True)
            dec = pretrained.auto_causal_lm(decoder)
        else:
            decoder = dict(decoder)
            assert "proto_model" in decoder and "proto_config" in decoder
            if decoder.pop("proto_model") != "bert-generation" or decoder.pop("proto_config") != "bert-generation":
                raise NotImplementedError("RRG_HF on the HIP path supports decoder proto_model / proto_config 'bert-generation'")
            dec_args = dict(decoder.pop("proto_config_args", None) or {})
            if dl:                                                   # RRG_HF.py:73-79
                tok = dl.dataset.seq.tokenizer
                dec_args.update(vocab_size=tok.vocab_size, unk_token_id=tok.unk_token_id, bos_token_id=tok.cls_token_id,
                                eos_token_id=tok.sep_token_id, pad_token_id=tok.pad_token_id)
            dec = BertGenerationDecoder(decoder_config(dec_args))
        self.model = _VisionEncoderDecoder(encoder, dec)
        if dl:                                                       # RRG_HF.py:96-100
            tok = dl.dataset.seq.tokenizer
            self.model.config.decoder_start_token_id = tok.cls_token_id
            self.model.config.pad_token_id = tok.pad_token_id
        else:
            self.model.config.decoder_start_token_id = self.model.decoder.config.bos_token_id
            self.model.config.pad_token_id = self.model.decoder.config.pad_token_id
        assert self.model.decoder.config.is_decoder and self.model.decoder.config.add_cross_attention
        self.eval_func = evaluation

    def encode(self, images, images_mask=None, **kwargs):
        """-> (encoder_hidden_states bf16 [B, S or N*S, D_dec], encoder_attention_mask or None)"""
        images = images.cuda()
        arena = arena_of(self)
        arena.refresh()
        mask = None
        if images.dim() == 5:                                        # RRG_HF.py:118-147
            B, N, C, H, W = images.shape
            m = torch.ones((B, N), dtype=torch.bool, device=images.device) if images_mask is None else images_mask.cuda().bool()
            flat = self.model.encoder(images.reshape(B * N, C, H, W))
            S, D = flat.shape[1], flat.shape[2]
            hidden = flat.reshape(B, N * S, D)
            mask = m.unsqueeze(-1).expand(B, N, S).reshape(B, N * S)
        elif images.dim() == 4:                                      # RRG_HF.py:160-172
            hidden = self.model.encoder(images)
        else:
            raise NotImplementedError(f"Unexpected images.dim() = {images.dim()}")
        if hasattr(self.model, "enc_to_dec_proj"):
            pj = self.model.enc_to_dec_proj
            hidden = ops.linear(hidden.contiguous(), arena.shadow(pj.weight), pj.bias, wgrad_buf=arena.grad(pj.weight),
                                bgrad_buf=arena.grad(pj.bias), anchor=pj.weight)
        return hidden, mask

    def forward(self, input_ids, attention_mask, images, images_mask=None, epoch=None, iteration=None, **kwargs):
        arena_of(self).refresh()
        hidden, mask = self.encode(images, images_mask)
        input_ids = input_ids.cuda()
        out = self.model.decoder(input_ids=input_ids, attention_mask=attention_mask.cuda(), encoder_hidden_states=hidden,
                                 encoder_attention_mask=mask, labels=input_ids, **kwargs)
        return vars(out)

    def __repr__(self):
        s = "model: RRG_HF\n"
        s += "(encoder):" + type(self.model.encoder).__name__ + "(" + str(dict(self.model.encoder.config)) + ")\n"
        s += "(decoder):" + type(self.model.decoder).__name__ + "(" + str(dict(self.model.decoder.config)) + ")\n"
        s += "{}\n".format(get_n_params(self))
        return s
