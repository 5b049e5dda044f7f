"""Decode driver of RRG_HF -- ref:vilmedic/blocks/huggingface/encoder_decoder/vision_evaluation.py and
vision_multi_evaluation.py (greedy / beam search from decoder_start_token_id over the encoder's hidden states)."""
import torch
import torch.nn as nn


def evaluation(models, config, dl, **kwargs):
    models = [m if not isinstance(m, nn.DataParallel) else m.module for m in models]
    model = models[0]
    vedm = model.model
    tokenizer = dl.dataset.seq.tokenizer if hasattr(dl.dataset, "seq") else dl.dataset.tokenizer
    max_len = getattr(dl.dataset.seq, "tokenizer_max_len", None) if hasattr(dl.dataset, "seq") else dl.dataset.tokenizer_max_len
    bos = vedm.config.decoder_start_token_id
    dcfg = vedm.decoder.config
    gen = dict(bos_token_id=bos, eos_token_id=dcfg.eos_token_id, pad_token_id=vedm.config.pad_token_id, max_length=max_len)
    if getattr(config, "length_penalty", None) is not None:
        gen["length_penalty"] = config.length_penalty
    if getattr(config, "beam_width", None) is not None:
        gen["num_beams"] = config.beam_width
    refs, hyps = [], []
    with torch.no_grad():
        for batch in dl:
            batch = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
            enc, enc_mask = model.encode(batch["images"], batch.get("images_mask"))
            B = enc.shape[0]
            out = vedm.decoder.generate(input_ids=torch.full((B, 1), bos, dtype=torch.long, device=enc.device),
                                        encoder_hidden_states=enc, encoder_attention_mask=enc_mask, **gen)
            for h, r in zip(out.tolist(), batch["input_ids"].tolist()):      # one device-to-host copy each, not one per row
                hyps.append(tokenizer.decode(h, skip_special_tokens=True, clean_up_tokenization_spaces=False))
                refs.append(tokenizer.decode(r, skip_special_tokens=True, clean_up_tokenization_spaces=False))
    return {"refs": refs, "hyps": hyps}
