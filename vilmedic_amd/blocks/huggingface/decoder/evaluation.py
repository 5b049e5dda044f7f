"""RRG decode driver -- ref:vilmedic/blocks/huggingface/decoder/evaluation.py:20-85."""
import torch
import torch.nn as nn


def get_special_token_ids(model, tokenizer):
    bos_token_id = model.config.bos_token_id
    eos_token_id = model.config.eos_token_id
    pad_token_id = model.config.pad_token_id
    if None in [bos_token_id, eos_token_id, pad_token_id]:
        bos_token_id = tokenizer.vocab[tokenizer.cls_token]
        eos_token_id = tokenizer.vocab[tokenizer.sep_token]
        pad_token_id = tokenizer.vocab[tokenizer.pad_token]
    return bos_token_id, eos_token_id, pad_token_id


def evaluation(models, config, dl, **kwargs):
    models = [m if not isinstance(m, nn.DataParallel) else m.module for m in models]
    hf_model = models[0].dec.decoder
    try:
        ref_str = "input_ids"
        tokenizer = dl.dataset.tokenizer
        max_len = dl.dataset.tokenizer_max_len
    except AttributeError:
        ref_str = "decoder_input_ids"
        tokenizer = dl.dataset.tgt_tokenizer
        max_len = dl.dataset.tgt_tokenizer_max_len
    bos_token_id, eos_token_id, pad_token_id = get_special_token_ids(hf_model, tokenizer)
    ref_list, hyp_list = [], []
    gen = dict(bos_token_id=bos_token_id, eos_token_id=eos_token_id, pad_token_id=pad_token_id, max_length=max_len)
    if getattr(config, "length_penalty", None) is not None:
        gen["length_penalty"] = config.length_penalty
    if getattr(config, "beam_width", None) is not None:
        gen["num_beams"] = config.beam_width
    with torch.no_grad():
        for batch in dl:
            batch = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
            batch_size = batch[ref_str].shape[0]
            start = torch.ones((batch_size, 1), dtype=torch.long).cuda() * bos_token_id
            if len(models) > 1:       # n-best ensembling: summed logits (ref: beam_search.py:243-262, bin/ensemble.py:72-80)
                eo = [dict(zip(("encoder_hidden_states", "encoder_attention_mask"), m.encode(**batch))) for m in models]
                hyps = hf_model.generate(input_ids=start, hf_models=[m.dec.decoder for m in models], encoders_outputs=eo, **gen)
            else:
                encoder_outputs, encoder_attention_mask = models[0].encode(**batch)
                hyps = hf_model.generate(input_ids=start, encoder_hidden_states=encoder_outputs,
                                         encoder_attention_mask=encoder_attention_mask, **gen)
            for h, r in zip(hyps.tolist(), batch[ref_str].tolist()):      # one device-to-host copy each, not one per row
                hyp_list.append(tokenizer.decode(h, skip_special_tokens=True, clean_up_tokenization_spaces=False))
                ref_list.append(tokenizer.decode(r, skip_special_tokens=True, clean_up_tokenization_spaces=False))
    return {"refs": ref_list, "hyps": hyp_list}
