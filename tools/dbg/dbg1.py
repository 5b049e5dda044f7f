import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import golden_recipes as R
from oracle import torch_ref as O
from test_hip_models_gpu import build_decoder, cosine, rel_l2
g = torch.load('/root/repo/tests/golden/g3_decoder_tiny.pt', weights_only=False)
cfg=g['cfg']; dev=torch.device('cuda:0')
dec, st = build_decoder(cfg, g['seed'])
ids, am = R.make_reports(g["B"], g["L"], cfg["vocab_size"], seed=g["seed"])
gen = torch.Generator().manual_seed(g["seed"] + 1)
enc = torch.randn(g["B"], g["S"], cfg["hidden_size"], generator=gen); enc[~g["enc_mask"]] = 0.0
enc_d = enc.to(dev).to(torch.bfloat16).requires_grad_(True)
dec.train()
out = dec(input_ids=ids.to(dev), attention_mask=am.to(dev), encoder_outputs=enc_d, encoder_attention_mask=g["enc_mask"].to(dev))
out['loss'].backward()
sto = {k: v.clone().requires_grad_(True) for k,v in st.items()}
loss, logits = O.decoder_forward(ids, am, enc, g['enc_mask'], sto, cfg)
loss.backward()
named = dict(dec.decoder.named_parameters())
for n in sorted(sto):
    got = named[n].grad.float().cpu(); ref = sto[n].grad
    c, r = cosine(got, ref), rel_l2(got, ref)
    flag = '' if (c>0.999 and r<3e-2) else '   <<<<<'
    print(f"{n:70s} cos={c:.5f} rel={r:.4f}{flag}")
