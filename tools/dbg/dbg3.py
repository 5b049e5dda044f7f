import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import golden_recipes as R
from oracle import torch_ref as O
from test_hip_models_gpu import build_decoder
from vilmedic_amd.generation import DecodeState
g=torch.load('/root/repo/tests/golden/g7_decode.pt',weights_only=False)
cfg,rc=g['cfg'],g['recipe']; dev=torch.device('cuda:0')
dec,st=build_decoder(cfg,g['seed'],**rc); dec.eval()
gen=torch.Generator().manual_seed(g['seed']+1)
enc=torch.randn(g['B'],g['S'],cfg['hidden_size'],generator=gen); enc[~g['enc_mask']]=0.0
ref=g['beams1_lp1.0']['sequences']
state=DecodeState(dec.decoder, enc.to(dev), g['enc_mask'].to(dev), 1, g['max_len'])
h=O.decoder_hidden(ref[:, :-1], None, enc, g['enc_mask'], st, cfg)
ol=O.lm_logits(h, st).float()
for t in range(ref.shape[1]-1):
    lg=state.step(ref[:,t].to(dev), t).cpu()
    d=(lg-ol[:,t]).abs().max(dim=1)[0]
    top2=ol[:,t].topk(2)[0]
    print(t, 'maxdiff/row', [round(x,3) for x in d.tolist()], 'oracle top-2 gap', [round(x,3) for x in (top2[:,0]-top2[:,1]).tolist()], 'argmax eq', (lg.argmax(-1)==ol[:,t].argmax(-1)).tolist())
