"""decode-step latency with and without HIP-graph capture (B=64, 12-layer d=768 decoder, greedy and beam-4, 64 tokens)"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench
from vilmedic_amd import generation
dev = torch.device("cuda")
model = bench.build_model(dev).eval()
dec = model.dec.decoder
B, S, T = 64, 197, 64
enc = torch.randn(B, S, 768, device=dev).bfloat16()
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
start = torch.zeros(B, 1, dtype=torch.long, device=dev)
def run(nb):
    kw = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=T)
    if nb > 1: kw["num_beams"] = nb
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = dec.generate(input_ids=start, encoder_hidden_states=enc, encoder_attention_mask=mask, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out
for graph in (False, True):
    generation.DECODE_GRAPH = graph
    for nb in (1, 4):
        t_first, o1 = run(nb)
        t_second, o2 = run(nb)
        steps = o2.shape[1] - 1
        print(f"graph={graph} beams={nb}: first call {t_first*1e3:.0f} ms, second call {t_second*1e3:.1f} ms = {t_second/steps*1e3:.2f} ms/step over {steps} steps; same ids: {torch.equal(o1, o2)}", flush=True)
    if not graph: ref = {nb: run(nb)[1] for nb in (1, 4)}
for nb in (1, 4):
    print("graph vs eager ids identical, beams", nb, torch.equal(run(nb)[1], ref[nb]))
