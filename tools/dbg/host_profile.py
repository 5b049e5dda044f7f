"""cProfile of the host side of the training step (where do the ~23 ms of enqueue time per step go?)"""
import cProfile, pstats, sys, io, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench
from vilmedic_amd import ops
from vilmedic_amd.optim import FusedAdam
dev = torch.device("cuda")
model = bench.build_model(dev); model.train(); ops.manual_seed(1)
opt = FusedAdam(model, lr=1e-4)
images, ids, am = bench.synthetic_batch(64, 128, 30522, dev, 0)
def step():
    out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
    opt.zero_grad(); out["loss"].backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
