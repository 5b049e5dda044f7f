"""cost of the two-phase (decoder, then encoder) backward used by ArenaDDP against a single backward call, on one GPU."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench
from vilmedic_amd import ops
from vilmedic_amd.optim import FusedAdam
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model = bench.build_model(dev); model.train(); ops.manual_seed(1)
opt = FusedAdam(model, lr=1e-4)
images, ids, am = bench.synthetic_batch(64, 128, 30522, dev, 0)
def step(split):
    out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
    opt.zero_grad()
    if split:
        feats, leaf = model._split
        out["loss"].backward()
        feats.backward(leaf.grad)
        model._split = None
    else:
        out["loss"].backward()
    opt.step()
def run(tag, split, K=8):
    for _ in range(2): step(split)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step(split)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{tag}: enqueue {1e3*(t1-t0)/K:.2f}  complete {1e3*(time.perf_counter()-t0)/K:.2f} ms/step", flush=True)
run("plain", False)
model.split_backward = True
run("split, joins at both backward ends", True)
ops._side["defer"] = True
def step2():
    out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
    opt.zero_grad()
    feats, leaf = model._split
    out["loss"].backward()
    feats.backward(leaf.grad)
    model._split = None
    ops.join_side()
    opt.step()
for _ in range(2): step2()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step2()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"split, single join: enqueue {1e3*(t1-t0)/8:.2f}  complete {1e3*(time.perf_counter()-t0)/8:.2f} ms/step", flush=True)
ops._side["defer"] = False
from vilmedic_amd.parallel import ArenaDDP
class FakeDist:
    def get_world_size(self): return 1
    def get_backend(self): return "nccl"
ddp = ArenaDDP(model, FakeDist())
ddp._start = lambda s, e, c: []; ddp._wait = lambda w: None
def step3():
    out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
    opt.zero_grad(); ddp.backward(out["loss"]); opt.step()
for _ in range(2): step3()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step3()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"ArenaDDP(fake dist), no comm: enqueue {1e3*(t1-t0)/8:.2f}  complete {1e3*(time.perf_counter()-t0)/8:.2f} ms/step  split_at={ddp.split_at}", flush=True)
