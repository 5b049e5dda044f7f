"""is the training step CPU (launch) bound?  enqueue time of K steps (no sync) vs GPU completion time"""
import sys, time, torch
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
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step, complete {1e3*(t2-t0)/K:.2f} ms/step")
# forward only / backward only CPU time
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = [model(input_ids=ids, attention_mask=am, images=images, return_logits=False) for _ in range(3)]
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"fwd enqueue {1e3*(t1-t0)/3:.2f} ms, complete {1e3*(t2-t0)/3:.2f}")
t0 = time.perf_counter()
for o in outs: o["loss"].backward()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"bwd enqueue {1e3*(t1-t0)/3:.2f} ms, complete {1e3*(t2-t0)/3:.2f}")
