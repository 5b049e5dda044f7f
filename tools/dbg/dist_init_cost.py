"""does merely initialising the nccl process group slow the (host-heavy) step?"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
import bench
from vilmedic_amd import ops
from vilmedic_amd.optim import FusedAdam
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model = bench.build_model(dev); model.train(); ops.manual_seed(1)
opt = FusedAdam(model, lr=1e-4)
images, ids, am = bench.synthetic_batch(64, 128, 30522, dev, 0)
def step():
    out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
    opt.zero_grad(); out["loss"].backward(); opt.step()
def run(tag, K=8):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{tag}: enqueue {1e3*(t1-t0)/K:.2f}  complete {1e3*(time.perf_counter()-t0)/K:.2f} ms/step", flush=True)
run("no dist")
import torch.distributed as dist
if len(sys.argv) > 1 and sys.argv[1] == "nodevid":
    dist.init_process_group("nccl", rank=0, world_size=1)
else:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
run("after init_process_group")
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
run("after first collective")
dist.destroy_process_group()
run("after destroy")
