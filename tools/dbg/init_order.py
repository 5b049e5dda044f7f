"""does the order  init_process_group  vs  model / activation allocation  matter for step time?"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
import torch.distributed as dist
import bench
from vilmedic_amd import ops
from vilmedic_amd.optim import FusedAdam
mode = sys.argv[1]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if mode == "init_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = bench.build_model(dev); model.train(); ops.manual_seed(1)
opt = FusedAdam(model, lr=1e-4)
images, ids, am = bench.synthetic_batch(64, 128, 30522, dev, 0)
def step():
    out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
    opt.zero_grad(); out["loss"].backward(); opt.step()
if mode == "warm_then_init":
    for _ in range(2): step()
    torch.cuda.synchronize()
if mode in ("model_first", "warm_then_init"):
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
if mode == "init_first_nodevid":
    dist.init_process_group("nccl", rank=0, world_size=1)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); print(f"{mode}: {1e3*(time.perf_counter()-t0)/8:.2f} ms/step", flush=True)
print(os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), flush=True)
dist.destroy_process_group()
