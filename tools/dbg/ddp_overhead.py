"""where does ArenaDDP's single-GPU overhead come from?  (run with one process; init a 1-rank nccl group)"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch.distributed as dist
import bench
from vilmedic_amd import ops
from vilmedic_amd.optim import FusedAdam
from vilmedic_amd.parallel import ArenaDDP
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = bench.build_model(dev); model.train(); ops.manual_seed(1)
ddp = ArenaDDP(model, dist)
opt = FusedAdam(model, lr=1e-4)
images, ids, am = bench.synthetic_batch(64, 128, 30522, dev, 0)

def run(tag, K=6):
    def step():
        out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
        opt.zero_grad(); ddp.backward(out["loss"]); opt.step()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); print(f"{tag}: enqueue {1e3*(t1-t0)/K:.2f} complete {1e3*(time.perf_counter()-t0)/K:.2f} ms/step", flush=True)

orig_start, orig_wait = ddp._start, ddp._wait
ddp._start = lambda s, e, c: []; ddp._wait = lambda w: None
run("BEFORE any collective: split backward, no comm, no casts")
ddp._start, ddp._wait = orig_start, orig_wait
run("full ddp")
ddp._start = lambda s, e, c: []; ddp._wait = lambda w: None
run("split backward, no comm, no casts")
def start_casts(s, e, c):
    g = ddp.arena.gflat
    ops.cast_to_bf16(g[s:e], ddp._wire[s:e]); return [(s, e)]
def wait_casts(works):
    for s, e in works: ops.cast_to_f32(ddp._wire[s:e], ddp.arena.gflat[s:e])
ddp._start, ddp._wait = start_casts, wait_casts
run("split backward + casts only")
ddp._start, ddp._wait = orig_start, orig_wait
ddp.bf16_wire = False
run("full ddp, fp32 wire")
# raw collective timing
w = ddp._wire
for n in (w.numel(), w.numel() // 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): dist.all_reduce(w[:n], op=dist.ReduceOp.AVG)
    torch.cuda.synchronize(); print(f"all_reduce {n*2/1e6:.0f} MB bf16: {1e3*(time.perf_counter()-t0)/5:.2f} ms")
model.split_backward = False
ddp.split_at = None
ddp.bf16_wire = True
run("no split (backward then allreduce)")
dist.destroy_process_group()
