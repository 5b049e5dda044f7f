"""torch.profiler pass over one RRG training step: which aten copies / adds / fills are still issued by torch (not by the HIP
library) and from which Python frames -- the list that drove the LN-fork / arena-view clean-ups.  GPU box only."""
import os, sys, torch
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
for _ in range(2): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::zero_", "aten::fill_", "aten::to", "aten::_to_copy", "aten::mul", "aten::detach")]
for e in sorted(rows, key=lambda e: -e.count)[:30]:
    print(e.key, e.count, str(e.input_shapes)[:120])
print("----- by stack")
ka2 = prof.key_averages(group_by_stack_n=6)
for e in sorted([e for e in ka2 if e.key in ("aten::copy_", "aten::clone")], key=lambda e: -e.count)[:12]:
    print(e.key, e.count)
    for s in e.stack[:6]: print("     ", s[-110:])
