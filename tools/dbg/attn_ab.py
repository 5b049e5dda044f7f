"""A/B of the head-resident attention kernels against the tile kernels (VM_ATTN_TILE=1) in one process: max abs
difference of o / dq|dk|dv and HIP-event timings.  usage: python tools/dbg/attn_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vilmedic_amd import ops

dev = torch.device("cuda")
torch.manual_seed(0)


def run(kind, B, H, Lq, Lk, causal, p, tile, iters=20):
    if tile: os.environ["VM_ATTN_TILE"] = "1"
    else: os.environ.pop("VM_ATTN_TILE", None)
    from vilmedic_amd._lib import lib
    lib().vm_reload_env()
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    km = torch.ones(B, Lk, dtype=torch.uint8, device=dev)
    km[:, Lk - 5:] = 0 if kind != "vit" else 1
    if kind in ("vit", "self"):
        qkv = (torch.randn(B, Lq, 3 * D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
        args = (qkv,)
        f = lambda: ops.self_attention(qkv, km if kind == "self" else None, H, causal, p)
    else:
        q = (torch.randn(B, Lq, D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
        kv = (torch.randn(B, Lk, 2 * D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
        args = (q, kv)
        f = lambda: ops.cross_attention(q, kv, km, H, p)
    do = (torch.randn(B, Lq, D, device=dev, generator=g)).bfloat16()
    ops.manual_seed(7)
    o = f()
    o.backward(do)
    grads = [a.grad.clone() for a in args]
    for a in args: a.grad = None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    outs = []
    ev[0].record()
    for _ in range(iters): outs.append(f())
    ev[1].record()
    for x in outs: x.backward(do)
    ev[2].record()
    torch.cuda.synchronize()
    return o.detach().float(), [x.float() for x in grads], ev[0].elapsed_time(ev[1]) / iters * 1e3, ev[1].elapsed_time(ev[2]) / iters * 1e3


for kind, B, H, Lq, Lk, causal, p in [("vit", 64, 12, 197, 197, False, 0.0), ("self", 64, 12, 128, 128, True, 0.1), ("cross", 64, 12, 128, 197, False, 0.1),
                                      ("self", 3, 2, 37, 37, True, 0.0), ("cross", 2, 3, 50, 131, False, 0.0), ("vit", 2, 2, 256, 256, False, 0.0),
                                      ("cross", 2, 2, 300, 77, False, 0.0)]:
    o1, g1, f1, b1 = run(kind, B, H, Lq, Lk, causal, p, True)
    o2, g2, f2, b2 = run(kind, B, H, Lq, Lk, causal, p, False)
    err_o = (o1 - o2).abs().max().item()
    err_g = max((a - b).abs().max().item() for a, b in zip(g1, g2))
    gmax = max(a.abs().max().item() for a in g1)
    print(f"{kind:5s} B{B} H{H} Lq{Lq} Lk{Lk} causal={int(causal)} p={p}: |do|={err_o:.3g} |dgrad|={err_g:.3g} (max {gmax:.3g})  "
          f"fwd {f1:7.1f} -> {f2:7.1f} us   bwd {b1:7.1f} -> {b2:7.1f} us", flush=True)
