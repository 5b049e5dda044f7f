#!/usr/bin/env python3
"""Which pieces of the data-parallel step survive HIP-graph capture on this stack (RCCL through torch.distributed on a 1-rank group)?
Each variant runs in its own process (a crash inside RCCL / the capture must not take the others down):
    python tools/ddp_graph_probe.py            # all variants
    python tools/ddp_graph_probe.py <variant>  # one, in this process"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = ["sync_sum_f32", "sync_avg_bf16", "async_avg_bf16", "side_stream_async", "two_collectives_two_streams", "arena_ddp_tiny", "bench_model"]


def run(variant):
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    from vilmedic_amd import ops
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.ones(1 << 20, device=dev)
    xb = torch.ones(1 << 20, device=dev, dtype=torch.bfloat16)
    dist.all_reduce(x)                      # communicator warm-up outside any capture
    dist.all_reduce(xb, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()

    def body():
        if variant == "sync_sum_f32":
            dist.all_reduce(x)
        elif variant == "sync_avg_bf16":
            dist.all_reduce(xb, op=dist.ReduceOp.AVG)
        elif variant == "async_avg_bf16":
            w = dist.all_reduce(xb, op=dist.ReduceOp.AVG, async_op=True)
            x.mul_(2.0)
            w.wait()
        elif variant == "side_stream_async":
            with ops.side_context(dev):
                w = dist.all_reduce(xb, op=dist.ReduceOp.AVG, async_op=True)
                w.wait()
            ops.join_side()
        elif variant == "two_collectives_two_streams":
            w1 = dist.all_reduce(xb[: 1 << 19], op=dist.ReduceOp.AVG, async_op=True)
            with ops.side_context(dev):
                w2 = dist.all_reduce(xb[1 << 19:], op=dist.ReduceOp.AVG, async_op=True)
                w2.wait()
            w1.wait()
            ops.join_side()

    if variant in ("arena_ddp_tiny", "bench_model"):
        os.environ["VM_FORCE_DDP"] = "1"
        import bench
        from vilmedic_amd.graph import GraphedTrainStep
        from vilmedic_amd.optim import FusedAdam
        from vilmedic_amd.parallel import ArenaDDP
        model = bench.build_model(dev)
        B, L, V = (16, 128, bench.DEC_12L["vocab_size"]) if variant == "bench_model" else (4, 32, bench.DEC_12L["vocab_size"])
        model.train()
        opt = FusedAdam(model, lr=1e-4)
        ddp = ArenaDDP(model, dist)
        ddp.attach_optimizer(opt)
        images, ids, am = bench.synthetic_batch(B, L, V, dev, seed=0)

        def step(input_ids=ids, attention_mask=am, images=images):
            out = model(input_ids=input_ids, attention_mask=attention_mask, images=images, return_logits=False)
            opt.zero_grad()
            opt.gate = out["loss"].detach()
            ddp.backward(out["loss"])
            opt.step()
            return out["loss"]
        gs = GraphedTrainStep(step, dict(input_ids=ids, attention_mask=am, images=images), optimizer=opt, warmup=2)
        for i in range(6):
            loss = gs(input_ids=ids, attention_mask=am, images=images)
            torch.cuda.synchronize()
            print(f"  step {i}: loss {float(loss):.4f} graph={'yes' if gs.graph is not None else 'no'}", flush=True)
        print(f"PROBE {variant}: ok", flush=True)
        dist.destroy_process_group()
        return
    body()                                   # once eagerly
    torch.cuda.synchronize()
    with ops.capture(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"PROBE {variant}: ok (x[0]={float(x[0]):.1f}, xb[0]={float(xb[0]):.1f})", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in VARIANTS:
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), v], capture_output=True, text=True, timeout=400)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip()]
            ok = any(l.startswith(f"PROBE {v}: ok") for l in tail)
            print(f"=== {v}: rc={r.returncode} {'OK' if ok else 'FAILED'}")
            for l in tail[-(4 if ok else 40):]:
                print("    " + l[:300])
