#!/usr/bin/env python3
"""bench.py -- RRG training throughput (image-report pairs/s) on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], SURVEY §8d "C2"): ViT-B/16 image encoder + 12-layer d=768 BERT-generation decoder
with cross-attention, bf16 activations / fp32 master weights, per-GPU batch 64, 224x224 synthetic images, 128-token
synthetic reports, V=30522, dropout 0.1 as in the shipped YAMLs; a step = forward + backward + (gradient all-reduce over
RCCL when N>1) + fused Adam.  One process per GPU, weak scaling (fixed per-GPU batch).

The JSON line also carries
  roofline     -- the bf16 MFMA GEMM family (the dominant kernels): algorithmic FLOPs / HIP-event time of those launches,
                  measured live on the launch stream by the library's event profiler over extra (untimed) steps;
  cpu_baseline -- the CPU oracle (oracle/torch_ref.py, a port of the reference's math) timed on this box's host cores
                  on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VIT_B16 = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               image_size=224, patch_size=16, num_channels=3, layer_norm_eps=1e-12)
DEC_12L = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               vocab_size=30522, max_position_embeddings=514, layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1,
               eos_token_id=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, initializer_range=0.02)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
PMC_PROFILE = "profiles/r06_pmc_gemm.txt"      # separate rocprofv3 --pmc passes of the dominant shapes (traffic is not measurable in-process)
PMC_FAMILY = "profiles/r06_pmc_gemm_family.json"   # FETCH_SIZE / WRITE_SIZE passes over a whole bench run, summed over the GEMM family (tools/pmc_family.py)


def committed_traffic():
    """HBM-side bytes of the GEMM family per step from the committed PMC passes (they cannot be collected in-process), with the digest of
    the library they were measured on: ``current_build`` says whether that is the library this process loaded"""
    try:
        with open(os.path.join(ROOT, PMC_FAMILY)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    from vilmedic_amd import build
    fam = d.get("gemm_family", {})
    return {"bytes_per_step": fam.get("traffic_MB_per_step", 0.0) * 1e6, "launches_per_step": fam.get("launches_per_step"),
            "fetch_MB_per_step": fam.get("fetch_MB_per_step"), "write_MB_per_step": fam.get("write_MB_per_step"),
            "profile": PMC_FAMILY, "profile_build_digest": (d.get("build_digest") or "")[:16],
            "current_build": d.get("build_digest") == build._lib_digest()}


def flops_per_pair_fwd(S=197, L=128, d=768, V=30522, layers=12):
    """36.80 GMAC fwd per pair at C2."""
    vit = 196 * 768 * d + layers * (12 * S * d * d + 2 * S * S * d)
    dec = layers * (4 * L * d * d + 2 * L * L * d + 2 * L * d * d + 2 * S * d * d + 2 * L * S * d + 8 * L * d * d)
    head = L * d * V
    return 2.0 * (vit + dec + head)


def gemm_family_algorithmic_bytes(B=64, S=197, L=128, d=768, ff=3072, V=30528, layers=12):
    """bytes a step's GEMM family must move at the least (bf16 operands and outputs once per product, fp32 weight gradients written once,
    the epilogue operands -- residual, GELU pre-activation -- once): per linear of M rows, K inputs, N outputs the forward, dgrad and wgrad
    products move 6 (MK + MN) + 8 NK bytes.  29.2 GB at the benched configuration."""
    def lin(M, K, N, extra=0):
        return 6 * (M * K + M * N) + 8 * N * K + extra
    Me, Md = B * S, B * L
    enc = layers * (lin(Me, d, 3 * d) + lin(Me, d, d, 2 * Me * d) + lin(Me, d, ff, 4 * Me * ff) + lin(Me, ff, d, 2 * Me * d))
    dec = layers * (lin(Md, d, 3 * d) + 3 * lin(Md, d, d, 2 * Md * d) + lin(Md, d, ff, 4 * Md * ff) + lin(Md, ff, d, 2 * Md * d))
    return enc + dec + lin(Me, d, 2 * d * layers) + lin(Md, d, V) + lin(B * (S - 1), d, d)


def dominant_shape_roofline(dump_path):
    """the single largest forward GEMM shape (QKV projection of the ViT, 12608 x 2304 x 768, bias epilogue): achieved rate from
    this run's per-launch HIP events, HBM-side traffic per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2
    gfx950 correction + WRITE_SIZE, profiles/r04_pmc_gemm.txt; rounds 2 / 3: r02_n_pmc_gemm.txt, r03_k_pmc_gemm.txt)."""
    tag, M, N, K = "M12608_N2304_K768_l00", 12608, 2304, 768
    ms = n = 0.0
    try:
        for line in open(dump_path):
            f = line.split()
            if len(f) >= 5 and f[0] == "0" and f[1].startswith(tag):
                n += float(f[2]); ms += float(f[3])
    except OSError:
        return None
    if n == 0:
        return None
    # HBM-side bytes need rocprofv3 --pmc passes (tools/pmc_kernels.sh, not possible in-process): reported under a separate key that names the
    # committed profile, never mixed with the figures measured in this run
    dur = ms / n * 1e-3
    algo = 2.0 * (M * K + N * K + M * N)
    return {"kernel": "gemm_fast_kernel<0,0,...> C[12608,2304] = A[12608,768] . B[2304,768]^T + bias (ViT QKV projection)",
            "bound": "mfma", "achieved": round(2.0 * M * N * K / dur / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(2.0 * M * N * K / dur / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "avg_launch_ms": round(dur * 1e3, 4),
            "launches": int(n), "algorithmic_bytes": algo, "traffic": None, "traffic_from_committed_profile": PMC_PROFILE,
            "hbm_frac_of_8TBps_algorithmic": round(algo / dur / 8e12, 4)}


def build_model(device):
    from vilmedic_amd.models.rrg.RRG import RRG
    torch.manual_seed(0)
    model = RRG(decoder=dict(proto=None, **DEC_12L),
                cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **VIT_B16))
    return model.to(device)


def synthetic_batch(B, L, V, device, seed):
    import golden_recipes as R
    images = R.make_images(B, 224, seed=seed).to(device)
    ids, am = R.make_reports(B, L, V, seed=seed)
    return images, ids.to(device), am.to(device)


def _cpu_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))        # cgroup-limited boxes report hundreds of CPUs; oversubscribing them stalls the oracle


C1_CNN = dict(layer_type="basic", embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2], hidden_act="relu")
C1_DEC = dict(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072, vocab_size=4000,
              max_position_embeddings=514, layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1, eos_token_id=2)


def _resnet_shapes(cfg, prefix):
    """parameter / buffer shapes of the HF-style ResNet the oracle restates (oracle.torch_ref.hf_resnet_forward)"""
    sh = {}

    def conv_bn(p, cin, cout, k):
        sh[p + ".convolution.weight"] = (cout, cin, k, k)
        for n in ("weight", "bias", "running_mean", "running_var"):
            sh[p + ".normalization." + n] = (cout,)
    conv_bn(prefix + "embedder.embedder", 3, cfg["embedding_size"], 7)
    cin = cfg["embedding_size"]
    for si, (depth, cout) in enumerate(zip(cfg["depths"], cfg["hidden_sizes"])):
        for li in range(depth):
            p = f"{prefix}encoder.stages.{si}.layers.{li}"
            stride = (2 if si > 0 else 1) if li == 0 else 1
            if cin != cout or stride != 1:
                conv_bn(p + ".shortcut", cin, cout, 1)
            conv_bn(p + ".layer.0", cin, cout, 3)
            conv_bn(p + ".layer.1", cout, cout, 3)
            cin = cout
    return sh


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline_child():
    """The CPU path timed beside the GPU number: the oracle (a port of the reference's math, fp32, stock PyTorch CPU
    kernels) on this box's host cores, in its own process (no GPU context).  (1) BASELINE configs[1] -- the benched workload -- at
    B = 2: median of 6 steps after 1 warm-up, all threads: the JSON line's ``value``.  (2) BASELINE configs[0] ("C1": ResNet-18 +
    2-layer decoder, B = 4, 64 tokens, V = 4000) exactly: median of 20 steps after 3 warm-ups with all threads, and of 5 steps with ONE
    thread.  Bounded: about 30 s of CPU work."""
    import golden_recipes as R
    from oracle import torch_ref as O
    threads = _cpu_threads()

    def run(step, warm, n, nthreads):
        torch.set_num_threads(nthreads)
        for _ in range(warm):
            step()
        ts = []
        for _ in range(n):
            t0 = time.time()
            step()
            ts.append(time.time() - t0)
        return _median(ts)

    # ---- (1) the benched workload (C2)
    vcfg, dcfg = dict(VIT_B16), dict(DEC_12L)
    st = {"enc.model." + k: v for k, v in R.rand_state(R.vit_shapes(vcfg), 0, std=0.02).items()}
    st.update({"dec.decoder." + k: v for k, v in R.rand_state(R.decoder_shapes(dcfg), 1, std=0.02).items()})
    st = {k: v.requires_grad_(True) for k, v in st.items()}
    B, L = 2, 128
    images = R.make_images(B, 224, seed=0)
    ids, am = R.make_reports(B, L, dcfg["vocab_size"], seed=0)
    opt = torch.optim.Adam(list(st.values()), lr=1e-4)

    def step_c2():
        loss, _ = O.rrg_vit_forward(images, ids, am, st, vcfg, dcfg)
        opt.zero_grad()
        loss.backward()
        opt.step()
    t_c2 = run(step_c2, 1, 6, threads)
    del st, opt
    # ---- (2) C1 exactly
    g = torch.Generator().manual_seed(0)
    st1 = {}
    for k, shp in _resnet_shapes(C1_CNN, "enc.model.").items():
        if k.endswith("running_var") or k.endswith("normalization.weight"):
            st1[k] = torch.ones(shp)
        elif k.endswith("running_mean") or k.endswith("normalization.bias"):
            st1[k] = torch.zeros(shp)
        else:
            st1[k] = torch.randn(shp, generator=g) * (2.0 / (shp[1] * shp[2] * shp[3])) ** 0.5
    st1["enc.visual_projection.weight"] = torch.randn(768, 512, generator=g) * 0.02
    st1["enc.visual_projection.bias"] = torch.zeros(768)
    st1.update({"dec.decoder." + k: v for k, v in R.rand_state(R.decoder_shapes(C1_DEC), 2, std=0.02).items()})
    train = [v.requires_grad_(True) for k, v in st1.items() if "running" not in k]
    img1 = R.make_images(4, 224, seed=1)
    ids1, am1 = R.make_reports(4, 64, C1_DEC["vocab_size"], seed=1)
    opt1 = torch.optim.Adam(train, lr=1e-4)

    def step_c1():
        loss, _ = O.rrg_cnn_forward(img1, ids1, am1, st1, C1_CNN, C1_DEC, training=True)
        opt1.zero_grad()
        loss.backward()
        opt1.step()
    t_c1_all = run(step_c1, 3, 20, threads)
    t_c1_one = run(step_c1, 1, 5, 1)
    print(json.dumps({
        "value": round(B / t_c2, 3), "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": f"oracle/torch_ref.py rrg_vit_forward + backward + Adam on the benched model (ViT-B/16 + 12-layer decoder, V=30522), B={B}, "
                  f"L={L}, fp32, median of 6 steps after 1 warm-up, {threads} torch threads",
        "c1": {"workload": "BASELINE configs[0]: HF ResNet-18 (512 ch) + projection + 2-layer decoder, B=4, 224x224, 64 tokens, V=4000, fp32, "
                           "forward + backward + Adam (oracle rrg_cnn_forward)",
               "pairs_per_s_all_threads": round(4 / t_c1_all, 3), "threads": threads, "steps": "median of 20 after 3 warm-ups",
               "pairs_per_s_one_thread": round(4 / t_c1_one, 3), "steps_one_thread": "median of 5 after 1 warm-up"}}), flush=True)


def cpu_baseline(timeout_s=240.0):
    """bounded: the oracle runs in a child process that is killed after ``timeout_s``"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], capture_output=True, text=True,
                           timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "pairs/s", "cores": _cpu_threads(), "kind": "port", "sample": "oracle child failed: " + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "pairs/s", "cores": _cpu_threads(), "kind": "port",
                "sample": f"oracle (B=2 fwd+bwd+Adam) did not finish one step pair within {timeout_s:.0f} s on this host"}


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: re-run this command line as N ranks (one per GPU, RCCL over xGMI) under
    torch.distributed.run on 127.0.0.1 and return its exit code.  Rank 0's JSON line is the child's stdout."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} requested but this node exposes {have} GPU(s) "
              f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}): RCCL needs one distinct device per rank, "
              f"not starting {n} ranks", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def allreduce_alone_ms(ddp, barrier, iters=5):
    """the gradient exchange with nothing to hide behind: cast to the wire dtype, the RCCL all-reduces of the whole arena, cast back
    (what ArenaDDP.finish() does), per step.  The timed steps overlap all but the front encoder bucket of this with the backward pass."""
    ddp.finish()
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        ddp.finish()
    barrier()
    return (time.perf_counter() - t0) / iters * 1e3


def secondary_metrics(model, device):
    """SURVEY §8(d)'s secondary figures and north_star's two named roofline targets, measured live on this GPU (rank 0, N = 1):
    the decode step (greedy / beam-4, both step dtypes) against its HBM floor, the contrastive loss at the C3 size (global batch 2048,
    768 features) and the decoder's cross-attention core (B = 64, 12 heads, 128 queries x 197 keys) against MFMA and HBM peaks."""
    import ctypes as C
    from vilmedic_amd import ops
    from vilmedic_amd._lib import lib
    from vilmedic_amd.blocks.losses.selfsup import _SimilarityLossFn
    out = {}
    L_ = lib()

    def kernel_ms(fn, iters):
        """vmhip kernel time per call (HIP events on the launch stream, every kernel alone) and launches per call"""
        L_.vm_prof_reset(); L_.vm_prof_enable(1)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        L_.vm_prof_enable(0)
        tot, n_tot = 0.0, 0
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        for f in range(8):
            if L_.vm_prof_read(f, C.byref(ms), C.byref(work), C.byref(n)) == 0:
                tot += ms.value; n_tot += n.value
        L_.vm_prof_reset()
        return tot / iters, n_tot // iters

    # ---- decode step: weights are read once per step (bf16: the decoder's shadows + tied LM head; fp32: the master copies)
    dec = model.dec.decoder
    was_training = model.training
    model.eval()
    B, S, T = 64, 197, 65
    n_dec = sum(p.numel() for p in dec.parameters())
    g = torch.Generator(device=device).manual_seed(3)
    enc = torch.randn(B, S, 768, device=device, generator=g).bfloat16()
    mask = torch.ones(B, S, dtype=torch.bool, device=device)
    start = torch.zeros(B, 1, dtype=torch.long, device=device)
    n_layers, D = len(dec.bert.encoder.layer), 768
    kv_elems = n_layers * (B * S * 2 * D)            # cross K|V of every layer, read once per step (shared by the beams of a sample)
    out["decode"] = {"batch": B, "new_tokens": T - 1, "weight_bytes_bf16": 2 * n_dec, "cross_kv_bytes_bf16": 2 * kv_elems,
                     "floor_note": "HBM floor per step (SURVEY 8d) = decoder weights + tied LM head read once + the cross K|V of all layers + the "
                                   "self K|V cache at half its final length, 2 B/element bf16 and 4 B/element fp32, at 8 TB/s"}
    for beams, dtype in ((1, "bf16"), (1, "fp32"), (4, "bf16"), (4, "fp32")):
        kw = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=T, decode_dtype=dtype)
        if beams > 1:
            kw["num_beams"] = beams
        with torch.no_grad():
            dec.generate(input_ids=start, encoder_hidden_states=enc, encoder_attention_mask=mask, **kw)      # builds the graphs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids = dec.generate(input_ids=start, encoder_hidden_states=enc, encoder_attention_mask=mask, **kw)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = max(1, ids.shape[1] - 1)
        ms_step = dt / steps * 1e3
        self_elems = n_layers * (B * beams * (T // 2) * 2 * D)
        floor_ms = (n_dec + kv_elems + self_elems) * (2 if dtype == "bf16" else 4) / 8e12 * 1e3
        out["decode"][f"beams{beams}_{dtype}"] = {"tokens_per_s": round(B * steps / dt, 1), "ms_per_step": round(ms_step, 3),
                                                   "floor_ms": round(floor_ms, 4), "hbm_frac": round(floor_ms / ms_step, 4)}
    model.train(was_training)

    # ---- contrastive similarity loss, C3 size (forward + backward)
    Bc, Dc = 2048, 768
    a = torch.randn(Bc, Dc, device=device, generator=g).requires_grad_(True)
    b = torch.randn(Bc, Dc, device=device, generator=g).requires_grad_(True)

    def contrastive_step():
        a.grad = b.grad = None
        r, c = _SimilarityLossFn.apply(a, b, True, 10.0, 1e-8)
        (r.mean() + c.mean()).backward()
    for _ in range(3):
        contrastive_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        contrastive_step()
    torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t0) / 20 * 1e6
    k_ms, k_n = kernel_ms(contrastive_step, 5)
    flop = 6.0 * Bc * Bc * Dc
    out["contrastive_B2048_D768"] = {"kernel_us_fwd_bwd": round(k_ms * 1e3, 1), "vmhip_launches": k_n, "wall_us_fwd_bwd": round(wall_us, 1),
                                      "gflop": round(flop * 1e-9, 1), "mfma_frac": round(flop / (k_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}

    # ---- attention cores at the step's shapes (forward; dropout 0.1 as in the step)
    H, dh = 12, 64
    shapes = {"cross_attention_fwd": (128, 197, False, "cross"), "vit_self_attention_fwd": (197, 197, False, "self"),
              "causal_self_attention_fwd": (128, 128, True, "self")}
    for name, (Lq, Lk, causal, kind) in shapes.items():
        km = torch.ones(B, Lk, dtype=torch.uint8, device=device) if name != "vit_self_attention_fwd" else None
        if kind == "cross":
            q = (torch.randn(B, Lq, H * dh, device=device, generator=g) * 0.5).bfloat16()
            kv = (torch.randn(B, Lk, 2 * H * dh, device=device, generator=g) * 0.5).bfloat16()
            f = lambda: ops.cross_attention(q, kv, km, H, 0.1)
        else:
            qkv = (torch.randn(B, Lq, 3 * H * dh, device=device, generator=g) * 0.5).bfloat16()
            f = lambda: ops.self_attention(qkv, km, H, causal, 0.1)
        with torch.no_grad():
            for _ in range(3):
                f()
            k_ms, _ = kernel_ms(f, 20)
        us = k_ms * 1e3
        flop = 4.0 * B * H * Lq * Lk * dh * (0.5 if causal else 1.0)
        byt = 2.0 * (2 * B * Lq * H * dh + 2 * B * Lk * H * dh)
        out[name] = {"us": round(us, 2), "mfma_frac": round(flop / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     "hbm_frac": round(byt / (us * 1e-6) / 8e12, 4), "algorithmic_mb": round(byt / 1e6, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the decode / contrastive / attention-core figures (rank 0, N = 1)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VM_TRAIN_GRAPH", "0")),
                    help="1: replay the whole step from one captured HIP graph (vilmedic_amd.graph); 0: eager launches")
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child()
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))        # `python bench.py --gpus N` alone: start the N ranks ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus})")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but this node exposes {torch.cuda.device_count()} device(s); "
                         "RCCL needs one distinct GPU per rank")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from vilmedic_amd import ops
    from vilmedic_amd._lib import lib
    from vilmedic_amd.optim import FusedAdam
    from vilmedic_amd.parallel import ArenaDDP

    # The model, its arena and the optimizer state are allocated BEFORE the RCCL communicator exists: on this stack
    # (ROCm 7.0 / RCCL 2.26, dmabuf IPC) device memory allocated after init_process_group made every step 5.6 ms slower
    # (measured in round 1: 35.65 vs 30.02 ms/step on one GPU with a 1-rank group).
    model = build_model(device)
    model.train()
    ops.manual_seed(1234 + rank)
    opt = FusedAdam(model, lr=1e-4)
    dist, wire = None, None
    if world > 1 or os.environ.get("VM_FORCE_DDP"):
        from vilmedic_amd.parallel import default_bf16_wire           # fp32 wire by default (exact mean); VM_DDP_WIRE=bf16 opts into the compressed wire
        if default_bf16_wire():
            wire = torch.empty(opt.arena.numel, dtype=torch.bfloat16, device=device)     # bf16 staging of the gradient all-reduce      # VM_FORCE_DDP: exercise the RCCL path on a single GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    ddp = ArenaDDP(model, dist, wire=wire) if dist is not None else None
    if ddp is not None:
        ddp.attach_optimizer(opt)            # Adam reads the averaged bf16 wire buffer (vm_adam_step_wire): no cast pass back to fp32
    B, L, V = args.batch, args.seq, DEC_12L["vocab_size"]
    images, ids, am = synthetic_batch(B, L, V, device, seed=rank)

    def eager_step(input_ids=ids, attention_mask=am, images=images):
        out = model(input_ids=input_ids, attention_mask=attention_mask, images=images, return_logits=False)
        opt.zero_grad()
        opt.gate = out["loss"].detach()      # NaN / Inf loss -> the update is skipped on the device (no host read of the loss)
        if ddp is not None:
            ddp.backward(out["loss"])        # two-phase backward: decoder all-reduce overlaps the ViT backward
        else:
            out["loss"].backward()
        opt.step()
        return out["loss"]

    step = eager_step
    # (under data parallelism the capture holds the RCCL collectives too: the decoder range and the mark-started encoder buckets keep their
    # side-stream fork / join inside the graph -- what a multi-GPU node replays is one graph launch per step and rank)
    if args.graph:
        from vilmedic_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(eager_step, dict(input_ids=ids, attention_mask=am, images=images), optimizer=opt, warmup=min(3, max(1, args.warmup - 1)))
        step = lambda: graphed(input_ids=ids, attention_mask=am, images=images)

    for _ in range(args.warmup):
        loss = step()

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # host side: the step enqueues ~1100 launches in 23-27 ms of Python against ~29 ms of GPU time, so a cyclic-GC pass over the
    # (static) module / arena object graph in the middle of a step stalls the GPU: park the long-lived objects in the permanent
    # generation once the warm-up has built them
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue = time.perf_counter() - t0          # the host's share: Python / ctypes time to enqueue the K steps (no synchronisation inside)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.unfreeze()
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.detach())

    roof = None
    if not args.no_roofline:
        # EVERY rank runs the two extra (untimed) steps -- they contain the gradient all-reduce when N > 1 -- but only rank 0
        # records and reads the per-launch events
        L_ = lib()
        if rank == 0:
            L_.vm_prof_reset()
            L_.vm_prof_enable(1)
        side = ops.SIDE_STREAM
        ops.SIDE_STREAM = False          # per-launch durations are taken with every kernel alone on the GPU (the timed
        for _ in range(2):               # steps above overlap parameter-gradient kernels with the dgrad chain)
            eager_step()
        torch.cuda.synchronize()
        ops.SIDE_STREAM = side
        if rank == 0:
            L_.vm_prof_enable(0)
            ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            fam = {}
            for f, name in enumerate(["gemm", "attention", "layernorm", "loss", "elementwise", "optimizer"]):
                L_.vm_prof_read(f, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n))
                fam[name] = (ms.value, work.value, n.value)
            gms, gwork, gn = fam["gemm"]
            ach = gwork / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "gemm_fast_kernel<LA,LB,...> (every forward / dgrad launch of vm_gemm_bf16; the LM head on gemm_p8_kernel) + "
                                             "gemm_p8w_kernel (the grouped weight + bias gradient launches of vm_wgrad_grouped, 256 x 256 tiles)",
                    "achieved": round(ach, 1),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                    "traffic_note": "HBM-side bytes cannot be collected in-process: traffic = (FETCH_SIZE x 2 + WRITE_SIZE) of every GEMM-family kernel of a "
                                    "step / launches per step, from the committed rocprofv3 --pmc passes named in traffic_source (tools/pmc_family.py)",
                    "launches_per_step": gn // 2, "avg_launch_ms": round(gms / max(gn, 1), 4),
                    "family_ms_per_step": {k: round(v[0] / 2, 3) for k, v in fam.items()}}
            ct = committed_traffic()
            if ct is not None and ct["bytes_per_step"] > 0:
                roof["traffic"] = round(ct["bytes_per_step"] / max(ct["launches_per_step"] or (gn // 2), 1))          # bytes per launch, like ``achieved``
                roof["traffic_source"] = ct
                roof["algorithmic_bytes"] = round(gemm_family_algorithmic_bytes(B=B, L=L) / max(ct["launches_per_step"] or (gn // 2), 1))
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes"], 3)
            dump = os.environ.get("VM_PROF_DUMP") or os.path.join(tempfile.gettempdir(), f"vm_prof_{os.getpid()}.txt")
            L_.vm_prof_dump(dump.encode())
            roof["dominant_shape"] = dominant_shape_roofline(dump)
            L_.vm_prof_reset()
        barrier()

    rccl = None
    if ddp is not None:
        ar_ms = allreduce_alone_ms(ddp, barrier)
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}")
        rccl = {"rccl_ranks": world, "devices": names, "backend": dist.get_backend(), "all_reduce_ms_per_step": round(ar_ms, 3),
                "all_reduce_note": "whole-arena gradient exchange ALONE (casts + 4 pipelined all-reduces + casts back); inside the timed steps "
                                   "all but the front encoder bucket overlaps the backward pass",
                "wire_dtype": "bf16" if ddp.bf16_wire else "fp32", "wire_bytes_per_step": int(opt.arena.numel * (2 if ddp.bf16_wire else 4)),
                "encoder_buckets_started_from_backward_marks": ddp.mark_starts}

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            secondary = secondary_metrics(model, device)
        except Exception as e:      # the headline line must survive a failing side measurement; the failure is reported, not hidden
            secondary = {"error": f"{type(e).__name__}: {e}"}
        if isinstance(secondary, dict):
            # the other launch mode of the SAME step, after the timed region: 10 iterations replayed from one captured HIP graph when the
            # timed steps were eager launches (and the reverse) -- the rate that does not depend on the host, next to the one that does
            try:
                from vilmedic_amd.graph import GraphedTrainStep
                if args.graph:
                    other, name = eager_step, "eager"
                else:
                    g2 = GraphedTrainStep(eager_step, dict(input_ids=ids, attention_mask=am, images=images), optimizer=opt, warmup=1)
                    other, name = (lambda: g2(input_ids=ids, attention_mask=am, images=images)), "hip-graph replay"
                for _ in range(3):
                    other()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    other()
                torch.cuda.synchronize()
                secondary["other_launch_mode"] = {"launch_mode": name, "ms_per_step": round((time.perf_counter() - t0) * 100.0, 3), "steps": 10}
            except Exception as e:
                secondary["other_launch_mode"] = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / elapsed
        step_flops = 3.0 * flops_per_pair_fwd(L=L) * B
        line = {
            "metric": "image-report pairs/sec training (RRG, 224px x 128tok)", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "host_enqueue_ms_per_step": round(1e3 * enqueue / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "config/RRG: ViT-B/16 + 12-layer BERT-generation decoder (d=768, h=12, ff=3072, V=30522), "
                                   "bf16, 224x224 images, 128-token reports, dropout 0.1, fwd+bwd+Adam",
                       "per_gpu_batch": B, "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}",
                       "ddp_wire": (("bf16" if ddp.bf16_wire else "fp32") if ddp is not None else None)},
            "model_tflops_per_s": round(step_flops * args.steps / elapsed / 1e12 * world, 1),
            "launch_mode": "hip-graph replay" if args.graph else "eager",
            "final_loss": round(final_loss, 4),
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary,
        }
        if rccl is not None:
            line.update(rccl)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
