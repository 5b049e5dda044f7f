#!/usr/bin/env python3
"""Host-side cost of one training step, measured WITHOUT a GPU.

The C2 step is 587 kernel launches; on the MI355X it takes 25 ms of GPU time and 22 ms of Python / ctypes / autograd time to
enqueue, so every further kernel gain is taxed by the host.  This tool times exactly that host part on any machine:
it builds a STUB of libvmhip.so (every C-ABI entry point of include/vmhip.h returns 0 and counts the call; nothing is computed),
points ``vilmedic_amd._lib`` at it, replaces the handful of ``torch.cuda`` stream calls by no-ops, and runs the real model code
(RRG: ViT-B/16 + 12-layer decoder, the bench's architecture, at batch 1 so that the few torch ops of the step cost nothing on
the CPU) under cProfile.  What it reports is the Python work per step and per launch -- the same code path the GPU run
executes, minus the kernels.  It lives in tools/ and patches the package from outside: the product has no such switch.

    python tools/host_profile.py [--steps 20] [--top 25] [--trace out.txt]

``--trace`` writes the sequence of C-ABI calls of one step (entry point + scalar arguments, pointers dropped): two builds of the
host code launch the same kernels with the same arguments iff their traces are equal -- a GPU-free equivalence check for host
refactors.
"""
import argparse
import contextlib
import cProfile
import ctypes as C
import os
import pstats
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def build_stub(sigs):
    """a shared library with every symbol of the C ABI: returns 0 (size queries: 1 MiB), counts calls"""
    d = tempfile.mkdtemp(prefix="vmstub_")
    src = ["#include <stddef.h>", "long vm_stub_calls = 0;"]
    for name, (res, _args) in sigs.items():
        if res is None:
            src.append(f"void {name}() {{ }}")
        elif res is C.c_char_p:
            src.append(f'const char* {name}() {{ return "stub"; }}')
        elif res is C.c_size_t:
            src.append(f"size_t {name}() {{ return (size_t)1 << 20; }}")
        else:
            src.append(f"int {name}() {{ vm_stub_calls++; return 0; }}")
    path = os.path.join(d, "stub.c")
    open(path, "w").write("\n".join(src) + "\n")
    so = os.path.join(d, "libvmstub.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-o", so, path])
    return so


class _Stream:
    cuda_stream = 0

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record_event(self, e=None):
        return e

    def synchronize(self):
        pass


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def synchronize(self):
        pass

    def wait(self, *a):
        pass


def _unguard(module, cls, needle):
    """re-define ``cls.__init__`` of ``module`` with its device check removed (the product refuses CPU tensors, rightly)"""
    import inspect
    import textwrap
    k = getattr(module, cls)
    src = textwrap.dedent(inspect.getsource(k.__init__))
    assert needle in src
    ns = {}
    exec(compile(src.replace(needle, "False"), module.__file__, "exec"), module.__dict__, ns)
    k.__init__ = ns["__init__"]


def patch(trace=None):
    from vilmedic_amd import _lib
    so = build_stub(_lib.SIGNATURES)
    L = C.CDLL(so)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.vm_sizeof_gemm_epilogue.restype = C.c_int
    real_sizeof = C.sizeof(_lib.GemmEpilogue)

    class Lib:
        """attribute access like a CDLL; optionally records (name, scalar args)"""

        def __getattr__(self, name):
            fn = getattr(L, name)
            if name == "vm_sizeof_gemm_epilogue":
                return lambda: real_sizeof
            if trace is None:
                setattr(self, name, fn)
                return fn

            def rec(*a):
                trace.append((name,) + tuple(x for x in a if isinstance(x, (int, float)) and not isinstance(x, bool)))
                return fn(*a)
            setattr(self, name, rec)
            return rec

    _lib._lib = Lib()
    _lib.stream = lambda: C.c_void_p(0)
    _lib.ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    for mod in list(sys.modules.values()):          # modules that did ``from ._lib import ptr, stream`` keep their own names
        if getattr(mod, "__name__", "").startswith("vilmedic_amd") and mod is not _lib:
            for n in ("ptr", "stream"):
                if hasattr(mod, n):
                    setattr(mod, n, getattr(_lib, n))
    _unguard(sys.modules["vilmedic_amd.arena"], "ParamArena", 'dev.type != "cuda"')
    one = _Stream()
    torch.cuda.current_stream = lambda *a, **k: one
    torch.cuda.Stream = lambda *a, **k: _Stream()
    torch.cuda.Event = _Event
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.record_stream = lambda self, s: None
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    real_zero = torch.Tensor.zero_
    torch.Tensor.zero_ = lambda self: self if self.numel() > (1 << 20) else real_zero(self)   # the 0.9 GB gradient memset is asynchronous on the GPU
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--trace", default=None)
    args = ap.parse_args()
    import vilmedic_amd  # noqa: F401  (import every module before patching their ptr / stream names)
    from vilmedic_amd import ops, optim  # noqa: F401
    from vilmedic_amd.models import RRG
    trace = [] if args.trace else None
    L = patch(trace)
    calls = C.c_long.in_dll(L, "vm_stub_calls")

    import bench
    cfg = bench.c2_config() if hasattr(bench, "c2_config") else None
    if cfg is None:
        dec = dict(proto=None, add_cross_attention=True, attention_probs_dropout_prob=0.1, bos_token_id=0, eos_token_id=2, hidden_act="gelu",
                   hidden_dropout_prob=0.1, hidden_size=768, initializer_range=0.02, intermediate_size=3072, is_decoder=True, layer_norm_eps=1e-5,
                   max_position_embeddings=514, num_attention_heads=12, num_hidden_layers=args.layers, pad_token_id=1, vocab_size=30522)
        cnn = dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, image_size=224, patch_size=16, hidden_size=768,
                   num_hidden_layers=args.layers, num_attention_heads=12, intermediate_size=3072)
    else:
        dec, cnn = cfg
    torch.manual_seed(0)
    model = RRG(decoder=dec, cnn=cnn)
    model.train()
    opt = optim.FusedAdam(model, lr=1e-4)
    B, Lq = args.batch, 128
    ids = torch.randint(3, 30522, (B, Lq))
    ids[:, 0] = 0
    am = torch.ones(B, Lq, dtype=torch.long)
    images = torch.randn(B, 3, 224, 224)

    def step():
        out = model(input_ids=ids, attention_mask=am, images=images)
        out["loss"].backward()
        opt.step()
        opt.zero_grad()

    for _ in range(3):
        step()
    if trace is not None:
        trace.clear()
        step()
        with open(args.trace, "w") as f:
            for t in trace:
                f.write(" ".join(str(x) for x in t) + "\n")
        print("trace of one step:", len(trace), "C-ABI calls ->", args.trace)
    import gc
    gc.collect()
    gc.freeze()
    c0 = calls.value
    wall, cpu = [], []
    for _ in range(args.steps):                 # per step: wall clock and CPU time of this thread (immune to preemption on a shared host)
        t0, u0 = time.perf_counter(), time.thread_time()
        step()
        wall.append(time.perf_counter() - t0)
        cpu.append(time.thread_time() - u0)
    n = (calls.value - c0) / args.steps
    wall.sort()
    cpu.sort()
    print("host time per step: min %.2f ms, median %.2f ms wall; min %.2f ms, median %.2f ms thread CPU   C-ABI calls per step: %.0f   %.1f us per call"
          % (wall[0] * 1e3, wall[len(wall) // 2] * 1e3, cpu[0] * 1e3, cpu[len(cpu) // 2] * 1e3, n, cpu[len(cpu) // 2] / max(n, 1) * 1e6))
    if args.top <= 0:
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
