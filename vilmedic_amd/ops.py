"""Python host side of the HIP hot path: thin wrappers over the C ABI (include/vmhip.h) and the
``torch.autograd.Function``s that pair each forward kernel with its backward.

Design notes
  * activations are bf16 row-major ``[rows, features]``; parameters are fp32 ``nn.Parameter``s (HF names)
    living in a flat arena (``vilmedic_amd.arena``) with a bf16 *shadow* the GEMMs read;
  * weight / bias gradients are ACCUMULATED IN PLACE into the fp32 ``.grad`` views of the arena by the
    wgrad GEMM (fp32 atomics, split-K) and the column-sum kernel -- autograd only routes activation
    gradients, so no per-step gradient tensors are allocated (sized for 288 GB HBM: everything resident);
  * dropout masks are regenerated from a counter-based RNG (seed per call site), never stored;
  * there is NO CPU fallback: every op raises on non-device tensors.
"""
import bisect
import contextlib
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import GemmEpilogue, VM_BF16, VM_F32, check, lib, ptr, stream

BF16 = torch.bfloat16

# ----------------------------------------------------------------------------- RNG seeds for dropout
_seed_state = {"base": 0x1234ABCD, "counter": 0}


# HIP-graph captures (generation.DecodeState, graph.GraphedTrainStep) are checked per THREAD: in the default "global" mode an event query of
# another thread -- the watchdog of an RCCL process group polls its collectives' events all the time -- invalidates a capture that happens
# to be open ("operation not permitted when stream is capturing"), i.e. a validation decode inside a data-parallel run aborted at random.
CAPTURE_MODE = "thread_local"


@contextlib.contextmanager
def capture(graph):
    """``torch.cuda.graph(graph)`` for this package's captures: per-thread error mode (above), and the cyclic garbage collector held off while
    the stream is capturing -- a collection that happens to run inside the capture and finds an OLD CUDAGraph (the decode graphs of a model
    that went out of scope) destroys it there, which HIP rejects ("operation not permitted when stream is capturing", raised from
    ~CUDAGraph: the process aborts).  torch.cuda.graph collects once before the capture begins; this keeps it that way until it ends."""
    import gc
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
            yield
    finally:
        if was:
            gc.enable()


def manual_seed(seed):
    _seed_state["base"] = int(seed) & 0xFFFFFFFF
    _seed_state["counter"] = 0


def next_seed():
    _seed_state["counter"] += 1
    return ((_seed_state["base"] << 32) ^ (_seed_state["counter"] * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF


# Device-side seed counter: every dropout site passes (launch-time seed, pointer to this counter) and the kernels add the two when
# they RUN.  Eagerly the counter stays 0 and the host seeds differ per call; in a captured training step (graph.py) the host seeds
# are frozen into the graph and ``advance_seed_dev()`` -- itself a node of the graph -- makes every replay draw fresh masks, with the
# forward and backward kernels of one replay still agreeing on them.
_seed_dev = {}


def seed_dev(device):
    t = _seed_dev.get(device)
    if t is None:
        t = _seed_dev[device] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


def advance_seed_dev(device):
    seed_dev(device).add_(0x2545F4914F6CDD1D)        # odd increment: the 64-bit counter cycles through all values


# ----------------------------------------------------------------------------- raw kernel wrappers
def gemm(A, a_layout, B, b_layout, C_out, M, N, K, *, lda=None, ldb=None, ldc=None, bias=None, act=0, aux_out=None,
         mul_gelu_z=None, dropout_p=0.0, dropout_seed=0, residual=None, ldr=None, alpha=1.0, alpha_dev=None, accumulate=False,
         split_k=1):
    e = GemmEpilogue()
    e.bias = ptr(bias) if bias is not None else None
    e.act = act
    e.aux_out = ptr(aux_out) if aux_out is not None else None
    e.mul_gelu_z = ptr(mul_gelu_z) if mul_gelu_z is not None else None
    e.dropout_p = dropout_p
    e.dropout_seed = dropout_seed
    e.dropout_seed_dev = seed_dev(C_out.device).data_ptr() if dropout_p > 0 else None
    e.residual = ptr(residual) if residual is not None else None
    e.ldr = ldr if ldr is not None else (residual.stride(0) if residual is not None else 0)
    e.alpha = alpha
    e.alpha_dev = ptr(alpha_dev) if alpha_dev is not None else None
    e.out_dtype = VM_F32 if C_out.dtype == torch.float32 else VM_BF16
    e.accumulate = 1 if accumulate else 0
    e.split_k = split_k
    if split_k > 1:
        ws = _workspace(split_k * M * (ldc if ldc is not None else C_out.stride(0)) * 4, C_out.device)
        e.workspace = ptr(ws)
        e.workspace_bytes = ws.numel()
    lda = lda if lda is not None else A.stride(0)
    ldb = ldb if ldb is not None else B.stride(0)
    ldc = ldc if ldc is not None else C_out.stride(0)
    check(lib().vm_gemm_bf16(ptr(A), lda, a_layout, ptr(B), ldb, b_layout, ptr(C_out), ldc, M, N, K, C.byref(e), stream()),
          "vm_gemm_bf16")
    return C_out


def gemm_grouped(problems, a_layout, b_layout, accumulate=False):
    """[(A, B, C_out, M, N, K), ...] independent products of one layout, 8 per launch (vm_gemm_grouped); shapes the grouped kernels
    cannot take (K % 64, alignment) run as single vm_gemm_bf16 launches instead"""
    arr = (_lib.GemmProblem * len(problems))()
    ok = True
    for q, (A, B, Cm, M, N, K) in zip(arr, problems):
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), Cm.data_ptr(), Cm.stride(0), M, N, K
        ok = ok and K % 64 == 0 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0 and Cm.stride(0) % 8 == 0 and not (a_layout == 1 and b_layout == 0) \
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0 and Cm.data_ptr() % 16 == 0
    if ok:
        out = VM_F32 if problems[0][2].dtype == torch.float32 else VM_BF16
        check(lib().vm_gemm_grouped(arr, len(problems), a_layout, b_layout, out, int(accumulate), stream()), "vm_gemm_grouped")
    else:
        for A, B, Cm, M, N, K in problems:
            gemm(A, a_layout, B, b_layout, Cm, M, N, K, accumulate=accumulate)


_ws_cache = {}


def _workspace(nbytes, device):
    """persistent split-K scratch (caller-provided per the C ABI); grows monotonically, one per (device, stream):
    launches that share it are ordered by their stream"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# ----------------------------------------------------------------------------- side stream for parameter gradients
# In a backward pass only the activation-gradient (dgrad) chain is on the critical path: the weight / bias gradients are
# consumed by the optimizer alone.  They are therefore launched on a second HIP stream, ordered after the kernel that
# produced their inputs; the GPU then fills the thin last round of workgroups of a dgrad GEMM with wgrad workgroups
# (and vice versa) instead of idling CUs.  The main stream re-joins the side stream when the backward pass ends
# (autograd engine callback) -- or on join_side() for launches made outside a backward pass.
_side = {"streams": {}, "pending": False, "callback_queued": False, "defer": False}
SIDE_STREAM = os.environ.get("VM_SIDE_STREAM", "1") != "0"


def _side_stream(device):
    st = _side["streams"].get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side["streams"][device] = st
    return st


def join_side():
    """main stream waits for everything queued on the side stream so far (no host sync); queued parameter gradients go first"""
    _side["callback_queued"] = False
    flush_param_grads()
    if _side["pending"]:
        for dev, st in _side["streams"].items():
            torch.cuda.current_stream(dev).wait_stream(st)
        _side["pending"] = False


@contextlib.contextmanager
def side_context(device, defer_join=None):
    """make the side stream the current stream (after everything enqueued on the main stream so far) -- for callers
    that order their own work behind the parameter-gradient kernels without stalling the main stream (ArenaDDP starts
    its gradient all-reduce from here).  ``defer_join`` sets/clears the flag that suppresses the automatic end-of-backward
    join; the caller then calls join_side() itself."""
    if defer_join is not None:
        _side["defer"] = bool(defer_join)
    if not SIDE_STREAM or torch.device(device).type != "cuda":
        yield
        return
    side = _side_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        yield
    _side["pending"] = True


@contextlib.contextmanager
def collective_context(device):
    """where ArenaDDP enqueues its gradient all-reduces.  Eagerly: the side stream (side_context), so the collectives are ordered behind the
    weight-gradient GEMMs without the main stream waiting for them.  While a HIP graph is being captured: the capturing stream itself, after it
    has joined the side stream -- a collective issued from a forked side stream inside a capture crashes at capture_end on this stack
    (ROCm 7.0 / RCCL 2.26; tools/ddp_graph_probe.py: sync, async and bf16 AVG all-reduces capture fine from the origin stream, any of them from the
    side stream segfaults); RCCL's own stream is still forked from and joined into the graph by the async work handle, so the all-reduce
    overlaps whatever the capturing stream enqueues after it."""
    if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
        join_side()
        yield
        return
    with side_context(device):
        yield


@contextlib.contextmanager
def on_side(*inputs):
    """launch the enclosed kernels on the side stream, after everything enqueued on the current stream so far.
    ``inputs`` are the tensors those kernels read: they are recorded on the side stream so the caching allocator does
    not hand their memory out again before the side stream is done with it."""
    if not SIDE_STREAM or not inputs:
        yield
        return
    dev = inputs[0].device
    side = _side_stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    for t in inputs:
        t.record_stream(side)
    with torch.cuda.stream(side):
        yield
    _side["pending"] = True
    if not _side["callback_queued"] and not _side["defer"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side)
            _side["callback_queued"] = True
        except RuntimeError:            # not inside a backward pass
            join_side()


WGRAD_SLOTS = int(os.environ.get("VM_WGRAD_SLOTS", "512"))


def _split_k_for(out_tiles, k_tiles):
    """as many K-splits as fit in ONE resident wave of workgroups (256 CUs x 2): tiles * split <= 512 -- rounding up
    instead would start a nearly empty second round (measured: 576 workgroups run 1.5x longer than 432) -- while
    keeping >= 8 K-tiles per split"""
    want = max(1, WGRAD_SLOTS // max(1, out_tiles))
    return max(1, min(want, max(1, k_tiles // 8)))


def wgrad(dY, X, dW, *, ld_dy=None, ld_x=None, alpha_dev=None, n_out=None):
    """dW[N,K] += dY[M,N]^T @ X[M,K]   (fp32 accumulate, split-K over M): the single-problem form (shapes the grouped launch
    cannot take: rows % 64 != 0)."""
    M, N = dY.shape
    N = n_out if n_out is not None else N
    K = X.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    gemm(dY, 1, X, 1, dW, N, K, M, lda=ld_dy or dY.stride(0), ldb=ld_x or X.stride(0), ldc=dW.stride(0),
         accumulate=True, split_k=_split_k_for(tiles, (M + 63) // 64), alpha_dev=alpha_dev)


def colsum(x, out, rows=None, cols=None, scale_dev=None):
    rows = rows if rows is not None else x.shape[0]
    cols = cols if cols is not None else x.shape[1]
    check(lib().vm_colsum_bf16(ptr(x), x.stride(0), ptr(out), rows, cols, ptr(scale_dev) if scale_dev is not None else None,
                               stream()), "vm_colsum_bf16")


# ----------------------------------------------------------------------------- backward marks
# An identity op whose backward tells a registered listener "the backward pass has just passed this point".  Models put marks between
# their layers (nn.ViTModel: in front of every encoder layer); ArenaDDP listens and starts the gradient all-reduce of the layers behind
# a mark while the layers in front of it are still being differentiated.  Without a listener no mark is inserted (single-GPU runs).
_bwd_mark = {"cb": None}


class Fn(torch.autograd.Function):
    """``torch.autograd.Function`` whose ``apply`` goes straight to the C++ implementation.  The stock Python ``apply`` first
    binds default arguments and unwraps functorch wrappers of every argument (``_functorch.utils.unwrap_dead_wrappers``): 7 % of
    the host time of a training step here (199 applies of ~10 arguments each; ``tools/host_profile.py``), for transforms this
    package never runs under."""

    @classmethod
    def apply(cls, *args):
        return super(torch.autograd.Function, cls).apply(*args)


class _BackwardMark(Fn):
    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        cb = _bwd_mark["cb"]
        if cb is not None:
            cb(ctx.tag)
        return g, None


def backward_mark(x, tag):
    if _bwd_mark["cb"] is None or not torch.is_grad_enabled() or not x.requires_grad:
        return x
    return _BackwardMark.apply(x, tag)


# ----------------------------------------------------------------------------- grouped parameter gradients
# Weight / bias gradients feed only the optimizer, so nothing in a backward pass waits for them.  Instead of launching one
# split-K GEMM (+ slab reduce) and one column-sum kernel per nn.Linear as its backward node runs, the backward nodes QUEUE
# (dY, X, dW, db) and the queue is flushed as ONE grouped launch (vm_wgrad_grouped) once it holds enough output tiles to fill
# the chip without splitting the contraction -- typically the 4-6 linears of one transformer layer -- on the side stream, so it
# still overlaps the activation-gradient chain.  Flushed at the latest when the backward pass ends, before a gradient
# all-reduce starts (ArenaDDP) and before the optimizer reads the gradients.
GROUP_WGRAD = os.environ.get("VM_WGRAD_GROUP", "1") != "0"
# flush threshold in 128 x 128 output tiles.  The wide-tile kernel (csrc/gemm_p8w.hip: 256 x 256 tiles, one workgroup per CU, up to 16 problems per
# launch) wants ONE full round of the 256 CUs: the linears of two transformer layers (encoder 2 x 432 = 864 -> 216 tiles of 256 x 256, decoder
# 2 x 504 = 1008 -> 252); the 128 x 128-tile kernel 400 (512 resident workgroups, VM_WGRAD_P8=0)
GROUP_TILES = int(os.environ.get("VM_WGRAD_GROUP_TILES", "400" if os.environ.get("VM_WGRAD_P8", "2") == "0" else "840"))
# [r6] flush policy of the wide-tile kernel.  It is ONE persistent workgroup per CU (160 KB of LDS, all 512 registers of a SIMD lane) walking the
# launch's 256 x 256 tiles: a launch costs ceil(tiles / 256) rounds whatever the tiles belong to -- and while it runs, no main-stream workgroup can
# share a CU with it.  Three rules, measured on one box (profiles/r06_d_wgrad_flush_policy.txt; ms per step / GEMM family alone):
#   threshold  flush the first time the queue holds >= 840 tiles of 128 x 128 (rounds 4-5): launches of 216-243 tiles (84-95 % of a round), the LM
#              head alone (360 = two rounds at 70 %), the patch embedding's 9 tiles alone (164 us at 4 % of the chip)          23.79 / 17.43
#   fill       flush BEFORE the problem that would open a new round once the queue fills its rounds to >= 0.9, never mix contraction lengths:
#              252-tile launches, LM head + last decoder layer 486 / 512; best kernels, but a full round leaves the activation-gradient chain
#              on the main stream no CU at all while it runs                                                                      23.75 / 16.95
#   hybrid     the threshold rule, except that a launch is not cut while it fills its rounds to < 0.8 (the LM head takes the last layer's FC
#              gradients along: 432 / 512), and a flush of < 16 tiles goes to the split-K GEMM instead (patch embedding)          23.63 / 17.31
# The step is CU-time-bound either way; "hybrid" (default) keeps ~40 CUs free for the main stream in most launches.  VM_WGRAD_FLUSH picks the rule.
WGRAD_FLUSH_MODE = os.environ.get("VM_WGRAD_FLUSH", "hybrid") if os.environ.get("VM_WGRAD_P8", "2") != "0" else "threshold"
WGRAD_FLUSH_FILL = WGRAD_FLUSH_MODE == "fill"
WGRAD_FLUSH_HYBRID = WGRAD_FLUSH_MODE == "hybrid"      # the threshold rule, but a launch is not cut while it fills its rounds to < 0.8 (the LM head alone)
WGRAD_FILL = float(os.environ.get("VM_WGRAD_FILL", "0.9"))
WGRAD_MAX_PROBLEMS = 16                          # P8W_MAX_GROUP of csrc/gemm_args.h: more problems would split the launch
WGRAD_TRACE = os.environ.get("VM_WGRAD_TRACE", "") == "1"
_pg = {"items": [], "tiles": 0, "ptrs": set(), "t256": 0}
WGRAD_CHECK = os.environ.get("VM_WGRAD_CHECK", "") == "1"
# first-touch tracking: a gradient buffer that no kernel has written since its arena zeroed the gradients may be STORED instead of accumulated
# (vm_wgrad_problem.overwrite: the read half of 0.9 GB of fp32 read-modify-write per step).  ``touched``: buffers written since then;
# ``shared``: buffers other kernels add into (the tied word embedding: vm_embedding_bwd) -- always accumulated; ``ranges``: arenas whose gradients
# have been zeroed through ParamArena.zero_grad at least once (before that nothing is known about a buffer's contents).
_touch = {"lo": [], "hi": [], "shared": [], "ranges": []}      # touched intervals (sorted, disjoint starts) / shared intervals / tracked arenas


def _span(t):
    """[lo, hi) byte interval a tensor's elements lie in (a column slice of a wider matrix spans its rows' strides, not numel)"""
    lo = t.data_ptr()
    if t.is_contiguous() or t.numel() == 0:                      # (the common case, asked ~250 times per step)
        return lo, lo + t.numel() * t.element_size()
    last = sum((n - 1) * st for n, st in zip(t.shape, t.stride()))
    return lo, lo + (last + 1) * t.element_size()


def forget_range(lo, hi):
    """an arena whose gradient buffer covered [lo, hi) is gone: nothing recorded about that memory may vouch for whatever the caching
    allocator hands out there next (ParamArena registers this as its finalizer)"""
    _touch["ranges"] = [r for r in _touch["ranges"] if r[1] <= lo or r[0] >= hi]
    _touch["shared"] = [r for r in _touch["shared"] if r[1] <= lo or r[0] >= hi]
    keep = [(a, b) for a, b in zip(_touch["lo"], _touch["hi"]) if b <= lo or a >= hi]
    _touch["lo"], _touch["hi"] = [a for a, _ in keep], [b for _, b in keep]


def grads_zeroed(gflat):
    """ParamArena.zero_grad() just zeroed ``gflat``: every gradient view inside it is clean again"""
    lo, hi = _span(gflat)
    if (lo, hi) not in _touch["ranges"]:
        _touch["ranges"] = [r for r in _touch["ranges"] if r[1] <= lo or r[0] >= hi] + [(lo, hi)]
    keep = []
    for a, b in zip(_touch["lo"], _touch["hi"]):          # (an interval that reaches beyond the arena keeps its outside parts)
        if b <= lo or a >= hi:
            keep.append((a, b))
        else:
            if a < lo:
                keep.append((a, lo))
            if b > hi:
                keep.append((hi, b))
    _touch["lo"], _touch["hi"] = [a for a, _ in keep], [b for _, b in keep]


def _touch_insert(lo, hi):
    """add [lo, hi) to the touched set, kept as SORTED DISJOINT intervals (overlapping ones are merged: a view nested in an earlier, larger
    one must not hide that one from a later neighbour test)"""
    los, his = _touch["lo"], _touch["hi"]
    i = bisect.bisect_left(los, lo)
    if i > 0 and his[i - 1] > lo:
        i -= 1
        lo, hi = los[i], max(hi, his[i])
        del los[i], his[i]
    while i < len(los) and los[i] < hi:
        hi = max(hi, his[i])
        del los[i], his[i]
    los.insert(i, lo)
    his.insert(i, hi)


def _touch_overlaps(lo, hi):
    los, his = _touch["lo"], _touch["hi"]
    i = bisect.bisect_right(los, lo)
    return (i > 0 and his[i - 1] > lo) or (i < len(los) and los[i] < hi)


def mark_touched(t):
    """``t`` (a gradient buffer) has been or will be written by something else than an overwriting weight-gradient GEMM"""
    _touch_insert(*_span(t))


def _first_touch(t):
    """True at most once per zeroing, for a buffer inside a tracked arena that overlaps nothing written since (intervals, not start pointers:
    a fused Q|K|V gradient view and the view of K alone are the same memory)"""
    lo, hi = _span(t)
    clean = not _touch_overlaps(lo, hi)
    if clean:
        clean = not any(a < hi and lo < b for a, b in _touch["shared"])
    _touch_insert(lo, hi)
    return clean and any(a <= lo and hi <= b for a, b in _touch["ranges"])


def _eligible(dY, X, dW, ld_dy, ld_x):
    return (dY.shape[0] % 64 == 0 and ld_dy % 8 == 0 and ld_x % 8 == 0 and dW.stride(0) % 8 == 0 and dY.data_ptr() % 16 == 0
            and X.data_ptr() % 16 == 0 and dW.data_ptr() % 16 == 0 and dW.dtype == torch.float32)


def param_grads(dY, X, dW, db=None, *, ld_dy=None, ld_x=None, alpha_dev=None, cols=None):
    """dW[N,K] += dY[M,N]^T X[M,K]  and  db[N] += colsum(dY)   (either may be None); ``cols``: leading columns of dY that count
    (the LM head's padded vocabulary).  Queued for the next grouped launch when possible, launched on the side stream otherwise."""
    if dW is None and db is None:
        return
    N = cols if cols is not None else dY.shape[1]
    ld_dy = ld_dy or dY.stride(0)
    ld_x = ld_x or (X.stride(0) if X is not None else 0)
    if GROUP_WGRAD and dW is not None and _eligible(dY, X, dW, ld_dy, ld_x):
        if dW.data_ptr() in _pg["ptrs"]:            # the same parameter twice in one backward graph: never in one launch (two owners of a tile)
            flush_param_grads()
        K = X.shape[1]
        t256 = ((N + 255) // 256) * ((K + 255) // 256)
        if WGRAD_FLUSH_FILL and _pg["items"]:
            have = _pg["t256"]
            rounds = (have + 255) // 256
            # (a problem of another contraction length -- the decoder's 8192 rows / the encoder's 12608 -- never joins the queue: a workgroup's
            #  tiles are dealt round-robin, and a launch that mixes 128- and 197-K-tile outputs runs as long as its longest pairing)
            if (len(_pg["items"]) >= WGRAD_MAX_PROBLEMS or dY.shape[0] != _pg["items"][0][0].shape[0]
                    or (have + t256 > 256 * rounds and have >= WGRAD_FILL * 256 * rounds)):
                flush_param_grads()                 # the queue fills its rounds: the new problem starts the next launch
        fw = _first_touch(dW)
        fb = _first_touch(db) if db is not None else True
        if not (fw and fb):                         # one flag for both outputs of a problem; accumulating into a clean buffer is always right
            fw = False
        _pg["items"].append((dY, X, dW, db, ld_dy, ld_x, alpha_dev, N, K, fw))
        _pg["ptrs"].add(dW.data_ptr())
        _pg["tiles"] += ((N + 127) // 128) * ((K + 127) // 128)
        _pg["t256"] += t256
        if not WGRAD_FLUSH_FILL and _pg["tiles"] >= GROUP_TILES and not (
                WGRAD_FLUSH_HYBRID and _pg["t256"] < 0.8 * 256 * ((_pg["t256"] + 255) // 256) and len(_pg["items"]) < WGRAD_MAX_PROBLEMS):
            flush_param_grads()
        else:
            _ensure_end_of_backward_flush()
        return
    for t in (dW, db):
        if t is not None:
            mark_touched(t)
    with on_side(*(t for t in (dY, X, alpha_dev) if t is not None)):
        if dW is not None:
            wgrad(dY, X, dW[:N] if dW.shape[0] != N else dW, ld_dy=ld_dy, ld_x=ld_x, alpha_dev=alpha_dev, n_out=N)
        if db is not None:
            colsum(dY, db, rows=dY.shape[0], cols=N, scale_dev=alpha_dev)


_lnq = {"items": [], "ptrs": set()}


def flush_ln_reduce():
    """dgamma / dbeta of every LayerNorm backward queued so far: one batched reduce launch on the side stream"""
    items = _lnq["items"]
    if not items:
        return
    _lnq["items"], _lnq["ptrs"] = [], set()
    arr = (_lib.LnReduceProblem * len(items))()
    for q, (ws, gg, gb, rows, cols) in zip(arr, items):
        q.ws, q.dgamma, q.dbeta, q.rows, q.cols = ws.data_ptr(), gg.data_ptr(), gb.data_ptr(), rows, cols
    with on_side(*(it[0] for it in items)):
        check(lib().vm_layernorm_bwd_reduce_batched(arr, len(items), stream()), "vm_layernorm_bwd_reduce_batched")


def flush_param_grads():
    """launch everything queued by param_grads() as one grouped weight-gradient GEMM, and the queued LayerNorm reduces (side stream)"""
    flush_ln_reduce()
    items = _pg["items"]
    if not items:
        return
    if WGRAD_TRACE:
        import sys
        print(f"[wgrad flush] {len(items)} problems, {_pg['t256']} tiles of 256 x 256 ({_pg['t256'] / (256 * max(1, (_pg['t256'] + 255) // 256)):.2f} of "
              f"{max(1, (_pg['t256'] + 255) // 256)} round(s)): " + " ".join(f"{it[7]}x{it[8]}/{it[0].shape[0]}" for it in items), file=sys.stderr)
    lonely = (WGRAD_FLUSH_FILL or WGRAD_FLUSH_HYBRID) and _pg["t256"] < 16
    _pg["items"], _pg["tiles"], _pg["ptrs"], _pg["t256"] = [], 0, set(), 0
    if lonely:
        # [r6] a flush that holds a handful of tiles (the ViT patch embedding's 768 x 768 gradient over 12544 rows, alone behind the last encoder
        # layer: 9 tiles) would walk its whole contraction with 4 % of the chip -- 164 us on the grouped kernel; the split-K GEMM + column sum take ~30
        for dY, X, dW, db, ld_dy, ld_x, alpha_dev, N, K, first in items:
            with on_side(*(t for t in (dY, X, alpha_dev) if t is not None)):
                wgrad(dY, X, dW[:N] if dW.shape[0] != N else dW, ld_dy=ld_dy, ld_x=ld_x, alpha_dev=alpha_dev, n_out=N)
                if db is not None:
                    colsum(dY, db, rows=dY.shape[0], cols=N, scale_dev=alpha_dev)
        return
    arr = (_lib.WgradProblem * len(items))()
    tensors = []
    if WGRAD_CHECK:             # VM_WGRAD_CHECK=1 (debug): a buffer about to be STORED into must still hold the zeros of zero_grad
        for dY, X, dW, db, ld_dy, ld_x, alpha_dev, N, K, first in items:
            if first and (float(dW[:N].abs().max()) != 0.0 or (db is not None and float(db[:N].abs().max()) != 0.0)):
                raise RuntimeError("first-touch bookkeeping: an overwriting weight-gradient launch targets a buffer that is not zero "
                                   f"(dW {tuple(dW.shape)} at {dW.data_ptr():#x}): some writer did not call ops.mark_touched")
    for q, (dY, X, dW, db, ld_dy, ld_x, alpha_dev, N, K, first) in zip(arr, items):
        q.dY, q.ld_dy, q.X, q.ld_x = dY.data_ptr(), ld_dy, X.data_ptr(), ld_x
        q.dW, q.ld_dw, q.db = dW.data_ptr(), dW.stride(0), (db.data_ptr() if db is not None else None)
        q.rows, q.n_out, q.k_in = dY.shape[0], N, K
        q.overwrite = 1 if first else 0
        q.alpha_dev = alpha_dev.data_ptr() if alpha_dev is not None else None
        tensors += [t for t in (dY, X, alpha_dev) if t is not None]
    with on_side(*tensors):
        check(lib().vm_wgrad_grouped(arr, len(items), stream()), "vm_wgrad_grouped")


def reset_host_state():
    """forget everything queued on the host for launches that will never happen (an aborted graph capture: graph.GraphedTrainStep):
    the weight-gradient and LayerNorm-reduce queues, the pending side-stream join, the masked-gradient hand-off table; the first-touch
    records of every tracked arena are invalidated (nothing is known about the buffers until the next zero_grad)."""
    _pg["items"], _pg["tiles"], _pg["ptrs"], _pg["t256"] = [], 0, set(), 0
    _lnq["items"], _lnq["ptrs"] = [], set()
    _masked.clear()
    _side["pending"], _side["callback_queued"], _side["defer"] = False, False, False
    for lo, hi in _touch["ranges"]:
        _touch_insert(lo, hi)                    # everything counts as touched: weight gradients accumulate until the arena is zeroed again


def _ensure_end_of_backward_flush():
    """something is queued: make sure it is flushed (and the side stream joined) when the running backward pass ends"""
    if _side["callback_queued"] or _side["defer"]:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join_side)
        _side["callback_queued"] = True
    except RuntimeError:                # not inside a backward pass: do it now
        join_side()


def cast_to_bf16(src, dst=None):
    dst = dst if dst is not None else torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib().vm_cast_f32_to_bf16(ptr(src), ptr(dst), src.numel(), stream()), "vm_cast_f32_to_bf16")
    return dst


def cast_to_f32(src, dst=None):
    dst = dst if dst is not None else torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(lib().vm_cast_bf16_to_f32(ptr(src), ptr(dst), src.numel(), stream()), "vm_cast_bf16_to_f32")
    return dst


def dropout_apply(x, p, seed):
    out = torch.empty_like(x)
    check(lib().vm_dropout_apply_bf16(ptr(x), ptr(out), x.numel(), p, seed, ptr(seed_dev(x.device)), stream()), "vm_dropout_apply_bf16")
    return out


def add_bf16(a, b):
    out = torch.empty_like(a)
    check(lib().vm_add_bf16(ptr(a), ptr(b), ptr(out), a.numel(), stream()), "vm_add_bf16")
    return out


def feature_mask(feats2d):
    rows, cols = feats2d.shape
    m = torch.empty(rows, dtype=torch.uint8, device=feats2d.device)
    check(lib().vm_feature_mask(ptr(feats2d), ptr(m), rows, cols, stream()), "vm_feature_mask")
    return m


def _grad_buf(p):
    """fp32 accumulation buffer of a parameter (its arena ``.grad`` view; re-created if a caller set it to None)."""
    if p.grad is None:
        view = getattr(p, "_vm_grad_view", None)
        if view is not None:
            view.zero_()
            p.grad = view
        else:
            p.grad = torch.zeros_like(p)
    return p.grad


def _2d(x):
    return x.reshape(-1, x.shape[-1])


# ----------------------------------------------------------------------------- Linear (+bias, +dropout, +residual)
class LinearFn(Fn):
    """y = dropout(x W^T + b) + residual.   ``w_sh`` is the bf16 shadow [N,K] of the fp32 parameter(s);
    ``w_params`` / ``b_params`` are lists of the fp32 parameters whose contiguous arena grads receive dW / db
    (several when Q,K,V projections are fused into one GEMM)."""

    @staticmethod
    def forward(ctx, x, w_sh, bias_f32, residual, dropout_p, wgrad_buf, bgrad_buf, anchor, out_f32=False, seed=0):
        x2 = _2d(x)
        M, K = x2.shape
        N = w_sh.shape[0]
        y = torch.empty(M, N, dtype=torch.float32 if out_f32 else BF16, device=x.device)      # fp32: straight from the accumulators
        gemm(x2, 0, w_sh, 0, y, M, N, K, bias=bias_f32, dropout_p=dropout_p, dropout_seed=seed,
             residual=_2d(residual) if residual is not None else None)
        ctx.save_for_backward(x2, w_sh)
        ctx.meta = (dropout_p, seed, wgrad_buf, bgrad_buf, residual is not None, x.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w_sh = ctx.saved_tensors
        dropout_p, seed, wgrad_buf, bgrad_buf, has_res, xshape = ctx.meta
        dy2 = _2d(dy.contiguous())
        if dy2.dtype != BF16:               # fp32 output (projection heads): the gradient GEMMs take bf16 operands
            dy2 = dy2.to(BF16)
        M, N = dy2.shape
        K = x2.shape[1]
        dpre = _masked_grad(dy2, dropout_p, seed) if dropout_p > 0 else dy2
        param_grads(dpre, x2, wgrad_buf, bgrad_buf)       # queued: grouped launch on the side stream, overlapping the dgrad chain
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=BF16, device=dy.device)
            gemm(dpre, 0, w_sh, 1, dx, M, K, N)
            dx = dx.view(xshape)
        return dx, None, None, (dy if has_res else None), None, None, None, None, None, None


def linear(x, w_sh, bias, *, residual=None, dropout_p=0.0, wgrad_buf=None, bgrad_buf=None, anchor=None, out_f32=False):
    seed = next_seed() if dropout_p > 0 else 0
    y = LinearFn.apply(x, w_sh, bias, residual, dropout_p, wgrad_buf, bgrad_buf, anchor, out_f32, seed)
    if dropout_p > 0:
        y._vm_drop = (dropout_p, seed)         # read by layer_norm(): its backward then also writes the masked gradient (_masked_grad)
    return y


# The gradient of y = dropout(linear(x)) + residual arrives from the LayerNorm backward that consumed y; that kernel can write the masked
# copy  keep ? dy / (1 - p) : 0  next to dy (vm_layernorm_bwd_partial_dropout), which saves one pass over dy per linear.  LayerNormFn
# registers the masked tensor under the address of the dy it returns; the linear's backward picks it up when exactly that tensor arrives.
_masked = {}


def _masked_grad(dy2, dropout_p, seed):
    hit = _masked.pop(dy2.data_ptr(), None)
    if hit is not None and hit[1] == (dropout_p, seed) and hit[0].shape == dy2.shape:
        return hit[0]
    return dropout_apply(dy2, dropout_p, seed)


# ----------------------------------------------------------------------------- MLP: FC1 + erf-GELU + FC2 (+dropout) + residual
class MlpFn(Fn):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, dropout_p, g_w1, g_b1, g_w2, g_b2, anchor, seed=0):
        x2 = _2d(x)
        M, K = x2.shape
        F = w1.shape[0]
        z = torch.empty(M, F, dtype=BF16, device=x.device)
        a = torch.empty(M, F, dtype=BF16, device=x.device)
        gemm(x2, 0, w1, 0, a, M, F, K, bias=b1, act=1, aux_out=z)
        y = torch.empty(M, K, dtype=BF16, device=x.device)
        gemm(a, 0, w2, 0, y, M, K, F, bias=b2, dropout_p=dropout_p, dropout_seed=seed,
             residual=_2d(residual) if residual is not None else None)
        ctx.save_for_backward(x2, z, a, w1, w2)
        ctx.meta = (dropout_p, seed, g_w1, g_b1, g_w2, g_b2, residual is not None, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, z, a, w1, w2 = ctx.saved_tensors
        dropout_p, seed, g_w1, g_b1, g_w2, g_b2, has_res, xshape = ctx.meta
        dy2 = _2d(dy.contiguous())
        M, K = dy2.shape
        F = w1.shape[0]
        dpre = _masked_grad(dy2, dropout_p, seed) if dropout_p > 0 else dy2
        param_grads(dpre, a, g_w2, g_b2)
        dz = torch.empty(M, F, dtype=BF16, device=dy.device)
        gemm(dpre, 0, w2, 1, dz, M, F, K, mul_gelu_z=z)          # da * gelu'(z) fused in the dgrad epilogue
        param_grads(dz, x2, g_w1, g_b1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=BF16, device=dy.device)
            gemm(dz, 0, w1, 1, dx, M, K, F)
            dx = dx.view(xshape)
        return dx, None, None, None, None, (dy if has_res else None), None, None, None, None, None, None, None


def mlp(x, w1, b1, w2, b2, *, residual=None, dropout_p=0.0, grads=(None, None, None, None), anchor=None):
    seed = next_seed() if dropout_p > 0 else 0
    y = MlpFn.apply(x, w1, b1, w2, b2, residual, dropout_p, *grads, anchor, seed)
    if dropout_p > 0:
        y._vm_drop = (dropout_p, seed)
    return y


# ----------------------------------------------------------------------------- LayerNorm
_ln_ws = {}


def _ln_ws_floats(rows, cols):
    """workspace floats of the LayerNorm backward for this shape (a pure function of the shape: asked once, not per launch)"""
    n = _ln_ws.get((rows, cols))
    if n is None:
        n = _ln_ws[(rows, cols)] = lib().vm_layernorm_bwd_ws(rows, cols) // 4
    return n


class LayerNormFn(Fn):
    """y = LN(x).  ``fork`` fuses the gradient sum of a residual fork into the backward kernel instead of leaving it to
    an autograd elementwise add:
      fork="in"  (pre-LN block):  returns (y, x_alias); the block feeds x_alias to its residual, so x has ONE consumer
                 and backward gets (dy, dres):  dx = LN'(dy) + dres;
      fork="out" (post-LN block): returns (y, y_alias); the next sub-layer reads y, its residual reads y_alias, and
                 backward gets two gradients of y that are summed in fp32 inside the kernel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, g_gamma, g_beta, fork, drop=None):
        x2 = _2d(x)
        rows, cols = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib().vm_layernorm_fwd(ptr(x2), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps, stream()),
              "vm_layernorm_fwd")
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.meta = (g_gamma, g_beta, x.shape, fork, drop)
        ctx.set_materialize_grads(False)
        y = y.view(x.shape)
        if fork == "in":
            return y, x.view(x.shape)
        if fork == "out":
            return y, y.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, d2=None):
        x2, gamma, mean, rstd = ctx.saved_tensors
        g_gamma, g_beta, xshape, fork, drop = ctx.meta
        rows, cols = x2.shape
        dres = None
        if fork == "in":
            dres, d2 = d2, None
        if dy is None and d2 is not None:
            dy, d2 = d2, None
        if dy is None:                      # the normalised output was not used: only the pass-through gradient
            return dres, None, None, None, None, None, None, None
        dy2 = _2d(dy.contiguous())
        dx = torch.empty_like(x2)
        ws = torch.empty(_ln_ws_floats(rows, cols), dtype=torch.float32, device=dy.device)
        if g_gamma is None:   # frozen affine: still need dx; send the param grads to scratch
            g_gamma = torch.zeros(cols, dtype=torch.float32, device=dy.device)
            g_beta = torch.zeros(cols, dtype=torch.float32, device=dy.device)
        dxd = torch.empty_like(x2) if drop is not None else None       # masked copy for the linear that produced x (see _masked_grad)
        check(lib().vm_layernorm_bwd_partial_dropout(ptr(dy2), ptr(_2d(d2.contiguous())) if d2 is not None else None,
                                                     ptr(_2d(dres.contiguous())) if dres is not None else None,
                                                     ptr(x2), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dxd) if dxd is not None else None,
                                                     drop[0] if drop else 0.0, drop[1] if drop else 0,
                                                     ptr(seed_dev(dy.device)) if drop else None, rows, cols, ptr(ws), stream()),
              "vm_layernorm_bwd")
        if dxd is not None:
            if len(_masked) > 8:
                _masked.clear()
            _masked[dx.data_ptr()] = (dxd, drop)
        # the dgamma / dbeta reduction only feeds the optimizer: queued, and reduced together with the step's other LayerNorms in one
        # launch when the parameter-gradient queue is flushed (flush_param_grads)
        if g_gamma.data_ptr() in _lnq["ptrs"] or len(_lnq["items"]) >= 64:
            flush_ln_reduce()
        _lnq["items"].append((ws, g_gamma, g_beta, rows, cols))
        _lnq["ptrs"].add(g_gamma.data_ptr())
        _ensure_end_of_backward_flush()
        return dx.view(xshape), None, None, None, None, None, None, None


FUSE_LN_DROPOUT = os.environ.get("VM_LN_DROPOUT_FUSE", "1") != "0"     # 0: separate vm_dropout_apply_bf16 pass in the linear's backward (A/B, tests)


def layer_norm(x, gamma, beta, eps, g_gamma=None, g_beta=None, fork=None):
    return LayerNormFn.apply(x, gamma, beta, eps, g_gamma, g_beta, fork, getattr(x, "_vm_drop", None) if FUSE_LN_DROPOUT else None)


# ----------------------------------------------------------------------------- attention
class AttentionFn(Fn):
    """q [B,Lq,*] k,v [B,Lk,*] are column slices (views) of projection outputs; heads are contiguous 64-wide blocks."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, H, causal, dropout_p, scale=None):
        B, Lq = q.shape[0], q.shape[1]
        Lk = k.shape[1]
        dh = q.shape[2] // H
        o = torch.empty(B, Lq, H * dh, dtype=BF16, device=q.device)
        stats = torch.empty(B, H, Lq, 2, dtype=torch.float32, device=q.device)
        seed = next_seed() if dropout_p > 0 else 0
        scale = dh ** -0.5 if scale is None else scale
        check(lib().vm_attention_fwd(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), o.stride(1), ptr(stats),
                                     ptr(key_mask) if key_mask is not None else None, B, H, Lq, Lk, dh, scale, int(causal),
                                     dropout_p, seed, ptr(seed_dev(q.device)) if dropout_p > 0 else None, None, 0, stream()), "vm_attention_fwd")
        ctx.save_for_backward(q, k, v, o, stats, key_mask)
        ctx.meta = (H, causal, dropout_p, seed, scale)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, stats, key_mask = ctx.saved_tensors
        H, causal, dropout_p, seed, scale = ctx.meta
        B, Lq = q.shape[0], q.shape[1]
        Lk = k.shape[1]
        dh = q.shape[2] // H
        d_o = d_o.contiguous()
        # gradients are written in the same packed layout as the inputs (one buffer when q,k,v share storage rows)
        dq = torch.empty(B, Lq, H * dh, dtype=BF16, device=q.device)
        dk = torch.empty(B, Lk, H * dh, dtype=BF16, device=q.device)
        dv = torch.empty(B, Lk, H * dh, dtype=BF16, device=q.device)
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        check(lib().vm_attention_bwd(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), o.stride(1),
                                     ptr(d_o), d_o.stride(1), ptr(stats), ptr(key_mask) if key_mask is not None else None,
                                     ptr(dq), dq.stride(1), ptr(dk), dk.stride(1), ptr(dv), dv.stride(1),
                                     B, H, Lq, Lk, dh, scale, int(causal), dropout_p, seed, ptr(seed_dev(q.device)) if dropout_p > 0 else None, ptr(delta), stream()), "vm_attention_bwd")
        return dq, dk, dv, None, None, None, None, None


class PackedSelfAttentionFn(Fn):
    """Self-attention on the fused QKV projection output [B,L,3*D]; the gradient comes back packed [B,L,3*D]
    so the QKV dgrad/wgrad run as single GEMMs."""

    @staticmethod
    def forward(ctx, qkv, key_mask, H, causal, dropout_p):
        B, L, D3 = qkv.shape
        D = D3 // 3
        dh = D // H
        o = torch.empty(B, L, D, dtype=BF16, device=qkv.device)
        stats = torch.empty(B, H, L, 2, dtype=torch.float32, device=qkv.device)
        seed = next_seed() if dropout_p > 0 else 0
        scale = dh ** -0.5
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        check(lib().vm_attention_fwd(ptr(q), D3, ptr(k), D3, ptr(v), D3, ptr(o), D, ptr(stats),
                                     ptr(key_mask) if key_mask is not None else None, B, H, L, L, dh, scale, int(causal),
                                     dropout_p, seed, ptr(seed_dev(q.device)) if dropout_p > 0 else None, None, 0, stream()), "vm_attention_fwd")
        ctx.save_for_backward(qkv, o, stats, key_mask)
        ctx.meta = (H, causal, dropout_p, seed, scale)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, stats, key_mask = ctx.saved_tensors
        H, causal, dropout_p, seed, scale = ctx.meta
        B, L, D3 = qkv.shape
        D = D3 // 3
        dh = D // H
        d_o = d_o.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, L, dtype=torch.float32, device=qkv.device)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
        check(lib().vm_attention_bwd(ptr(q), D3, ptr(k), D3, ptr(v), D3, ptr(o), D, ptr(d_o), D, ptr(stats),
                                     ptr(key_mask) if key_mask is not None else None, ptr(dq), D3, ptr(dk), D3, ptr(dv), D3,
                                     B, H, L, L, dh, scale, int(causal), dropout_p, seed, ptr(seed_dev(q.device)) if dropout_p > 0 else None, ptr(delta), stream()), "vm_attention_bwd")
        return dqkv, None, None, None, None


class PackedCrossAttentionFn(Fn):
    """q [B,Lq,D] from the decoder, kv [B,Lk,2*D] = fused K|V projection of the encoder features (any row stride: the
    view of one layer inside the all-layer projection of CrossKVAllFn).  ``dkv_out``: optional pre-allocated gradient
    slot with kv's shape (written in place and returned as kv's gradient)."""

    @staticmethod
    def forward(ctx, q, kv, key_mask, H, dropout_p, dkv_out):
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        dh = D // H
        if kv.stride(2) != 1 or kv.stride(0) != Lk * kv.stride(1):
            kv = kv.contiguous()
        ldkv = kv.stride(1)
        o = torch.empty(B, Lq, D, dtype=BF16, device=q.device)
        stats = torch.empty(B, H, Lq, 2, dtype=torch.float32, device=q.device)
        seed = next_seed() if dropout_p > 0 else 0
        scale = dh ** -0.5
        k, v = kv[..., :D], kv[..., D:]
        check(lib().vm_attention_fwd(ptr(q), D, ptr(k), ldkv, ptr(v), ldkv, ptr(o), D, ptr(stats),
                                     ptr(key_mask) if key_mask is not None else None, B, H, Lq, Lk, dh, scale, 0,
                                     dropout_p, seed, ptr(seed_dev(q.device)) if dropout_p > 0 else None, None, 0, stream()), "vm_attention_fwd")
        ctx.save_for_backward(q, kv, o, stats, key_mask)
        ctx.meta = (H, dropout_p, seed, scale, dkv_out)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, kv, o, stats, key_mask = ctx.saved_tensors
        H, dropout_p, seed, scale, dkv_out = ctx.meta
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        dh = D // H
        d_o = d_o.contiguous()
        dq = torch.empty_like(q)
        dkv = dkv_out if dkv_out is not None else torch.empty(B, Lk, 2 * D, dtype=BF16, device=q.device)
        ldkv, lddkv = kv.stride(1), dkv.stride(1)
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        k, v = kv[..., :D], kv[..., D:]
        dk, dv = dkv[..., :D], dkv[..., D:]
        check(lib().vm_attention_bwd(ptr(q), D, ptr(k), ldkv, ptr(v), ldkv, ptr(o), D, ptr(d_o), D, ptr(stats),
                                     ptr(key_mask) if key_mask is not None else None, ptr(dq), D, ptr(dk), lddkv, ptr(dv), lddkv,
                                     B, H, Lq, Lk, dh, scale, 0, dropout_p, seed, ptr(seed_dev(q.device)) if dropout_p > 0 else None, ptr(delta), stream()), "vm_attention_bwd")
        return dq, dkv, None, None, None, None


class CrossKVAllFn(Fn):
    """The K|V projections of ALL decoder layers' cross-attention in one GEMM: enc [B,S,De] x W_all^T [n*2D, De] -> one
    buffer [B*S, n*2D]; layer i reads the strided view [..., i*2D:(i+1)*2D].  Backward: every layer's attention kernel
    writes dK|dV straight into its slice of one gradient buffer, then ONE dgrad GEMM (contraction n*2D) yields d_enc --
    replacing n dgrad GEMMs plus n-1 gradient-accumulation adds on the encoder output -- and one wgrad GEMM / column
    sum yields all the weight / bias gradients (needs the layers' K,V parameters adjacent in the arena)."""

    @staticmethod
    def forward(ctx, enc, w_all, b_all, n_layers, wgrad_buf, bgrad_buf, anchor, need_grad):
        e2 = _2d(enc)
        M, K = e2.shape
        N = w_all.shape[0]
        kv = torch.empty(M, N, dtype=BF16, device=enc.device)
        gemm(e2, 0, w_all, 0, kv, M, N, K, bias=b_all)
        dkv = torch.empty(M, N, dtype=BF16, device=enc.device) if need_grad else None
        ctx.save_for_backward(e2, w_all)
        ctx.set_materialize_grads(False)      # no zero tensors for the (non-differentiable) gradient-slot outputs
        ctx.meta = (dkv, n_layers, wgrad_buf, bgrad_buf, enc.shape)
        B, S = enc.shape[0], enc.shape[1]
        per = N // n_layers
        outs = tuple(kv.view(B, S, N)[..., i * per:(i + 1) * per] for i in range(n_layers))
        if dkv is None:
            return outs + (None,) * n_layers
        slots = tuple(dkv.view(B, S, N)[..., i * per:(i + 1) * per] for i in range(n_layers))
        ctx.mark_non_differentiable(*slots)
        return outs + slots

    @staticmethod
    def backward(ctx, *grads):
        e2, w_all = ctx.saved_tensors
        dkv, n, wgrad_buf, bgrad_buf, eshape = ctx.meta
        M, K = e2.shape
        N = w_all.shape[0]
        per = N // n
        B, S = eshape[0], eshape[1]
        for i in range(n):                      # normally every gradient IS the slot (written in place by the attention backward)
            g, slot = grads[i], dkv.view(B, S, N)[..., i * per:(i + 1) * per]
            if g is None:
                slot.zero_()
            elif g.data_ptr() != slot.data_ptr():
                slot.copy_(g)
        param_grads(dkv, e2, wgrad_buf, bgrad_buf)
        d_enc = None
        if ctx.needs_input_grad[0]:
            d_enc = torch.empty(M, K, dtype=BF16, device=e2.device)
            gemm(dkv, 0, w_all, 1, d_enc, M, K, N)
            d_enc = d_enc.view(eshape)
        return d_enc, None, None, None, None, None, None, None


def cross_kv_all(enc, w_all, b_all, n_layers, wgrad_buf=None, bgrad_buf=None, anchor=None):
    """-> (list of n K|V views [B,S,2D], list of n gradient slots or Nones)"""
    need_grad = torch.is_grad_enabled() and (enc.requires_grad or (anchor is not None and anchor.requires_grad))
    out = CrossKVAllFn.apply(enc, w_all, b_all, n_layers, wgrad_buf, bgrad_buf, anchor, need_grad)
    return list(out[:n_layers]), list(out[n_layers:])


HEAD_DIMS = (32, 64, 96, 128)          # head widths the attention kernels are instantiated for


def padded_head_dim(dh):
    """the kernel head width that carries heads of ``dh`` columns (dh itself when supported)"""
    for w in HEAD_DIMS:
        if dh <= w:
            return w
    raise ValueError(f"head_dim {dh} > {HEAD_DIMS[-1]} is outside the HIP attention kernels")


def _attention_padded_heads(q, k, v, key_mask, H, causal, dropout_p):
    """heads narrower than a kernel head width (BertGenerationConfig's default 16 heads on hidden_size 768 = 48 columns:
    ref:config/RRG/baseline-HF.yml:26-30): every head is zero-padded to the next width -- QK^T and PV are unchanged by zero columns, the
    softmax scale stays dh^-1/2 -- and the context is cut back.  torch pads / slices (autograd routes the gradients); not a tuned path."""
    B, Lq, D = q.shape
    dh = D // H
    w = padded_head_dim(dh)

    def pad(t):
        return torch.nn.functional.pad(t.reshape(t.shape[0], t.shape[1], H, dh), (0, w - dh)).reshape(t.shape[0], t.shape[1], H * w)
    o = AttentionFn.apply(pad(q), pad(k), pad(v), key_mask, H, causal, dropout_p, dh ** -0.5)
    return o.view(B, Lq, H, w)[..., :dh].reshape(B, Lq, D)


def self_attention(qkv, key_mask, H, causal, dropout_p=0.0):
    D = qkv.shape[-1] // 3
    if D // H not in HEAD_DIMS:
        return _attention_padded_heads(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], key_mask, H, causal, dropout_p)
    return PackedSelfAttentionFn.apply(qkv, key_mask, H, causal, dropout_p)


def cross_attention(q, kv, key_mask, H, dropout_p=0.0, dkv_out=None):
    D = q.shape[-1]
    if D // H not in HEAD_DIMS:
        return _attention_padded_heads(q, kv[..., :D], kv[..., D:], key_mask, H, False, dropout_p)
    return PackedCrossAttentionFn.apply(q, kv, key_mask, H, dropout_p, dkv_out)


# ----------------------------------------------------------------------------- embeddings
class EmbeddingFn(Fn):
    @staticmethod
    def forward(ctx, anchor, ids, word, pos, past_len, padding_idx, g_word, g_pos):
        B, L = ids.shape
        D = word.shape[1]
        out = torch.empty(B, L, D, dtype=BF16, device=word.device)
        check(lib().vm_embedding_fwd(ptr(ids), ptr(word), ptr(pos), ptr(out), B, L, D, past_len, stream()), "vm_embedding_fwd")
        ctx.save_for_backward(ids)
        ctx.meta = (padding_idx, g_word, g_pos, D, past_len)
        if g_word is not None:
            sp = _span(g_word)
            if sp not in _touch["shared"]:
                _touch["shared"].append(sp)                 # the scatter-add of the backward pass shares this buffer with the tied LM head's weight gradient
        return out

    @staticmethod
    def backward(ctx, d_out):
        (ids,) = ctx.saved_tensors
        padding_idx, g_word, g_pos, D, past_len = ctx.meta
        B, L = ids.shape
        if g_word is not None:
            d_out = d_out.contiguous()
            gp = g_pos[past_len:] if past_len else g_pos
            mark_touched(g_word)
            # parameter gradients only: side stream, which also orders this scatter-add after the tied LM-head wgrad that
            # accumulates into the same g_word there (same stream -> no race, and the main stream never waits)
            with on_side(d_out, ids):
                check(lib().vm_embedding_bwd(ptr(ids), ptr(d_out), ptr(g_word), ptr(gp), B, L, D,
                                             padding_idx if padding_idx is not None else -1, stream()), "vm_embedding_bwd")
        return None, None, None, None, None, None, None, None


def embedding(anchor, ids, word, pos, *, past_len=0, padding_idx=None, g_word=None, g_pos=None):
    return EmbeddingFn.apply(anchor, ids, word, pos, past_len, padding_idx, g_word, g_pos)


class EmbeddingExFn(Fn):
    """BERT / RoBERTa embeddings: (word[ids] + type_row) + pos[pos_ids or t + past_len]  (vm_embedding_fwd_ex / _bwd_ex).
    ``pos_ids``: int64 [B, L] or None; ``type_row``: the fp32 parameter whose row 0 is added (or None); ``pos_pad``: position row
    that receives no gradient (RoBERTa's nn.Embedding(padding_idx), -1: none); ``pos_offset``: position of column t without pos_ids
    and the "regular" position the backward sums in registers."""

    @staticmethod
    def forward(ctx, anchor, ids, pos_ids, word, pos, type_w, past_len, padding_idx, pos_offset, pos_pad, g_word, g_pos, g_type):
        B, L = ids.shape
        D = word.shape[1]
        out = torch.empty(B, L, D, dtype=BF16, device=word.device)
        check(lib().vm_embedding_fwd_ex(ptr(ids), ptr(pos_ids), ptr(word), ptr(pos), ptr(type_w), ptr(out), VM_BF16, B, L, D, past_len,
                                        stream()), "vm_embedding_fwd_ex")
        ctx.save_for_backward(ids, pos_ids)
        ctx.meta = (padding_idx, pos_offset, pos_pad, g_word, g_pos, g_type, D)
        if g_word is not None:
            sp = _span(g_word)
            if sp not in _touch["shared"]:
                _touch["shared"].append(sp)                 # (as EmbeddingFn: the tied LM head's weight gradient accumulates into the same buffer)
        return out

    @staticmethod
    def backward(ctx, d_out):
        ids, pos_ids = ctx.saved_tensors
        padding_idx, pos_offset, pos_pad, g_word, g_pos, g_type, D = ctx.meta
        B, L = ids.shape
        if g_word is not None:
            d_out = d_out.contiguous()
            mark_touched(g_word)
            with on_side(d_out, ids, *([pos_ids] if pos_ids is not None else [])):
                check(lib().vm_embedding_bwd_ex(ptr(ids), ptr(pos_ids), ptr(d_out), ptr(g_word), ptr(g_pos), ptr(g_type), B, L, D,
                                                padding_idx if padding_idx is not None else -1, pos_offset, pos_pad, stream()),
                      "vm_embedding_bwd_ex")
        return (None,) * 13


def embedding_ex(anchor, ids, pos_ids, word, pos, type_w, *, past_len=0, padding_idx=None, pos_offset=0, pos_pad=-1,
                 g_word=None, g_pos=None, g_type=None):
    return EmbeddingExFn.apply(anchor, ids, pos_ids, word, pos, type_w, past_len, padding_idx, pos_offset, pos_pad, g_word, g_pos, g_type)


# ----------------------------------------------------------------------------- dense -> erf-GELU (LM-head transform of BERT / RoBERTa)
class LinearGeluFn(Fn):
    """y = gelu(x W^T + b); the pre-activation z is kept for the backward (dz = dy * gelu'(z): vm_gelu_bwd_bf16)."""

    @staticmethod
    def forward(ctx, x, w_sh, bias_f32, wgrad_buf, bgrad_buf, anchor):
        x2 = _2d(x)
        M, K = x2.shape
        N = w_sh.shape[0]
        z = torch.empty(M, N, dtype=BF16, device=x.device)
        y = torch.empty(M, N, dtype=BF16, device=x.device)
        gemm(x2, 0, w_sh, 0, y, M, N, K, bias=bias_f32, act=1, aux_out=z)
        ctx.save_for_backward(x2, w_sh, z)
        ctx.meta = (wgrad_buf, bgrad_buf, x.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w_sh, z = ctx.saved_tensors
        wgrad_buf, bgrad_buf, xshape = ctx.meta
        dy2 = _2d(dy.contiguous())
        M, N = dy2.shape
        K = x2.shape[1]
        dz = torch.empty_like(z)
        check(lib().vm_gelu_bwd_bf16(ptr(dy2), ptr(z), ptr(dz), dz.numel(), stream()), "vm_gelu_bwd_bf16")
        param_grads(dz, x2, wgrad_buf, bgrad_buf)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=BF16, device=dy.device)
            gemm(dz, 0, w_sh, 1, dx, M, K, N)
            dx = dx.view(xshape)
        return dx, None, None, None, None, None


def linear_gelu(x, w_sh, bias, *, wgrad_buf=None, bgrad_buf=None, anchor=None):
    return LinearGeluFn.apply(x, w_sh, bias, wgrad_buf, bgrad_buf, anchor)


# ----------------------------------------------------------------------------- ViT patch embedding (+cls +pos)
class PatchEmbedFn(Fn):
    """``cls``: fp32 [ns, D] special tokens in front of the patches (ViT: [CLS]; DeiT: [CLS], distillation), ``pos`` [(n + ns), D]."""

    @staticmethod
    def forward(ctx, anchor, images, w_sh, bias, cls, pos, patch, g_w, g_b, g_cls, g_pos, ns=1):
        B, Cc, Hh, Ww = images.shape
        n = (Hh // patch) * (Ww // patch)
        D, Kd = w_sh.shape
        cols = torch.empty(B * n, Kd, dtype=BF16, device=images.device)
        check(lib().vm_im2col_patches(ptr(images), ptr(cols), B, Cc, Hh, Ww, patch, stream()), "vm_im2col_patches")
        pe = torch.empty(B * n, D, dtype=BF16, device=images.device)
        gemm(cols, 0, w_sh, 0, pe, B * n, D, Kd, bias=bias)
        out = torch.empty(B, n + ns, D, dtype=BF16, device=images.device)
        check(lib().vm_vit_assemble_ex(ptr(pe), ptr(cls), ptr(pos), ptr(out), B, n, ns, D, stream()), "vm_vit_assemble")
        ctx.save_for_backward(cols)
        ctx.meta = (B, n, D, g_w, g_b, g_cls, g_pos, ns)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (cols,) = ctx.saved_tensors
        B, n, D, g_w, g_b, g_cls, g_pos, ns = ctx.meta
        if g_w is not None:
            d_out = d_out.contiguous()
            dpe = torch.empty(B * n, D, dtype=BF16, device=d_out.device)
            check(lib().vm_vit_assemble_bwd_ex(ptr(d_out), ptr(dpe), ptr(g_cls), ptr(g_pos), B, n, ns, D, stream()), "vm_vit_assemble_bwd")
            param_grads(dpe, cols, g_w.view(D, -1), g_b)
        return (None,) * 12


def patch_embed(anchor, images, w_sh, bias, cls, pos, patch, grads=(None, None, None, None), ns=1):
    return PatchEmbedFn.apply(anchor, images, w_sh, bias, cls, pos, patch, *grads, ns)


# ----------------------------------------------------------------------------- LM head + shifted CE (fused fwd/bwd)
class LmHeadLossFn(Fn):
    """logits = h E^T + b (bf16, padded leading dim); loss = sum_rows w_row CE(logits[b,t], ids[b,t+1]).

    Default (w = 1/(B(L-1))): mean CE of logits[:, :-1] vs ids[:, 1:] with pads INCLUDED (ref: decoder_model.py:46 passes
    labels=input_ids).  With ``row_weight`` / ``banned`` / ``top_k`` it is the SCST policy-gradient loss
    (ref: blocks/rl/SCST.py:14-45,142-170).  The gradient wrt the logits is produced in the same pass that computes the
    loss; backward only runs the dgrad / wgrad GEMMs, with dL/dloss folded in through a device scalar."""

    @staticmethod
    def forward(ctx, h, emb_sh, bias, ids, V, g_emb, g_bias, want_logits, row_weight, banned, top_k, need_grad=True):
        B, L, D = h.shape
        Vp = emb_sh.shape[0]
        h2 = _2d(h)
        logits = torch.empty(B * L, Vp, dtype=BF16, device=h.device)
        gemm(h2, 0, emb_sh, 0, logits, B * L, V, D, bias=bias)
        loss_sum = torch.zeros(1, dtype=torch.float32, device=h.device)
        inv = 1.0 / (B * (L - 1)) if row_weight is None else 1.0
        # need_grad comes from the caller: grad mode is always off INSIDE an autograd.Function.forward; under no_grad (validation
        # loss) the gradient half of the fused kernel -- a second full write of the [B*L, Vp] matrix -- is skipped
        dlogits = (torch.empty_like(logits) if want_logits else logits) if need_grad else None
        row_logp = torch.empty(B * L, dtype=torch.float32, device=h.device)
        thr = None
        ban = (C.c_int32 * 4)(*(list(banned) + [0] * (4 - len(banned)))) if banned else None
        if top_k:                     # per-row k-th largest live logit: exact radix select on the bf16 logits (no fp32 copy, no sort)
            thr = torch.empty(B * L, dtype=torch.float32, device=h.device)
            check(lib().vm_topk_threshold_bf16(ptr(logits), Vp, B * L, V, max(1, min(int(top_k), V - (len(banned) if banned else 0))), ban, len(banned) if banned else 0, ptr(thr),
                                               stream()), "vm_topk_threshold_bf16")
        check(lib().vm_ce_shift_fwd_bwd(ptr(logits), Vp, ptr(ids), B, L, V, ptr(loss_sum), ptr(row_logp),
                                        ptr(dlogits) if dlogits is not None else None, inv,
                                        ptr(row_weight) if row_weight is not None else None, ban, len(banned) if banned else 0,
                                        ptr(thr) if thr is not None else None, stream()), "vm_ce_shift_fwd_bwd")
        ctx.save_for_backward(h2, emb_sh, dlogits)
        ctx.meta = (B, L, D, V, Vp, g_emb, g_bias)
        loss = (loss_sum * inv).squeeze(0)
        out_logits = logits.view(B, L, Vp)[..., :V] if want_logits else None
        row_logp = row_logp.view(B, L)
        ctx.mark_non_differentiable(row_logp, *([out_logits] if out_logits is not None else []))
        return loss, out_logits, row_logp

    @staticmethod
    def backward(ctx, dloss, _dlogits_unused, _dlogp_unused):
        h2, emb_sh, dlogits = ctx.saved_tensors
        B, L, D, V, Vp, g_emb, g_bias = ctx.meta
        # dloss (dL/dloss, a device scalar: 1 for loss.backward(), 1/grad_accu or a loss scale otherwise) is folded
        # into the GEMM / column-sum epilogues through a device pointer -- no host sync, no extra pass over dlogits.
        sc = dloss.detach().to(torch.float32).contiguous()
        M = B * L
        if g_emb is not None:      # dE[V,D] += dlogits^T h (pad columns of dlogits are zero), d_bias[V] += colsum(dlogits)
            param_grads(dlogits, h2, g_emb, g_bias, alpha_dev=sc, cols=V)
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty(M, D, dtype=BF16, device=h2.device)
            gemm(dlogits, 0, emb_sh, 1, dh, M, D, Vp, alpha_dev=sc)
            dh = dh.view(B, L, D)
        return (dh,) + (None,) * 11


def lm_head_loss(h, emb_sh, bias, ids, V, g_emb=None, g_bias=None, want_logits=True, row_weight=None, banned=None, top_k=None):
    """-> (loss, logits or None, row_logp [B,L] = log p(ids[b,t+1] | prefix) under the filtered distribution)"""
    need_grad = torch.is_grad_enabled() and (h.requires_grad or g_emb is not None)
    return LmHeadLossFn.apply(h, emb_sh, bias, ids, V, g_emb, g_bias, want_logits, row_weight, banned, top_k, need_grad)


def lm_logits_f32(h2, emb_sh, bias, V):
    """fp32 logits for the decode step (hf:generation/utils.py:3384 upcasts to fp32 before log_softmax)."""
    M, D = h2.shape
    out = torch.empty(M, V, dtype=torch.float32, device=h2.device) if V % 4 == 0 else \
        torch.empty(M, (V + 3) // 4 * 4, dtype=torch.float32, device=h2.device)
    gemm(h2, 0, emb_sh, 0, out, M, V, D, bias=bias)
    return out[:, :V]
