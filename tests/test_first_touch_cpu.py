"""Host logic behind vm_wgrad_problem.overwrite (vilmedic_amd/ops.py): a weight-gradient launch may STORE into a gradient buffer instead of
read-add-writing only when nothing has written that memory since its arena zeroed the gradients.  Wrong "clean" answers lose gradient
contributions (accumulation steps, tied parameters, a view nested in a larger one), so the interval bookkeeping is tested on its own.
ref: what autograd's AccumulateGrad does for nn.Linear weights (the reference never overwrites: torch accumulates into .grad)."""
import pytest
import torch

from vilmedic_amd import ops


@pytest.fixture()
def fresh():
    saved = {k: list(v) for k, v in ops._touch.items()}
    for k in ops._touch:
        ops._touch[k] = []
    yield
    for k, v in saved.items():
        ops._touch[k] = v


def test_first_touch_is_true_once_per_zeroing(fresh):
    g = torch.zeros(1000)
    a, b = g[0:100], g[100:300]
    assert not ops._first_touch(a)              # nothing is known about an arena that never reported a zeroing
    ops.grads_zeroed(g)
    assert ops._first_touch(a) and ops._first_touch(b)
    assert not ops._first_touch(a) and not ops._first_touch(b)      # second micro-batch of an accumulation step: must accumulate
    ops.grads_zeroed(g)
    assert ops._first_touch(a)


def test_overlapping_views_share_their_memory(fresh):
    g = torch.zeros(1000)
    ops.grads_zeroed(g)
    qkv, k_only, v_only, after = g[0:300], g[100:200], g[200:300], g[300:400]
    assert ops._first_touch(qkv)
    assert not ops._first_touch(k_only)          # nested in the fused view written before
    assert not ops._first_touch(v_only)          # ... and a later neighbour of that nested view is still inside the fused one
    assert ops._first_touch(after)
    ops.grads_zeroed(g)
    assert ops._first_touch(k_only)
    assert not ops._first_touch(qkv)             # the fused view contains memory already written
    assert len(ops._touch["lo"]) == len(ops._touch["hi"]) and all(a < b for a, b in zip(ops._touch["lo"], ops._touch["hi"]))
    assert all(ops._touch["hi"][i] <= ops._touch["lo"][i + 1] for i in range(len(ops._touch["lo"]) - 1))     # sorted and disjoint


def test_marked_and_shared_buffers_always_accumulate(fresh):
    g = torch.zeros(1000)
    ops.grads_zeroed(g)
    emb, w = g[0:200], g[200:400]
    ops._touch["shared"].append(ops._span(emb))          # what the embedding forward registers for the tied word embedding
    assert not ops._first_touch(emb)
    ops.grads_zeroed(g)
    assert not ops._first_touch(emb)                     # shared survives every zeroing
    ops.mark_touched(w)                                  # e.g. the column-sum fallback wrote it
    assert not ops._first_touch(w)


def test_buffers_outside_tracked_arenas_and_two_arenas(fresh):
    g1, g2, stray = torch.zeros(500), torch.zeros(500), torch.zeros(64)
    ops.grads_zeroed(g1)
    assert not ops._first_touch(stray)                   # not arena memory: its contents are unknown
    assert not ops._first_touch(g2[0:10])                # the second arena has not been zeroed yet
    ops.grads_zeroed(g2)
    assert ops._first_touch(g2[10:20])
    assert ops._first_touch(g1[0:10])
    ops.grads_zeroed(g1)                                 # zeroing one arena leaves the other's history alone
    assert not ops._first_touch(g2[10:20])
    assert ops._first_touch(g1[0:10])


def test_span_of_a_column_slice_covers_its_row_strides():
    """a [rows, 64] column slice of a [rows, 256] matrix spans (rows - 1) * 256 + 64 elements, not rows * 64"""
    from vilmedic_amd import ops
    t = torch.zeros(8, 256)
    lo, hi = ops._span(t[:, 64:128])
    assert lo == t.data_ptr() + 64 * 4 and hi - lo == ((8 - 1) * 256 + 64) * 4
    assert ops._span(t) == (t.data_ptr(), t.data_ptr() + t.numel() * 4)
    assert ops._span(torch.zeros(0)) [1] == ops._span(torch.zeros(0))[0]


def test_forget_range_and_reset_host_state():
    """records about a freed arena's memory are dropped (ParamArena's finalizer), and an aborted graph capture leaves no queued launches
    and no buffer that still counts as clean (graph.GraphedTrainStep's except path)"""
    from vilmedic_amd import ops
    saved = {k: (list(v) if isinstance(v, list) else v) for k, v in ops._touch.items()}
    try:
        g = torch.zeros(1024)
        ops._touch.update(lo=[], hi=[], shared=[], ranges=[])
        ops.grads_zeroed(g)
        assert ops._first_touch(g[:128]) is True and ops._first_touch(g[:128]) is False
        ops._touch["shared"].append(ops._span(g[512:640]))
        ops.forget_range(*ops._span(g))
        assert ops._touch["ranges"] == [] and ops._touch["shared"] == [] and ops._touch["lo"] == []
        assert ops._first_touch(g[128:256]) is False             # untracked memory is never "clean"
        ops.grads_zeroed(g)
        ops._pg["items"].append("stale"), ops._pg["ptrs"].add(1), ops._lnq["items"].append("stale")
        ops._side["pending"] = True
        ops.reset_host_state()
        assert ops._pg["items"] == [] and ops._pg["ptrs"] == set() and ops._pg["tiles"] == 0 and ops._lnq["items"] == []
        assert ops._side["pending"] is False and ops._side["callback_queued"] is False
        assert ops._first_touch(g[256:384]) is False             # after an aborted capture nothing stores until the next zero_grad
        ops.grads_zeroed(g)
        assert ops._first_touch(g[256:384]) is True
    finally:
        ops._touch.update(saved)
        ops._pg.update(items=[], tiles=0, ptrs=set())
        ops._lnq.update(items=[], ptrs=set())
