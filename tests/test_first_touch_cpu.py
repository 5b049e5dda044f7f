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
