"""Launch budget of the training step, checked without a GPU: tools/host_profile.py runs the real host code (RRG: ViT + decoder,
forward + backward + fused Adam) against a stub of the C ABI that only counts calls.  A host change that adds launches per
transformer layer -- the step is 577 launches at 12 + 12 layers and nearly host-bound -- fails here, on the CPU suite."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _calls(layers):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_profile.py"), "--steps", "2", "--top", "0", "--layers", str(layers)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"C-ABI calls per step: (\d+)", out.stdout)
    assert m, out.stdout
    return int(m.group(1))


def test_launches_per_layer_and_fixed_part():
    c2, c4 = _calls(2), _calls(4)
    per_layer_pair = (c4 - c2) / 2            # one ViT layer + one decoder layer (self + cross attention), forward and backward
    fixed = c2 - 2 * per_layer_pair           # embeddings, patch projection, all-layer cross K|V, LM head + loss, Adam, flushes
    # (the weight-gradient queue flushes every two layers since round 4 -- one full round of 256 x 256 tiles -- so the split into a per-layer
    # and a fixed part is only approximate: the flush count does not grow linearly at 2 / 4 layers; the projected 12 + 12-layer total is the budget)
    assert per_layer_pair <= 46, (c2, c4)
    assert fixed <= 27, (c2, c4)
    assert fixed + 12 * per_layer_pair <= 577


def test_call_trace_is_deterministic(tmp_path):
    """the C-ABI call trace (entry point + scalar arguments) of a step is what host refactors are checked against: two runs must agree"""
    outs = []
    for k in range(2):
        path = os.path.join(str(tmp_path), f"trace{k}.txt")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_profile.py"), "--steps", "1", "--top", "0", "--layers", "2", "--trace", path],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(path).read())
    assert outs[0] == outs[1] and outs[0].count("vm_gemm_bf16") > 20
    assert "vm_adam_step_dev" in outs[0] and "vm_ce_shift_fwd_bwd" in outs[0]
