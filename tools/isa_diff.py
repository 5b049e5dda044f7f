#!/usr/bin/env python3
"""Compare the generated gfx950 device code of the library's kernels between a git revision and the working tree:

    python tools/isa_diff.py <rev> [file.hip ...]      (default: every vilmedic_amd/csrc/*.hip)

Each source is compiled to assembly at both revisions (hipcc cross-compiles without a GPU), comments and the per-function block
label indices are stripped, and kernels are compared by mangled name.  "identical" means instruction-identical: a host-side or
gated-off change can then be committed without re-validating the kernels on the GPU (used at the end of round 1, when the GPU
budget was spent, to show that adding the experimental GEMM variants left all 21 existing GEMM kernels untouched)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernels(asm):
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", asm, re.S | re.M):
        body = "\n".join(re.sub(r";.*$", "", line).rstrip() for line in m.group(2).splitlines())
        out[m.group(1)] = re.sub(r"\.LBB\d+_", ".LBBx_", body)
    return out


def compile_tree(tree, rel, out):
    from vilmedic_amd.build import FLAGS
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, *FLAGS, "-S", "--cuda-device-only", os.path.join(tree, rel), "-o", out], check=True, capture_output=True)
    return kernels(open(out).read())


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    rev = sys.argv[1]
    files = sys.argv[2:] or sorted("vilmedic_amd/csrc/" + f for f in os.listdir(os.path.join(ROOT, "vilmedic_amd", "csrc")) if f.endswith(".hip"))
    changed = 0
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        subprocess.run("git archive {} vilmedic_amd/csrc include | tar -x -C {}".format(rev, old), shell=True, check=True, cwd=ROOT)
        for rel in files:
            if not os.path.exists(os.path.join(old, rel)):
                print(f"{rel}: new file")
                continue
            a = compile_tree(old, rel, os.path.join(tmp, "a.s"))
            b = compile_tree(ROOT, rel, os.path.join(tmp, "b.s"))
            diff = [k for k in a if a[k] != b.get(k)]
            changed += len(diff)
            print(f"{rel}: {len(a) - len(diff)} of {len(a)} kernels identical, {len([k for k in b if k not in a])} new"
                  + ("".join("\n    changed: " + k for k in diff)))
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
