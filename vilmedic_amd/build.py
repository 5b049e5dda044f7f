"""Build libvmhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m vilmedic_amd.build [--force]

Cross-compiles without a GPU.  The .so is git-ignored but travels with the
gpurun snapshot, so the GPU box loads the prebuilt library.
"""
import hashlib
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libvmhip.so")
STAMP = os.path.join(CSRC, ".build_stamp")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
         "-Wno-unused-result", "-Wno-inline-asm"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in _sources() + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(CSRC, "..", "..", "include", "vmhip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _stale():
    """a source newer than the library: the stamp (a tracked file) can be restored by a checkout that also rewrote sources, and would then
    vouch for a library built from other code"""
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(CSRC, "..", "..", "include", "vmhip.h"))
    return any(os.path.getmtime(f) > t for f in files)


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(CSRC, src[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed:\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, "w").write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
