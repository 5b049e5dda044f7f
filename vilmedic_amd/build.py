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
    for f in _sources() + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):       # (build_digest.inc, the generated file, is not part of it)
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(CSRC, "..", "..", "include", "vmhip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


DIGEST_INC = os.path.join(CSRC, "build_digest.inc")      # generated: the digest the library is built from, compiled into runtime.hip
MARK = b"VMDIGEST:"


def _lib_digest():
    """the source digest compiled into libvmhip.so (``vm_build_digest()``), read from the file's bytes -- no loading, no mtimes: a copy or a
    checkout that does not preserve timestamps (the gpurun snapshot) cannot make a matching library look stale, and a stamp file restored
    next to a library built from other sources cannot vouch for it"""
    try:
        blob = open(LIB, "rb").read()
    except OSError:
        return None
    i = blob.find(MARK)
    return blob[i + len(MARK):i + len(MARK) + 64].decode("ascii", "replace") if i >= 0 else None


def build(force=False, verbose=True):
    dig = _digest()
    if not force and _lib_digest() == dig:
        if not (os.path.exists(STAMP) and open(STAMP).read() == dig):
            open(STAMP, "w").write(dig)
        return LIB
    with open(DIGEST_INC, "w") as f:
        f.write('#define VM_BUILD_DIGEST "%s"\n' % dig)
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
