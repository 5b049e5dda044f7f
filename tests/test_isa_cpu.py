"""Static checks of the generated gfx950 code for the hot GEMM instantiations (hipcc cross-compiles without a GPU): register
budget for two workgroups per CU, no spills, and the software-pipelined K-tile schedule (LDS reads slotted between MFMAs) that
profiles/r01_e_gemm_pipe_ab.txt measured -- a source or flag change that silently loses it shows up here, not only in a bench."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def gemm_fast_asm(tmp_path_factory):
    if not shutil.which(HIPCC):
        pytest.skip("hipcc not available")
    from vilmedic_amd.build import FLAGS
    out = tmp_path_factory.mktemp("isa") / "gemm_fast.s"
    cmd = [HIPCC, *FLAGS, "-S", "--cuda-device-only", os.path.join(ROOT, "vilmedic_amd", "csrc", "gemm_fast.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text()


def _kernel_meta(asm):
    meta = {}
    for block in asm.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        meta[name.group(1)] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                               for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count")}
    return meta


def _hot(name):
    # gemm_fast_kernel<LA, LB, 2, 2, 2, 64, MF, 1>: 4 waves, 2-stage ring, k-tile 64, pipelined K-tile (the training step's kernels)
    return re.match(r"_Z16gemm_fast_kernelILi[01]ELi[01]ELi2ELi2ELi2ELi64ELi[45]ELi1EEv8GemmArgs$", name)


def test_hot_gemm_kernels_fit_two_workgroups_per_cu_without_spills(gemm_fast_asm):
    meta = {k: v for k, v in _kernel_meta(gemm_fast_asm).items() if _hot(k)}
    assert len(meta) == 6, sorted(meta)                      # 4 layouts of the 128x128 tile + 2 of the 160x128 tile
    for name, m in meta.items():
        assert m["vgpr_spill_count"] == 0, (name, m)
        assert m["vgpr_count"] <= 256, (name, m)             # 512 registers per SIMD lane / 2 resident waves per SIMD


def test_pipelined_k_tile_schedule_is_present(gemm_fast_asm):
    """in the main-loop block of the row-major 128x128 kernel: all MFMAs of a K-tile (32), LDS reads BETWEEN MFMAs (the second
    k-half's fragments), and no more than two full LDS waits"""
    start = gemm_fast_asm.index("_Z16gemm_fast_kernelILi0ELi0ELi2ELi2ELi2ELi64ELi4ELi1EEv8GemmArgs:")
    body = gemm_fast_asm[start:gemm_fast_asm.index(".end_amdhsa_kernel", start)] if ".end_amdhsa_kernel" in gemm_fast_asm[start:] else gemm_fast_asm[start:]
    blocks = re.split(r"\n\.LBB\d+_\d+:", body)
    loop = max(blocks, key=lambda b: b.count("v_mfma_f32_16x16x32_bf16") if "ds_read_b128" in b else -1)
    ops = [l.split()[0] for l in loop.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    mfma = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    assert len(mfma) == 32, len(mfma)
    reads_between = [i for i, o in enumerate(ops) if o.startswith("ds_read") and mfma[0] < i < mfma[-1]]
    assert len(reads_between) >= 8, len(reads_between)
    full_waits = sum(1 for l in loop.splitlines() if "s_waitcnt" in l and "lgkmcnt(0)" in l)
    assert full_waits <= 2, full_waits
