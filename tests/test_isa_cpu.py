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


HOT = re.compile(r"_Z(16gemm_fast_kernel|19gemm_grouped_kernel)ILi[01]ELi[01]ELi2ELi2ELi2ELi64ELi[45]ELi[14]E(Li[01]E)?Ev(8GemmArgs|13GemmGroupArgs)$")


def _loop(asm, name):
    """the basic block of kernel ``name`` with the most MFMAs (its main loop), as a list of (opcode, full line)"""
    start = asm.index(name + ":")
    body = asm[start:asm.index(".Lfunc_end", start)]
    blocks = re.split(r"\n\.LBB\d+_\d+:", body)
    loop = max(blocks, key=lambda b: b.count("v_mfma_f32_16x16x32_bf16"))
    return [(l.split()[0], l.strip()) for l in loop.splitlines() if l.strip() and not l.strip().startswith((";", "."))]


def test_hot_gemm_kernels_fit_two_waves_per_simd_without_spills(gemm_fast_asm):
    """every kernel the training step can launch: the 128- / 160-row tiles (row-major: PIPE 1, strided: PIPE 4) and the grouped kernels"""
    meta = {k: v for k, v in _kernel_meta(gemm_fast_asm).items() if HOT.match(k)}
    assert len(meta) == 9, sorted(meta)      # 4 layouts x 128 rows + 2 layouts x 160 rows + 3 grouped (TN with bias gradient, NT, NN)
    for name, m in meta.items():
        assert m["vgpr_spill_count"] == 0, (name, m)
        assert m["sgpr_spill_count"] <= 16, (name, m)        # a few SGPRs parked in VGPR lanes (v_writelane) are harmless; scratch is not
        assert m["vgpr_count"] <= 256, (name, m)             # 512 registers per SIMD lane / 2 resident waves per SIMD


def test_pipelined_k_tile_schedule_is_present(gemm_fast_asm):
    """row-major 128 x 128 kernel (PIPE 1): all MFMAs of a K-tile (32), LDS reads BETWEEN MFMAs (the second k-half's fragments), and
    no more than two full LDS waits"""
    ops = _loop(gemm_fast_asm, "_Z16gemm_fast_kernelILi0ELi0ELi2ELi2ELi2ELi64ELi4ELi1EEv8GemmArgs")
    mfma = [i for i, (o, _) in enumerate(ops) if o.startswith("v_mfma")]
    assert len(mfma) == 32, len(mfma)
    reads_between = [i for i, (o, _) in enumerate(ops) if o.startswith("ds_read") and mfma[0] < i < mfma[-1]]
    assert len(reads_between) >= 8, len(reads_between)
    # (besides the barrier's own "vmcnt(0) lgkmcnt(0)": since the last K-tile is peeled -- the epilogue operands go out in front of it --
    # the barrier, the DMA issue and the MFMAs are one basic block)
    assert sum(1 for o, l in ops if o == "s_waitcnt" and "lgkmcnt(0)" in l and "vmcnt" not in l) <= 2


@pytest.mark.parametrize("name,n_mfma", [
    ("_Z16gemm_fast_kernelILi0ELi1ELi2ELi2ELi2ELi64ELi4ELi4EEv8GemmArgs", 32),       # dgrad, 128 x 128
    ("_Z16gemm_fast_kernelILi0ELi1ELi2ELi2ELi2ELi64ELi5ELi4EEv8GemmArgs", 40),       # dgrad, 160 x 128
    ("_Z16gemm_fast_kernelILi1ELi1ELi2ELi2ELi2ELi64ELi4ELi4EEv8GemmArgs", 32),       # wgrad
    ("_Z19gemm_grouped_kernelILi1ELi1ELi2ELi2ELi2ELi64ELi4ELi4ELi1EEv13GemmGroupArgs", 40),  # grouped wgrad with the bias-gradient MFMAs
    ("_Z19gemm_grouped_kernelILi0ELi1ELi2ELi2ELi2ELi64ELi4ELi4ELi0EEv13GemmGroupArgs", 32),  # grouped NN (per-image products)
])
def test_cross_tile_register_pipeline_keeps_the_dma_in_flight(gemm_fast_asm, name, n_mfma):
    """PIPE 4 main loops: one barrier per K-tile with the DMA issue right behind it, NO vmcnt wait between the DMA issue and the end
    of the loop (the builtin LDS-DMA made hipcc put vmcnt(0) in front of the first transpose read: tile t+2 was waited for at once),
    LDS reads slotted between MFMAs, nothing spilled inside the loop"""
    ops = _loop(gemm_fast_asm, name)
    names = [o for o, _ in ops]
    assert sum(o.startswith("v_mfma") for o in names) == n_mfma
    assert names.count("s_barrier") == 1
    assert not any(o.startswith("scratch_") for o in names)
    bar = names.index("s_barrier")
    dma = [i for i, o in enumerate(names) if o.startswith("global_load_lds")]
    assert dma and min(dma) > bar
    assert not any(o == "s_waitcnt" and "vmcnt" in l for o, l in ops[max(dma):]), [l for o, l in ops[max(dma):] if o == "s_waitcnt"]
    mf = [i for i, o in enumerate(names) if o.startswith("v_mfma")]
    assert sum(1 for i, o in enumerate(names) if o.startswith("ds_read") and mf[0] < i < mf[-1]) >= 16


def _asm_of(tmp_path_factory, src):
    if not shutil.which(HIPCC):
        pytest.skip("hipcc not available")
    from vilmedic_amd.build import FLAGS
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    cmd = [HIPCC, *FLAGS, "-S", "--cuda-device-only", os.path.join(ROOT, "vilmedic_amd", "csrc", src + ".hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text()


def test_wide_tile_weight_gradient_kernel_keeps_two_k_tiles_in_flight(tmp_path_factory):
    """gemm_p8w_kernel<2> (round 4: the grouped weight gradients of the step): 256 x 256 tile = 64 MFMAs per wave and K-tile, both operands by
    LDS-DMA (8 requests per wave and K-tile), ONE counted wait per K-tile that leaves the two younger K-tiles in flight (a vmcnt(0) here is
    what the compiler's own waits looked like: every K-tile then waited for the tile just requested), no register spilled anywhere"""
    asm = _asm_of(tmp_path_factory, "gemm_p8w")
    name = next(k for k in _kernel_meta(asm) if "gemm_p8w_kernelILi2E" in k)
    m = _kernel_meta(asm)[name]
    assert m["vgpr_spill_count"] == 0 and m["vgpr_count"] <= 256, m
    ops = _loop(asm, name)
    names = [o for o, _ in ops]
    assert sum(o.startswith("v_mfma") for o in names) == 64
    assert sum(o.startswith("global_load_lds") for o in names) == 8
    assert names.count("s_barrier") == 4                                   # two barrier pairs per K-tile (the two wave groups run one apart)
    vm = [l for o, l in ops if o == "s_waitcnt" and "vmcnt" in l]
    assert vm == ["s_waitcnt vmcnt(8)"], vm
    assert not any(o.startswith("scratch_") for o in names)


def test_wide_tile_forward_kernel_main_loop_is_clean(tmp_path_factory):
    """gemm_p8_kernel<0, 0, 8, 2, EPI 1, OPS 0> (the LM head since round 6: wave-private epilogue): the K loop holds its 64 MFMAs, one counted
    wait, no scratch access -- and the kernel as a whole no longer spills (the staged epilogue of round 4 parked 6 VGPRs in scratch); the
    epilogue's element-wise code exists ONCE (a runtime pass loop: unrolled, MF x 2 copies of every variant measured 20 us per tile against
    7 us, profiles/r06_a_wide_tile_epilogue.txt), so the whole kernel stays below 40 KB"""
    asm = _asm_of(tmp_path_factory, "gemm_p8")
    meta = _kernel_meta(asm)
    name = next(k for k in meta if "gemm_p8_kernelILi0ELi0ELi8ELi2ELi1ELi0E" in k)
    assert meta[name]["vgpr_spill_count"] == 0 and meta[name]["vgpr_count"] <= 256, meta[name]
    ops = _loop(asm, name)
    names = [o for o, _ in ops]
    assert sum(o.startswith("v_mfma") for o in names) == 64
    assert [l for o, l in ops if o == "s_waitcnt" and "vmcnt" in l] == ["s_waitcnt vmcnt(4)"]
    assert not any(o.startswith("scratch_") for o in names)
    start = asm.index(name + ":")
    size = int(re.search(r"; codeLenInByte = (\d+)", asm[asm.index(".Lfunc_end", start):]).group(1))
    assert size <= 40 * 1024, size
    # every wave-private-epilogue instantiation the cost model can pick (MF 5..8, row-major / k-major B, with / without a pre-loaded operand):
    # nothing spilled inside a K loop
    for k in meta:
        if "gemm_p8_kernel" in k and "ELi2ELi1ELi" in k:
            assert not any(o.startswith("scratch_") for o, _ in _loop(asm, k)), k


def _vregs(tok):
    """VGPR numbers named by one operand token: v7 -> {7}, v[12:15] -> {12..15}"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(x) for x in re.findall(r"\bv(\d+)\b", tok)}
    return out


def _blocks(body):
    """basic blocks of one function's assembly: {label: (instructions, successor labels)}; the entry block is labelled None"""
    blocks, order, cur = {}, [], None
    blocks[cur] = []
    order.append(cur)
    for raw in body.splitlines():
        l = raw.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if l.startswith(";;#ASM"):
            blocks[cur].append(l)                       # inline-asm markers: only loads inside them are the hand-placed ones
            continue
        if not l or l.startswith((";", ".", "//")):
            continue
        blocks[cur].append(l)
    succ = {}
    for k, lab in enumerate(order):
        ins = blocks[lab]
        nxt = order[k + 1] if k + 1 < len(order) else None
        out, falls = [], True
        for l in ins:
            op = l.split()[0]
            if op.startswith("s_cbranch"):
                out.append(l.split()[-1])
            elif op == "s_branch":
                out.append(l.split()[-1])
                falls = False
            elif op == "s_endpgm":
                falls = False
        if falls and nxt is not None:
            out.append(nxt)
        succ[lab] = out
    return blocks, succ, order


def _reads(l):
    """VGPRs an instruction READS: every vector operand but the destination (stores, compares and AGPR writes have no VGPR destination;
    accumulating ops read theirs)"""
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if not ops or not ops[0]:
        return set()
    no_dst = op.startswith(("global_store", "scratch_store", "ds_write", "ds_store", "buffer_store", "flat_store", "v_cmp", "v_accvgpr_write",
                            "v_readlane", "v_readfirstlane", "global_atomic", "v_swap"))
    reads_dst = any(t in op for t in ("mac", "dot2c", "v_swap", "v_permlane"))
    srcs = ops if (no_dst or reads_dst or not ops[0].startswith("v")) else ops[1:]
    out = set()
    for o in srcs:
        out |= _vregs(o)
    return out


def _check_pending_loads(name, body):
    """forward may-analysis over the CFG: the set of VGPRs that are destinations of an inline-asm global_load_dwordx4 still in flight (no
    vmcnt(0) since).  Any instruction that READS such a register -- a copy, a spill, an AGPR move, an address computation, an ALU operand --
    would see stale data.  (A plain redefinition ends the register's pending state: the kernels have several mutually exclusive issue
    sites -- one K-tile / peeled last K-tile / late operands -- that reuse the same registers, and the analysis does not correlate branches.)"""
    blocks, succ, order = _blocks(body)
    entry = {lab: None for lab in order}              # None = not reached yet
    entry[None] = frozenset()
    work, n_loads = [None], 0
    seen_loads = set()
    while work:
        lab = work.pop()
        pending = dict.fromkeys(entry[lab], "")
        in_asm = False
        for l in blocks[lab]:
            if l.startswith(";;#ASM"):
                in_asm = l.startswith(";;#ASMSTART")
                continue
            op = l.split()[0]
            if op == "s_waitcnt" and "vmcnt(0)" in l:
                pending.clear()
                continue
            if in_asm and op.startswith("global_load_dwordx4") and l.rstrip().endswith("off") and "lds" not in op:
                dst, rest = l.split(None, 1)[1].split(",", 1)
                assert not (_vregs(rest) & set(pending)), (name, lab, l, "address built from an in-flight destination")
                for r in _vregs(dst):
                    pending[r] = l
                if (lab, l) not in seen_loads:
                    seen_loads.add((lab, l))
                    n_loads += 1
                continue
            if pending and not op.startswith("s_"):
                hit = _reads(l) & set(pending)
                assert not hit, (name, lab, l, "reads a register whose load is still in flight", pending[next(iter(hit))])
                for r in _vregs(l.partition(" ")[2].split(",")[0]) if l.partition(" ")[2].strip().startswith("v") else ():
                    pending.pop(r, None)
        out = frozenset(pending)
        for t in succ[lab]:
            if t not in entry:
                continue
            new = out if entry[t] is None else (entry[t] | out)
            if new != entry[t]:
                entry[t] = new
                work.append(t)
    return n_loads


def test_early_epilogue_operand_loads_are_not_touched_before_their_wait(gemm_fast_asm):
    """The epilogue operands (bias, residual, gelu' input) are requested by inline-asm ``global_load_dwordx4`` in front of the last K-tile and
    completed by an explicit ``s_waitcnt vmcnt(0)`` at the head of the epilogue (gemm_fast.hip).  The compiler believes the destination
    registers are defined when the asm returns: a copy, a spill or a move to an AGPR placed between the request and the wait would read
    stale data.  For EVERY gemm_fast_kernel / gemm_grouped_kernel / gemm_pair_kernel instantiation: no VGPR spill, and on no path of the
    control-flow graph does an instruction name a destination register of such a load before the next vmcnt(0)."""
    meta = _kernel_meta(gemm_fast_asm)
    names = [k for k in meta if re.match(r"_Z(16gemm_fast_kernel|19gemm_grouped_kernel|16gemm_pair_kernel)", k)]
    assert len(names) >= 9, names
    total = 0
    for name in names:
        assert meta[name]["vgpr_spill_count"] == 0, (name, meta[name])
        start = gemm_fast_asm.index(name + ":")
        total += _check_pending_loads(name, gemm_fast_asm[start:gemm_fast_asm.index(".Lfunc_end", start)])
    assert total >= 2 * len(names), total
