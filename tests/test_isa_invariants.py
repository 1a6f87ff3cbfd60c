"""Audit of the gfx950 ISA hipcc emits for the attention kernels (cross-compiles without a GPU).  These are properties the
kernels' correctness or speed depends on but that no numeric test can see:

* inside the tile loops (between the first and the last s_barrier of a kernel) the ONLY waits on vmcnt are the kernels' own
  asm statements.  vmcnt is one in-order counter: a compiler-placed wait for one of ITS loads also drains the LDS-DMA
  prefetch of the next tile, which it cannot see -- round 2's kernels lost their whole prefetch overlap to exactly that;
* the dK/dV kernel keeps its accumulators and K / V fragments in asm-owned AGPRs: nothing the compiler emits may read or
  write an AGPR (a spill or a v_accvgpr_* into that range would be silent corruption);
* no scratch, no spills."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "specforge_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asm(src):
    out_dir = os.path.join(ROOT, "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, src.replace(".hip", ".s"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        tmp = f"{out}.{os.getpid()}.tmp"       # (parallel test workers compile the same unit: publish the result atomically)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-I", CSRC, "-ffp-contract=fast",
                        "-fno-honor-nans", "-w", "-S", "--cuda-device-only", "-o", tmp, os.path.join(CSRC, src)]
                       + (["-fno-slp-vectorize"] if src.startswith("sf_attn_w1") else []), check=True)     # (specforge_amd/build.py EXTRA_FLAGS)
        os.replace(tmp, out)
    return open(out).read()


def _kernels(txt, pattern):
    """{mangled name: (body lines, metadata dict)} of the kernels whose name matches"""
    lines = txt.split("\n")
    res = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*%s\S*):" % pattern, l)
        if not m:
            continue
        end = next(j for j in range(i, len(lines)) if lines[j].startswith("\t.set") and "uses_flat_scratch" in lines[j])
        meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(m.group(1)), txt, re.S).group(1)
        md = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", meta)}
        res[m.group(1)] = (lines[i:end], md)
    return res


def _compiler_lines(body):
    """(line, inside_asm) for instruction lines"""
    inasm = False
    for l in body:
        if "ASMSTART" in l:
            inasm = True
            continue
        if "ASMEND" in l:
            inasm = False
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        yield t, inasm


def _loop_region(body):
    """lines of the basic blocks that belong to the loop nest around the kernel's first s_barrier (LLVM annotates every block
    with `in Loop: Header=BBx_y`, the header itself with `Loop Header` / `Parent Loop BBx_y`)"""
    blocks, cur = [], None          # (header ids this block belongs to, own id, lines)
    for l in body:
        m = re.match(r"^(?:\.L|; %bb\.)(BB\d+_\d+|\d+):?", l)
        if m or re.match(r"^; %bb\.", l):
            own = re.match(r"^\.L(BB\d+_\d+):", l)
            cur = dict(own=own.group(1) if own else None, hdrs=set(re.findall(r"Header=(BB\d+_\d+)", l)), lines=[], is_hdr="Loop Header" in l)
            blocks.append(cur)
        elif cur is not None and l.lstrip().startswith(";") and ("Loop Header" in l or "Parent Loop" in l or "Child Loop" in l):
            cur["is_hdr"] = cur["is_hdr"] or "Loop Header" in l
            cur["hdrs"] |= set(re.findall(r"Parent Loop (BB\d+_\d+)", l))
        if cur is not None:
            cur["lines"].append(l)
    bb = next((b for b in blocks if any(x.strip().startswith("s_barrier") for x in b["lines"])), None)
    if bb is None:
        return []
    nest = set(bb["hdrs"]) | ({bb["own"]} if bb["is_hdr"] and bb["own"] else set())
    for b in blocks:                # parents of the loops found so far
        if b["own"] in nest:
            nest |= b["hdrs"]
    out = []
    for b in blocks:
        if (b["own"] in nest and b["is_hdr"]) or (b["hdrs"] & nest):
            out += b["lines"]
    return out


hipcc = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@hipcc
@pytest.mark.parametrize("src,pattern", [("sf_attn.hip", "attn_fwd_kernel"), ("sf_attn.hip", "attn_bwd_dq_kernel"),
                                         ("sf_attn_dkv.hip", "attn_bwd_dkv_kernel")])
def test_no_compiler_vmcnt_wait_inside_the_tile_loops(src, pattern):
    ks = _kernels(_asm(src), pattern)
    assert ks, "kernel not found in the ISA"
    for name, (body, md) in ks.items():
        region = _loop_region(body)
        assert region, (name, "no loop around the barrier")
        ins = list(_compiler_lines(region))
        assert md["vgpr_spill_count"] == 0 and md["sgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, (name, md)
        bad = [t for t, inasm in ins if not inasm and t.startswith("s_waitcnt") and "vmcnt" in t]
        assert not bad, (name, bad[:4])
        own = [t for t, inasm in ins if inasm and t.startswith("s_waitcnt") and "vmcnt" in t]
        assert own, (name, "the loop's own vmcnt wait is missing")
        assert sum(1 for t, _ in ins if "v_mfma" in t) >= 16, name


@hipcc
def test_dkv_register_bank_is_untouched_by_the_compiler():
    ks = _kernels(_asm("sf_attn_dkv.hip"), "attn_bwd_dkv_kernel")
    assert len(ks) == 2                     # head_dim 64 and 128
    for name, (body, md) in ks.items():
        bad = [t for t, inasm in _compiler_lines(body) if not inasm and ("accvgpr" in t or re.search(r"[\s,\[]a\[?\d", t))]
        assert not bad, (name, bad[:4])
        mf = [t for t, inasm in _compiler_lines(body) if "v_mfma" in t]
        hd = 128 if "ILi128" in name else 64
        per_tile = 2 * (2 * hd // 16) + 2 * (4 * hd // 32)        # score + gradient MFMAs per tile: each product once
        assert len(mf) == 2 * per_tile, (name, len(mf))            # x2: masked and unmasked tile bodies
        # bank = 2 * hd/32 accumulator tiles of 16 + 2 * hd/16 fragments of 4 AGPRs on top of the compiler's VGPRs
        bank = 32 * (hd // 32) + 8 * (hd // 16)
        assert md["vgpr_count"] - bank <= 256 and md["vgpr_count"] <= 512, (name, md["vgpr_count"])


@hipcc
@pytest.mark.parametrize("src,pattern,bank", [("sf_attn_w1.hip", "attn_fwd_w1_kernel", 192), ("sf_attn_w1.hip", "attn_bwd_dq_w1_kernel", 256),
                                              ("sf_attn_w1_dkv.hip", "attn_bwd_dkv_pair_kernel", 192)])
def test_head_dim_256_kernels_keep_the_compiler_out_of_their_register_bank(src, pattern, bank):
    """The one-wave-per-SIMD head_dim-256 kernels (round 5) keep their accumulators and B-operand fragments in asm-owned AGPRs
    (AgprBank).  hipcc does not know the bank is taken: the moment its own pressure passes 256 VGPRs anywhere in the kernel it
    spills into a0.. -- the accumulators -- silently (seen twice while the forward's epilogue was written: 48 address values parked in
    a0..a47 for the whole tile loop).  So: no compiler-emitted instruction may name an AGPR, no scratch, no spills, the only vmcnt
    waits between the first and the last MFMA are the kernels' own, and every MFMA is one of the planned stream."""
    ks = _kernels(_asm(src), pattern)
    assert len(ks) == 1, list(ks)
    (name, (body, md)), = ks.items()
    assert md["vgpr_spill_count"] == 0 and md["sgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, md
    ins = list(_compiler_lines(body))
    bad = [t for t, inasm in ins if not inasm and ("accvgpr" in t or re.search(r"[\s,\[]a\[?\d", t))]
    assert not bad, (name, bad[:4])
    assert md["vgpr_count"] - bank <= 256 and md["vgpr_count"] <= 512, md["vgpr_count"]     # the bank + what the compiler uses
    mf = [i for i, (t, _) in enumerate(ins) if "v_mfma" in t]
    inner = ins[mf[0]:mf[-1]]
    assert not [t for t, inasm in inner if not inasm and t.startswith("s_waitcnt") and "vmcnt" in t], name
    assert [t for t, inasm in ins if inasm and t.startswith("s_waitcnt") and "vmcnt" in t], (name, "the loop's own vmcnt wait is missing")
    assert not [t for t, _ in ins if t.startswith("scratch_")], name


_W4_UNITS = ["sf_gemm256w4_i%d.hip" % i for i in range(7)]
_w4_compiled = []


def _compile_w4_units_once():
    """all seven translation units to ISA concurrently (one at a time they add ~4 minutes to a cold CPU suite)"""
    if not _w4_compiled:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 2)) as ex:
            list(ex.map(_asm, _W4_UNITS))
        _w4_compiled.append(True)


@hipcc
@pytest.mark.parametrize("unit", _W4_UNITS)
def test_nt_gemm_main_loop_is_the_planned_stream_in_every_epilogue_variant(unit):
    """The 4-wave NT GEMM is compiled once per epilogue form (plain, fp32, row addend, d(SwiGLU), SwiGLU forward, teacher reduction), and
    hipcc's register allocation of the whole kernel moves with the epilogue code.  Whatever the epilogue: the K loop of every
    instantiation must be the planned stream -- 128 MFMAs, 32 fragment reads, 16 LDS-DMA pieces (asm), 3 barriers -- with no scratch
    access, no accumulator copy and no compiler-inserted vmcnt wait inside it."""
    _compile_w4_units_once()
    txt = _asm(unit)
    ks = _kernels(txt, "gemm_nt_256w4_kernel")
    assert len(ks) == 2, (unit, list(ks))                        # both operand-order plans
    for name, (body, md) in ks.items():
        blocks, cur = [], []
        for l in body:
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur)
                cur = []
            cur.append(l)
        blocks.append(cur)
        loop = max(blocks, key=lambda b: sum("v_mfma" in x for x in b))
        ins = [t for t, _ in _compiler_lines(loop)]
        count = lambda pat: sum(1 for t in ins if re.match(pat, t))
        assert count(r"v_mfma_f32_16x16x32_bf16") == 128, (name, count(r"v_mfma"))
        assert count(r"ds_read_b128") == 32 and count(r"buffer_load_dwordx4") == 16 and count(r"s_barrier") == 3, name
        assert not [t for t in ins if t.startswith("scratch_") or "accvgpr" in t], name
        assert not [t for t, inasm in _compiler_lines(loop) if not inasm and t.startswith("s_waitcnt") and "vmcnt" in t], name
        assert len(ins) <= 260, (name, len(ins))                 # 243-245 today: nothing crept into the loop
