"""A few seeded draws of each randomised sweep (tools/*_fuzz.py) under the SIMT interpreter: the sweeps themselves run on the GPU box
(profiles/r5_fuzz_summary.jsonl), this keeps their drawing code and the bodies they call alive in the CPU suite -- and puts a handful of
shapes nobody picked by hand in front of every change."""
import importlib.util
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def emu(emu_lib_path):
    from specforge_amd import _lib

    _lib._inject_library_for_tests(emu_lib_path)
    yield "cpu"
    _lib._inject_library_for_tests(None)


def test_attention_sweep_draws(emu):
    t = _tool("attn_fuzz")
    rng = random.Random(20)
    for _ in range(5):
        hd, B, S, nh, nkv, lengths, nsteps = t.draw(rng, 60)
        t.TA.test_ttt_attention_fwd_bwd(emu, hd, B, S, nh, nkv, lengths, nsteps)


def test_end_to_end_sweep_draws(emu):
    t = _tool("engine_fuzz")
    rng = random.Random(21)
    for i in range(3):
        c = t.draw(rng, True)
        c["S"] = min(c["S"], 24)
        c["lengths"] = [min(L, c["S"]) for L in c["lengths"]]
        t.TC.run_small_case(emu, c, seed=10 * i + 1)


def test_gemm_sweep_draws(emu):
    t = _tool("gemm_fuzz")
    t.DEV, t.SHRINK = "cpu", 12
    rng = random.Random(22)
    for i, fn in enumerate([t.case_nt, t.case_rowadd, t.case_tn, t.case_swiglu, t.case_nt, t.case_tn]):
        fn(rng, 100 * i + 1)


def test_kernel_sweep_draws(emu):
    t = _tool("kernel_fuzz")
    rng = random.Random(23)
    fns = dict(ce=t.TK.test_ce_fused, norm=t.TK.test_rmsnorm_fwd_bwd, teacher_reduce=t.TK.test_teacher_reduce_perm,
               teacher_gemm=t.TK.test_gemm_nt_teacher_matches_the_stored_logits)
    for _ in range(8):
        kind, kw = t.draw(rng, True)
        fns[kind](emu, **kw)
