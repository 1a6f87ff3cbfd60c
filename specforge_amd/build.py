"""Build the native libraries of specforge_amd.

* ``build_hip()``  -> specforge_amd/libsfhip.so : the product library, hipcc for gfx950.
* ``build_emu()``  -> tests/emu/libsfhip_emu.so : the same kernel sources compiled for the
  host against the SIMT interpreter (tests/emu/sf_emu.h).  Test infrastructure only.

hipcc cross-compiles without a GPU, so both build in the CPU-only container.  Objects are
cached by source mtime under build/.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
SOURCES = ["sf_core.hip", "sf_loss.hip", "sf_pointwise.hip", "sf_gemm.hip", "sf_gemm256.hip", "sf_gemm256w4.hip", "sf_gemm256w4_i0.hip", "sf_gemm256w4_i1.hip", "sf_gemm256w4_i2.hip", "sf_gemm256w4_i3.hip", "sf_gemm256w4_i4.hip", "sf_gemm256w4_i5.hip", "sf_gemm256w4_i6.hip", "sf_gemm256tn.hip", "sf_attn.hip", "sf_attn_dkv.hip", "sf_attn_w1.hip", "sf_attn_w1_dkv.hip"]
HIP_LIB = os.path.join(PKG, "libsfhip.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libsfhip_emu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOSTCXX = os.environ.get("SF_HOSTCXX", "/opt/rocm/lib/llvm/bin/clang++")


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "specforge_amd.h"))
    hdrs.append(os.path.join(ROOT, "tests", "emu", "sf_emu.h"))
    exp = os.path.join(ROOT, "tools", "experiments")
    hdrs += [os.path.join(exp, f) for f in os.listdir(exp) if f.endswith(".inc")]
    return [h for h in hdrs if os.path.exists(h)]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


# per-file extra flags.  sf_attn.hip: without -fno-honor-nans every fmaxf on an MFMA result is preceded by a canonicalising
# v_max_f32 x, x (the compiler cannot prove the accumulator is not a signalling NaN): 32 extra VALU per 64-key tile in the
# online softmax.  Infinities keep their meaning (the causal mask is -inf).
EXTRA_FLAGS = {"sf_attn.hip": ["-fno-honor-nans"], "sf_attn_dkv.hip": ["-fno-honor-nans"], 
               # one wave per SIMD, slot-planned: -O3's SLP vectoriser pairs the per-element softmax / dS arithmetic of DIFFERENT slots into
               # v_pk_* instructions placed where the later operand appears -- the pack slots then carry 14-16 instructions (the matrix pipe
               # idles) and packed fp32 VALU beside MFMAs is an anti-lever by itself (MI355X_MICROARCH.md)
               "sf_attn_w1.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "sf_attn_w1_dkv.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}

def _build(lib, objdir, compile_cmd, link_cmd, force=False):
    os.makedirs(objdir, exist_ok=True)
    deps = _deps()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + deps):
            jobs.append(compile_cmd + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _stale(lib, objs):
        _run(link_cmd + objs + ["-o", lib])
    return lib


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> specforge_amd/libsfhip.so"""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC,
             "-Wno-unused-result", "-ffp-contract=fast"]
    lib = _build(HIP_LIB, os.path.join(ROOT, "build", "hip"), [HIPCC] + flags,
                 [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"], force)
    if verbose:
        print("built", lib)
    return lib


def build_ablate(force=False, verbose=False):
    """TOOLS ONLY: hipcc -DSF_ABLATE -> tools/experiments/libsfhip_ablate.so -- the measured-and-rejected kernel variants
    and the SF_* environment knobs that select them (A/B timing; some give wrong results by design).  Never loaded by the
    package; tools inject it with ``_lib._inject_library_for_tests``."""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC, "-DSF_ABLATE",
             "-Wno-unused-result", "-ffp-contract=fast"]
    lib = _build(os.path.join(ROOT, "tools", "experiments", "libsfhip_ablate.so"), os.path.join(ROOT, "build", "ablate"),
                 [HIPCC] + flags, [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"], force)
    if verbose:
        print("built", lib)
    return lib


def build_emu(force=False, asan=False, verbose=False):
    """host clang++ -DSF_EMU -> tests/emu/libsfhip_emu.so (SIMT interpreter build)"""
    flags = ["-DSF_EMU", "-O2", "-g", "-std=c++17", "-fPIC", "-x", "c++", "-I", CSRC,
             "-I", os.path.join(ROOT, "tests", "emu"), "-pthread", "-Wno-unused-result"]
    link = [HOSTCXX, "-shared", "-fPIC", "-pthread"]
    lib, objdir = EMU_LIB, os.path.join(ROOT, "build", "emu")
    if asan:
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer"]
        link += ["-fsanitize=address", "-shared-libasan"]
        lib = EMU_LIB.replace(".so", "_asan.so")
        objdir += "_asan"
    lib = _build(lib, objdir, [HOSTCXX] + flags, link, force)
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    what = sys.argv[1:] or ["hip", "emu"]
    if "hip" in what:
        build_hip(force="--force" in what, verbose=True)
    if "emu" in what:
        build_emu(force="--force" in what, verbose=True)
    if "ablate" in what:
        build_ablate(force="--force" in what, verbose=True)
    if "asan" in what:
        build_emu(force="--force" in what, asan=True, verbose=True)
