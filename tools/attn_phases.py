"""Where the time of the slot-planned dQ kernel goes: per-wave cycle sums between the stamps of the tools build
(SfProf in sf_attn_common.h; SF_ATTN_PROF=1 selects the stamped instantiation).   python tools/attn_phases.py   (GPU box)

marks: 0 wait for the tile's DMA pieces, 1 barrier, 2 burst of the first 8 fragment reads, 3..10 the eight phases
A0 A1 G0 A2 A3 G1 G2 G3 of the tile (16/16/8/16/16/8/8/8 MFMAs), 11 loop overhead; slot 15 = s_memrealtime span."""
import ctypes
import json
import math
import os
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

L = _lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
L.sf_tool_attn_prof.restype = ctypes.c_int
L.sf_tool_attn_prof.argtypes = [ctypes.c_void_p]
dev = "cuda"
B, S, nh, nkv, hd = 8, 2048, 32, 8, 128
N = B * S
torch.manual_seed(0)
qkv = torch.randn(N, (nh + 2 * nkv) * hd, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
do = torch.randn(N, nh * hd, device=dev).to(torch.bfloat16)
dq = torch.empty(N, nh * hd, device=dev, dtype=torch.bfloat16)
lse = torch.rand(B, nh, S, device=dev) + 6.0
delta = torch.rand(B, nh, S, device=dev)
kw = dict(B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=1 / math.sqrt(hd))
grid = 8 * ((S // 256) * nh * B // 8)
buf = torch.zeros(grid * 4 * 16, dtype=torch.int64, device=dev)
run = lambda: ops.attn_bwd_dq(q, do, k, v, None, lse, delta, None, dq, **kw)
for _ in range(3):
    run()
torch.cuda.synchronize()
assert L.sf_tool_attn_prof(ctypes.c_void_p(buf.data_ptr())) == 0
names = ["wait_dma", "barrier", "burst", "A0", "A1", "G0", "A2", "A3", "G1", "G2", "G3", "loop"]
mfmas = dict(A0=16, A1=16, G0=8, A2=16, A3=16, G1=8, G2=8, G3=8)
for mode in (0, 1):
    os.environ["SF_ATTN_PROF"] = str(mode)
    buf.zero_()
    run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    out = dict(kernel="dq", mode=mode, kernel_ms=round(ms, 4))
    if mode:
        t = buf.view(grid, 4, 16).double().cpu()
        ticks = t[..., :12].sum(-1)
        rt = t[..., 15]
        out["memtime_MHz"] = round(float(ticks.sum() / rt.sum()) * 100.0, 1)
        out["resident_workgroups_mean"] = round(float(rt.max(dim=1).values.sum()) / 100.0 / (ms * 1e3), 1)
        tot = t.sum(dim=(0, 1))
        out["share"] = {n: round(float(tot[i] / tot[:12].sum()), 4) for i, n in enumerate(names)}
        # tiles a wave actually computes: A0 ticks > 0 per tile is not recorded; use MFMA-bound ticks as the yardstick
        out["ticks_per_mfma"] = {n: round(float(tot[3 + i]) / float(tot[3:11].sum()) * sum(mfmas.values()) / mfmas[n], 3)
                                 for i, n in enumerate(names[3:11])}
        out["in_tile_vs_all"] = round(float(tot[3:11].sum() / tot[:12].sum()), 4)
    print(json.dumps(out), flush=True)
L.sf_tool_attn_prof(ctypes.c_void_p(0))
