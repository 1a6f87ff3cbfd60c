"""Where the forward attention kernel's time goes: per-wave cycle sums of the tile-loop phases (tools build's SF_PROF_* stamps,
s_memtime at 100 MHz .. shader clock depending on the part; only ratios are used).   python tools/attn_phases.py   (GPU box)"""
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
o = torch.empty(N, nh * hd, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, nh, S, device=dev)
kw = dict(B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=1 / math.sqrt(hd))
grid = 8 * ((S // 128) * nh * B // 8)
buf = torch.zeros(grid * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.attn_fwd(q, k, v, [], [], None, o, lse, **kw)
torch.cuda.synchronize()
assert L.sf_tool_attn_prof(ctypes.c_void_p(buf.data_ptr())) == 0
names = ["wait_dma", "barrier", "stage_issue", "qk", "drain_max", "exp_pv", "loop_total"]
for mode in (0, 1, 2):
    os.environ["SF_ATTN_PROF"] = str(mode)
    buf.zero_()
    ops.attn_fwd(q, k, v, [], [], None, o, lse, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.attn_fwd(q, k, v, [], [], None, o, lse, **kw)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    out = dict(mode=mode, kernel_ms=round(ms, 4))
    if mode:
        t = buf.view(grid, 4, 8).double().cpu()
        ticks = t[..., :7].sum(-1)                      # s_memtime ticks spent in the tile loop, per wave
        rt = t[..., 7]                                  # s_memrealtime (100 MHz) span of the same region
        out["loop_ticks_mean"] = round(float(ticks.mean()), 1)
        out["loop_us_mean_realtime"] = round(float(rt.mean()) / 100.0, 2)
        out["memtime_MHz"] = round(float(ticks.sum() / rt.sum()) * 100.0, 1)
        # mean number of workgroups in their tile loop at once = sum of per-workgroup spans / kernel time
        out["resident_workgroups_mean"] = round(float(rt[:, 0].sum()) / 100.0 / (ms * 1e3), 1)
        if mode == 2:
            tot = t.sum(dim=(0, 1))
            out["share"] = {n: round(float(tot[i] / tot[:6].sum()), 4) for i, n in enumerate(names[:6])}
            out["ticks_per_tile"] = {n: round(float(tot[i] / (grid * 4) / 16.5), 1) for i, n in enumerate(names[:6])}
    print(json.dumps(out), flush=True)
L.sf_tool_attn_prof(ctypes.c_void_p(0))
