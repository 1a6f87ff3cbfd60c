"""Per-workgroup timeline of the NT GEMM (TOOLS build, SF_GEMM_CYC=2): every workgroup stores s_memrealtime (100 MHz) at
entry / loop begin / loop end / exit plus HW_ID at its tile origin.  Prints, per shape: mean prologue, loop, epilogue time
and the gap between consecutive workgroups on the same CU (exit of one -> entry of the next).   (GPU box)"""
import collections
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
os.environ["SF_GEMM_CYC"] = os.environ.get("CYC", "2")
dev = "cuda"
for (M, N, K) in [(16384, 4096, 14336), (16384, 32000, 4096), (16384, 4096, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(a, b, c)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gemm_nt(a, b, c)
    e.record()
    torch.cuda.synchronize()
    w = c.view(torch.int32)                      # [M, N/2] words
    tiles = w[::256].cpu()
    rec = []
    for tm in range(tiles.shape[0]):
        for tn in range((N + 255) // 256):
            o = tiles[tm, tn * 128: tn * 128 + 6].tolist()
            rec.append([x & 0xffffffff for x in o])
    t0 = min(r[0] for r in rec)
    pro = [(r[1] - r[0]) / 100 for r in rec]
    loop = [(r[2] - r[1]) / 100 for r in rec]
    epi = [(r[3] - r[2]) / 100 for r in rec]
    by_cu = collections.defaultdict(list)
    for r in rec:
        hw, xcc = r[4], r[5] & 0xf
        by_cu[(xcc, (hw >> 13) & 0x7, (hw >> 12) & 0x1, (hw >> 8) & 0xf)].append((r[0], r[3]))
    gaps = []
    for k, v in by_cu.items():
        v.sort()
        gaps += [(v[i + 1][0] - v[i][1]) / 100 for i in range(len(v) - 1)]
    mean = lambda x: round(sum(x) / max(1, len(x)), 2)
    print(json.dumps(dict(shape=[M, N, K], kernel_us=round(s.elapsed_time(e) * 1000, 1), wgs=len(rec), cus_seen=len(by_cu),
                          prologue_us=mean(pro), loop_us=mean(loop), epilogue_us=mean(epi), gap_us=mean(gaps),
                          gap_max=round(max(gaps), 2) if gaps else None, span_us=(max(r[3] for r in rec) - t0) / 100,
                          wg_total_us=mean([(r[3] - r[0]) / 100 for r in rec]))), flush=True)
