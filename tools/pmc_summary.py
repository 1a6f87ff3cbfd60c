"""Aggregate a rocprofv3 --pmc run (counter_collection.csv) per kernel family.
usage: python tools/pmc_summary.py <rocprof output dir> <out.json> [commit]   (the commit is stored as "_commit": bench.py stamps
roofline.traffic with it)"""
import csv, glob, json, re, sys
from collections import defaultdict

FAMILIES = [("gemm256w4", r"gemm_nt_256w4_kernel"), ("gemm_tn", r"gemm_tn_256w4p?_kernel"), ("gemm256", r"gemm_nt_256_kernel|gemm_nt_256v2"), ("gemm128", r"gemm_nt_kernel"),
            ("attn_fwd", r"attn_fwd_(w1_)?kernel"), ("attn_bwd_dq", r"attn_bwd_dq_(w1_)?kernel"), ("attn_bwd_dkv", r"attn_bwd_dkv_kernel|attn_bwd_dkv_rs_kernel|attn_bwd_dkv_pair_kernel"),
            ("attn_bwd_pre", r"attn_bwd_pre_kernel"), ("attn_bwd_diag", r"attn_bwd_diag_kernel"), ("transpose", r"transpose_kernel"), ("ce_fused", r"ce_fused_kernel"),
            ("swiglu_fwd", r"swiglu_fwd_kernel"), ("swiglu_bwd", r"swiglu_bwd_kernel"), ("adamw", r"adamw_kernel"),
            ("teacher_reduce", r"teacher_reduce_kernel"), ("rmsnorm_bwd", r"rmsnorm_bwd_kernel")]


def family(name):
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return None


def main(d, out, commit=None):
    # one csv per rocprofv3 run; several runs of the same command (one counter group each) may sit under `d`
    agg = defaultdict(lambda: defaultdict(float))
    nlaunch = defaultdict(int)
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        per_file = defaultdict(set)
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            per_file[fam].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
        for fam, ids in per_file.items():
            nlaunch[fam] = max(nlaunch[fam], len(ids))
    launches = {fam: range(n) for fam, n in nlaunch.items()}
    res = {}
    for fam, c in agg.items():
        e = {"launches": len(launches[fam])}
        e.update({k + "_sum": v for k, v in c.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            e["MfmaUtil_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "FETCH_SIZE" in c:
            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, calibrated on AdamW)
            e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0 / e["launches"]
        res[fam] = e
    if commit:
        res["_commit"] = commit
    json.dump(res, open(out, "w"), indent=1)
    for fam, e in res.items():
        if not isinstance(e, dict):
            continue
        print(fam, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
