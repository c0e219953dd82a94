#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 PMC passes written by tools/pmc_run.sh:
    python tools/pmc_summary.py gpurun_out/<tag> > profiles/<round>_pmc_summary.json
HBM traffic per launch (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, both KiB per
dispatch), MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES summed over SEs ... reported as busy cycles per
kernel-active cycle of the 1024 SIMDs via GRBM_GUI_ACTIVE), LDS bank-conflict share, L2 hit rate."""
import csv, glob, json, os, re, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(directory):
    files = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = re.sub(r"^void\s+", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name).replace("dm::", "")
            c = r["Counter_Name"]
            tot[name][c] += float(r["Counter_Value"])
            cnt[name][c] += 1
    return tot, cnt


def main():
    tag = sys.argv[1]
    from bench import csrc_sha
    out = {"_csrc_sha": csrc_sha(), "_tag": os.path.basename(tag)}
    if os.environ.get("PMC_ITERS"):  # iterations of the profiled loop (training step: tools/final_profiles.sh)
        out["_iterations"] = int(os.environ["PMC_ITERS"])
        out["_command"] = os.environ.get("PMC_CMD", "")
    ft, fc = load(tag + "_fetch")
    wt, wc = load(tag + "_write")
    mt, mc = load(tag + "_mfma")
    lt, lc = load(tag + "_l2")
    for k in sorted(ft):
        n = fc[k]["FETCH_SIZE"]
        row = {"launches": n,
               "fetch_bytes_per_launch_corrected": ft[k]["FETCH_SIZE"] / n * 1024.0 * 2.0,
               "write_bytes_per_launch": wt[k]["WRITE_SIZE"] / max(wc[k]["WRITE_SIZE"], 1) * 1024.0}
        if k in mt and mc[k].get("GRBM_GUI_ACTIVE"):
            m = mt[k]
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs; MFMA busy counts SIMD cycles
            gui = m["GRBM_GUI_ACTIVE"] / 8.0
            row["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / gui if gui else None
            row["lds_conflict_share"] = (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]) if m.get("SQ_LDS_IDX_ACTIVE") else None
            row["mfma_mops_f32_per_launch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] / mc[k]["SQ_INSTS_VALU_MFMA_MOPS_F32"]
        if k in lt:
            h, mi = lt[k]["TCC_HIT_sum"], lt[k]["TCC_MISS_sum"]
            row["l2_hit_rate"] = h / (h + mi) if h + mi else None
        out[k] = row
    allf = sum(ft[k]["FETCH_SIZE"] for k in ft) * 2048.0
    allw = sum(wt[k]["WRITE_SIZE"] for k in wt) * 1024.0
    h = sum(lt[k]["TCC_HIT_sum"] for k in lt); mi = sum(lt[k]["TCC_MISS_sum"] for k in lt)
    out["_total"] = {"fetch_bytes_corrected": allf, "write_bytes": allw, "l2_hit_rate": h / (h + mi) if h + mi else None}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
