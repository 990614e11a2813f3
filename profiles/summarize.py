#!/usr/bin/env python3
"""Distil a gpurun_out/prof_<tag>/ directory (profiles/run_profile.sh) into the
text summary committed as profiles/<tag>_summary.txt.

HBM traffic follows MI355X_MICROARCH.md section "HBM": FETCH_SIZE and WRITE_SIZE
are collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE
reports half the bytes of a 16-B-per-lane coalesced read, so it is doubled.
"""
import collections
import csv
import glob
import os
import sys


def rows(pat):
    out = []
    for f in glob.glob(pat):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    d = sys.argv[1]
    tag = os.path.basename(d.rstrip("/")).replace("prof_", "")
    lines = [f"rocprofv3 summary {tag}: python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "
             "(CornellBox-Original 1920x1080, 8 spp, 4 bounces, 1 x MI355X)", ""]
    lines.append("== kernel-trace --stats (kernel_stats.csv) ==")
    for r in rows(f"{d}/trace/*/*_kernel_stats.csv"):
        lines.append(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:10.1f} us  "
                     f"min {float(r['MinNs']) / 1e3:10.1f}  max {float(r['MaxNs']) / 1e3:10.1f}  {r['Percentage']}%")
    tr = [r for r in rows(f"{d}/trace/*/*_kernel_trace.csv") if "rt_render" in r["Kernel_Name"]]
    if tr:
        r = tr[-1]
        lines.append(f"render kernel: grid {r['Grid_Size_X']} x wg {r['Workgroup_Size_X']}, VGPR_Count {r['VGPR_Count']}, "
                     f"SGPR_Count {r['SGPR_Count']}, LDS_Block_Size {r['LDS_Block_Size']}, Scratch_Size {r['Scratch_Size']}")
    lines.append("")
    lines.append("== PMC passes (per render launch; mean over the accumulating launches, frames >= 1) ==")
    agg = collections.OrderedDict()
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_l2"):
        per = collections.defaultdict(list)
        for r in rows(f"{d}/{sub}/*/*_counter_collection.csv"):
            if "rt_render" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            v = v[1:] if len(v) > 2 else v  # drop the frames=0 launch (no image read)
            agg[k] = sum(v) / len(v)
            lines.append(f"{k:26s} {agg[k]:16.6g}   ({len(v)} launches)")
    lines.append("")
    if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
        fetch = agg["FETCH_SIZE"] * 1024 * 2  # KiB -> B, gfx950 x2 correction for 16 B/lane reads
        write = agg["WRITE_SIZE"] * 1024
        import json
        rec = {"bytes_per_launch": fetch + write, "source": f"profiles/{tag}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"}
        if "SQ_INSTS_VALU" in agg:
            rec["valu_instructions_per_launch"] = agg["SQ_INSTS_VALU"]
        if "SQ_THREAD_CYCLES_VALU" in agg and "SQ_ACTIVE_INST_VALU" in agg:
            rec["valu_lane_utilisation"] = agg["SQ_THREAD_CYCLES_VALU"] / (64 * agg["SQ_ACTIVE_INST_VALU"])
        if "GRBM_GUI_ACTIVE" in agg and tr:
            dur_ = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr) / len(tr) * 1e-9
            rec["shader_clock_ghz"] = agg["GRBM_GUI_ACTIVE"] / 8 / dur_ / 1e9
        json.dump(rec, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "latest_traffic.json"), "w"))
        lines.append(f"HBM traffic per launch: read {fetch / 1e6:.1f} MB (FETCH_SIZE x 1024 x 2) + write {write / 1e6:.1f} MB "
                     f"(WRITE_SIZE x 1024) = {(fetch + write) / 1e6:.1f} MB; algorithmic 66.4 MB")
    if "SQ_INSTS_VALU" in agg and "SQ_WAVES" in agg:
        lines.append(f"VALU instructions per launch {agg['SQ_INSTS_VALU']:.4g}; per wave {agg['SQ_INSTS_VALU'] / agg['SQ_WAVES']:.0f}; "
                     f"SALU {agg.get('SQ_INSTS_SALU', 0):.4g}; LDS {agg.get('SQ_INSTS_LDS', 0):.4g}; SMEM {agg.get('SQ_INSTS_SMEM', 0):.4g}")
    if "SQ_THREAD_CYCLES_VALU" in agg and "SQ_ACTIVE_INST_VALU" in agg:
        lines.append(f"VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = "
                     f"{agg['SQ_THREAD_CYCLES_VALU'] / (64 * agg['SQ_ACTIVE_INST_VALU']):.3f}")
    if "GRBM_GUI_ACTIVE" in agg and tr:
        dur = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr) / len(tr) * 1e-9
        lines.append(f"effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {agg['GRBM_GUI_ACTIVE'] / 8 / dur / 1e9:.2f} GHz")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_summary.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
