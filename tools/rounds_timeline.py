#!/usr/bin/env python3
"""Per-launch timeline of the LAST batch in a rocprofv3 --kernel-trace of tools/rounds_trace.py:
    python tools/rounds_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("render_persistent", "walk_kernel", "blend_frames"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
blends = [i for i, r in enumerate(rows) if "blend" in r["Kernel_Name"]]
first = blends[-2] + 1 if len(blends) > 1 else 0
t0 = int(rows[first]["Start_Timestamp"])
tot = {"render": 0.0, "walk": 0.0, "blend": 0.0}
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = "walk" if "walk" in r["Kernel_Name"] else "blend" if "blend" in r["Kernel_Name"] else "render"
    tot[k] += (e - s) / 1e6
    print(f"{k:7s} start {(s - t0) / 1e6:9.3f} ms   {(e - s) / 1e6:8.3f} ms")
print("sums:", {k: round(v, 3) for k, v in tot.items()}, "span", round((int(rows[-1]["End_Timestamp"]) - t0) / 1e6, 3), "ms")
