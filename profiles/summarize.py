#!/usr/bin/env python3
"""Distil a gpurun_out/prof_<tag>/ directory (profiles/run_profile.sh) into the
text summary committed as profiles/<tag>_summary.txt.

    python3 profiles/summarize.py gpurun_out/prof_<tag> [--algorithmic-bytes-per-frame N] [--frames-per-launch K]
                                  [--note "..."] [--no-latest]

HBM traffic follows MI355X_MICROARCH.md section "HBM": FETCH_SIZE and WRITE_SIZE
are collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE
reports half the bytes of a 16-B-per-lane coalesced read, so it is doubled.
Counters are averaged per launch of the render kernel (the first launch -- warm-up, natural
tile order -- is dropped); the blend kernel of a frame batch is listed beside it.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def rows(pat):
    """Rows of the NEWEST file that matches (gpurun merges a call's outputs into gpurun_out/, so a tag that was
    profiled twice holds both runs' files: only the last run counts)."""
    files = sorted(glob.glob(pat), key=os.path.getmtime)
    return list(csv.DictReader(open(files[-1]))) if files else []


def code_object_resources():
    """Registers, scratch and static LDS of every kernel, from the gfx950 code object's metadata (what the ISA really
    uses: the per-dispatch VGPR_Count / LDS_Block_Size columns of rocprofv3's kernel trace are allocation granules and
    the STATIC LDS only -- they read `48` and `0` for a kernel with 96 VGPRs and 30 KB of dynamic LDS).  Compiles the
    device code to assembly with the product flags (a minute or two; no GPU needed).  Returns {demangled name: dict}."""
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(HERE)
    out = {}
    with tempfile.TemporaryDirectory() as t:
        asm = os.path.join(t, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
               "-fno-slp-vectorize", "-I", os.path.join(root, "include"), "--cuda-device-only", "-S",
               os.path.join(root, "ray_tracer_2_amd", "csrc", "rt_kernel.hip"), "-o", asm]
        if subprocess.run(cmd, capture_output=True).returncode != 0:
            return out
        text = open(asm).read()
    md = text[text.index("amdhsa.kernels:"):]
    for ent in md.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", ent).group(1)
        name = g("name")
        dem = name
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                dem = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
                break
            except OSError:
                continue
        out[dem.replace(" ", "")] = {"vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")), "scratch": int(g("private_segment_fixed_size")),
                                     "static_lds": int(g("group_segment_fixed_size")), "agpr": int(ent.split()[0])}
    return out


def bench_line(d):
    """The JSON line bench.py printed during the trace pass (its stdout is in <dir>/trace.log), if the profiled
    command was bench.py."""
    try:
        for l in reversed(open(f"{d}/trace.log").read().splitlines()):
            if l.startswith("{") and '"metric"' in l:
                return json.loads(l)
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--algorithmic-bytes-per-frame", type=float, default=1920 * 1080 * 32 + 6528)
    ap.add_argument("--frames-per-launch", type=int, default=16)
    ap.add_argument("--note", default="")
    ap.add_argument("--demand-json", default="", help="tools/bench_scene.py BS_JSON output: demand bytes per segment from the stats counters")
    ap.add_argument("--no-resources", action="store_true", help="skip the code-object resource table (needs hipcc)")
    ap.add_argument("--no-latest", action="store_true", help="do not rewrite profiles/latest_traffic.json (not the headline config)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="frames rendered by the whole run (warm-up included): adds a section with the sums over ALL render / walk / "
                         "blend launches per frame -- for deferred-walk sequences, where a batch is many launches")
    args = ap.parse_args()
    d = args.dir
    tag = os.path.basename(d.rstrip("/")).replace("prof_", "")
    cmd = open(f"{d}/command.txt").read().strip() if os.path.exists(f"{d}/command.txt") else "?"
    lines = [f"rocprofv3 summary {tag}: {cmd}", ""]
    if args.note:
        lines += [args.note, ""]
    if os.path.exists(f"{d}/status.txt"):
        lines += ["passes: " + "; ".join(l.strip() for l in open(f"{d}/status.txt")), ""]
    lines.append("== kernel-trace --stats (kernel_stats.csv) ==")
    for r in rows(f"{d}/trace/*/*_kernel_stats.csv"):
        lines.append(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:10.1f} us  "
                     f"min {float(r['MinNs']) / 1e3:10.1f}  max {float(r['MaxNs']) / 1e3:10.1f}  {r['Percentage']}%")
    trace = rows(f"{d}/trace/*/*_kernel_trace.csv")
    # the render kernel = the rt_render instantiation launched most often (tools/bench_scene.py also runs ONE frame of
    # the counter-keeping instantiation at the end: not part of the averages)
    names = collections.Counter(r["Kernel_Name"] for r in trace if "rt_render" in r["Kernel_Name"])
    if not names:   # a wavefront sequence: its walk kernel is the one to describe
        names = collections.Counter(r["Kernel_Name"] for r in trace if "rt_wf_walk" in r["Kernel_Name"])
    render_name = names.most_common(1)[0][0] if names else "rt_render"
    tr = [r for r in trace if r["Kernel_Name"] == render_name]
    bl = [r for r in trace if "rt_blend" in r["Kernel_Name"]]
    dur = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / max(len(rs), 1) * 1e-9
    if tr:
        r = tr[-1]
        res = {} if args.no_resources else code_object_resources()
        lines.append(f"render kernel: {render_name}")
        lines.append(f"  launch: grid {r['Grid_Size_X']} work-items = {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])} workgroups x {r['Workgroup_Size_X']}")
        for name in sorted(set(x["Kernel_Name"] for x in trace if "rt_" in x["Kernel_Name"])):
            cr = res.get(name.replace(" ", ""))
            if cr:
                lines.append(f"  code object: {cr['vgpr']:3d} VGPRs, {cr['agpr']} AGPRs, {cr['sgpr']:3d} SGPRs, {cr['scratch']:3d} B scratch per lane, "
                             f"{cr['static_lds']} B static LDS   {name[:90]}")
        bl_json = bench_line(d)
        if bl_json and "launch" in bl_json.get("roofline", {}):
            ll = bl_json["roofline"]["launch"]
            lines.append(f"  dynamic LDS of the render launch (rt_last_launch): {ll['lds_bytes_per_workgroup']} B per workgroup, "
                         f"{ll['workgroups']} workgroups, scene staged in LDS: {ll['scene_in_lds']}, specialised instantiation: {ll['specialised']}")
        lines.append(f"  (rocprofv3's per-dispatch columns, for the record: VGPR_Count {r['VGPR_Count']}, Accum_VGPR_Count {r.get('Accum_VGPR_Count', '?')}, "
                     f"SGPR_Count {r['SGPR_Count']}, LDS_Block_Size {r['LDS_Block_Size']}, Scratch_Size {r['Scratch_Size']} -- allocation "
                     "granules / static LDS only, not the figures above)")
        steady = tr[1:] if len(tr) > 2 else tr
        lines.append(f"render kernel, launches after the first: avg {dur(steady) * 1e6:.1f} us per launch = "
                     f"{dur(steady) * 1e6 / args.frames_per_launch:.1f} us per frame ({args.frames_per_launch} frames per launch)"
                     + (f"; blend kernel avg {dur(bl) * 1e6:.1f} us per launch" if bl else ""))
    lines.append("")
    lines.append("== PMC passes (mean per launch; render kernel: launches after the first) ==")
    agg, agg_blend = collections.OrderedDict(), collections.OrderedDict()
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_l2"):
        per, perb = collections.defaultdict(list), collections.defaultdict(list)
        for r in rows(f"{d}/{sub}/*/*_counter_collection.csv"):
            if r["Kernel_Name"] == render_name or (not names and "rt_render" in r["Kernel_Name"]):
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            elif "rt_blend" in r["Kernel_Name"]:
                perb[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            v = v[1:] if len(v) > 2 else v
            agg[k] = sum(v) / len(v)
            extra = ""
            if k in perb:
                agg_blend[k] = sum(perb[k]) / len(perb[k])
                extra = f"   blend kernel {agg_blend[k]:14.6g}"
            lines.append(f"{k:26s} {agg[k]:16.6g}   ({len(v)} launches){extra}")
    lines.append("")
    rec = None
    if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
        fetch = agg["FETCH_SIZE"] * 1024 * 2  # KiB -> B, gfx950 x2 correction for 16 B/lane reads
        write = agg["WRITE_SIZE"] * 1024
        bfetch = agg_blend.get("FETCH_SIZE", 0.0) * 1024 * 2
        bwrite = agg_blend.get("WRITE_SIZE", 0.0) * 1024
        total = fetch + write + bfetch + bwrite
        algo = args.algorithmic_bytes_per_frame * args.frames_per_launch
        from ray_tracer_2_amd.build import source_hash
        rec = {"bytes_per_launch": total, "frames_per_launch": args.frames_per_launch, "source_hash": source_hash(),
               "source": f"profiles/{tag}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"}
        lines.append(f"HBM traffic per launch ({args.frames_per_launch} frames): render kernel read {fetch / 1e6:.1f} MB (FETCH_SIZE x 1024 x 2) + "
                     f"write {write / 1e6:.1f} MB (WRITE_SIZE x 1024)"
                     + (f"; blend kernel read {bfetch / 1e6:.1f} MB + write {bwrite / 1e6:.1f} MB" if agg_blend else "")
                     + f" = {total / 1e6:.1f} MB = {total / args.frames_per_launch / 1e6:.1f} MB per frame; "
                     f"algorithmic {algo / 1e6:.1f} MB per launch ({args.algorithmic_bytes_per_frame / 1e6:.1f} MB per frame) -> x{total / algo:.2f}")
    if "SQ_INSTS_VALU" in agg and "SQ_WAVES" in agg:
        lines.append(f"VALU instructions per launch {agg['SQ_INSTS_VALU']:.4g} ({agg['SQ_INSTS_VALU'] / args.frames_per_launch:.4g} per frame); "
                     f"per wave {agg['SQ_INSTS_VALU'] / agg['SQ_WAVES']:.0f}; "
                     f"SALU {agg.get('SQ_INSTS_SALU', 0):.4g}; LDS {agg.get('SQ_INSTS_LDS', 0):.4g}; SMEM {agg.get('SQ_INSTS_SMEM', 0):.4g}")
        if rec is not None:
            rec["valu_instructions_per_launch"] = agg["SQ_INSTS_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in agg and "SQ_ACTIVE_INST_VALU" in agg:
        lu = agg["SQ_THREAD_CYCLES_VALU"] / (64 * agg["SQ_ACTIVE_INST_VALU"])
        lines.append(f"VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {lu:.3f}")
        if rec is not None:
            rec["valu_lane_utilisation"] = lu
    if "SQ_WAIT_ANY" in agg and "SQ_WAVE_CYCLES" in agg:
        lines.append(f"wave time: waiting (s_waitcnt) {agg['SQ_WAIT_ANY'] / agg['SQ_WAVE_CYCLES']:.1%}, issue stalls "
                     f"{agg.get('SQ_WAIT_INST_ANY', 0) / agg['SQ_WAVE_CYCLES']:.1%}, issuing {agg.get('SQ_ACTIVE_INST_ANY', 0) / agg['SQ_WAVE_CYCLES']:.1%} "
                     "(SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; pmc_sq and pmc_sq2 are different runs)")
    if "TCC_HIT_sum" in agg and "TCC_MISS_sum" in agg:
        lines.append(f"L2 hit rate {agg['TCC_HIT_sum'] / (agg['TCC_HIT_sum'] + agg['TCC_MISS_sum']):.1%}")
    if "GRBM_GUI_ACTIVE" in agg and tr:
        steady = tr[1:] if len(tr) > 2 else tr
        clk = agg["GRBM_GUI_ACTIVE"] / 8 / dur(steady) / 1e9
        lines.append(f"effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {clk:.2f} GHz")
        if rec is not None:
            rec["shader_clock_ghz"] = clk
        if "SQ_INSTS_VALU" in agg:
            simds = 1024
            rate = agg["SQ_INSTS_VALU"] / dur(steady) / 1e9
            lines.append(f"VALU issue: {rate:.0f} G wave-instructions/s = {rate / (simds * clk / 2):.2f} of the guide's peak "
                         f"(one wave64 instruction per 2 cycles per SIMD x {simds} SIMDs x {clk:.2f} GHz = {simds * clk / 2:.0f} G/s)")
    whole_run_hbm = None
    if args.total_frames:
        F = args.total_frames
        fam = lambda name: ("walk" if "rt_walk" in name or "rt_wf_walk" in name else "shade" if "rt_wf_shade" in name else
                            "blend" if "rt_blend" in name else "render" if "rt_render" in name else None)
        lines.append("")
        lines.append(f"== whole run: sums over all render / walk / blend launches, per frame ({F} frames incl. warm-up) ==")
        tsum = collections.Counter()
        nl = collections.Counter()
        for r in trace:
            f = fam(r["Kernel_Name"])
            if f:
                tsum[f] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                nl[f] += 1
        tot_t = sum(tsum.values())
        lines.append("kernel time per frame: " + ", ".join(f"{k} {tsum[k] / F * 1e3:.3f} ms ({nl[k]} launches)" for k in ("render", "shade", "walk", "blend") if nl[k])
                     + f" = {tot_t / F * 1e3:.3f} ms")
        csum = collections.defaultdict(collections.Counter)
        for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_l2"):
            for r in rows(f"{d}/{sub}/*/*_counter_collection.csv"):
                f = fam(r["Kernel_Name"])
                if f:
                    csum[r["Counter_Name"]][f] += float(r["Counter_Value"])
        tc = lambda k: sum(csum[k].values())
        if "FETCH_SIZE" in csum and "WRITE_SIZE" in csum:
            hb = tc("FETCH_SIZE") * 2048 + tc("WRITE_SIZE") * 1024
            whole_run_hbm = hb / F
            lines.append(f"HBM traffic per frame: read {tc('FETCH_SIZE') * 2048 / F / 1e6:.1f} MB + write {tc('WRITE_SIZE') * 1024 / F / 1e6:.1f} MB = {hb / F / 1e6:.1f} MB "
                         f"(walk kernels: {(csum['FETCH_SIZE']['walk'] * 2048 + csum['WRITE_SIZE']['walk'] * 1024) / F / 1e6:.1f} MB) = "
                         f"{hb / tot_t / 1e9:.0f} GB/s = {hb / tot_t / 8e12:.3f} of the 8 TB/s peak")
        if "SQ_INSTS_VALU" in csum:
            lines.append(f"VALU instructions per frame: {tc('SQ_INSTS_VALU') / F:.4g} (" + ", ".join(f"{k} {csum['SQ_INSTS_VALU'][k] / F:.4g}" for k in ("render", "shade", "walk") if csum['SQ_INSTS_VALU'][k]) + ")")
        if "SQ_THREAD_CYCLES_VALU" in csum and "SQ_ACTIVE_INST_VALU" in csum:
            lu = lambda f: csum["SQ_THREAD_CYCLES_VALU"][f] / max(64 * csum["SQ_ACTIVE_INST_VALU"][f], 1)
            lines.append("VALU lane utilisation: " + ", ".join(f"{k} kernels {lu(k):.3f}" for k in ("render", "shade", "walk") if csum["SQ_ACTIVE_INST_VALU"][k])
                         + f", all {tc('SQ_THREAD_CYCLES_VALU') / (64 * tc('SQ_ACTIVE_INST_VALU')):.3f}")
        if "SQ_WAIT_ANY" in csum and "SQ_WAVE_CYCLES" in csum:
            lines.append("wave time in s_waitcnt: " + ", ".join(f"{k} kernels {csum['SQ_WAIT_ANY'][k] / max(csum['SQ_WAVE_CYCLES'][k], 1):.1%}"
                                                                for k in ("render", "shade", "walk") if csum["SQ_WAVE_CYCLES"][k])
                         + f", all {tc('SQ_WAIT_ANY') / max(tc('SQ_WAVE_CYCLES'), 1):.1%}")
        if "TCC_HIT_sum" in csum:
            lines.append(f"L2 hit rate {tc('TCC_HIT_sum') / (tc('TCC_HIT_sum') + tc('TCC_MISS_sum')):.1%}")
    if args.demand_json and os.path.exists(args.demand_json):
        dj = json.load(open(args.demand_json))
        lines.append("")
        lines.append(f"== demand vs counter bytes ({dj['scene']}, {dj['width']}x{dj['height']}, {dj['spp']} spp, {dj['bounces']} bounces) ==")
        lines.append(f"scene: {dj['triangles']} triangles, {dj['nodes']} nodes, {dj['meshes']} meshes = {dj.get('scene_bytes_reference_layout', 0) / 1e6:.1f} MB in the reference's layout")
        lines.append(f"un-profiled run: {dj['ms_per_frame']:.3f} ms/frame, {dj['rays_per_frame'] / 1e6:.2f} M rays/frame ({dj['rays_traversed_per_frame'] / 1e6:.2f} M traversed), {dj['mrays_per_s']:.0f} Mrays/s")
        if "demand_bytes_per_segment" in dj:
            lines.append(f"stats counters (wgsl:307,322): {dj['node_tests_per_segment']:.1f} box tests + {dj['triangle_tests_per_segment']:.1f} triangle tests per segment "
                         f"-> demand = box tests x 48 + triangle tests x 96 + meshes x 240 = {dj['demand_bytes_per_segment']:.0f} B per segment = "
                         f"{dj['demand_bytes_per_frame'] / 1e9:.2f} GB per frame")
            if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
                per_frame = (agg["FETCH_SIZE"] * 2048 + agg["WRITE_SIZE"] * 1024 + agg_blend.get("FETCH_SIZE", 0) * 2048 + agg_blend.get("WRITE_SIZE", 0) * 1024) / args.frames_per_launch
                if args.total_frames and whole_run_hbm is not None:
                    per_frame = whole_run_hbm   # a sequence of launches per batch: the sums over ALL launches, per frame
                lines.append(f"counter bytes (HBM): {per_frame / 1e6:.1f} MB per frame = {per_frame / dj['demand_bytes_per_frame']:.4f} of the demand bytes "
                             f"(the rest is served by LDS / L1 / L2); compulsory = image {dj['width'] * dj['height'] * 32 / 1e6:.1f} MB + scene once "
                             f"{dj.get('scene_bytes_reference_layout', 0) / 1e6:.1f} MB")
                lines.append(f"HBM roofline: {per_frame / (dj['kernel_ms_per_frame'] * 1e-3) / 1e9:.0f} GB/s of counter traffic = "
                             f"{per_frame / (dj['kernel_ms_per_frame'] * 1e-3) / 8e12:.4f} of the 8 TB/s peak; demand rate "
                             f"{dj['demand_bytes_per_frame'] / (dj['kernel_ms_per_frame'] * 1e-3) / 1e12:.2f} TB/s")
    if rec is not None and not args.no_latest:
        json.dump(rec, open(os.path.join(HERE, "latest_traffic.json"), "w"))
    out = os.path.join(HERE, f"{tag}_summary.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
