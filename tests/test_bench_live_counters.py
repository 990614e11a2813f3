"""bench.py measures roofline.traffic / roofline_valu in its own run: child passes under `rocprofv3 --pmc`, parsed here
against a stub profiler that writes counter files of the shape rocprofv3 writes (no GPU needed): launches after the
first are averaged, the blend kernel is added, FETCH_SIZE is KiB x 2 and WRITE_SIZE KiB (MI355X_MICROARCH.md), a failing
profiler gives a reason instead of a number."""
import argparse
import importlib.util
import os
import stat
import sys

from conftest import ROOT

STUB = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counters = []
i = a.index("--pmc") + 1
while not a[i].startswith("-"):
    counters.append(a[i]); i += 1
d = a[a.index("-d") + 1]
if os.environ.get("STUB_FAIL") == counters[0]:
    sys.stderr.write("stub: no counters today\n"); sys.exit(3)
os.makedirs(os.path.join(d, "host"), exist_ok=True)
render = "void rtd::rt_render_persistent_kernel<true, false, false, false, true, false>(rtd::RenderArgs)"
val = {"FETCH_SIZE": [9e9, 1000.0, 3000.0], "WRITE_SIZE": [9e9, 500.0, 700.0], "SQ_INSTS_VALU": [1.0, 64e9, 64e9],
       "SQ_ACTIVE_INST_VALU": [1.0, 4e9, 4e9], "SQ_THREAD_CYCLES_VALU": [1.0, 128e9, 128e9], "GRBM_GUI_ACTIVE": [1.0, 8 * 2.4e9 * 0.04, 8 * 2.4e9 * 0.04]}
blend = {"FETCH_SIZE": 100.0, "WRITE_SIZE": 10.0}
with open(os.path.join(d, "host", "1_counter_collection.csv"), "w") as f:
    f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n')
    disp = 0
    for k in range(3):
        for name, table in ((render, val), ("rtd::rt_blend_frames_kernel(rtd::BlendArgs)", blend)):
            disp += 1
            for c in counters:
                if c in table:
                    v = table[c][k] if isinstance(table[c], list) else table[c]
                    f.write(f'{disp},{disp},"Agent 2",1,1,1,327680,10,"{name}",256,0,0,96,0,96,"{c}",{v:.6f},{1000 + disp * 100000000},{1000 + disp * 100000000 + 40000000}\n')
'''


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_counter_passes_are_parsed_and_converted(tmp_path, monkeypatch):
    stub = tmp_path / "rocprofv3"
    stub.write_text(STUB)
    stub.chmod(stub.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    bench = load_bench()
    args = argparse.Namespace(width=1920, height=1080)
    rec, why = bench.live_traffic(args)
    assert why is None, why
    # launches 2 and 3 (the first is dropped): FETCH (1000 + 3000) / 2 + blend 100 KiB, x 1024 x 2; WRITE (500 + 700) / 2 + 10 KiB x 1024
    assert rec["read_bytes_per_launch"] == (2000.0 + 100.0) * 1024 * 2
    assert rec["write_bytes_per_launch"] == (600.0 + 10.0) * 1024
    assert rec["bytes_per_launch"] == rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"] and rec["frames_per_launch"] == 32
    v = rec["valu"]
    assert v["valu_instructions_per_launch"] == 64e9 and abs(v["valu_lane_utilisation"] - 0.5) < 1e-12
    assert abs(v["shader_clock_ghz"] - 2.4) < 1e-9 and rec["valu_error"] is None
    # a failing VALU pass leaves the traffic standing; a failing traffic pass gives a reason
    monkeypatch.setenv("STUB_FAIL", "SQ_INSTS_VALU")
    rec, why = bench.live_traffic(args)
    assert why is None and rec["valu"] is None and "exit 3" in rec["valu_error"] and "no counters today" in rec["valu_error"]
    monkeypatch.setenv("STUB_FAIL", "WRITE_SIZE")
    rec, why = bench.live_traffic(args)
    assert rec is None and "WRITE_SIZE" in why and "exit 3" in why
