#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (and ms/frame) of the path-trace hot
path on CornellBox-Original at 1920x1080, 8 spp, 4 bounces (BASELINE.json
configs[1]), on N GPUs of one node.

A "step" is one frame: one pass of the hot path (wgsl `main`) over the whole
image with `Params.frames` advancing by one per frame (progressive accumulation
on, as the reference's App::update does).  1 ray = 1 path segment = one
calculate_ray_collions call (wgsl:353), counted exactly on the device; the
segments whose hit is taken from the per-pixel primary-ray memo (no traversal)
are counted too and `Mrays_traversed/s` is reported beside the headline.

The window of K frames between two fences (barrier + synchronize) is timed `--repeats` times (default 5, each window
continuing the accumulation): `ms_per_step` and `value` are the MEDIAN window's, `timing` carries min / max / every window.
A window renders its K frames with rt_render_frames: frames are sampled in
batches (default 32 per launch) by ONE persistent launch over (frame, tile) work
items and blended in frame order by a dense second kernel -- bit-identical to K
single-frame launches (tests/test_gpu_frames.py).  The single-frame
(un-overlapped) latency and the first frame after a camera change are measured
outside the timed region and reported as extra keys.

`value` is the batched figure (`value_batched` is the same number under an
explicit name); `value_per_frame_launch` is the throughput with ONE launch per
frame -- what a host that presents every frame gets, the reference's only mode
(src/core/app.rs:285-340) -- measured right after the timed region.  Of the
counted rays `config.reused_fraction` are primary segments served from the
per-pixel memo without a traversal.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  The frame
is split into 8-row strips dealt round-robin to the ranks (strong scaling: the
frame is fixed); every batch each rank renders its strips and ONE gather over
xGMI assembles the accumulated frame on rank 0 (one gather per BATCH in the
timed region; the same frames with one launch and one gather per FRAME are
timed afterwards and reported as `value_per_frame_launch`).  value = rays of
the whole frames, all ranks, per second of the slowest rank.

Launch: `python bench.py --gpus N` starts its own N ranks (one child process
per GPU through torch.distributed.run, before this process touches the GPU);
under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

Prints one JSON line on rank 0 (contract in the task statement), with
`roofline` (compulsory HBM bytes per launch / measured launch time against the
8 TB/s peak -- this path is VALU-bound, so the fraction is small by
construction; `roofline_valu` gives the instruction-issue view against the
guide's peak) and `cpu_baseline` (the CPU oracle timed on the host cores on a
bounded sample).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (before anything initialises the HIP runtime: the library's pipelined frames want a hardware queue per stream, see
# ray_tracer_2_amd/__init__.py; a setting of the caller's wins)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

WIDTH, HEIGHT, SPP, BOUNCES = 1920, 1080, 8, 4
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(rt, arrays):
    """Oracle (CPU restatement of ray_tracer.wgsl, NOT wgpu/lavapipe) on a
    bounded sample of the same workload: evenly spaced 8-row strips of frame 0."""
    from oracle import oracle
    import numpy as np
    cores = oracle.hardware_threads()
    p = rt.make_params(WIDTH, HEIGHT, BOUNCES, SPP, skybox=1, frames=0)
    img = np.zeros((HEIGHT, WIDTH, 4), np.float32)
    n_strips = (HEIGHT + 7) // 8

    def run(strips):
        rows = np.concatenate([np.arange(s * 8, min(s * 8 + 8, HEIGHT)) for s in strips]).astype(np.uint32)
        t0 = time.perf_counter()
        _, st = oracle.render(p, arrays, image=img, rows=rows, threads=cores)
        return st.segments, time.perf_counter() - t0

    # calibrate on a few strips, then size the sample for ~12 s of wall time: evenly spaced strips of frame 0, or -- when
    # a whole frame takes less than that -- several whole frames (frames 0, 1, ...: other seeds, the same workload)
    cal = sorted(set(int(i * n_strips / 8) for i in range(8)))
    seg, t = run(cal)
    want = int(max(8, 12.0 / max(t / len(cal), 1e-6)))
    if want <= n_strips:
        strips = sorted(set(int(i * n_strips / want) for i in range(want)))
        seg, t = run(strips)
        what = f"{len(strips)} of {n_strips} evenly spaced 8-row strips of frame 0"
    else:
        n_frames = min(64, -(-want // n_strips))
        seg, t = 0, 0.0
        for f in range(n_frames):
            p.frames = -f   # (frames <= 0: a plain store; |frames| is the seed, wgsl:475)
            s1, t1 = run(range(n_strips))
            seg, t = seg + s1, t + t1
        what = f"{n_frames} whole frames (seeds 0..{n_frames - 1})"
    return {"value": seg / t / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{what} of the same workload ({seg} rays in {t:.1f} s); CPU restatement of ray_tracer.wgsl, not wgpu/lavapipe"}


def live_traffic(args):
    """HBM bytes per launch of the render (+ blend) kernel, measured IN THIS RUN: two child runs of this script's timed loop
    under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE each in a pass of its own, converted as MI355X_MICROARCH.md
    prescribes (KiB; FETCH_SIZE x 2 on gfx950) -- started BEFORE this process touches the GPU (a process that has
    initialised HIP must not start programs on this pool; the children are ordinary processes of their own).
    Returns (record, None) or (None, reason); the committed profile (profiles/latest_traffic.json) is the fallback."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    # (the product library must exist before the profiler starts: no compile under rocprofv3)
    from ray_tracer_2_amd import build
    if not os.path.exists(build.PRODUCT_SO):
        return None, "product library not built"
    frames = 32
    child = [sys.executable, os.path.abspath(__file__), "--steps", "64", "--warmup", "32", "--batch", str(frames),
             "--width", str(args.width), "--height", str(args.height),
             "--no-cpu-baseline", "--no-extras", "--no-per-frame-leg", "--no-live-traffic", "--repeats", "1"]
    out = tempfile.mkdtemp(prefix="rt2_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sums = {}

    def one_pass(tag, counters):
        """{counter: (render mean per launch, blend mean per launch, launches, kernel, mean render launch seconds)} or a reason"""
        d = os.path.join(out, tag)
        # (a process group of its own: on a timeout the profiler AND the program under it are ended, by that group's id)
        proc = subprocess.Popen([prof, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", *child], cwd="/tmp", env=env,
                                stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.communicate()
            return f"rocprofv3 --pmc {' '.join(counters)}: no result within 300 s"
        if proc.returncode != 0:
            return f"rocprofv3 --pmc {' '.join(counters)}: exit {proc.returncode}: " + err.decode(errors="replace")[-200:]
        rows = []
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(path)))
        if rows and "Dispatch_Id" in rows[0]:
            rows.sort(key=lambda x: int(x["Dispatch_Id"]))
        names = collections.Counter(x["Kernel_Name"] for x in rows if "rt_render" in x["Kernel_Name"])
        if not names:
            return f"rocprofv3 --pmc {' '.join(counters)}: no render kernel in the counter file"
        name = names.most_common(1)[0][0]
        res = {}
        for counter in counters:
            mine = [x for x in rows if x.get("Counter_Name") == counter and x["Kernel_Name"] == name]
            if not mine:
                return f"rocprofv3 --pmc: no {counter} rows for the render kernel"
            mine = mine[1:] if len(mine) > 2 else mine          # (launches after the first, as profiles/summarize.py)
            v = [float(x["Counter_Value"]) for x in mine]
            secs = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-9 for x in mine] if "End_Timestamp" in mine[0] else [0.0]
            b = [float(x["Counter_Value"]) for x in rows if x.get("Counter_Name") == counter and "rt_blend" in x["Kernel_Name"]]
            res[counter] = (sum(v) / len(v), sum(b) / len(b) if b else 0.0, len(v), name, sum(secs) / len(secs))
        return res

    valu, valu_error = None, None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            got = one_pass(counter, [counter])
            if isinstance(got, str):
                return None, got
            sums.update(got)
        # (a third pass for the bound that matters -- VALU issue; its failure does not void the traffic)
        got = one_pass("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"])
        if isinstance(got, str):
            valu_error = got
        else:
            secs = got["GRBM_GUI_ACTIVE"][4]
            valu = {"valu_instructions_per_launch": got["SQ_INSTS_VALU"][0],
                    "valu_lane_utilisation": got["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * got["SQ_ACTIVE_INST_VALU"][0]),
                    "shader_clock_ghz": got["GRBM_GUI_ACTIVE"][0] / 8.0 / secs / 1e9 if secs > 0 else None}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    fetch = (sums["FETCH_SIZE"][0] + sums["FETCH_SIZE"][1]) * 1024 * 2   # KiB -> B, gfx950: x 2
    write = (sums["WRITE_SIZE"][0] + sums["WRITE_SIZE"][1]) * 1024
    return {"valu": valu, "valu_error": valu_error, "kernel_name": sums["FETCH_SIZE"][3], "bytes_per_launch": fetch + write, "frames_per_launch": frames, "read_bytes_per_launch": fetch, "write_bytes_per_launch": write,
            "source": f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, over `bench.py --steps 64 --warmup 32 "
                      f"--batch {frames}` ({sums['FETCH_SIZE'][2]} launches of {sums['FETCH_SIZE'][3][:60]} after the first, plus the blend kernel); "
                      "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md)"}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the window of --steps frames is timed (ms_per_step = the median window)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the un-overlapped / first-frame measurements")
    ap.add_argument("--no-per-frame-leg", action="store_true",
                    help="skip the one-launch-per-frame leg after the timed region (profiling runs: one kind of launch only)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (CPU-staged rehearsal on one GPU)")
    ap.add_argument("--batch", type=int, default=64, help="frames per launch of rt_render_frames (1..64; 1 = one launch per frame)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc); use the committed profile")
    ap.add_argument("--variant", type=int, default=None, help="kernel variant (tuning; default: library default)")
    ap.add_argument("--blocks", type=int, default=None, help="persistent grid size (tuning)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Launched bare: start the N ranks as children (one process per GPU) and leave with their exit code.  Nothing
        # in this process has touched the GPU yet (no torch import, no HIP call), and it never does.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        args.gpus = world
    live, live_error = None, None
    # (never under a profiler: its preloaded tool library has initialised the GPU in THIS process already, and such a process
    # must not start programs on this pool; never when torch has been imported by whoever runs this as a module)
    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and "WORLD_SIZE" not in os.environ and not args.no_live_traffic and args.variant is None and args.blocks is None \
            and not os.environ.get("RT2_OPTIONS") and args.batch > 1 and not profiled and "torch" not in sys.modules:
        try:
            live, live_error = live_traffic(args)   # (before this process touches the GPU)
        except Exception as e:  # noqa: BLE001 -- the bench line must not depend on the profiler
            live, live_error = None, f"{type(e).__name__}: {e}"

    import torch
    import ray_tracer_2_amd as rt
    from ray_tracer_2_amd import parallel

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    device = local_rank % n_dev   # > 1 rank per device only in the gloo rehearsal
    torch.cuda.set_device(device)
    dist = None
    backend_note = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl" and n_dev < world:
            # RCCL cannot put two ranks on one device: rehearse with host-staged gloo gathers (the render path is the same)
            args.backend = "gloo"
            backend_note = f"gloo rehearsal: {world} ranks on {n_dev} device(s); RCCL needs one device per rank"
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    W, H = args.width, args.height
    arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
    tracer = rt.RayTracer(device=device, max_width=W, max_height=H)
    tracer.load_scene(arrays)
    tracer.set_option("batch_frames", max(1, min(64, args.batch)))
    if args.variant is not None:
        tracer.set_option("kernel_variant", args.variant)
    if args.blocks is not None:
        tracer.set_option("persistent_blocks", args.blocks)

    local_texels = tracer.strip_texels(W, H, rank, world)
    pad_texels = tracer.strip_texels(W, H, 0, world)
    if world > 1:
        # render straight into a torch tensor that RCCL will send
        local = torch.zeros((pad_texels, 4), dtype=torch.float32, device="cuda")
        tracer.bind_image(local.data_ptr(), pad_texels)
        gathered = torch.empty((world, pad_texels, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
        frame = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
        assembler = None
        # everything is ordered on torch's current stream (the one RCCL syncs with):
        # render -> gather -> assemble, no host synchronisation inside a step
        side = torch.cuda.Stream()   # a real (non-null) stream; collectives issued under it sync with it
        torch.cuda.set_stream(side)
        stream_ptr = side.cuda_stream
        tracer.set_stream(stream_ptr)
        if rank == 0:
            assembler = rt.RayTracer(device=device, max_width=8, max_height=8)
            assembler.bind_image(frame.data_ptr(), H * W)
            assembler.set_stream(stream_ptr)

    def render(f0, n, batch=None):
        """Frames f0 .. f0 + n - 1, in batches of `batch` (default --batch) frames; N > 1: one gather + assemble per
        batch.  batch = 1: one launch (and one gather) per frame."""
        batch = max(1, args.batch) if batch is None else batch
        p = rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=f0)
        if world == 1:
            if batch == 1:
                for f in range(f0, f0 + n):
                    p.frames = f
                    tracer.render(p)
            else:
                tracer.render_frames(p, n)
            return
        done = 0
        while done < n:
            nb = min(batch, n - done)
            p.frames = f0 + done
            tracer.render_strips_frames(p, nb, rank, world)
            if args.backend == "nccl":
                # one gather over RCCL + de-interleave on the root's device (ray_tracer_2_amd/parallel.py)
                parallel.gather_frame(dist, local, W, H, rank, world, gathered=gathered,
                                      assemble_fn=lambda g, w, h, n: assembler.assemble_strips(g.data_ptr(), w, h, n))
            else:  # rehearsal: stage through the host
                host = local.cpu()
                parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    gathered.copy_(torch.stack(parts))
                    assembler.assemble_strips(gathered.data_ptr(), W, H, world)
            done += nb

    def fence():
        tracer.synchronize()
        if world > 1:
            if rank == 0:
                assembler.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    render(0, args.warmup)
    fence()
    # The timed region: EXACTLY `steps` frames between two fences -- and that window `repeats` times over (default 5, each
    # continuing the accumulation), because one window of 20 frames is 23 ms of wall time: one sample.  ms_per_step / value
    # are the MEDIAN window's (N > 1: of the windows' max-over-ranks times); min / max go into the line beside it.
    windows = []
    for w_i in range(max(1, args.repeats)):
        tracer.reset_timing()
        t0 = time.perf_counter()
        render(args.warmup + w_i * args.steps, args.steps)
        fence()
        dt = time.perf_counter() - t0
        st = tracer.stats()
        windows.append((dt, float(st.segments), float(st.segments_reused), st.kernel_ms / max(st.launches, 1), st.frames / max(st.launches, 1)))
    frames_done = args.warmup + max(1, args.repeats) * args.steps

    if world > 1 and rank == 0 and os.environ.get("RT2_BENCH_VERIFY"):
        # the stitched frame must equal the single-GPU frame (same frames sequence)
        single = rt.RayTracer(device=device, max_width=W, max_height=H)
        single.load_scene(arrays)
        for f in range(frames_done):
            single.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=f))
        ref = torch.from_numpy(single.read_image(W, H)).reshape(H * W, 4)
        same = torch.equal(ref.view(torch.int32), frame.cpu().view(torch.int32))
        print(f"[verify] stitched frame bit-identical to 1-GPU frame: {same}", file=sys.stderr, flush=True)
        assert same
    launch_info = tracer.last_launch()                 # dynamic LDS per workgroup, grid, which instantiation ran
    wt = torch.tensor(windows, dtype=torch.float64)          # [window, (elapsed, rays, reused, launch ms, frames per launch)]
    if world > 1:
        dev_t = "cuda" if args.backend == "nccl" else "cpu"
        wmax, wsum = wt.to(dev_t).clone(), wt.to(dev_t).clone()
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(wsum, op=dist.ReduceOp.SUM)
        wt = torch.stack([wmax[:, 0], wsum[:, 1], wsum[:, 2], wmax[:, 3], wmax[:, 4]], dim=1).cpu()
    order = sorted(range(wt.shape[0]), key=lambda i: float(wt[i, 0]))
    mid = order[(len(order) - 1) // 2]                  # the median window (the lower one of an even count)
    elapsed, rays_med, reused_med = float(wt[mid, 0]), float(wt[mid, 1]), float(wt[mid, 2])
    launch_ms = float(wt[mid, 3])                       # render (+ blend) launch, HIP events on the tracer's stream
    frames_per_launch = float(wt[mid, 4])
    window_ms = [float(x) / args.steps * 1e3 for x in wt[:, 0]]

    # The same K frames once more with ONE launch (N > 1: and one gather + assemble) per frame: what a host that
    # presents every frame gets.  Outside the timed region of `value`; same barrier + synchronize bracket.
    elapsed_pf, rays_pf_local, launch_ms_pf = float("nan"), float("nan"), float("nan")
    if not args.no_per_frame_leg:
        render(frames_done, 4, batch=1)       # (untimed: the pipeline's streams and scratch images are made here)
        fence()
        tracer.reset_timing()
        t1 = time.perf_counter()
        render(frames_done + 4, args.steps, batch=1)
        fence()
        elapsed_pf = time.perf_counter() - t1
        st_pf = tracer.stats()
        # (calls that continue the accumulation may have rendered frames beyond the window with their own -- option
        # frame_ahead; the counters count what was launched: the window's share is rays per rendered frame x its frames)
        rays_pf_local = float(st_pf.segments) * args.steps / max(st_pf.frames + st_pf.frames_speculative, 1)
        launch_ms_pf = st_pf.kernel_ms / max(st_pf.launches, 1)

    rays, reused = rays_med, reused_med
    ranks_seen, rank_devices = 1, [torch.cuda.get_device_name(device)]
    if world > 1:
        t = torch.tensor([elapsed_pf, rays_pf_local, launch_ms_pf], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed_pf, rays_pf, launch_ms_pf = float(tmax[0]), float(t[1]), float(tmax[2])
        # who took part: the communicator's size and every rank's device (so that a SCALE record can be checked for N ranks
        # on N devices)
        ranks_seen = dist.get_world_size()
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: cuda:{device} {torch.cuda.get_device_name(device)}")
        rank_devices = names
    else:
        rays_pf = rays_pf_local

    extras = {}
    if world == 1 and rank == 0 and not args.no_extras:
        # (outside the timed region) one launch per frame WITHOUT the pipeline (option pipeline = 0: every launch ramps up and
        # drains alone, in-place blend) -- the latency of a frame, and what round 2 called the un-overlapped frame
        tracer.set_option("pipeline", 0)
        tracer.set_option("frame_ahead", 0)   # (every frame a launch of its own)
        seq = []
        for rep in range(3):
            tracer.synchronize()
            tracer.reset_timing()
            t1 = time.perf_counter()
            for f in range(64):
                tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=1 + f))
            tracer.synchronize()
            seq.append((time.perf_counter() - t1) / 64 * 1e3)
        s1 = tracer.stats()
        tracer.set_option("pipeline", 1)   # (back to the default depth)
        tracer.set_option("frame_ahead", -1)
        extras["ms_per_frame_unoverlapped"] = statistics.median(seq)
        extras["kernel_ms_unoverlapped"] = s1.kernel_ms / max(s1.launches, 1)
        # The reference's present loop (src/core/app.rs:285-340): render one frame, consume it, render the next -- here
        # "consume" = wait for the frame (rt_synchronize; a host copy of the 33 MB frame over PCIe would add ~0.6 ms to
        # every variant alike).  Nothing overlaps in such a loop, so by default a frame that finds the stream idle takes
        # the plain in-place launch (option pipeline_when_idle = 0); with 1 it pays the pipeline's scratch image, blend
        # kernel and event hops for nothing.  By default (option frame_ahead = -1) a call that continues an accumulation
        # and finds the stream idle renders the next frames with its own (batches of 2, then 3 at this size): the calls
        # that follow only blend theirs -- the average over the 32 frames is what is reported, the image after every call
        # is the same.
        import numpy as np
        present = {}
        for name, opts in (("default", {}), ("frame_ahead_off", {"frame_ahead": 0}),
                           ("frame_ahead_off_pipeline_when_idle", {"frame_ahead": 0, "pipeline_when_idle": 1}),
                           ("frame_ahead_off_pipeline_off", {"frame_ahead": 0, "pipeline": 0})):
            for k, v in opts.items():
                tracer.set_option(k, v)
            ts = []
            for rep in range(3):
                tracer.synchronize()
                t1 = time.perf_counter()
                for f in range(32):
                    tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=1 + f))
                    tracer.synchronize()
                ts.append((time.perf_counter() - t1) / 32 * 1e3)
            present[name] = statistics.median(ts)
            tracer.set_option("pipeline", 1)
            tracer.set_option("pipeline_when_idle", 0)
            tracer.set_option("frame_ahead", -1)
        extras["ms_per_frame_render_then_wait"] = present
        # A host that SHOWS every frame (the reference blits its storage texture every redraw) has to bring the 33 MB frame to
        # the host: render, rt_read_image, render, ... against rt_snapshot_image (device copy in stream order) +
        # rt_read_snapshot (the host copy on a stream of its own, while the next frame renders).  PCIe-inclusive: never `value`.
        shown = {}
        host_frame = np.empty((H, W, 4), np.float32)
        host_frame[:] = 0
        for name in ("render_then_read_image", "snapshot_read_under_next_frame"):
            ts = []
            for rep in range(3):
                tracer.synchronize()
                t1 = time.perf_counter()
                n_shown = 48
                tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=1))
                if name.startswith("snapshot"):
                    tracer.snapshot_image(W, H)
                for f in range(1, n_shown):
                    if name.startswith("snapshot"):
                        tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=1 + f))
                        tracer._check(tracer._L.rt_read_snapshot(tracer._h, host_frame.ctypes.data, host_frame.nbytes))
                        tracer.snapshot_image(W, H)
                    else:
                        tracer._check(tracer._L.rt_read_image(tracer._h, host_frame.ctypes.data, host_frame.nbytes))
                        tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=1 + f))
                if name.startswith("snapshot"):
                    tracer._check(tracer._L.rt_read_snapshot(tracer._h, host_frame.ctypes.data, host_frame.nbytes))
                else:
                    tracer._check(tracer._L.rt_read_image(tracer._h, host_frame.ctypes.data, host_frame.nbytes))
                ts.append((time.perf_counter() - t1) / n_shown * 1e3)
            shown[name] = statistics.median(ts)
        extras["ms_per_shown_frame_pcie_inclusive"] = shown
        # first frame after a camera change at full size: natural tile order, primary-ray table rebuilt
        cam_t = type(arrays.uniform.camera)
        cam0 = cam_t.from_buffer_copy(bytes(arrays.uniform.camera))
        firsts = []
        for rep in range(5):
            cam = cam_t.from_buffer_copy(bytes(cam0))
            cam.cam_to_world[3][0] = cam0.cam_to_world[3][0] + 1e-3 * (rep + 1)
            tracer.set_camera(cam)
            tracer.synchronize()
            t1 = time.perf_counter()
            tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=0))
            tracer.synchronize()
            firsts.append((time.perf_counter() - t1) * 1e3)
        tracer.set_camera(cam0)
        extras["ms_first_frame_after_camera_change"] = statistics.median(firsts)
        # a camera that moves EVERY frame (frames = 0 each call), the host running ahead: a frame whose camera is not the
        # previous frame's renders without the primary table, so nothing is rebuilt behind a barrier and the frames overlap
        moving = {}
        for name, per_slot in (("default", 1),):
            tracer.set_option("primary_per_slot", per_slot)
            ts = []
            for rep in range(3):
                tracer.synchronize()
                t1 = time.perf_counter()
                for f in range(48):
                    cam = cam_t.from_buffer_copy(bytes(cam0))
                    cam.cam_to_world[3][0] = cam0.cam_to_world[3][0] + 2e-3 * (f + 1 + 48 * rep)
                    tracer.set_camera(cam)
                    tracer.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=0))
                tracer.synchronize()
                ts.append((time.perf_counter() - t1) / 48 * 1e3)
            moving[name] = statistics.median(ts)
        tracer.set_option("primary_per_slot", 1)
        tracer.set_camera(cam0)
        extras["ms_per_frame_moving_camera"] = moving
        # independent single frames (every frame a new accumulation, as while the camera moves) pipelined across two
        # handles on the device: each has its own stream and image, so frame k + 1's launch takes the CUs that frame
        # k's draining waves free (the per-frame latency stays ms_per_frame_unoverlapped)
        second = rt.RayTracer(device=device, max_width=W, max_height=H)
        second.load_scene(arrays)
        pair = [tracer, second]
        for t in pair:
            t.render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=0))
            t.synchronize()
        two = []
        for rep in range(3):
            t1 = time.perf_counter()
            for i in range(64):
                pair[i & 1].render(rt.make_params(W, H, BOUNCES, SPP, skybox=1, frames=0))
            for t in pair:
                t.synchronize()
            two.append((time.perf_counter() - t1) / 64 * 1e3)
        second.close()
        extras["ms_per_independent_frame_two_handles"] = statistics.median(two)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays / elapsed / 1e6
        # compulsory bytes: per frame the frame's texels are read (frames >= 1) and
        # written once, 32 B per pixel, plus the scene once per launch.
        scene_bytes = arrays.meshes.nbytes + arrays.nodes.nbytes + arrays.triangles.nbytes + arrays.spheres.nbytes
        texels = local_texels if world > 1 else W * H
        algo_bytes = texels * 16 * 2 * frames_per_launch + scene_bytes
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        # HBM traffic and VALU counters per launch come from the committed rocprofv3 PMC passes of this same
        # command (profiles/run_profile.sh; PMC counters cannot be read from inside the process).  They
        # are reported only when they were measured on the build that is running (source hash, batch).
        traffic, traffic_src, valu, stale = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if live is not None and frames_per_launch > 1:
            traffic = live["bytes_per_launch"] * frames_per_launch / live["frames_per_launch"]
            traffic_src = live["source"] + (f"; per frame x {frames_per_launch:g} frames per launch"
                                            if frames_per_launch != live["frames_per_launch"] else "")
        if world == 1 and (W, H) == (WIDTH, HEIGHT) and os.path.exists(tpath):
            from ray_tracer_2_amd.build import source_hash
            tj = json.load(open(tpath))
            if tj.get("source_hash") != source_hash() or (tj.get("frames_per_launch", 1) > 1) != (frames_per_launch > 1) \
                    or args.variant is not None or args.blocks is not None or os.environ.get("RT2_OPTIONS"):
                stale = f"stale profile: {tj.get('source')} was taken on another build or launch mode"
            else:
                # counters are kept per frame (they do not depend on how many frames share a launch) and
                # scaled to this run's frames per launch
                scale = frames_per_launch / tj.get("frames_per_launch", 1)
                if traffic is None:   # (not measured in this run: the committed profile of the same sources)
                    traffic = tj["bytes_per_launch"] * scale
                    traffic_src = tj["source"] + (f", per frame x {frames_per_launch:g} frames per launch" if scale != 1 else "")
                if "valu_instructions_per_launch" in tj and not (live and live.get("valu")):
                    # The bound that matters: VALU issue.  Peak = the guide's: a SIMD-32
                    # issues one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md; that is what
                    # 157.3 TFLOP/s FP32 means).  issue_frac = wave-instructions issued / that peak;
                    # useful_lane_frac = issue_frac x active lanes per instruction / 64.
                    simds = torch.cuda.get_device_properties(device).multi_processor_count * 4
                    clock = tj.get("shader_clock_ghz", 2.4)
                    rate = tj["valu_instructions_per_launch"] * scale / (launch_ms * 1e-3) / 1e9
                    peak = simds * clock / 2.0
                    lanes = tj.get("valu_lane_utilisation")
                    valu = {"bound": "valu_issue", "achieved": rate, "peak": peak, "unit": "G wave-instructions/s",
                            "peak_definition": "MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD-32 x "
                                               f"{simds} SIMDs x {clock:.2f} GHz (measured shader clock)",
                            "frac": rate / peak, "issue_frac": rate / peak, "lane_utilisation": lanes,
                            "useful_lane_frac": (rate / peak * lanes) if lanes else None,
                            "measured_stream_peak": tj.get("measured_valu_stream_peak"),
                            "source": tj["source"].replace("FETCH_SIZE / WRITE_SIZE", "SQ_INSTS_VALU / SQ_THREAD_CYCLES_VALU / GRBM_GUI_ACTIVE")}
        if live and live.get("valu") and frames_per_launch > 1:
            lv = live["valu"]
            simds = torch.cuda.get_device_properties(device).multi_processor_count * 4
            clock = lv.get("shader_clock_ghz") or 2.4
            rate = lv["valu_instructions_per_launch"] * frames_per_launch / live["frames_per_launch"] / (launch_ms * 1e-3) / 1e9
            peak = simds * clock / 2.0
            lanes = lv["valu_lane_utilisation"]
            valu = {"bound": "valu_issue", "achieved": rate, "peak": peak, "unit": "G wave-instructions/s",
                    "peak_definition": "MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD-32 x "
                                       f"{simds} SIMDs x {clock:.2f} GHz (shader clock measured in this run: GRBM_GUI_ACTIVE / 8 / kernel time)",
                    "frac": rate / peak, "issue_frac": rate / peak, "lane_utilisation": lanes, "useful_lane_frac": rate / peak * lanes,
                    "valu_instructions_per_frame": lv["valu_instructions_per_launch"] / live["frames_per_launch"],
                    "measured_in_this_run": True,
                    "source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE, one pass over "
                              "the same child run as roofline.traffic; instructions per launch / this run's launch time"}
        # which kernel the library runs for this shape (rt_api.hip render_impl): batches always the persistent one
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        tiles = ((W + 7) // 8) * (texels // W // 8 if world > 1 else (H + 7) // 8)
        if live is not None and live.get("kernel_name"):
            # the name rocprofv3 reported for the render kernel of this run's own profiled child pass
            kernel, kernel_src = live["kernel_name"], "rocprofv3 (the live --pmc pass of this run)"
        else:
            # (no profiler pass in this run -- N > 1, --no-live-traffic, tuning options: named from rt_last_launch; template
            # arguments <LDS, STATS, TLAS, PARK, SIMPLE, HYB> of rt_kernel.hip)
            spec = "true" if launch_info["specialised"] else "false"
            lds = "true" if launch_info["scene_in_lds"] else "false"
            kernel = f"void rtd::rt_render_persistent_kernel<{lds}, false, false, false, {spec}, false>(rtd::RenderArgs)"
            if launch_info["one_wave_per_tile"]:
                kernel = f"void rtd::rt_render_tiles_kernel<{lds}, false, false, {spec}>(rtd::RenderArgs)"
            kernel_src = "assembled from rt_last_launch (no profiler pass in this run)"
        out = {
            "metric": "Mrays/s", "value": mrays, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            # the window of `steps` frames was timed `repeats` times; ms_per_step and value are the median window's
            "timing": {"repeats": len(window_ms), "ms_per_step_median": ms_per_step, "ms_per_step_min": min(window_ms),
                       "ms_per_step_max": max(window_ms), "ms_per_step_windows": window_ms},
            "ranks_seen": ranks_seen, "rank_devices": rank_devices,
            # the figures to track across rounds: ms per frame (= ms_per_step), camera paths per second and the rays that
            # were actually traversed (`value` also counts the primary segments served from the per-pixel memo)
            "Mpaths/s": W * H * SPP * args.steps / elapsed / 1e6,
            "Mrays_traversed/s": (rays - reused) / elapsed / 1e6,
            # the two launch modes under explicit names (value == value_batched when --batch > 1)
            "value_batched": mrays if frames_per_launch > 1 else None,
            "value_per_frame_launch": None if args.no_per_frame_leg else rays_pf / elapsed_pf / 1e6,
            "ms_per_frame_batched": ms_per_step if frames_per_launch > 1 else None,
            "ms_per_frame_per_launch": None if args.no_per_frame_leg else elapsed_pf / args.steps * 1e3,
            "value_definition": f"value = rays of {args.steps} frames / wall time with {frames_per_launch:g} frames per launch "
                                "(intermediate frames of a batch are not observable); value_per_frame_launch = the same frames "
                                "with one launch" + (" and one gather" if world > 1 else "") + " per frame, every frame observable, timed "
                                "right after (N = 1: consecutive launches pipelined across internal streams, four frames in flight, option pipeline; "
                                "ms_per_frame_unoverlapped = the same with the pipeline off = a frame's latency; N >= 2: a call that continues "
                                "the accumulation renders the next frames of the rank's small share with its own and the following calls blend "
                                "theirs, option frame_ahead -- every call still leaves its frame in the image, and the gather is per frame)",
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "CornellBox-Original.obj/.mtl of the reference through the loader + BVH builder "
                    "(committed as tests/golden/cornell_scene.npz); no random inputs: the seed is Params.frames",
            "config": {"workload": f"CornellBox-Original {W}x{H}, {SPP} spp, {BOUNCES} bounces "
                                   "(BASELINE configs[1]); 8 meshes / 32 triangles / 32 BVH nodes",
                       "frames": f"{args.warmup}..{frames_done - 1} (progressive accumulation; {max(1, args.repeats)} timed windows of {args.steps})",
                       "frames_overlapped": f"{frames_per_launch:g} frames per launch (rt_render_frames: (frame, tile) work items "
                                            "+ ordered blend kernel; bit-identical to one launch per frame)"
                                            if frames_per_launch > 1 else "no: one launch per frame",
                       "parallelism": "1 GPU" if world == 1 else f"8-row strips round-robin over {world} GPUs + 1 "
                                                                 f"{'RCCL' if args.backend == 'nccl' else 'gloo (host-staged)'} gather per "
                                                                 f"batch of {frames_per_launch:g} frames (value); one gather per frame in "
                                                                 "value_per_frame_launch",
                       "backend": backend_note or args.backend,
                       "rays_per_frame": rays / args.steps,
                       "reused_fraction": reused / rays if rays else None,
                       "rays_traversed_per_frame": (rays - reused) / args.steps,
                       "Mrays_traversed/s": (rays - reused) / elapsed / 1e6,
                       "Mpaths/s": W * H * SPP * args.steps / elapsed / 1e6},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src or stale,
                         "traffic_measured_in_this_run": live is not None and traffic is not None, "traffic_live_error": live_error,
                         "kernel": kernel, "kernel_name_source": kernel_src, "kernel_ms": launch_ms, "frames_per_launch": frames_per_launch,
                         "launch": launch_info,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "path is FP32-VALU-bound (SURVEY 8d, DESIGN.md): compulsory HBM bytes are 32 B per pixel per frame; "
                                 "kernel_ms is the launch (render + ordered blend of the batch) from HIP events on the "
                                 "tracer's stream"},
        }
        if valu is not None:
            out["roofline_valu"] = valu
        if live is not None and live.get("valu_error"):
            out["roofline_valu_live_error"] = live["valu_error"]
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rt, arrays)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
