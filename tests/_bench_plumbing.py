"""Helper of tests/test_gpu_frames.py::test_bench_rank_plumbing_on_torch_memory_and_stream (run as a script)."""
import os
import sys

import numpy as np
import torch   # first, as in bench.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

batch = int(sys.argv[1])
use_rccl = len(sys.argv) > 2 and sys.argv[2] == "rccl"   # the gather through torch.distributed's nccl backend, one-rank group
w, h, spp, nb, n, world = 200, 100, 2, 3, 8, (1 if use_rccl else 3)
if use_rccl:
    import socket
    import torch.distributed as dist
    from ray_tracer_2_amd import parallel
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
cornell = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
single = rt.RayTracer(0, w, h)
single.load_scene(cornell)
for f in range(n):
    single.render(rt.make_params(w, h, nb, spp, skybox=1, frames=f))
want = single.read_image(w, h)

side = torch.cuda.Stream()
with torch.cuda.stream(side):
    pad = single.strip_texels(w, h, 0, world)
    frame = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    assembler = rt.RayTracer(0, 8, 8)
    assembler.bind_image(frame.data_ptr(), h * w)
    assembler.set_stream(side.cuda_stream)
    ranks, locals_ = [], []
    for r in range(world):
        t = rt.RayTracer(0, w, h)
        t.load_scene(cornell)
        t.set_option("batch_frames", max(1, batch))
        loc = torch.zeros((pad, 4), dtype=torch.float32, device="cuda")
        t.bind_image(loc.data_ptr(), pad)
        t.set_stream(side.cuda_stream)
        ranks.append(t)
        locals_.append(loc)
    done = 0
    while done < n:
        p = rt.make_params(w, h, nb, spp, skybox=1, frames=done)
        for r, t in enumerate(ranks):
            t.render_strips_frames(p, batch, r, world)
        if use_rccl:   # bench.py's call, verbatim
            parallel.gather_frame(dist, locals_[0], w, h, 0, world,
                                  assemble_fn=lambda g, ww, hh, nn: assembler.assemble_strips(g.data_ptr(), ww, hh, nn))
        else:
            gathered = torch.stack(locals_)          # (ordered behind the renders: same stream)
            assembler.assemble_strips(gathered.data_ptr(), w, h, world)
        done += batch
    side.synchronize()
    got = frame.cpu().numpy().reshape(h, w, 4)
assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got.view(np.uint32) != want.view(np.uint32)).sum())
for t in ranks + [assembler, single]:
    t.close()
if use_rccl:
    dist.destroy_process_group()
print("plumbing ok")
