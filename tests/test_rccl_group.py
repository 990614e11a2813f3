"""The grouped ncclSend / ncclRecv gather of rt_render_multi (csrc/rt_rccl.h) against a stub RCCL that fails on
request (tests/cpp/fake_rccl.c) -- on the CPU: the gather's group logic makes no HIP call.

What must hold whatever fails inside the group: the group is closed before the call returns (an open group would
swallow every later RCCL call of the process), the communicators of the failed gather are aborted, the call returns
RT_ERR_DEVICE with the failing call named, and the next gather works.  The reference has no multi-GPU path
(SURVEY.md section 2, rows 15-16); the real library runs in tests/test_gpu_frames.py.

Each case runs in a child process: the stub's failure knobs are environment variables and its counters are per process."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import ctypes, json, os, sys
sys.path.insert(0, {root!r})
from ray_tracer_2_amd.build import build_fake_rccl
import ray_tracer_2_amd.lib as lib
so = build_fake_rccl()
L = lib.load_test()   # (rt_test_rccl_gather: include/rt_test_abi.h, the test library)
fake = ctypes.CDLL(so)   # the same mapping the product dlopens: shared counters
def state():
    out = (ctypes.c_int * 8)()
    fake.fake_rccl_state(out)
    return dict(zip("depth sends recvs live aborts destroys starts ends".split(), out))
res = []
for env in {envs!r}:
    for k in [k for k in os.environ if k.startswith("FAKE_RCCL_")]:
        del os.environ[k]
    os.environ.update(env)
    rc = L.rt_test_rccl_gather(so.encode(), {ranks})
    res.append(dict(rc=rc, err=L.rt_last_error(None).decode() if rc else "", **state()))
print(json.dumps(res))
"""


def run(envs, ranks=4):
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, envs=envs, ranks=ranks)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_gather_without_failures():
    (a,) = run([{}])
    assert a["rc"] == 0 and a["depth"] == 0 and a["sends"] == 4 and a["recvs"] == 4
    assert a["live"] == 0 and a["destroys"] == 4 and a["aborts"] == 0 and a["starts"] == a["ends"] == 1


@pytest.mark.parametrize("knob,k", [("FAKE_RCCL_FAIL_SEND", 1), ("FAKE_RCCL_FAIL_SEND", 3), ("FAKE_RCCL_FAIL_RECV", 2),
                                    ("FAKE_RCCL_FAIL_RECV", 4)])
def test_failure_inside_the_group_closes_it_and_aborts_the_communicators(knob, k):
    bad, good = run([{knob: str(k)}, {}])
    assert bad["rc"] == -3 and ("ncclSend" if "SEND" in knob else "ncclRecv") in bad["err"]   # RT_ERR_DEVICE
    assert bad["depth"] == 0 and bad["starts"] == bad["ends"] == 1     # the group was closed
    assert bad["live"] == 0 and bad["aborts"] == 4 and bad["destroys"] == 0
    # nothing was issued behind the failing call
    assert (bad["sends"], bad["recvs"]) == ((k, k - 1) if "SEND" in knob else (k, k))
    # ... and the process can gather again (fresh communicators, a fresh group)
    assert good["rc"] == 0 and good["depth"] == 0 and good["live"] == 0 and good["starts"] == good["ends"] == 2


def test_failing_group_end_and_group_start():
    end, start, good = run([{"FAKE_RCCL_FAIL_END": "1"}, {"FAKE_RCCL_FAIL_START": "1"}, {}])
    assert end["rc"] == -3 and "ncclGroupEnd" in end["err"] and end["depth"] == 0 and end["aborts"] == 4
    assert start["rc"] == -3 and "ncclGroupStart" in start["err"] and start["depth"] == 0 and start["sends"] == end["sends"]
    assert good["rc"] == 0 and good["depth"] == 0 and good["live"] == 0


def test_missing_library_is_an_error_not_a_crash():
    import ray_tracer_2_amd.lib as lib
    L = lib.load_test()
    assert L.rt_test_rccl_gather(b"/nonexistent/librccl.so", 2) == -6   # RT_ERR_IO
    assert b"cannot load librccl" in L.rt_last_error(None)
