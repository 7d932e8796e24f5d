"""GPU: the exchange step on the `nccl` backend (= RCCL on ROCm).  A one-GPU box admits one rank per device, so this is
a ONE-rank RCCL communicator: every collective WalkerComms issues (the packed delta all-reduce, the dd joins' all-gather
and broadcast, the max all-reduce, the flag broadcast, the barrier) really goes through RCCL on device buffers, and the
results are held to the numpy arithmetic of comms_mpi.f90:244-297,381-459.  (The 8-rank xGMI run is the driver's.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["MW_ROOT"])
from mc_water_ls_mw_amd.comms import WalkerComms

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
nb = 101
c = WalkerComms(nb, samplerun=True, device=torch.device("cuda", 0))
assert c._buf.is_cuda and c._stage.is_pinned()
rng = np.random.default_rng(5)
w, h, u = rng.random(nb), rng.random(nb) * 10, rng.random(nb)
w0, h0, u0 = w.copy(), h.copy(), u.copy()
c.sync(w, h, u)                                   # first call: last = 0, one rank: values unchanged
out["sync_first"] = float(max(np.abs(w - w0).max(), np.abs(h - h0).max(), np.abs(u - u0).max()))
w += 0.25; h[3] += 2.0; u[7] += 0.5               # increments since the last synchronisation
w1, h1, u1 = w.copy(), h.copy(), u.copy()
c.sync(w, h, u)
out["sync_second"] = float(max(np.abs(w - w1).max(), np.abs(h - h1).max(), np.abs(u - u1).max()))
out["last_sync_ok"] = bool(np.array_equal(c.eta_last_sync, w) and np.array_equal(c.hist_last_sync, h) and np.array_equal(c.uhist_last_sync, u))
e = rng.random(nb)
c2 = WalkerComms(nb, samplerun=False, device=torch.device("cuda", 0))
e_in = e.copy(); c2.allreduce_eta(e)
out["allreduce_eta"] = float(np.abs(e - e_in).max())
j = c.join_eta(w.copy(), 2)                       # one window: the table minus its middle bin
out["join_eta"] = float(np.abs(j - (w - w[nb // 2])).max())
ju = c.join_uhist(u.copy(), 2)
out["join_uhist"] = float(np.abs(ju - u).max())
out["get_max"] = c.get_max(3.5)
out["bcast_flag"] = [c.bcast_flag(True), c.bcast_flag(False)]
c.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print(json.dumps(out))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_exchange_step_on_a_one_rank_rccl_communicator():
    env = dict(os.environ, MW_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["sync_first"] == 0.0 and out["sync_second"] == 0.0 and out["last_sync_ok"]
    assert out["allreduce_eta"] == 0.0 and out["join_eta"] == 0.0 and out["join_uhist"] == 0.0
    assert out["get_max"] == 3.5 and out["bcast_flag"] == [True, False]


def test_bench_line_at_one_gpu_issues_the_rccl_all_reduce():
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--walkers", "32",
                          "--moves", "256", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and "nccl all-reduce" in line["config"]["exchange"] and "one-rank" in line["config"]["exchange"]
    assert line["value"] > 0
