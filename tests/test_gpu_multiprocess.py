"""GPU, two processes: the N > 1 paths end to end on a one-GPU box -- two ranks (gloo) sharing device 0, each with
its own engine context and walkers, exchanging through WalkerComms.  (The 8-GPU RCCL run is the driver's; this is
the same code with the backend swapped.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(args, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]                     # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_two_rank_farm_agrees_on_the_synchronised_tables():
    res = _launch(["-m", "mc_water_ls_mw_amd.farm", "--walkers", "16", "--cycles", "30", "--sync", "10",
                   "--backend", "gloo", "--share-device"])
    assert res["world"] == 2 and res["ranks_agree"] is True
    assert res["histogram_total"] > 0 and res["weight_max"] > 0
    assert 0.02 < res["acceptance"] < 0.9
    assert res["moves_per_s_all_ranks"] > res["moves_per_s"]


def test_two_rank_bench_line():
    res = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--walkers", "32",
                   "--moves", "256", "--backend", "gloo", "--share-device", "--no-cpu-baseline"])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["scaling"] == "weak" and res["unit"] == "interactions/s"
    assert res["value"] > res["per_gpu"] > 0
    assert res["roofline"]["kernel"] in ("k_move_energy", "k_model_energy") and res["cpu_baseline"] is None
    assert "gloo all-reduce" in res["config"]["exchange"]


def test_two_rank_dd_farm_stitches_its_windows():
    """parallel_strategy = 'dd' over two processes: one walker = one window each (leshift puts the start inside the
    overlap of both), no exchange during the run, the windows joined at the end by an all-gather over the ranks."""
    res = _launch(["-m", "mc_water_ls_mw_amd.farm", "--walkers", "1", "--cycles", "40", "--strategy", "dd", "--leshift",
                   "--eq-cycles", "3", "--flat-chk", "8", "--wl-schedule", "1", "--wl-minhist", "-1", "--no-thermalise",
                   "--backend", "gloo", "--share-device"])
    assert res["world"] == 2 and res["ranks_agree"] is True                 # both ranks hold the same joined weights
    lo, hi = res["joined_weight_range"]
    assert hi > lo and res["in_window"] == [True]
    assert [e["action"] for e in res["flatness_events"]][:2] == ["first reset", "halved"]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher environment: the parent -- before it imports torch or touches the GPU --
    starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's ONE json line and exits
    with the children's status (the rank / size bootstrap of comms_mpi.f90:26-71).  It must never come back as a 1-GPU run."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--walkers", "32", "--moves", "256", "--backend", "gloo", "--share-device", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-1500:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["value"] > res["per_gpu"] > 0
    assert "2 x 32 independent walkers" in res["config"]["parallelism"]


def test_every_rank_stops_when_one_rank_has_walkers_outside_their_windows():
    """Four windows over two ranks, the pair starting at mu ~ -0.1: the outer windows' walkers are not inside their windows after
    three equilibration cycles, and the run stops with the reference's message (mc_moves.F90:187-201) -- on EVERY rank, promptly:
    a rank that raised alone left the other one waiting in the windows' all-gather."""
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "mc_water_ls_mw_amd.farm", "--walkers", "2", "--cycles", "8", "--strategy", "dd",
           "--leshift", "--eq-cycles", "3", "--no-thermalise", "--backend", "gloo", "--share-device"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    # (here both ranks hold an outer window: each reports its own walkers, after the two have agreed to stop)
    assert out.stderr.count("MwError: Error : Not all walkers have reached their designated window after 3 MC cycles") == 2
    assert "join_eta" not in out.stderr                            # nobody went on to the windows' all-gather


def test_two_rank_restart_continues_the_synchronised_run(tmp_path):
    """Two ranks x four walkers, tables synchronised every 10 cycles, checkpoints at cycle 20: the restarted second half ends
    where the uninterrupted 40 cycles end (the checkpointed table is every rank's baseline of the delta exchange -- left at zero,
    the first synchronisation after a restart returned 8 x the weights)."""
    common = ["-m", "mc_water_ls_mw_amd.farm", "--walkers", "4", "--sync", "10", "--backend", "gloo", "--share-device"]
    full = _launch(common + ["--cycles", "40"])
    _launch(common + ["--cycles", "20", "--chkpt", "20", "--outdir", str(tmp_path)])
    rest = _launch(common + ["--cycles", "20", "--restart", "--outdir", str(tmp_path)])
    assert rest["ranks_agree"] is True and full["ranks_agree"] is True
    assert rest["weight_max"] == pytest.approx(full["weight_max"], rel=1e-9) and full["weight_max"] > 0
    assert rest["histogram_total"] == pytest.approx(full["histogram_total"], rel=1e-12)


def test_every_rank_stops_when_one_ranks_checkpoints_are_a_dump_behind(tmp_path):
    """Two ranks x two walkers, checkpoints at cycles 10 and 20 (files .1 and .2 in turn); rank 1's newer files are removed, so it
    would restart from cycle 10 and rank 0 from cycle 20.  The ranks agree before either stops (farm.run: every rank takes part in
    the broadcast of rank 0's cycle and in one get_max of "something is wrong here"): BOTH end with the message, at once -- a rank
    that raised alone left the other waiting in the collective until the launcher's watchdog gave up."""
    import glob
    import time
    common = ["-m", "mc_water_ls_mw_amd.farm", "--walkers", "2", "--sync", "10", "--backend", "gloo", "--share-device"]
    _launch(common + ["--cycles", "20", "--chkpt", "10", "--outdir", str(tmp_path)])
    newer = sorted(glob.glob(os.path.join(str(tmp_path), "checkpoint00[23].dat.2")))       # walkers 2, 3 = rank 1
    assert len(newer) == 2 and len(glob.glob(os.path.join(str(tmp_path), "checkpoint*.dat.*"))) == 8
    for f in newer:
        os.remove(f)
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + common + ["--cycles", "10", "--restart", "--outdir", str(tmp_path)]
    t0 = time.monotonic()
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and time.monotonic() - t0 < 120
    assert "rank 1 restarts from cycle 10, rank 0 from 20: checkpoint files out of step" in out.stderr
    assert "rank 0: another rank's checkpoint files are out of step" in out.stderr
