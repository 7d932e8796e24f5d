"""CPU, world_size 2 over gloo: the multi-walker weight/histogram exchange
(comms_mpi.f90:244-277, 461-530) with the reference's delta-since-last-sync semantics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NBINS = 101


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference_semantics(arrs, lasts):
    """Plain restatement of comms_mpi.f90:256-270 for a list of per-rank arrays."""
    total = sum(a - l for a, l in zip(arrs, lasts))
    return [total + l for l in lasts]


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mc_water_ls_mw_amd.comms import WalkerComms
    rng = np.random.default_rng(100 + rank)
    c_sep = WalkerComms(NBINS)
    c_fused = WalkerComms(NBINS)
    # rank 0 alone holds the weights read from eta_weights.dat: first sync acts as a broadcast
    w0 = np.linspace(0.0, 5.0, NBINS) if rank == 0 else np.zeros(NBINS)
    state = {k: dict(w=w0.copy(), h=np.zeros(NBINS), u=np.zeros(NBINS)) for k in ("sep", "fused")}
    c_sep.allreduce_eta(state["sep"]["w"])
    c_fused.sync(state["fused"]["w"], state["fused"]["h"], state["fused"]["u"])
    out = {"bcast": state["sep"]["w"].copy()}
    for it in range(4):
        dw, dh, du = rng.random(NBINS) * 0.05, rng.integers(0, 50, NBINS).astype(float), rng.random(NBINS)
        for k in state:
            state[k]["w"] += dw
            state[k]["h"] += dh
            state[k]["u"] += du
        c_sep.allreduce_eta(state["sep"]["w"])
        c_sep.allreduce_hist(state["sep"]["h"])
        c_sep.allreduce_uhist(state["sep"]["u"])
        c_fused.sync(state["fused"]["w"], state["fused"]["h"], state["fused"]["u"])
        if it == 1:   # histogram reset re-bases last_sync (mc_moves.F90:1977)
            for k, c in (("sep", c_sep), ("fused", c_fused)):
                state[k]["h"][:] = 0.0
                c.set_histogram(state[k]["h"])
        out[f"dw{it}"], out[f"dh{it}"], out[f"du{it}"] = dw, dh, du
    for k in state:
        for name, a in state[k].items():
            out[f"{k}_{name}"] = a
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


def test_delta_allreduce_two_walkers(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(world)]
    # broadcast of rank-0 weights
    assert np.array_equal(r[0]["bcast"], np.linspace(0.0, 5.0, NBINS))
    assert np.array_equal(r[1]["bcast"], r[0]["bcast"])
    # all ranks agree, fused == three separate calls, and both equal the reference semantics
    for name in ("w", "h", "u"):
        assert np.array_equal(r[0][f"sep_{name}"], r[1][f"sep_{name}"])
        assert np.array_equal(r[0][f"sep_{name}"], r[0][f"fused_{name}"])
        assert np.array_equal(r[1][f"sep_{name}"], r[1][f"fused_{name}"])
    w = np.linspace(0.0, 5.0, NBINS)
    h = np.zeros(NBINS)
    u = np.zeros(NBINS)
    for it in range(4):
        w = w + sum(r[k][f"dw{it}"] for k in range(world))
        h = h + sum(r[k][f"dh{it}"] for k in range(world))
        u = u + sum(r[k][f"du{it}"] for k in range(world))
        if it == 1:
            h = np.zeros(NBINS)
    assert np.allclose(r[0]["sep_w"], w, rtol=1e-14)
    assert np.allclose(r[0]["sep_h"], h, rtol=1e-14)
    assert np.allclose(r[0]["sep_u"], u, rtol=1e-14)


def test_single_process_is_identity():
    from mc_water_ls_mw_amd.comms import WalkerComms
    c = WalkerComms(NBINS)
    w = np.arange(NBINS, dtype=float)
    c.allreduce_eta(w)
    assert np.array_equal(w, np.arange(NBINS)) and np.array_equal(c.eta_last_sync, w)
    w += 1.0
    h = np.ones(NBINS)
    c.sync(w, h)
    assert np.array_equal(w, np.arange(NBINS) + 1.0) and np.array_equal(h, np.ones(NBINS))


def _dd_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mc_water_ls_mw_amd.comms import WalkerComms
    c = WalkerComms(NBINS)
    rng = np.random.default_rng(7 + rank)
    w = np.cumsum(rng.random(NBINS)) + 10.0 * rank          # each window has its own arbitrary offset
    u = np.exp(rng.normal(0, 1, NBINS)) * (3.0 ** rank)
    np.savez(os.path.join(tmp, f"dd{rank}.npz"), w=w, u=u, jw=c.join_eta(w, 2), ju=c.join_uhist(u, 2),
             mx=c.get_max(float(rank) * 1.5 - 1.0))
    dist.destroy_process_group()


def test_window_join_three_ranks(tmp_path):
    """'dd' strategy: comms_join_eta / comms_join_uhist / comms_get_max (comms_mpi.f90:279-459) over 3 ranks."""
    world = 3
    mp.spawn(_dd_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f"dd{k}.npz")) for k in range(world)]
    bpw, ov = NBINS // world, 2
    # plain restatement of the reference's serial loop
    jw, ju = r[0]["w"].copy(), r[0]["u"].copy()
    for k in range(1, world):
        end = k * bpw
        a, b = end - ov - 1, end + ov
        jw[end:] = r[k]["w"][end:] + (jw[a:b].mean() - r[k]["w"][a:b].mean())
        ju[end:] = r[k]["u"][end:] * np.exp(np.log(ju[a:b]).mean() - np.log(r[k]["u"][a:b]).mean())
    jw -= jw[NBINS // 2]
    for k in range(world):
        assert np.allclose(r[k]["jw"], jw, rtol=1e-13, atol=1e-13)      # every rank holds the joined function
        assert np.allclose(r[k]["ju"], ju, rtol=1e-12)
        assert float(r[k]["mx"]) == 2.0
    assert jw[NBINS // 2] == 0.0
    # the seams are continuous in the mean
    for k in range(1, world):
        end = k * bpw
        assert abs(np.diff(jw)[end - 1]) < 5.0


def _dd_rows_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mc_water_ls_mw_amd.comms import WalkerComms
    c = WalkerComms(NBINS)
    rows = np.load(os.path.join(tmp, "rows.npz"))
    np.savez(os.path.join(tmp, f"ddrows{rank}.npz"), jw=c.join_eta(rows["w"][2 * rank:2 * rank + 2], 2),
             ju=c.join_uhist(rows["u"][2 * rank:2 * rank + 2], 2))
    dist.destroy_process_group()


def test_window_join_of_a_farm_two_processes_two_walkers_each(tmp_path):
    """A farm hands the join one row per walker: walker k of process p is window 2p + k, so two processes of two
    walkers stitch exactly what four single-window ranks would (done here without a process group)."""
    from mc_water_ls_mw_amd.comms import WalkerComms
    rng = np.random.default_rng(11)
    w = np.cumsum(rng.random((4, NBINS)), axis=1) + 7.0 * np.arange(4)[:, None]
    u = np.exp(rng.normal(0, 1, (4, NBINS)))
    np.savez(tmp_path / "rows.npz", w=w, u=u)
    mp.spawn(_dd_rows_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    single = WalkerComms(NBINS)
    jw, ju = single.join_eta(w, 2), single.join_uhist(u, 2)
    bpw = NBINS // 4
    ref = w[0].copy()
    for k in range(1, 4):
        end = k * bpw
        ref[end:] = w[k][end:] + (ref[end - 3:end + 2].mean() - w[k][end - 3:end + 2].mean())
    ref -= ref[NBINS // 2]
    assert np.allclose(jw, ref, rtol=1e-13, atol=1e-12)
    for k in range(2):
        r = np.load(tmp_path / f"ddrows{k}.npz")
        assert np.array_equal(r["jw"], jw) and np.array_equal(r["ju"], ju)
