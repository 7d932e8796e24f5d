#!/usr/bin/env python3
"""bench.py -- mW interactions/s on the 4096-molecule ice-Ih workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], "ih4096_t015"): every rank holds `--walkers`
independent walkers, each a 4096-molecule hexagonal-ice box (8x8x8 of the 8-atom
orthorhombic cell, nearest O-O 2.73 A, Gaussian displacement sigma 0.15 A, seed
20250228 + walker index; walker 0 is exactly the golden fixture ih4096_t015).
One STEP = one pass of the hot path over that batch:
  (1) full-box energy of every walker        (compute_model_energy, molint.F90:407-499)
  (2) `--moves` trial translations per walker, old and new local energy of each
      (the two compute_local_real_energy calls of a move, mc_moves.F90:1010,1083)
  (3) N > 1 only: the multi-walker weight/histogram all-reduce (comms_mpi.f90:244-530).
Positions, lists and requests are resident in HBM before the timed region;
results stay on the device.  An interaction is one in-range pair or one in-range
triplet as the reference enumerates them; the counts come from
the kernels themselves (integers, checked against the oracle in tests/).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MOL = 4096
NBINS = 101            # examples/ice1_gen_weights/ice.input
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def make_walkers(first_walker, count, sigma):
    from mc_water_ls_mw_amd import lattice as lat
    h0, x0 = lat.ice_box("ih", (8, 8, 8), 0.0)
    xs = [lat.thermalise(x0, sigma, 20250228 + first_walker + w) for w in range(count)]
    return h0, xs


def cpu_baseline(h, xyz, moves_per_walker, budget_s=10.0):
    """The reference's own Fortran (oracle/_ref, kind "reference") -- or, if that
    build did not travel, the C restatement ("port") -- timed on one host core on a
    bounded sample of the same workload: one walker of it."""
    from mc_water_ls_mw_amd import lattice as lat
    from oracle import COracle, RefOracle
    C = COracle()
    iv = C.ivects(h)
    nn, jn, vn = C.neighbours(xyz, iv)
    _, cf = C.model_energy(xyz, iv, nn, jn, vn, counts=True)
    imol, _ = lat.trial_moves(xyz, 4096, seed=1)
    # interactions of those local-energy calls (counts are data, not timing)
    per_call = [C.local_energy(int(i), xyz, iv, nn, jn, vn, counts=True)[1].sum() for i in imol]
    i_local = float(np.sum(per_call))
    if RefOracle.available():
        R = RefOracle()
        R.load([h], [xyz])
        kind = "reference"
        t_full = lambda n: R.time_model_energy(1, n)          # noqa: E731
        t_loc = lambda n: R.time_local_energy(1, n, imol)     # noqa: E731
    else:
        kind = "port"
        t_full = lambda n: [C.model_energy(xyz, iv, nn, jn, vn) for _ in range(n)]   # noqa: E731
        t_loc = lambda n: [C.local_energy_all(xyz, iv, nn, jn, vn) for _ in range(n)]  # noqa: E731
        _, cl = C.local_energy_all(xyz, iv, nn, jn, vn, counts=True)
        i_local = float(cl.sum())

    def rate(fn, budget):
        fn(1)
        t0 = time.perf_counter(); fn(2); dt = (time.perf_counter() - t0) / 2
        n = max(3, int(budget / max(dt, 1e-6)))
        t0 = time.perf_counter(); fn(n); return (time.perf_counter() - t0) / n, n

    sec_full, n_full = rate(t_full, budget_s / 2)
    sec_loc, n_loc = rate(t_loc, budget_s / 2)
    ncalls = len(imol) if kind == "reference" else len(xyz)
    sec_call = sec_loc / ncalls
    i_call = i_local / ncalls
    # same mix as one GPU step of one walker: 1 full-box + 2 local energies per trial move
    t_step = sec_full + 2 * moves_per_walker * sec_call
    i_step = float(cf.sum()) + 2 * moves_per_walker * i_call
    return {
        "value": i_step / t_step, "unit": "interactions/s", "cores": 1, "kind": kind,
        "sample": f"1 walker of the workload: {n_full} full-box evaluations + {n_loc * ncalls} local-energy calls "
                  f"on one host core, combined in the step's mix (1 full-box + 2x{moves_per_walker} local)",
        "full_box_interactions_per_s": float(cf.sum()) / sec_full,
        "local_energy_interactions_per_s": i_call / sec_call,
        "full_box_ms": sec_full * 1e3, "local_call_us": sec_call * 1e6,
        "cpu": _cpu_model(),
    }


def cpu_all_cores(args):
    """The same bounded sample on every host core this process may use, one process per core (the Fortran modules
    hold global state): the aggregate is what the reference's MPI build would deliver on this host's CPUs."""
    import subprocess
    ncores = _cpu_share()
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--moves", str(args.moves), "--sigma", str(args.sigma),
           "--cpu-budget", str(min(args.cpu_budget, 8.0))]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(ncores)]
    vals = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode == 0:
            try:
                vals.append(json.loads(out.strip().splitlines()[-1]))
            except (ValueError, IndexError):
                pass
    if not vals:
        return None
    return {"value": float(sum(v["value"] for v in vals)), "unit": "interactions/s", "cores": len(vals), "kind": vals[0]["kind"],
            "sample": f"{len(vals)} concurrent processes, each the single-core sample", "cpu": vals[0]["cpu"]}


def _cpu_share():
    """Host cores this job may really use: the cgroup CPU quota if there is one, else the affinity mask, and never
    more than 16 processes (a one-GPU box's share of the host)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                if f[0] != "max":
                    n = min(n, max(1, int(int(f[0]) / int(f[1]))))
            else:
                q = int(f[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 16))


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--walkers", type=int, default=512, help="independent 4096-molecule walkers per GPU")
    ap.add_argument("--moves", type=int, default=2048, help="trial moves per walker per step")
    ap.add_argument("--sigma", type=float, default=0.15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)   # one process of the all-cores CPU leg
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 path)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank computes on device 0 (use with --backend gloo)")
    args = ap.parse_args()

    if args.cpu_worker:                       # child of cpu_all_cores(): never touches the GPU
        h, xs = make_walkers(0, 1, args.sigma)
        print(json.dumps(cpu_baseline(h, xs[0], args.moves, args.cpu_budget)), flush=True)
        return
    all_cores = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        all_cores = cpu_all_cores(args)       # before this process initialises the GPU

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.energy import EnergyModule

    W, M = args.walkers, args.moves
    h, xs = make_walkers(rank * W, W, args.sigma)
    em = EnergyModule(N_MOL, W, device=local_rank)
    for b in range(W):
        em.hmatrix[b] = h
        em.ljr[b] = xs[b]
    # untimed set-up: mirror cells + positions, build every walker's Verlet list on the GPU
    em._chk(em.L.mw_init(em.device, em.nwater, em.num_lattices, em.maxneigh))
    em._live = True
    t_setup = time.perf_counter()
    for b in range(1, W + 1):
        em.compute_ivects(b)
        em._upload(b)
    em.timer_start(4000)
    mn, mx = em.build_neighbours_batch(1, W)
    em.timer_stop(4000)
    list_ms = em.timer_ms(4000)
    # trial moves, walker-major
    ils = np.repeat(np.arange(1, W + 1, dtype=np.int32), M)
    imol = np.empty(W * M, dtype=np.int32)
    trial = np.empty((W * M, 3))
    for b in range(W):
        imol[b * M:(b + 1) * M], trial[b * M:(b + 1) * M] = lat.trial_moves(xs[b], M, seed=1 + rank * W + b)
    em.moves_upload(ils, imol, trial)
    t_setup = time.perf_counter() - t_setup

    comms = WalkerComms(NBINS, samplerun=True,
                        device=torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu"))
    weight, hist, uhist = np.zeros(NBINS), np.zeros(NBINS), np.zeros(NBINS)

    def step(k, timed):
        if timed and k < 2000:
            em.timer_start(2 * k); em.model_energy_launch(1, W); em.timer_stop(2 * k)
            em.timer_start(2 * k + 1); em.moves_launch(); em.timer_stop(2 * k + 1)
        else:
            em.model_energy_launch(1, W)
            em.moves_launch()
        if world > 1:
            hist[(k * 7 + rank) % NBINS] += 1.0
            weight[(k * 7 + rank) % NBINS] += 0.05
            uhist[(k * 3 + rank) % NBINS] += 0.5
            comms.sync(weight, hist, uhist)

    def fence():
        em.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    fence()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every walker of every rank must have ended with the same shared weights / histograms
        chk = torch.tensor(np.concatenate([weight, hist, uhist]), dtype=torch.float64,
                           device="cuda" if args.backend == "nccl" else "cpu")
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit("ranks disagree on the synchronised weights/histograms")
        expect = float(args.warmup + args.steps) * world
        if abs(float(hist.sum()) - expect) > 1e-9:
            raise SystemExit(f"histogram total {hist.sum()} != {expect}: the delta all-reduce lost or duplicated increments")

    # ---- accounting (outside the timed region) ----------------------------------------
    npairs, ntrip = em.model_energy_counts_total(1, W)
    i_full = npairs + ntrip
    io, so, inw, sn = em.moves_counts()
    i_moves = io + inw
    entries = em.neighbour_total(1, W)
    bytes_full = W * N_MOL * (24 + 8) + 8 * entries             # N*(24 + 8*nbar + 8)
    bytes_moves = 24 * (2 * W * M) + 32 * (so + sn)             # 24 + 8*slots + 24*slots per evaluation
    nk = min(args.steps, 2000)
    ms_full = float(np.mean([em.timer_ms(2 * k) for k in range(nk)]))
    ms_moves = float(np.mean([em.timer_ms(2 * k + 1) for k in range(nk)]))
    e_walker0 = em.model_energy_fetch(1, 1)[0]
    name, cus, mem = em.device_info()

    # the ceiling a plain device-to-device copy reaches on this box (report against nominal AND this)
    copy_gbs = None
    if rank == 0:
        nbytes = 1 << 30
        a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            b.copy_(a)
        t1.record(); torch.cuda.synchronize()
        copy_gbs = 2.0 * nbytes * 10 / (t0.elapsed_time(t1) * 1e-3) / 1e9          # bytes read + bytes written
        del a, b

    if rank == 0:
        sanity = None
        gold = os.path.join(ROOT, "tests", "golden", "ih4096_t015.npz")
        if os.path.exists(gold) and args.sigma == 0.15:
            ref = float(np.load(gold)["model_energy"])
            sanity = abs(e_walker0 - ref) / abs(ref)
            if sanity > 1e-10:
                raise SystemExit(f"walker 0 energy {e_walker0!r} differs from the golden vector {ref!r}")
        dominant = "k_model_energy" if ms_full >= ms_moves else "k_move_energy"
        dom_bytes, dom_ms = (bytes_full, ms_full) if dominant == "k_model_energy" else (bytes_moves, ms_moves)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        side = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(side):
            try:
                tj = json.load(open(side))
                if tj.get("walkers") == W and tj.get("moves") == M:
                    traffic = tj.get(dominant)
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": "mW interactions/sec/GPU (full-box + single-move ΔE), 4096-mol ice; 1/2/4/8-GPU replica scaling",
            "value": (i_full + i_moves) * args.steps * world / elapsed,
            "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "ih4096_t015: 4096-molecule ice-Ih boxes (8x8x8 x 8-atom cell, sigma 0.15 A), "
                            "per step: full-box energy of every walker + old/new local energy of every trial move",
                "walkers_per_gpu": W, "moves_per_walker": M, "molecules": N_MOL,
                "parallelism": f"replica farm: {world} x {W} independent walkers, one process per GPU",
                "exchange": (f"one {args.backend} all-reduce of 3 x 101 f64 per step" if world > 1 else "none at N=1"),
            },
            "per_gpu": (i_full + i_moves) * args.steps / elapsed,
            "roofline": {
                "bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
            },
            "kernels": {
                "k_model_energy": {"avg_ms": ms_full, "interactions_per_launch": i_full,
                                   "interactions_per_s": i_full / (ms_full * 1e-3),
                                   "algorithmic_GBps": bytes_full / (ms_full * 1e-3) / 1e9,
                                   "atoms_per_s": W * N_MOL / (ms_full * 1e-3)},
                "k_move_energy": {"avg_ms": ms_moves, "interactions_per_launch": i_moves,
                                   "interactions_per_s": i_moves / (ms_moves * 1e-3),
                                   "algorithmic_GBps": bytes_moves / (ms_moves * 1e-3) / 1e9,
                                   "evaluations_per_s": 2 * W * M / (ms_moves * 1e-3)},
                "k_build_neighbours": {"ms_for_all_walkers": list_ms, "nn_min": mn, "nn_max": mx},
            },
            "device": {"name": name, "compute_units": cus, "hbm_bytes": mem},
            "walker0_rel_err_vs_golden": sanity,
            "setup_s": t_setup,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(h, xs[0], M, args.cpu_budget)
            out["cpu_baseline"]["all_cores"] = all_cores
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    em.energy_deinit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
