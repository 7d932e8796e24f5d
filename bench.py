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


# ---- roofline bookkeeping ------------------------------------------------------------------------------
SIMDS, CUS, CLOCK_HZ = 1024, 256, 2.4e9     # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock
VALU_F64_PEAK = SIMDS * CLOCK_HZ / 4.0      # wave64 f64 instructions / s: 16 lanes per clock per SIMD (78.6 TFLOP/s FMA)


def load_counters(W, M):
    """profiles/counters.json: raw rocprofv3 counters per kernel for one bench configuration (tools/collect_counters.py
    from separate --pmc passes).  Returns {} unless it was taken on this configuration."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        cj = json.load(open(path))
    except (OSError, ValueError):
        return {}
    return cj if cj.get("walkers") == W and cj.get("moves") == M else {}


def kernel_roofline(name, alg_bytes, avg_ms, units, counters, copy_gbs):
    """The roofline block of one kernel.  `bound` names the ceiling the counters say binds -- FP64 VALU issue, the LDS
    pipe, or HBM traffic, each as a fraction of its own ceiling -- and `achieved / peak / frac` are measured against THAT
    ceiling, so `frac` cannot exceed 1:
      valu: wave64 vector instructions per second (SQ_INSTS_VALU of this binary on this workload -- exact and the same in
            every launch -- over the LIVE average launch time) against 1024 SIMDs x 2.4 GHz / 4 cycles per FP64 instruction;
      hbm:  counter traffic (2 FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch over the live launch time against 8 TB/s;
      lds:  SQ_LDS_IDX_ACTIVE cycles against the launch's cycles.
    The SURVEY.md 8(d) convention (algorithmic bytes per launch / launch time against 8 TB/s) stays in the line as
    `algorithmic_GBps` / `convention_frac`: for kernels that serve their operands from LDS it is a throughput
    normalisation and can exceed 1, it says nothing about HBM."""
    sec = avg_ms * 1e-3
    alg = alg_bytes / sec / 1e9
    r = {"kernel": name, "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
         "algorithmic_GBps": alg, "convention_frac": alg / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
         "avg_launch_ms": avg_ms,
         "convention": "algorithmic_GBps = SURVEY.md 8(d) algorithmic bytes per launch / launch time; NOT the kernel's HBM traffic",
         "hbm_traffic_frac": None}
    c = counters.get(name)
    if c:
        cyc_grbm = c["GRBM_GUI_ACTIVE"] / float(counters.get("xcds", 8))   # the counter sums the XCDs; over-counts sub-0.3-ms launches
        cpu_ = counters.get("cycles_per_us")                    # shader clock during the counter passes (from the long kernel)
        cyc_valu = c["dur_us_sq1"] * cpu_ if cpu_ and "dur_us_sq1" in c else cyc_grbm     # the launch's cycles in the pass that
        cyc_lds = c["dur_us_sq2"] * cpu_ if cpu_ and "dur_us_sq2" in c else cyc_grbm      # counted VALU / LDS activity
        r["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None
        if r["traffic"] is not None:
            r["hbm_traffic_frac"] = r["traffic"] / sec / (HBM_PEAK_GBS * 1e9)
        busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cyc_valu)
        issue = c["SQ_INSTS_VALU"] / sec                        # wave64 instructions per second at the LIVE launch time
        r["valu"] = {"busy_duration": busy, "busy_grbm": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cyc_grbm),
                     "insts_per_unit": c["SQ_INSTS_VALU"] * 64.0 / units, "wave_insts_per_launch": c["SQ_INSTS_VALU"],
                     "wave_insts_per_s": issue, "peak_wave_insts_per_s_f64": VALU_F64_PEAK,
                     "unit_of_work": "molecule" if name == "k_model_energy" else "trial move (old + new)"}
        lds_d = c["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc_lds) if "SQ_LDS_IDX_ACTIVE" in c else None
        r["lds"] = {"busy_duration": lds_d, "busy_grbm": c["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc_grbm) if "SQ_LDS_IDX_ACTIVE" in c else None,
                    "bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
                    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE") else None}
        ceilings = {"valu": busy, "lds": lds_d or 0.0, "hbm": r["hbm_traffic_frac"] or 0.0}
        r["bound"] = max(ceilings, key=ceilings.get)
        if r["bound"] == "valu":
            r["achieved"], r["peak"], r["unit"] = issue / 1e9, VALU_F64_PEAK / 1e9, "G wave64-instructions/s (FP64 VALU issue)"
        elif r["bound"] == "hbm":
            r["achieved"], r["peak"], r["unit"] = r["traffic"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            r["achieved"], r["peak"], r["unit"] = lds_d, 1.0, "LDS pipe busy (SQ_LDS_IDX_ACTIVE / CU cycles)"
        r["frac"] = r["achieved"] / r["peak"]
        r["counters_tag"] = counters.get("tag")
        r["counters_note"] = "instruction / byte counts per launch from profiles/counters.json (rocprofv3 --pmc passes of this binary on this configuration); launch time measured live"
    else:
        # no counter summary for this configuration: nothing says what binds, so no ceiling is claimed (frac stays null);
        # the SURVEY.md 8(d) convention is in algorithmic_GBps / convention_frac
        r["bound"] = ("unknown: no counter summary for this configuration under profiles/"
                      + ("; not hbm -- the algorithmic rate exceeds this box's measured copy ceiling (operands served from LDS / L2)"
                         if copy_gbs and alg > copy_gbs else ""))
        r["achieved"], r["peak"], r["unit"] = alg, HBM_PEAK_GBS, "GB/s (SURVEY.md 8(d) convention, not traffic)"
    r["measured_copy_GBps"] = copy_gbs
    return r


def timed(em, slot, fn, reps):
    """Average of a HIP-event bracket over enough launches to last >= 30 ms, after >= 100 ms of the same launches
    (a few sub-millisecond launches on an idle GPU are timed at its idle clock)."""
    t0 = time.perf_counter()
    n_warm = 0
    while time.perf_counter() - t0 < 0.1 or n_warm < 2:
        fn(); em.sync()
        n_warm += 1
    per = (time.perf_counter() - t0) / n_warm
    reps = max(reps, min(2000, int(0.03 / max(per, 1e-6)) + 1))
    em.timer_start(slot)
    for _ in range(reps):
        fn()
    em.timer_stop(slot)
    return em.timer_ms(slot) / reps


def secondary_measurements(device):
    """The figures BASELINE.json's other configurations ask for, measured live after the timed run (own engine
    contexts, HIP events on the engine's stream): configs[2] 1536-molecule Ic/Ih pairs on the single-move path,
    configs[4] 32768-molecule boxes (list rebuild + full energy, 64 per launch and one per launch)."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    out = []

    # configs[4]: 32768-molecule ice Ih
    h, x = lat.ice_box("ih", (16, 16, 16), 0.15, seed=20250228)
    ideal = lat.ice_box("ih", (16, 16, 16), 0.0)[1]
    N = len(x)
    gold = os.path.join(ROOT, "tests", "golden", "ih32768_t015.npz")
    ref = float(np.load(gold)["model_energy"]) if os.path.exists(gold) else None
    for B in (64, 1):
        xs = [x] + [lat.thermalise(ideal, 0.15, 500 + b) for b in range(1, B)]
        em = load_boxes([h] * B, xs, device=device)
        try:
            ms_list = timed(em, 0, lambda: em.build_neighbours_launch(1, B), 5)
            ms_full = timed(em, 1, lambda: em.model_energy_launch(1, B), 10)
            entries = em.neighbour_total(1, B)
            npairs, ntrip = em.model_energy_counts_total(1, B)
            e = em.model_energy_fetch(1, 1)[0]
            b_list, b_full = B * N * (24 + 4) + 8 * entries, B * N * (24 + 8) + 8 * entries
            out.append({"name": f"configs[4] ih32768_t015 x {B} box{'es' if B > 1 else ''} per launch", "molecules": N, "boxes": B,
                        "list_rebuild_ms": ms_list, "list_algorithmic_GBps": b_list / ms_list / 1e6,
                        "list_convention_frac": b_list / ms_list / 1e6 / HBM_PEAK_GBS,
                        "full_energy_ms": ms_full, "full_algorithmic_GBps": b_full / ms_full / 1e6,
                        "full_convention_frac": b_full / ms_full / 1e6 / HBM_PEAK_GBS,
                        "full_interactions_per_s": (npairs + ntrip) / (ms_full * 1e-3),
                        "box1_rel_err_vs_golden": None if ref is None else abs(e - ref) / abs(ref),
                        "note": "positions (786 KiB) do not fit LDS: gathered through L2" + ("" if B > 1 else "; one box per launch is latency, not throughput")})
        finally:
            em.energy_deinit()

    # configs[2]: the 1536-molecule Ic <-> Ih lattice-switch pair, single-move path
    g1p, g2p = (os.path.join(ROOT, "tests", "golden", f) for f in ("ic1536.npz", "ih1536.npz"))
    if os.path.exists(g1p) and os.path.exists(g2p):
        g1, g2 = np.load(g1p), np.load(g2p)
        W, M = 256, 1536
        hs, xs = [], []
        for w in range(W):
            hs += [g1["h"], g2["h"]]
            xs += [lat.thermalise(g1["xyz"], 0.12, 1000 + w), lat.thermalise(g2["xyz"], 0.12, 2000 + w)]
        em = load_boxes(hs, xs, device=device)
        try:
            ils = np.repeat(np.arange(1, 2 * W + 1, dtype=np.int32), M)
            mv = [lat.trial_moves(xs[b], M, seed=b) for b in range(2 * W)]
            em.moves_upload(ils, np.concatenate([m[0] for m in mv]), np.concatenate([m[1] for m in mv]))
            ms = timed(em, 2, em.moves_launch, 10)
            io, so, inw, sn = em.moves_counts()
            nmv = 2 * W * M
            b_mv = 24 * 2 * nmv + 32 * (so + sn)
            out.append({"name": "configs[2] 1536-molecule Ic/Ih lattice-switch pairs, single-move path", "walkers": W,
                        "moves_per_lattice": M, "ms_per_launch": ms, "interactions_per_s": (io + inw) / (ms * 1e-3),
                        "move_evaluations_per_s": 2 * nmv / (ms * 1e-3), "algorithmic_GBps": b_mv / ms / 1e6,
                        "convention_frac": b_mv / ms / 1e6 / HBM_PEAK_GBS,
                        "convention": "SURVEY.md 8(d) algorithmic bytes / time / 8 TB/s: most of those bytes are LDS reads of a staged box, so this is NOT a share of HBM peak and may exceed 1"})
        finally:
            em.energy_deinit()
    return out


# ---- the production driver (k_sweep, the replica farm) in the driver-run line --------------------------------------
def load_sweep_counters():
    """profiles/sweep_counters.json: SQ_INSTS_VALU per move of k_sweep per case (tools/sweep_counters.sh: rocprofv3 --pmc over this
    binary; the count is exact for a given chain), keyed by case name."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "sweep_counters.json")))
    except (OSError, ValueError):
        return {}


def sweep_roofline(case, moves, kernel_ms, counters):
    """k_sweep against the ceiling that binds it, FP64 VALU issue (every counter pass of rounds 3-4: VALU busy 0.5-0.7, HBM traffic
    ~50 B per move on the LDS-resident cases): wave64 vector instructions per move from the committed counter pass of this case
    x moves of this run / the LIVE in-kernel time, against 1024 SIMDs x 2.4 GHz / 4."""
    c = counters.get(case)
    r = {"kernel": "k_sweep", "case": case, "bound": "valu" if c else "unknown: no counter summary for this case under profiles/",
         "achieved": None, "peak": VALU_F64_PEAK / 1e9, "unit": "G wave64-instructions/s (FP64 VALU issue)", "frac": None, "traffic": None,
         "kernel_ms": kernel_ms}
    if c and kernel_ms:
        r["valu_insts_per_move"] = c["SQ_INSTS_VALU_per_move"]
        r["achieved"] = c["SQ_INSTS_VALU_per_move"] * moves / (kernel_ms * 1e-3) / 1e9
        r["frac"] = r["achieved"] / r["peak"]
        r["traffic"] = c.get("hbm_bytes_per_move")
        r["counters_tag"] = counters.get("tag")
    return r


def chain_self_check(device, name):
    """Walker 0's chain on the device against the ORACLE's (tests/golden/<name>.npz, tests/golden/make_chain_fixtures.py): same
    molecule, same outcome (accepted / lattice switch / volume move) move by move, energies to 1e-10 relative -- with one move at a
    time (the build a farm of thousands runs) and with the look-ahead the launch would pick for a handful of walkers; for walkers in
    global memory (chain_ih4096) also on the moment path a launch that fills the chip takes (MW_SWEEP_MOMENTS=2 forces it for the
    two walkers here: what the timed 2048-walker launch runs).  Raises SystemExit on a mismatch, like the check of the timed move kernel."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import MuGrid, WalkerFarm
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    ref = g["log"]
    nmoves = len(ref)
    worst, checked = 0.0, 0
    variants = [("1", None), (None, None)] + ([("1", "2"), ("2", "2")] if not name.startswith("chain_farm48") else [])
    for ahead, moments in variants:
        if ahead is None:
            os.environ.pop("MW_SWEEP_AHEAD", None)
        else:
            os.environ["MW_SWEEP_AHEAD"] = ahead
        if moments is None:
            os.environ.pop("MW_SWEEP_MOMENTS", None)
        else:
            os.environ["MW_SWEEP_MOMENTS"] = moments
        if name.startswith("chain_farm48"):
            z1, z2 = (np.load(os.path.join(ROOT, "tests", "golden", f)) for f in ("ic48.npz", "ih48.npz"))
            nw, nlat = 4, 2
            hs, xs = [], []
            for w in range(nw):
                for l, z in enumerate((z1, z2)):
                    hs.append(z["h"]); xs.append(lat.thermalise(z["xyz"], float(g["sigma_ang"]), 7919 * w + l))
            em = load_boxes(hs, xs, device=device)
            farm = WalkerFarm(em, 2, float(g["temperature"]), float(g["max_trans_ang"]),
                              grid=MuGrid(int(g["nbins"]), -float(g["mu_range"]), float(g["mu_range"])),
                              pressure_au=float(g["pressure_atm"]) / 2.90363081e8)
            npt = bool(int(g["npt"]))
            farm.options(record=True, samplerun=False, always_switch=True, npt=npt, wl_factor=float(g["wl_factor"]))
            if npt:
                farm.moves(trans_prob=0.5, vol_prob=1.0 / em.nwater, dv_max_ang=float(g["dv_max_ang"]))
            farm.set_states(1)
            seed = int(g["seed"])
        else:
            h, x0 = lat.ice_box("ih", (8, 8, 8), 0.0)
            nw, nlat = 2, 1
            xs = [lat.thermalise(x0, float(g["sigma_ang"]), int(g["thermalise_seed"]) + w) for w in range(nw)]
            em = load_boxes([h] * nw, xs, device=device)
            farm = WalkerFarm(em, 1, float(g["temperature"]), float(g["max_trans_ang"]))
            farm.set_states(1)
            seed = int(g["seed"])
        try:
            log = farm.sweep(nmoves, seed=seed, move0=0, first_walker=1, count=1, log=True)[0]
        finally:
            em.energy_deinit()
        if not (np.array_equal(log[:, 0], ref[:, 0]) and np.array_equal(log[:, 1], ref[:, 1])):
            k = int(np.argmax((log[:, 0] != ref[:, 0]) | (log[:, 1] != ref[:, 1])))
            raise SystemExit(f"{name} (look-ahead {ahead or 'auto'}): walker 0 leaves the oracle's chain at move {k}: "
                             f"molecule / outcome {log[k, :2]} against {ref[k, :2]}")
        err = float(np.max(np.abs(log[:, 2:6] - ref[:, 2:6]) / np.maximum(np.abs(ref[:, 2:6]), 1e-300) * (ref[:, 2:6] != 0.0)))
        if not err <= 1e-10:
            raise SystemExit(f"{name} (look-ahead {ahead or 'auto'}): walker 0's energies differ from the oracle's chain (max rel {err:.3e})")
        worst, checked = max(worst, err), checked + nmoves
    os.environ.pop("MW_SWEEP_AHEAD", None)
    os.environ.pop("MW_SWEEP_MOMENTS", None)
    return {"fixture": f"tests/golden/{name}.npz", "moves_checked": checked, "max_rel_err_energies": worst,
            "accepted": int((ref[:, 1].astype(int) & 1).sum()), "lattice_switches": int(((ref[:, 1].astype(int) >> 1) & 1).sum()),
            "volume_moves": int(((ref[:, 1].astype(int) >> 2) & 1).sum())}


def production_driver_measurements(device, backend):
    """BASELINE.json configs[3] as the product runs it: `mc_water_ls_mw_amd.farm.run` -- 48-molecule Ic/Ih walkers, Wang-Landau update
    and a lattice-switch attempt after every move, list rebuilds every 10 cycles, the table exchange every 25 on the process group
    of this run (one rank at N = 1: the collective is executed, on RCCL with --backend nccl) -- for 8192 walkers (NVT and NPT, one
    shared table and the reference's own arithmetic), for 8 and for 1 (the reference's own operating point: one chain's speed); and
    the driver alone on 2048 x 4096-molecule boxes.  End-to-end wall time (lists, exchange, host loop) and k_sweep's share of it
    from HIP events; each workload's chain is first held to the oracle's (chain_self_check)."""
    import torch
    from mc_water_ls_mw_amd import farm as mwfarm
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    counters = load_sweep_counters()
    checks = {n: chain_self_check(device, n) for n in ("chain_farm48_nvt", "chain_farm48_npt", "chain_ih4096")}
    out = {"walker0_chain_checked": int(sum(c["moves_checked"] for c in checks.values() if c)), "chain_checks": checks, "runs": []}
    z1, z2 = (np.load(os.path.join(ROOT, "tests", "golden", f)) for f in ("ic48.npz", "ih48.npz"))
    cdev = torch.device("cuda", device) if backend == "nccl" else None
    for walkers, cycles in ((8192, 100), (8, 500), (1, 500)):
        for npt in (False, True):
            for regauge in ((None, False) if walkers > 64 else (True,)):      # (a handful of walkers: one shared table asked for explicitly)
                res = mwfarm.run([z1["h"], z2["h"]], [z1["xyz"], z2["xyz"]], walkers, cycles, mpi_sync_int=25, device=device,
                                 comms=WalkerComms(NBINS, device=cdev), npt=npt, regauge=regauge, time_kernels=True)
                case = ("farm48_npt" if npt else "farm48_nvt") if walkers > 64 else f"chain48_{'npt' if npt else 'nvt'}_{walkers}"
                out["runs"].append({
                    "name": f"configs[3] replica farm: {walkers} x 48-molecule Ic/Ih walker{'s' if walkers > 1 else ''}, {'NPT' if npt else 'NVT'}, "
                            f"WL update + switch attempt per move, exchange every 25 cycles"
                            + (", one shared table (regauge)" if res["regauge"] else ", the reference's exchange arithmetic (comms_mpi.f90:256-270)"),
                    "walkers": walkers, "cycles": cycles, "ensemble": "npt" if npt else "nvt", "regauge": res["regauge"],
                    "moves_per_s": res["moves_per_s"], "local_energy_evaluations_per_s": 4.0 * res["moves_per_s"],
                    "us_per_move_per_walker": 1e6 * walkers / res["moves_per_s"],
                    "wall_s": res["wall_s"], "k_sweep_ms": res["sweep_kernel_ms"], "k_sweep_share_of_wall": res["sweep_kernel_ms"] * 1e-3 / res["wall_s"],
                    "k_sweep_moves_per_s": res["moves"] / (res["sweep_kernel_ms"] * 1e-3), "launches": res["sweep_launches"],
                    "acceptance": res["acceptance"], "switches_per_walker": res["switches_per_walker"], "weight_max": res["weight_max"],
                    "histogram_total": res["histogram_total"], "drift_walker1_Ha": res["drift_walker1_Ha"],
                    "roofline": sweep_roofline(case, res["moves"], res["sweep_kernel_ms"], counters)})
    # the driver alone on large boxes: 2048 walkers x 4096 molecules, one lattice (positions and lists in HBM, gathered through L2)
    h, x0 = lat.ice_box("ih", (8, 8, 8), 0.0)
    W, nmv = 2048, 300
    em = load_boxes([h] * W, [lat.thermalise(x0, 0.1, w) for w in range(W)], device=device)
    try:
        farm = WalkerFarm(em, 1, 200.0, 1.1)
        farm.set_states(1)
        farm.sweep_launch(20, seed=1, move0=0)
        em.sync()
        em.timer_start(0)
        farm.sweep_launch(nmv, seed=1, move0=20)
        em.timer_stop(0)
        ms = em.timer_ms(0)
        st = farm.state(1)
        fresh = em.model_energy_batch(1, 1)
        out["runs"].append({"name": "k_sweep alone: 2048 x 4096-molecule ice-Ih walkers (one lattice), 300 moves each",
                            "walkers": W, "moves_per_s": W * nmv / (ms * 1e-3), "local_energy_evaluations_per_s": 2.0 * W * nmv / (ms * 1e-3),
                            "k_sweep_ms": ms, "drift_walker1_Ha": [st["model_energy"][0] - fresh[0]],
                            "roofline": sweep_roofline("ih4096_2048", W * nmv, ms, counters)})
    finally:
        em.energy_deinit()
    return out


def self_launch(n, json_fd):
    """`python bench.py --gpus N` without a launcher: start N children of this script, one rank per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the environment torch.distributed.run would give
    them), relay rank 0's ONE JSON line and return the worst exit status.  Called before anything in this process has
    touched the GPU; the children are ordinary child processes (no exec from a process that holds the device)."""
    import signal
    import socket
    import subprocess
    import threading
    import time
    # The port is held (bound, not listening) until every child exists: picking a number and closing the socket at once left a
    # window in which another job could take it -- and the same number keys the RCCL id file.  SO_REUSEADDR lets rank 0's store
    # bind it while this socket is still open; it is closed as soon as the children have been started.
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    procs = []

    def end_children():
        """Terminate, then kill, exactly the children started here."""
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_term(signum, frame):                       # a SIGTERM to the parent must not leave the ranks holding their GPUs
        raise KeyboardInterrupt

    old_term = signal.signal(signal.SIGTERM, on_term)
    chunks = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # ranks above 0 print nothing on stdout by design; whatever a failing one does print there goes to stderr, not away
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr))
        sk.close()
        # rank 0's output is drained by a thread while all ranks are watched: a rank that dies (a bad device index, an
        # allocation that fails) would otherwise leave the others in a collective until its watchdog gives up -- the
        # survivors are ended here instead (exactly the children started above), and the failure is reported at once
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                end_children()
                break
            time.sleep(0.1)
    except BaseException:                             # interrupted (SIGINT / SIGTERM) or failed to start a rank
        end_children()
        raise
    finally:
        sk.close()
        signal.signal(signal.SIGTERM, old_term)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10.0)
    out0 = b"".join(c for c in chunks if c)
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.strip()]
    if any(codes):
        sys.stderr.write(f"bench.py --gpus {n}: rank exit codes {codes}\n")
        return max((c if c > 0 else 1) for c in codes if c)
    if len(lines) != 1:
        sys.stderr.write(f"bench.py --gpus {n}: rank 0 printed {len(lines)} lines instead of one\n")
        return 1
    os.write(json_fd, (lines[0] + "\n").encode())
    return 0


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
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (configs[2] 1536 LS pairs, configs[4] 32768 boxes) after the timed run")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N = 1 only: skip the per-step weight/histogram all-reduce (by default a one-rank process group "
                         "is created so that the N = 1 line times the same step as the N > 1 lines)")
    args = ap.parse_args()

    # Libraries print to stdout too (RCCL announces its version when the communicator is created): the process's
    # stdout carries ONE JSON line, everything else goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    if args.cpu_worker:                       # child of cpu_all_cores(): never touches the GPU
        h, xs = make_walkers(0, 1, args.sigma)
        emit(cpu_baseline(h, xs[0], args.moves, args.cpu_budget))
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    # Rank / size bootstrap (what comms_initialise does with MPI_Comm_rank / _size, comms_mpi.f90:26-71): the launcher's
    # environment if there is one -- it must then agree with --gpus -- otherwise this process IS the launcher.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                     # nothing here has touched the GPU (torch is not even imported yet)
            raise SystemExit(self_launch(args.gpus, json_fd))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                             f"(python bench.py --gpus {args.gpus} launches them itself when WORLD_SIZE is unset)")
    all_cores = None
    if world == 1 and not args.no_cpu_baseline:
        all_cores = cpu_all_cores(args)       # before this process initialises the GPU

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    exchange = world > 1 or not args.no_exchange
    exchange_note = None
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))   # (see WalkerComms: the exchange must not queue behind the engine's kernels)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        except Exception as err:                      # noqa: BLE001  (N = 1 only: the bench still measures the kernels)
            if world > 1:
                raise
            exchange, exchange_note = False, f"none at N=1 ({args.backend} process group unavailable: {err})"

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
    t_setup = time.perf_counter()
    em.setup_boxes()                                    # cells + positions of all walkers: a handful of transfers
    em.timer_start(4000)
    mn, mx = em.build_neighbours_batch(1, W)
    em.timer_stop(4000)
    list_cold_ms = em.timer_ms(4000)                    # the very first launches of the process, GPU at its idle clock
    list_ms = timed(em, 4001, lambda: em.build_neighbours_launch(1, W), 10)      # steady state: what a farm pays every list_update_int cycles
    # trial moves, walker-major
    ils = np.repeat(np.arange(1, W + 1, dtype=np.int32), M)
    imol = np.empty(W * M, dtype=np.int32)
    trial = np.empty((W * M, 3))
    for b in range(W):
        imol[b * M:(b + 1) * M], trial[b * M:(b + 1) * M] = lat.trial_moves(xs[b], M, seed=1 + rank * W + b)
    gold_path = os.path.join(ROOT, "tests", "golden", "ih4096_t015.npz")
    gold = np.load(gold_path) if (rank == 0 and args.sigma == 0.15 and os.path.exists(gold_path)) else None
    n_gold = 0
    if gold is not None:        # walker 0 IS the fixture's configuration: its first moves are the fixture's trial moves
        n_gold = min(M, len(gold["trial_imol"]))
        imol[:n_gold], trial[:n_gold] = gold["trial_imol"][:n_gold], gold["trial_xyz"][:n_gold]
    em.moves_upload(ils, imol, trial)
    t_setup = time.perf_counter() - t_setup

    comms = WalkerComms(NBINS, samplerun=True,
                        device=torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu"))
    weight, hist, uhist = np.zeros(NBINS), np.zeros(NBINS), np.zeros(NBINS)

    def step(k, timed):
        # one host call per step: full-box energies of all walkers, then their trial moves (timed steps: with the engine's
        # HIP events around each kernel, on the engine's stream)
        em.step_launch(1, W, 2 * k if (timed and k < 2000) else -1)
        if exchange:
            hist[(k * 7 + rank) % NBINS] += 1.0
            weight[(k * 7 + rank) % NBINS] += 0.05
            uhist[(k * 3 + rank) % NBINS] += 0.5
            comms.sync(weight, hist, uhist)

    def fence():
        em.sync()
        torch.cuda.synchronize()
        if dist.is_initialized():
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

    if exchange:
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
    # The step's full-box launch also leaves every molecule's moments behind for the step's move kernel (its moment path, DESIGN.md
    # 3.2): ~50 us of stores on top of the evaluation.  The kernel of configs[1] by itself -- compute_model_energy of every walker,
    # nothing else -- is timed here, outside the step loop, and is what `roofline_model_energy` describes.
    for k in range(24):                              # one event pair per launch, each launch behind a launch of the move kernels -- the
        em.moves_launch()                            # conditions the kernel has inside a step (a bracket of back-to-back launches lets their
        em.timer_start(4010 + k)                     # tails and heads overlap: 7 % less per launch; launches on an idle GPU: 8 % more)
        em.model_energy_launch(1, W)
        em.timer_stop(4010 + k)
    ms_full_plain = float(np.mean([em.timer_ms(4010 + k) for k in range(4, 24)]))
    moves_err = None
    if gold is not None and n_gold:    # what the LAST timed k_move_energy launch left on the device, against the reference's values
        eo, en = em.moves_fetch()
        moves_err = float(max(np.max(np.abs(eo[:n_gold] - gold["trial_old"][:n_gold]) / np.abs(gold["trial_old"][:n_gold])),
                              np.max(np.abs(en[:n_gold] - gold["trial_new"][:n_gold]) / np.abs(gold["trial_new"][:n_gold]))))
        if not moves_err <= 1e-10:
            raise SystemExit(f"walker 0: old/new local energies of the timed move kernel differ from the golden vector (max rel {moves_err:.3e})")
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
        if gold is not None:
            ref = float(gold["model_energy"])
            sanity = abs(e_walker0 - ref) / abs(ref)
            if sanity > 1e-10:
                raise SystemExit(f"walker 0 energy {e_walker0!r} differs from the golden vector {ref!r}")
        dominant = "k_model_energy" if ms_full >= ms_moves else "k_move_energy"
        counters = load_counters(W, M)
        rl = {"k_model_energy": kernel_roofline("k_model_energy", bytes_full, ms_full_plain, W * N_MOL, counters, copy_gbs),
              "k_move_energy": kernel_roofline("k_move_energy", bytes_moves, ms_moves, W * M, counters, copy_gbs)}
        entries_bytes = W * N_MOL * (24 + 4) + 8 * entries         # N*(24 + 4 + 8*nbar)
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
                "exchange": (f"one {args.backend} all-reduce of 3 x 101 f64 per step"
                             + (" (one-rank process group)" if world == 1 else "")) if exchange
                            else (exchange_note or "none at N=1 (--no-exchange)"),
            },
            "per_gpu": (i_full + i_moves) * args.steps / elapsed,
            "roofline": rl[dominant],
            "roofline_model_energy": rl["k_model_energy"],     # the kernel of configs[1] itself, whichever dominates the step
            "kernels": {
                "k_model_energy": {"avg_ms": ms_full, "interactions_per_launch": i_full,
                                   "interactions_per_s": i_full / (ms_full * 1e-3),
                                   "algorithmic_GBps": bytes_full / (ms_full * 1e-3) / 1e9,
                                   "atoms_per_s": W * N_MOL / (ms_full * 1e-3),
                                   "note": "inside the step: the evaluation + every molecule's moments for the step's move kernel "
                                           "(80 B per molecule written); `plain_avg_ms` is the evaluation alone, timed after the step loop",
                                   "plain_avg_ms": ms_full_plain, "plain_interactions_per_s": i_full / (ms_full_plain * 1e-3),
                                   "plain_algorithmic_GBps": bytes_full / (ms_full_plain * 1e-3) / 1e9},
                "k_move_energy": {"avg_ms": ms_moves, "interactions_per_launch": i_moves,
                                   "interactions_per_s": i_moves / (ms_moves * 1e-3),
                                   "algorithmic_GBps": bytes_moves / (ms_moves * 1e-3) / 1e9,
                                   "evaluations_per_s": 2 * W * M / (ms_moves * 1e-3)},
                "k_build_neighbours": {"ms_for_all_walkers": list_ms, "first_build_ms": list_cold_ms, "nn_min": mn, "nn_max": mx,
                                       "algorithmic_GBps": entries_bytes / (list_ms * 1e-3) / 1e9,
                                       "convention_frac": entries_bytes / (list_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            },
            "device": {"name": name, "compute_units": cus, "hbm_bytes": mem},
            "walker0_rel_err_vs_golden": sanity,
            "walker0_moves_max_rel_err": moves_err, "walker0_moves_checked": n_gold,
            "setup_s": t_setup,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(h, xs[0], M, args.cpu_budget)
            out["cpu_baseline"]["all_cores"] = all_cores
        else:
            out["cpu_baseline"] = None
    em.energy_deinit()
    if rank == 0:
        if not args.no_secondary and world == 1:
            out["secondary"] = secondary_measurements(local_rank)
            out["production_driver"] = production_driver_measurements(local_rank, args.backend if exchange else "none")
            out["walker0_chain_checked"] = out["production_driver"]["walker0_chain_checked"]
        emit(out)

    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
