"""Replica farm: many multicanonical walkers per GPU, one process per GPU, periodic delta all-reduce of the
weight / histogram tables -- BASELINE.json configs[3] (examples/ice1_gen_weights) on the device-resident driver.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m mc_water_ls_mw_amd.farm --walkers 1024 --cycles 500

Per cycle, as mc_cycle does (mc_moves.F90:117-321) with volume moves off: rebuild the Verlet lists every
``list_update_int`` cycles, ``nwater`` translation moves per walker (each followed by mc_update_wl_bins and a
lattice-switch attempt), and every ``mpi_sync_int`` cycles the synchronisation of weights and histograms over all
walkers of all GPUs (comms_allreduce_eta/hist, comms_mpi.f90:244-277,461-494).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import time

import numpy as np


#: farms of more walkers than this (all ranks together) exchange one shared table by default (``regauge``)
REGAUGE_ABOVE = 64


def run(h_pair, x_pair, walkers, cycles, temperature=200.0, nbins=101, mu_range=400.0, wl_factor=0.05,
        list_update_int=10, mpi_sync_int=250, sigma_ang=0.05, seed=2025, device=0, comms=None, rank=0,
        samplerun=False, weight=None, npt=False, pressure_atm=1.0,
        flat_chk_int=10000, wl_schedule=0, wl_flattol=0.05, wl_minhist=20, wl_useinvt=False, file_wl_factor=None,
        deltaG_int=100000, max_mc_cycles=None, eq_mc_cycles=0, outdir=None, thermalise=True, regauge=None,
        parallel_strategy="mw", window_overlap=2, leshift=False, input_ref_enthalpy=None, wl_swetnam=False, wl_alpha=1.0,
        eq_adjust_mc=False, mc_target_ratio=0.5, monitor_int=1000, mc_max_trans_ang=1.1, mc_dv_max_ang=0.924,
        latt_sync_int=10000, chkpt_dump_int=None, restart=False, minu=False, time_kernels=False):
    """Run `cycles` MC cycles of `walkers` two-lattice walkers on this GPU.  Returns a dict of results.

    ``flat_chk_int`` ... ``file_wl_factor``: the Wang-Landau schedule (mc_check_flatness, :291-294;
    :mod:`mc_water_ls_mw_amd.schedule`); ``deltaG_int``: free-energy estimate of a sample run (:302-306);
    ``outdir``: where wlf.dat and the tagged tables go (nothing is written when None).
    ``regauge``: False is the reference's exchange arithmetic to the letter (comms_mpi.f90:256-270: every rank's
    increment carries the window minimum it subtracted); True sums the increments proper and subtracts the minimum
    once (WalkerFarm.synchronise): one shared table in the reference's own gauge, identical to the reference for one
    walker.  None (default) = True for more than ``REGAUGE_ABOVE`` = 64 walkers in all, False up to there: the
    reference's scheme multiplies a uniform offset by -(walkers - 1) at every synchronisation (harmless for its 8 ranks
    over a run, 1.9e10 after four synchronisations of 8192 walkers; tests/test_sweep.py shows the growth), so a farm of
    hundreds of walkers that followed it to the letter would be faithful and useless.
    ``parallel_strategy``: 'mw' (every walker samples the whole range, tables synchronised every mpi_sync_int cycles) or
    'dd' (mc_moves.F90:659-709: every walker of every GPU is one window of ``world x walkers``, overlapping its
    neighbours by ``window_overlap`` bins, with its own increment and flatness check and no exchange; the windows are
    stitched at the end as mc_monitor_stats does, :1851-1925).  ``leshift`` / ``input_ref_enthalpy``: userparams.f90:41,57;
    ``wl_swetnam`` / ``wl_alpha``: mc_moves.F90:1636-1653.  ``eq_adjust_mc`` / ``mc_target_ratio`` / ``monitor_int``: every
    monitor_int cycles the stored energies are replaced by freshly computed ones and, below eq_mc_cycles, every walker's
    step sizes are tuned toward the target acceptance ratio (mc_monitor_stats, :1724-1732,1783-1787).  ``latt_sync_int``:
    every so many cycles lattice 2 of every walker is re-imposed from its lattice 1 (mc_check_chain_synchronisation, :296-300).
    ``chkpt_dump_int`` / ``restart`` (with ``outdir``): every walker writes the reference's own checkpoint file --
    ``checkpointRRR.dat.{1,2}`` alternately, R = its global index (mc_checkpoint_write, :324-390; at most 1000 walkers in
    all) -- and a restarted run takes cycle number, step sizes, increment, tables, cells, reference and current positions
    and the active lattice from the newer readable one (mc_checkpoint_load, :393-501) and runs ``cycles`` MORE cycles.
    ``minu``: the reference's compile-time ``-DMINU`` variant as a run option (an accepted move also takes the walker to
    the lattice of lower enthalpy, mc_moves.F90:1119-1140,1385-1401).
    ``time_kernels``: HIP events around every launch of the Monte Carlo driver (read back long after the launch has ended, so
    nothing waits for them): ``sweep_kernel_ms`` / ``sweep_launches`` in the result say what share of the wall time is k_sweep."""
    from . import lattice as lat
    from .energy import EnergyModule
    from .schedule import WangLandauSchedule, WindowSchedules, delta_g_from_hist, log_unbiased_norm
    from .sweep import MuGrid, WalkerFarm

    if parallel_strategy not in ("mw", "dd"):
        raise ValueError("Unknown parallel_strategy")                      # mc_moves.F90:720
    dd = parallel_strategy == "dd"
    nwalk_all = walkers * (comms.world_size if comms is not None else 1)
    if regauge is None:
        regauge = nwalk_all > REGAUGE_ABOVE
    if not dd and not regauge and not samplerun and nwalk_all > REGAUGE_ABOVE and rank == 0:
        import sys
        print(f"mc_water_ls_mw_amd.farm: {nwalk_all} walkers exchange their weights with the reference's own arithmetic "
              "(comms_mpi.f90:256-270, asked for with regauge=False / --no-regauge), whose uniform offset grows by the number of "
              "walkers at every synchronisation: the table reaches the end of the double range within a few dozen synchronisations "
              "(8192 walkers: 1e307 after 80).", file=sys.stderr)
    n = len(x_pair[0])
    em = EnergyModule(n, 2 * walkers, device=device)
    for w in range(walkers):
        for l in range(2):
            em.hmatrix[2 * w + l] = h_pair[l]
            em.ljr[2 * w + l] = (lat.thermalise(x_pair[l], sigma_ang, 7919 * (rank * walkers + w) + l)
                                 if thermalise else np.asarray(x_pair[l], dtype=np.float64))
    start_cycle, chk, restart_factors = 0, None, None
    if restart:                                                            # mc_checkpoint_load
        from . import io as mwio
        chk = [mwio.latest_checkpoint(outdir, rank * walkers + w)[1] for w in range(walkers)]
        start_cycle = chk[0]["cycle"]
        # The ranks AGREE before any of them stops: a rank that raised on its own left the others waiting in the broadcast
        # below or in the first synchronisation.  Every rank takes part in the broadcast of rank 0's cycle number (mc_cycle_num is
        # rank 0's, :441: ranks that picked files of different cycles would pair their collectives wrongly) and in one get_max
        # of "something is wrong here"; then all of them raise, the bad ones saying what they found.
        bad = None
        if any(c["cycle"] != start_cycle for c in chk):
            bad = ("the walkers' checkpoint files are not of the same cycle: " + ", ".join(str(c["cycle"]) for c in chk[:8]))
        if comms is not None and comms.world_size > 1:
            agreed = comms.bcast_int(start_cycle, 0)
            if bad is None and agreed != start_cycle:
                bad = f"rank {rank} restarts from cycle {start_cycle}, rank 0 from {agreed}: checkpoint files out of step"
            anybad = comms.get_max(0.0 if bad is None else 1.0)
            if bad is None and anybad:
                bad = f"rank {rank}: another rank's checkpoint files are out of step (its message says which); cycle here {start_cycle}"
        if bad is not None:
            raise ValueError(bad)
        for w, c in enumerate(chk):
            if c["nwater"] != n or len(c["hmatrix"]) != 2:
                raise ValueError("checkpoint does not match this run")
            em.hmatrix[2 * w:2 * w + 2] = c["hmatrix"]
            em.ljr[2 * w:2 * w + 2] = c["ljr"]
    last_cycle = start_cycle + cycles
    em.setup_boxes()                                   # cells + positions of every box in a handful of transfers
    try:
        list_rows = em.build_neighbours_batch(1, 2 * walkers)          # (shortest, longest) row of any box
        em.model_energy_batch(1, 2 * walkers)
        grid = MuGrid(nbins, -mu_range, mu_range)
        farm = WalkerFarm(em, 2, temperature, mc_max_trans_ang, grid=grid, weight=weight, pressure_au=pressure_atm / 2.90363081e8)
        comms = farm.local_comms() if comms is None else comms             # one exchange object throughout
        if weight is not None and chk is None:
            # mc_init (mc_moves.F90:738-776): rank 0 reads eta_weights.dat, the others hold zeros, and comms_allreduce_eta hands
            # everybody the table -- which is thereby also the baseline of the delta scheme.  Here every walker starts with the
            # table, so the baseline is set directly (left at zero, the first synchronisation of W walkers returned W x the table).
            comms.set_weights(np.asarray(weight, dtype=np.float64))
        skw = dict(wl_schedule=wl_schedule, wl_flattol=wl_flattol, wl_minhist=wl_minhist, wl_useinvt=wl_useinvt,
                   wl_swetnam=wl_swetnam, samplerun=samplerun, outdir=outdir)
        if dd:                                                             # :659-709: one window per walker
            sched = WindowSchedules(grid, comms.world_size * walkers, window_overlap, comms.rank * walkers, walkers,
                                    wl_factor, **skw)
            farm.set_windows(sched.windows)
            farm.dd(True, eq_mc_cycles)
            wrow = np.tile(farm.weight, (walkers, 1))                      # :808-812: only the window's part of the weights
            for k, w_ in enumerate(sched.windows):
                wrow[k, :w_["start_bin"] - 1] = 0.0
                wrow[k, w_["end_bin"]:] = 0.0
            farm.set_tables_range(1, weight=wrow)
        else:
            sched = WangLandauSchedule(grid.nbins, wl_factor, **skw)
        sched.adopt_file_factor(file_wl_factor)                            # mc_moves.F90:751-760,816-821
        if leshift:                                                        # main.f90:146-150
            ref = input_ref_enthalpy
            if ref is None or not np.any(np.abs(ref) > np.finfo(np.float64).tiny):
                ref = farm.starting_enthalpy(1, npt)
            farm.leshift(ref)
        if minu:
            farm.minu(True)
        if wl_swetnam:
            farm.swetnam(True, wl_alpha, wl_factor)
            farm.set_factors(wl_factor=np.full(walkers, sched.wl_factors[0] if dd else sched.wl_factor))
        lun = 0.0
        if samplerun:                                                      # :778-806
            lun = log_unbiased_norm(np.zeros(grid.nbins) if weight is None else weight, grid.av_binwidth,
                                    cycles if max_mc_cycles is None else max_mc_cycles, eq_mc_cycles,
                                    comms.world_size * walkers, n)

        def set_options(cyc):
            if dd:                                                 # per-window increments (Swetnam's live on the device)
                farm.options(record=cyc >= eq_mc_cycles, samplerun=samplerun, always_switch=True, npt=npt,
                             wl_factor=0.0, log_unbiased_norm=lun)
                if not wl_swetnam:
                    farm.set_factors(wl_factor=sched.move_factors(cyc, n))
            else:
                farm.options(record=cyc >= eq_mc_cycles, samplerun=samplerun, always_switch=True, npt=npt,
                             wl_factor=sched.move_factor(cyc, n), log_unbiased_norm=lun)    # :1615,1655-1657

        if npt:                                                    # io.f90:171-172: vol 1/N against trans 0.5
            farm.moves(trans_prob=0.5, vol_prob=1.0 / n, dv_max_ang=mc_dv_max_ang)
        farm.set_reference()                                       # ref_hmatrix / ref_ljr (init.f90:90,106) for the chain synchronisation
        if chk is not None:                                        # the rest of mc_checkpoint_load
            farm._ref_h = np.array([h_pair[l] for _ in range(walkers) for l in range(2)], dtype=np.float64)   # init.f90:90: the input cells
            farm._s_ref = np.einsum("bnd,bdk->bnk", np.concatenate([c["ref_ljr"] for c in chk]), np.linalg.inv(em.hmatrix))
            farm.set_tables_range(1, weight=np.array([c["weight"] for c in chk]), histogram=np.array([c["histogram"] for c in chk]),
                                  unbiased_hist=np.array([c["unbiased_hist"] for c in chk]) if samplerun else None)
            comms.set_histogram(chk[0]["histogram"])               # :455-458
            # The loader does not re-base the weights (:455-458 name the histograms only): the reference's baseline is what
            # mc_init's all-reduce left (:775), i.e. the eta_weights.dat the interrupted run dumped -- or zero without that
            # file, and then R ranks come back from their first synchronisation with R x the table.  A farm has no such file
            # to fall back on: the baseline is the checkpointed table itself (exactly the synchronised table when checkpoints
            # fall on synchronisation cycles, as they do with the reference's default intervals).
            if not dd:                                             # (rank 0's, for everybody: the delta scheme needs ONE baseline)
                comms.set_weights(comms._bcast(np.asarray(chk[0]["weight"], dtype=np.float64)))
            if samplerun:
                comms.set_uhistogram(chk[0]["unbiased_hist"])
            step_t0 = np.array([c["mc_max_trans"] for c in chk]); step_v0 = np.array([c["mc_dv_max"] for c in chk])
            # wl_factor / wl_invt_active come back on EVERY rank (mc_checkpoint_load, :447-448,462-464): with 'dd' every
            # walker is a rank with an increment, a 1/t flag and a first-cycle state of its own
            for k, sc in enumerate(sched.scheds if dd else [sched]):
                sc.wl_factor = float(chk[k]["wl_factor"])
                sc.invt_active = bool(chk[k]["wl_invt_active"])
                if sc.wl_factor < sc.orig_wl_factor:
                    sc.firstcycle = False
            if dd or wl_swetnam:                                   # per-walker increments (and Swetnam's visit totals) live on the device
                # sumhist = sum(histogram) is what the loader sets (:475) -- but only past the 'mw' branch's early return
                # (:469-472): a restarted 'mw' run keeps the module's initial sumhist = 0 (:94)
                farm.set_factors(wl_factor=[c["wl_factor"] for c in chk],
                                 sumhist=[float(np.sum(c["histogram"])) if dd else 0.0 for c in chk])
                restart_factors = farm.factors()
        # :703-704: a window on one side of mu = 0 fixes the lattice; every walker's state in two transfers
        farm.set_states([chk[w]["ls"] if chk is not None else ((sched.windows[w]["ls"] or 1) if dd else 1) for w in range(walkers)])
        from .lattice import ANG_TO_BOHR
        step_t = np.full(walkers, mc_max_trans_ang * ANG_TO_BOHR)  # mc_max_trans / mc_dv_max of every walker, bohr (io.f90:165-166)
        step_v = np.full(walkers, mc_dv_max_ang * ANG_TO_BOHR)
        if chk is not None:
            step_t, step_v = step_t0, step_v0
            farm.set_steps(step_t, step_v)
        nchk = 0

        def write_checkpoints(c_):                                 # mc_checkpoint_write, :324-390
            from . import io as mwio
            nonlocal nchk
            if comms.world_size * walkers > 1000:
                raise ValueError("checkpoint files are numbered with three digits: at most 1000 walkers in all")
            h = farm.sync_cells() if npt else np.array(em.hmatrix)
            x = np.zeros((2 * walkers, n, 3))
            em._chk(em.L.mw_download_positions_range(1, 2 * walkers, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
            ref_ljr = np.einsum("bnk,bkd->bnd", farm._s_ref, h)    # ref_ljr follows the cell (fractional reference kept)
            wt, hi, uh = farm.tables_range()
            # (the windows' increments are the schedule's: the device copy lags a halving until the next stretch starts)
            facs = farm.factors()[0] if wl_swetnam else (sched.wl_factors if dd else np.full(walkers, sched.wl_factor))
            for k in range(walkers):
                st = farm.state(k + 1)
                mwio.write_checkpoint(os.path.join(outdir, "checkpoint%03d.dat.%d" % (comms.rank * walkers + k, 1 + nchk % 2)),
                                      dict(nwater=n, cycle=c_, mc_max_trans=step_t[k], mc_dv_max=step_v[k], wl_factor=facs[k],
                                           histogram=hi[k], weight=wt[k], wl_invt_active=sched.invt_active, samplerun=samplerun,
                                           unbiased_hist=uh[k], hmatrix=h[2 * k:2 * k + 2], ref_ljr=ref_ljr[2 * k:2 * k + 2],
                                           ljr=x[2 * k:2 * k + 2], ls=st["ls"]))
            nchk += 1
        def check_flags_everywhere():
            """farm.check_flags on every rank, and every rank stops if one of them has to (the reference's ranks agree before they
            stop, mc_moves.F90:187-201; a rank that raised alone left the others waiting in their next collective)."""
            err = None
            try:
                farm.check_flags()
            except Exception as e:                                 # noqa: BLE001 -- re-raised below, after the ranks have agreed
                err = e
            stop = comms.get_max(1.0 if err is not None else 0.0) if comms.world_size > 1 else float(err is not None)
            if err is not None:
                raise err
            if stop:
                raise RuntimeError("another rank's walkers failed their window / image-vector check (its message says which)")

        mon_acc = mon_vatt = mon_vacc = np.zeros(walkers, dtype=np.int64)
        mon_cycle = 0
        t0 = time.perf_counter()
        synced, events, delta_g = None, [], None
        def ends_a_stretch(c):
            """Does the host have something to do after cycle c (or before cycle c + 1)?"""
            return (c == last_cycle or (c + 1) % list_update_int == 0 or c % mpi_sync_int == 0 or c % flat_chk_int == 0
                    or c % monitor_int == 0 or c % latt_sync_int == 0 or (chkpt_dump_int is not None and c % chkpt_dump_int == 0)
                    or (samplerun and c % deltaG_int == 0) or c + 1 == eq_mc_cycles or sched.invt_active)

        cyc = mon_cycle = start_cycle
        nlaunch, kernel_ms = 0, 0.0
        while cyc < last_cycle:
            first = cyc + 1
            if first % list_update_int == 0:                       # mc_moves.F90:217-222
                if npt:
                    farm.sync_cells()                              # device-side volume moves changed the cells
                list_rows = em.build_neighbours_batch(1, 2 * walkers)      # checked: fails loudly on list overflow
            # Cycles between two host actions go out as ONE launch (the move counter simply runs on): a launch
            # stages a walker's positions and list rows in LDS, which is amortised over n moves per cycle only.
            # (In 1/t mode the increment changes every cycle, so cycles stay single.)
            cyc = first
            while not ends_a_stretch(cyc):
                cyc += 1
            set_options(first)
            if time_kernels:
                slot = 3000 + nlaunch % 64
                if nlaunch >= 64:
                    kernel_ms += em.timer_ms(slot)                 # (the launch 64 launches ago: long finished)
                em.timer_start(slot)
            farm.sweep_launch(n * (cyc - first + 1), seed=seed + rank, move0=(first - 1) * n)
            if time_kernels:
                em.timer_stop(slot)
            nlaunch += 1
            if cyc % mpi_sync_int == 0:                            # mc_moves.F90:258-276
                em.sync()
                if npt or dd:
                    check_flags_everywhere()
                if not dd:                                         # (:270-272: the windows exchange nothing)
                    synced = farm.synchronise(comms, regauge=regauge)
            if cyc % monitor_int == 0:                             # mc_monitor_stats, :281-284
                em.sync()
                acc, vatt, vacc = farm.counters()
                if eq_adjust_mc and cyc < eq_mc_cycles:            # :1729-1732, every walker its own (each rank of the reference does)
                    att_t = (cyc - mon_cycle) * n - (vatt - mon_vatt)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        atr = (acc - mon_acc) / att_t
                        avr = (vacc - mon_vacc) / (vatt - mon_vatt)
                    step_t = np.where(att_t > 0, np.maximum(step_t * atr / mc_target_ratio, 0.1), step_t)
                    step_v = np.where(vatt - mon_vatt > 0, np.maximum(step_v * avr / mc_target_ratio, 0.0001), step_v)
                    farm.set_steps(step_t, step_v)
                mon_acc, mon_vatt, mon_vacc, mon_cycle = acc, vatt, vacc, cyc
                em.model_energy_batch(1, 2 * walkers)              # :1783-1787: stored energies <- computed ones
            if cyc % flat_chk_int == 0:                            # :291-294
                em.sync()
                if dd:
                    events += [dict(action=what, cycle=cyc, walker=comms.rank * walkers + k - 1)   # 'walker' = the window's rank
                               for k, what in sched.check_flatness(cyc, n, farm)]
                else:
                    ev = sched.check_flatness(cyc, n, farm, comms)
                    if ev["action"] != "none":
                        events.append(ev)
            if cyc % latt_sync_int == 0:                           # :296-300
                em.sync()
                farm.chain_synchronise()
            if samplerun and cyc % deltaG_int == 0:                # :302-306
                em.sync()
                if dd:                                             # comms_join_uhist (:2535)
                    joined_u = comms.join_uhist(farm.tables_range()[2], window_overlap)
                else:
                    synced = farm.synchronise(comms, regauge=regauge)  # comms_allreduce_uhist (:2532) with the rest
                    joined_u = synced[2]
                dg, per, normp = delta_g_from_hist(joined_u, grid.binwidth, n, temperature, beta_dh=farm.beta_dh())
                delta_g = dict(cycle=cyc, kT=dg, **{"per_molecule_" + k: v for k, v in per.items()})
                if outdir is not None and comms.rank == 0:         # :2590-2613
                    with open(os.path.join(outdir, "unbiased_histogram_%010d.dat" % cyc), "w") as fh:
                        for m_, p_ in zip(grid.mu_bin, normp):
                            fh.write(f"  {float(m_)!r}        {float(p_)!r}\n")
            if chkpt_dump_int is not None and cyc % chkpt_dump_int == 0:   # :311-315
                em.sync()
                write_checkpoints(cyc)
        em.sync()
        if npt or dd:
            check_flags_everywhere()                               # every walker, not only the ones read out below
        wall = time.perf_counter() - t0
        if time_kernels:
            kernel_ms += sum(em.timer_ms(3000 + k % 64) for k in range(max(0, nlaunch - 64), nlaunch))
        joined = None
        if dd:                                                     # mc_monitor_stats, :1883-1886
            facs = farm.factors()[0] if wl_swetnam else sched.wl_factors
            joined = dict(weight=comms.join_eta(farm.tables_range()[0], window_overlap),
                          wl_factor=comms.get_max(float(np.max(facs))))
            if samplerun:
                joined["unbiased_hist"] = comms.join_uhist(farm.tables_range()[2], window_overlap)
        states = [farm.state(w) for w in range(1, min(walkers, 32) + 1)]
        fresh = em.model_energy_batch(1, 2)
        out = dict(walkers=walkers, cycles=cycles, molecules=n, moves=walkers * cycles * n, wall_s=wall,
                   moves_per_s=walkers * cycles * n / wall,
                   acceptance=float(np.mean([s["accepted"] for s in states])) / (cycles * n),
                   switches_per_walker=float(np.mean([farm.switches(w) for w in range(1, min(walkers, 32) + 1)])),
                   drift_walker1_Ha=[states[0]["model_energy"][l] - fresh[l] for l in range(2)],
                   volume_moves_walker1=farm.volume_moves(1) if npt else None,
                   histogram_total=None if synced is None else float(synced[1].sum()),
                   weight_max=None if synced is None else float(synced[0].max()),
                   wl_factor=(joined["wl_factor"] if dd else float(farm.factors(1, 1)[0][0]) if wl_swetnam else sched.wl_factor),
                   wl_invt_active=sched.invt_active, flatness_events=events, delta_g=delta_g,
                   ref_enthalpy=farm.ref_enthalpy,
                   max_trans_bohr=step_t[:32].tolist(), dv_max_bohr=step_v[:32].tolist(),
                   list_rows_last_rebuild=list(list_rows), regauge=bool(regauge),
                   sweep_kernel_ms=kernel_ms if time_kernels else None, sweep_launches=nlaunch)
        out["tables"] = synced
        out["restart_factors"] = None if restart_factors is None else [np.asarray(a).tolist() for a in restart_factors]
        out["joined"] = joined
        if dd:
            out["windows"] = sched.windows
            out["in_window"] = farm.factors()[2].tolist()
        out["walker1_tables"] = farm.tables(1)
        out["walker1_positions"] = [farm.positions(1), farm.positions(2)]
        out["first_walkers"] = [dict(positions=[farm.positions(2 * w - 1), farm.positions(2 * w)], tables=farm.tables(w),
                                     **farm.state(w)) for w in range(1, min(walkers, 4) + 1)]
        return out
    finally:
        em.energy_deinit()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, default=512)
    ap.add_argument("--cycles", type=int, default=100)
    ap.add_argument("--sync", type=int, default=25)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-device", action="store_true")
    ap.add_argument("--npt", action="store_true", help="volume moves too (mc_ensemble = 'npt')")
    ap.add_argument("--wl-factor", type=float, default=0.05)
    ap.add_argument("--flat-chk", type=int, default=10000, help="flat_chk_int: cycles between flatness checks")
    ap.add_argument("--wl-schedule", type=int, default=0, choices=[0, 1, 2])
    ap.add_argument("--wl-flattol", type=float, default=0.05)
    ap.add_argument("--wl-minhist", type=int, default=20)
    ap.add_argument("--wl-useinvt", action="store_true")
    ap.add_argument("--outdir", default=None, help="directory for wlf.dat / eta_weights.dat_* / histogram.dat_*")
    ap.add_argument("--strategy", default="mw", choices=["mw", "dd"], help="parallel_strategy")
    ap.add_argument("--window-overlap", type=int, default=2)
    ap.add_argument("--eq-cycles", type=int, default=0, help="eq_mc_cycles")
    ap.add_argument("--leshift", action="store_true")
    ap.add_argument("--minu", action="store_true", help="the reference's -DMINU variant: accepted moves end in the lattice of lower enthalpy")
    ap.add_argument("--wl-swetnam", action="store_true")
    ap.add_argument("--wl-alpha", type=float, default=1.0)
    ap.add_argument("--chkpt", type=int, default=None, help="chkpt_dump_int: cycles between checkpoint files (needs --outdir)")
    ap.add_argument("--restart", action="store_true", help="continue from the checkpoint files in --outdir")
    ap.add_argument("--latt-sync", type=int, default=10000, help="latt_sync_int")
    ap.add_argument("--no-thermalise", action="store_true", help="every walker starts from the input configuration itself (as the ranks of the reference do)")
    ap.add_argument("--eq-adjust", action="store_true", help="eq_adjust_mc: tune the step sizes during equilibration")
    ap.add_argument("--monitor", type=int, default=1000, help="monitor_int")
    ap.add_argument("--temperature", type=float, default=200.0, help="Kelvin")
    ap.add_argument("--pressure", type=float, default=1.0, help="atmospheres (with --npt)")
    ap.add_argument("--mu-range", type=float, default=400.0, help="the order parameter runs over -mu_range .. +mu_range (101 bins)")
    ap.add_argument("--list-update", type=int, default=10, help="list_update_int: cycles between Verlet-list rebuilds")
    ap.add_argument("--samplerun", action="store_true", help="fixed weights, unbiased histogram (examples/ice1_sample); needs --weights")
    ap.add_argument("--weights", default=None, help="eta_weights.dat: the starting weights (mc_moves.F90:738-770)")
    ap.add_argument("--delta-g", type=int, default=100000, help="deltaG_int: cycles between free-energy estimates of a sample run")
    ap.add_argument("--regauge", dest="regauge", action="store_true", default=None,
                    help="exchange step sums the weight increments proper and subtracts the window minimum once: one shared table "
                         "(default for more than 64 walkers in all)")
    ap.add_argument("--no-regauge", dest="regauge", action="store_false",
                    help="the reference's exchange arithmetic to the letter, comms_mpi.f90:256-270 (default up to 64 walkers in all)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from .comms import WalkerComms
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))   # (see WalkerComms: the exchange must not queue behind the engine's kernels)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    z1, z2 = np.load(os.path.join(gold, "ic48.npz")), np.load(os.path.join(gold, "ih48.npz"))
    comms = WalkerComms(101, device=torch.device("cuda", local) if (world > 1 and args.backend == "nccl") else None)
    weight, file_factor = None, None
    if args.samplerun and args.weights is None:
        raise SystemExit("--samplerun needs --weights (the reference stops without eta_weights.dat too)")
    if args.weights is not None:
        from . import io as mwio
        file_factor, _, weight = mwio.read_table(args.weights)
        if len(weight) != 101:
            raise SystemExit(f"{args.weights}: {len(weight)} bins, this farm runs the examples' 101")
    res = run([z1["h"], z2["h"]], [z1["xyz"], z2["xyz"]], args.walkers, args.cycles, mpi_sync_int=args.sync,
              samplerun=args.samplerun, weight=weight, file_wl_factor=file_factor, deltaG_int=args.delta_g,
              temperature=args.temperature, pressure_atm=args.pressure, mu_range=args.mu_range, list_update_int=args.list_update,
              device=local, comms=comms, rank=rank, npt=args.npt, wl_factor=args.wl_factor, flat_chk_int=args.flat_chk,
              wl_schedule=args.wl_schedule, wl_flattol=args.wl_flattol, wl_minhist=args.wl_minhist,
              wl_useinvt=args.wl_useinvt, outdir=args.outdir, regauge=args.regauge, parallel_strategy=args.strategy,
              window_overlap=args.window_overlap, eq_mc_cycles=args.eq_cycles, leshift=args.leshift,
              wl_swetnam=args.wl_swetnam, wl_alpha=args.wl_alpha, eq_adjust_mc=args.eq_adjust, monitor_int=args.monitor,
              thermalise=not args.no_thermalise, chkpt_dump_int=args.chkpt, restart=args.restart, latt_sync_int=args.latt_sync, minu=args.minu)
    tabs = res.pop("tables")
    res.pop("walker1_tables"), res.pop("walker1_positions"), res.pop("first_walkers")
    joined = res.pop("joined")
    if joined is not None:
        tabs = (joined["weight"], joined["weight"])
        res["joined_weight_range"] = [float(joined["weight"].min()), float(joined["weight"].max())]
    res.pop("windows", None)
    if world > 1:
        t = torch.tensor(np.concatenate(tabs[:2]), dtype=torch.float64,
                         device=torch.device("cuda", local) if args.backend == "nccl" else "cpu")
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["ranks_agree"] = bool(torch.equal(lo, hi))
        agg = torch.tensor([res["moves_per_s"]], dtype=torch.float64, device=t.device)
        dist.all_reduce(agg)
        res["moves_per_s_all_ranks"] = float(agg.item())
    if rank == 0:
        res["world"] = world
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
