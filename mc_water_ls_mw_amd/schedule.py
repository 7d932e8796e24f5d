"""Wang-Landau increment schedule and free-energy read-out of the multicanonical runs: the host-side layer that sits
between the device-resident move driver and the exchange step.

    mc_check_flatness              mc_moves.F90:1936-2185   WangLandauSchedule.check_flatness
    1/t clamp of the increment     mc_moves.F90:1655-1657   WangLandauSchedule.move_factor
    increment from eta_weights.dat mc_moves.F90:751-760,816 WangLandauSchedule.adopt_file_factor
    log_unbiased_norm              mc_moves.F90:778-806     log_unbiased_norm
    mc_compute_deltaG_from_hist    mc_moves.F90:2498-2621   delta_g_from_hist

All of it is O(nbins) arithmetic once every flat_chk_int (default 10^4) cycles: it stays on the host, on the tables the
farm already brings over for the exchange step.  Parallel strategy 'mw' (every walker samples the whole range):
:class:`WangLandauSchedule`, one for the whole job.  Window decomposition ('dd'): every walker is a rank of the
reference with its own window, histogram, increment and first-cycle state, nothing is exchanged
(:class:`WindowSchedules`, mc_moves.F90:2002-2016,2114-2126).

Adam Swetnam's per-move increment (wl_swetnam, mc_moves.F90:1636-1653) lives on the device (mw_sweep_swetnam); here it
only switches the flatness test off, as wl_invt_active does (:2018).
"""
from __future__ import annotations

import math
import os

import numpy as np

KB = 1.0 / 3.1577465e5        # Hartree / K   (constants.f90:39)
HART_TO_KJPM = 2625.49962     # constants.f90:49
HART_TO_EV = 27.211396181     # constants.f90:47


class WangLandauSchedule:
    """State: wl_factor, orig_wl_factor, firstcycle, histogram_reset, wl_invt_active (mc_moves.F90:80-85,1958)."""

    def __init__(self, nbins, wl_factor, wl_schedule=0, wl_flattol=0.05, wl_minhist=20, wl_useinvt=False,
                 wl_swetnam=False, samplerun=False, start_bin=1, end_bin=None, invt_dump_int=500000, outdir=None):
        self.wl_swetnam = bool(wl_swetnam)
        if wl_schedule not in (0, 1, 2):
            raise ValueError("Error - unknown wl_schedule value")          # mc_moves.F90:2051
        self.nbins = int(nbins)
        self.wl_factor = float(wl_factor)
        self.orig_wl_factor = float(wl_factor)                             # :734
        self.wl_schedule, self.wl_flattol, self.wl_minhist = int(wl_schedule), float(wl_flattol), int(wl_minhist)
        self.wl_useinvt, self.samplerun = bool(wl_useinvt), bool(samplerun)
        self.start_bin, self.end_bin = int(start_bin), int(self.nbins if end_bin is None else end_bin)
        self.invt_dump_int = int(invt_dump_int)
        self.outdir = outdir
        self.firstcycle = True
        self.histogram_reset = False
        self.invt_active = False

    def adopt_file_factor(self, file_factor):
        """mc_init: an existing eta_weights.dat carries the increment it was written with; the smaller one wins, a
        sample run has none, and a smaller-than-input increment means the first refinement cycle is over."""
        if file_factor is not None and file_factor > 1e-10:                # :757-760
            self.wl_factor = min(self.wl_factor, float(file_factor))
            if self.samplerun:
                self.wl_factor = 0.0
        if self.wl_factor < self.orig_wl_factor:                           # :816-821
            self.firstcycle = False

    def move_factor(self, cycle, nwater):
        """The increment mc_update_wl_bins applies during MC cycle ``cycle`` (constant within a cycle)."""
        if self.invt_active:                                               # :1655-1657
            self.wl_factor = min(self.wl_factor, float(self.nbins) / float(cycle * nwater))
        return self.wl_factor

    def check_flatness(self, cycle, nwater, farm, comms=None):
        """mc_check_flatness on a WalkerFarm.  Returns a dict describing what happened (for the log)."""
        from . import io as mwio
        if self.samplerun:                                                 # :1962
            return dict(action="none")
        hist = farm.allreduce_hist(comms)                                  # :1965-1967 (every walker now holds the sum)
        if hist.sum() < np.finfo(np.float64).tiny:                         # :1962 (the reference tests its local sum first)
            return dict(action="none")
        out = dict(action="checked", cycle=cycle)
        mini = _nint(hist.min())                                           # :1972  nint(minval(histogram))
        if self.firstcycle and not self.histogram_reset and mini > self.wl_minhist:      # :1973-1980
            self.histogram_reset = True
            farm.reset_histogram(comms)
            out["action"] = "first reset"
            return out
        win = hist[self.start_bin - 1:self.end_bin]
        av = _seqsum(win) / float(len(win))                                # :1983-1989
        out["most_pct"], out["least_pct"] = 100.0 * hist.max() / av, 100.0 * hist.min() / av   # :1997-1998
        if not (self.invt_active or self.wl_swetnam):                      # :2018
            flat = self._is_flat(win, av)
            if comms is not None:                                          # :2055
                flat = comms.bcast_flag(flat)
            out["flat"] = flat
            if flat:
                farm.shift_weights()                                       # :2062-2066  weight -= weight(nbins/2+1)
                if self.outdir is not None and (comms is None or comms.rank == 0):
                    w, h, _ = farm.tables(1)
                    fresh = self.firstcycle or not os.path.exists(os.path.join(self.outdir, "wlf.dat"))
                    mwio.append_wlf(self.outdir, [(cycle, self.wl_factor), (cycle, 0.5 * self.wl_factor)], replace=fresh)
                    mwio.write_tagged_tables(self.outdir, "%.12f" % self.wl_factor, self.wl_factor, farm.grid.mu_bin, w, hist)
                farm.reset_histogram(comms)                                # :2105-2106
                self.wl_factor *= 0.5                                      # :2107
                self.firstcycle = False                                    # :2112
                out["action"] = "halved"
            wl_invt = float(self.nbins) / float(cycle * nwater)            # :2136-2143
            if self.wl_factor < wl_invt and self.wl_factor > np.finfo(np.float64).tiny and self.wl_useinvt:
                self.invt_active = True
                self.wl_factor = wl_invt
                out["invt"] = True
        else:                                                              # :2145-2181: periodic dumps only
            if self.wl_swetnam:                                            # rank 0's own increment lives on the device
                self.wl_factor = float(farm.factors(1, 1)[0][0])
            if self.outdir is not None and (comms is None or comms.rank == 0) and cycle % self.invt_dump_int == 0:
                w, _, _ = farm.tables(1)
                mwio.append_wlf(self.outdir, [(cycle, self.wl_factor)], replace=False)
                mwio.write_tagged_tables(self.outdir, "%020d" % cycle, self.wl_factor, farm.grid.mu_bin, w, hist)
            out["action"] = "invt" if self.invt_active else "swetnam"
        out["wl_factor"] = self.wl_factor
        return out


    def _is_flat(self, win, av):
        if self.wl_schedule == 0:                                          # within wl_flattol of the mean, :2024-2031
            return not bool(np.any(np.abs(win - av) / av > self.wl_flattol))
        if self.wl_schedule == 1:                                          # every bin visited wl_minhist times, :2033-2039
            return not (_nint(win.min()) < self.wl_minhist)
        return not bool(np.any(win < (1.0 - self.wl_flattol) * av))        # above (1 - wl_flattol) of the mean, :2041-2048

    def check_window(self, cycle, nwater, hist):
        """mc_check_flatness as ONE rank of a 'dd' run sees it (no exchange, its own histogram, the window
        start_bin..end_bin; flat: histogram reset and increment halved, but no weight shift and no wlf.dat,
        :2114-2126).  Returns (what happened, whether the rank's histogram is to be zeroed)."""
        if self.samplerun or _seqsum(hist) < np.finfo(np.float64).tiny:    # :1962
            return "none", False
        if self.firstcycle and not self.histogram_reset and _nint(hist.min()) > self.wl_minhist:   # :1972-1980
            self.histogram_reset = True
            return "first reset", True
        win = hist[self.start_bin - 1:self.end_bin]
        av = _seqsum(win) / float(len(win))
        if self.invt_active or self.wl_swetnam:
            return "invt" if self.invt_active else "swetnam", False
        what, zero = "checked", False
        if self._is_flat(win, av):
            self.wl_factor *= 0.5
            self.firstcycle = False
            what, zero = "halved", True
        wl_invt = float(self.nbins) / float(cycle * nwater)                # :2136-2143
        if self.wl_factor < wl_invt and self.wl_factor > np.finfo(np.float64).tiny and self.wl_useinvt:
            self.invt_active = True
            self.wl_factor = wl_invt
        return what, zero


class WindowSchedules:
    """parallel_strategy = 'dd' for a farm: walker k of this process is rank ``rank0 + k`` of ``size`` windows, each with
    its own WangLandauSchedule state (mc_moves.F90:80-85 are per-rank variables)."""

    def __init__(self, grid, size, overlap, rank0, count, wl_factor, **kw):
        self.windows = [grid.window(rank0 + k, size, overlap) for k in range(count)]
        self.scheds = [WangLandauSchedule(grid.nbins, wl_factor, start_bin=w["start_bin"], end_bin=w["end_bin"], **kw)
                       for w in self.windows]
        self.overlap, self.size = int(overlap), int(size)

    @property
    def wl_factors(self):
        return np.array([s.wl_factor for s in self.scheds])

    @property
    def invt_active(self):
        return any(s.invt_active for s in self.scheds)

    def adopt_file_factor(self, f):
        for s in self.scheds:
            s.adopt_file_factor(f)

    def move_factors(self, cycle, nwater):
        return np.array([s.move_factor(cycle, nwater) for s in self.scheds])

    def check_flatness(self, cycle, nwater, farm):
        """Every walker looks at its own histogram over its own window; returns the list of (walker, what)."""
        _, hist, _ = farm.tables_range(1, len(self.scheds))
        events, dirty = [], False
        for k, s in enumerate(self.scheds):
            what, zero = s.check_window(cycle, nwater, hist[k])
            if zero:
                hist[k] = 0.0
                dirty = True
            if what not in ("none", "checked"):
                events.append((k + 1, what))
        if dirty:
            farm.set_tables_range(1, histogram=hist)
        return events


def _nint(x):
    """Fortran nint(): halves round away from zero (numpy.rint rounds them to even)."""
    x = float(x)
    return int(math.floor(x + 0.5)) if x >= 0.0 else -int(math.floor(-x + 0.5))


def _seqsum(a):
    """Left-to-right double-precision sum, as a Fortran do loop forms it (numpy's pairwise sum differs in the last bits)."""
    s = 0.0
    for v in a:
        s += float(v)
    return s


def log_unbiased_norm(weight, av_binwidth, max_mc_cycles, eq_mc_cycles, nranks, nwater):
    """log of the expected total of the unbiased histogram, summed in an overflow-proof way (mc_moves.F90:778-806);
    mc_update_wl_bins divides every unbiased increment by exp() of it (:1628)."""
    nbins = len(weight)
    hits_per_bin = (float(max_mc_cycles) - float(eq_mc_cycles)) * float(nranks * nwater) / float(nbins)
    incr = hits_per_bin * av_binwidth
    lun = math.log(incr) + float(weight[0])
    for k in range(1, nbins):
        wk = float(weight[k])
        if lun > wk + math.log(incr):
            lun = lun + math.log(1.0 + incr * math.exp(wk - lun))
        else:
            lun = math.log(incr) + wk + math.log(1.0 + math.exp(lun - wk) / incr)
    return lun


def delta_g_from_hist(joined, binwidth, nwater, temperature, beta_dh=0.0):
    """mc_compute_deltaG_from_hist (mc_moves.F90:2546-2601) on the joined (all-reduced or window-stitched) unbiased
    histogram: G(lattice 2) - G(lattice 1).  ``beta_dh`` = beta (H_ref(2) - H_ref(1)) when leshift is on (:2586).
    Returns (deltaG in kT, per-molecule values in kT / J mol^-1 / meV, normalised P(mu))."""
    joined, binwidth = np.asarray(joined, dtype=np.float64), np.asarray(binwidth, dtype=np.float64)
    nbins = len(joined)
    pnorm = _seqsum(joined * binwidth)                                     # :2547-2550
    normp = joined / pnorm                                                 # :2551-2553
    pa = _seqsum(normp[:nbins // 2] * binwidth[:nbins // 2])               # :2566-2571
    pb = _seqsum(normp[nbins // 2:] * binwidth[nbins // 2:])               # :2574-2579
    dg = math.log(pa / pb) + beta_dh                                       # :2584-2586
    kt = KB * temperature
    per = dict(kT=dg / nwater, J_per_mole=kt * HART_TO_KJPM * 1000.0 * dg / nwater, meV=kt * HART_TO_EV * 1000.0 * dg / nwater)
    return dg, per, normp
