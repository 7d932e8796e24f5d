"""Device-resident translation-move driver: many independent walkers advance their Markov
chains on the GPU (mc_water_translation, mc_moves.F90:966-1213).

:class:`MuGrid` restates the overlap-parameter grid of mc_init (mc_moves.F90:571-656);
:class:`WalkerFarm` wraps the ``mw_sweep_*`` C-ABI entries over an initialised
:class:`~mc_water_ls_mw_amd.energy.EnergyModule` whose boxes are grouped ``nlat`` per walker.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

KB = 1.0 / 3.1577465e5          # constants.f90:39, Hartree / Kelvin
_dp = ctypes.POINTER(ctypes.c_double)


def _seqsum(a):
    """Left-to-right sum, as Fortran's sum() over a short array forms it here (flang: sequential)."""
    t = 0.0
    for v in a:
        t += float(v)
    return t


def _ipow(r, n):
    acc, b = 1.0, r
    while n > 0:
        if n & 1:
            acc *= b
        b *= b
        n >>= 1
    return acc


class MuGrid:
    """Geometric bin widths either side of a unit middle bin (mc_moves.F90:571-656)."""

    def __init__(self, nbins=101, mu_min=-400.0, mu_max=400.0):
        if nbins % 2 == 0:
            nbins += 1                                     # mc_moves.F90:555
        self.nbins, self.mu_min, self.mu_max = nbins, float(mu_min), float(mu_max)
        ns = nbins // 2
        self.a_pos = self.a_neg = 1.0
        self.r_pos = self._ratio(self.a_pos, abs(mu_max) - 0.5, ns)
        self.r_neg = self._ratio(self.a_neg, abs(mu_min) - 0.5, ns)
        self.mu_bin = np.zeros(nbins)
        self.binwidth = np.zeros(nbins)
        mu_u, k = -0.5, 0
        for ibin in range(ns, 0, -1):                      # :625-633
            mu_l = mu_u - self.a_neg * _ipow(self.r_neg, k)
            self.mu_bin[ibin - 1] = 0.5 * (mu_u + mu_l)
            self.binwidth[ibin - 1] = mu_u - mu_l
            mu_u, k = mu_l, k + 1
        self.mu_bin[ns] = 0.0                              # :636-637
        self.binwidth[ns] = 1.0
        mu_l, k = 0.5, 0
        for ibin in range(ns + 2, nbins + 1):              # :641-649
            mu_u = mu_l + self.a_pos * _ipow(self.r_pos, k)
            self.mu_bin[ibin - 1] = 0.5 * (mu_u + mu_l)
            self.binwidth[ibin - 1] = mu_u - mu_l
            mu_l, k = mu_u, k + 1
        self.av_binwidth = self.binwidth.sum() / nbins      # :652-656
        # 'mw' strategy: every walker spans the whole range (mc_moves.F90:711-718)
        self.start_bin, self.end_bin = 1, nbins
        self.my_mu_min, self.my_mu_max = self.mu_min, self.mu_max

    def window(self, rank, size, overlap):
        """parallel_strategy = 'dd' (mc_moves.F90:659-709): the window of rank `rank` of `size` -- bins (1-based,
        inclusive), limits of mu, and the lattice that has to be active there (None: either)."""
        bpw = self.nbins // size                                           # :663
        bw = self.binwidth
        if rank == 0:                                                      # :665-673
            s, e = 1, bpw + overlap
            lo, hi = self.mu_min, self.mu_min + _seqsum(bw[:e])
        elif rank < size - 1:                                              # :677-688
            s, e = rank * bpw - overlap, (rank + 1) * bpw + overlap
            lo, hi = self.mu_min + _seqsum(bw[:s - 1]), self.mu_min + _seqsum(bw[:e])
        else:                                                              # :690-698
            s, e = rank * bpw - overlap, self.nbins
            lo, hi = self.mu_min + _seqsum(bw[:s - 1]), self.mu_max
        if size == 1:
            e, hi = bpw + overlap, self.mu_min + _seqsum(bw[:bpw + overlap])   # (rank 0's branch is the only one taken)
        ls = 1 if hi < 0.0 else (2 if lo > 0.0 else None)                  # :703-704
        return dict(start_bin=s, end_bin=e, mu_min=lo, mu_max=hi, ls=ls)

    def restricted(self, w):
        """A copy of the grid whose walker is confined to window `w` (my_start_bin .. my_mu_max of that rank)."""
        import copy
        g = copy.copy(self)
        g.start_bin, g.end_bin, g.my_mu_min, g.my_mu_max = w["start_bin"], w["end_bin"], w["mu_min"], w["mu_max"]
        return g

    @staticmethod
    def _ratio(a, s, ns):                                  # :583-596
        r = 1.1
        for _ in range(1000001):
            tmpsum = a * (1.0 - _ipow(r, ns)) / (1.0 - r)
            r_new = r * (s / tmpsum) ** (1.0 / ns)
            if abs(r_new - r) <= 2.0 * np.finfo(float).eps:
                break
            r = r_new
        return r


class WalkerFarm:
    def __init__(self, em, nlat, temperature, max_trans_ang=1.1, grid=None, weight=None, eta_interp=True,
                 pressure_au=0.0):
        from .lattice import ANG_TO_BOHR
        self.em, self.L, self.nlat = em, em.L, int(nlat)
        if em.num_lattices % nlat:
            raise ValueError("boxes do not split into walkers")
        self.nwalkers = em.num_lattices // nlat
        self.beta = 1.0 / (KB * temperature)                # mc_moves.F90:998
        self.max_trans = max_trans_ang * ANG_TO_BOHR        # io.f90:165
        self.pressure = pressure_au
        self.grid = grid if grid is not None else MuGrid()
        g = self.grid
        self.weight = np.zeros(g.nbins) if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
        self.eta_interp = bool(eta_interp)
        self.dref = 0.0                                      # ref_enthalpy(1) - ref_enthalpy(2) when leshift is on
        self.ref_enthalpy = None
        mb, bw = np.ascontiguousarray(g.mu_bin), np.ascontiguousarray(g.binwidth)
        em._chk(self.L.mw_sweep_configure(
            self.nlat, ctypes.c_double(self.beta), ctypes.c_double(self.max_trans), g.nbins, int(self.eta_interp),
            g.start_bin, g.end_bin, ctypes.c_double(g.r_pos), ctypes.c_double(g.a_pos), ctypes.c_double(g.r_neg),
            ctypes.c_double(g.a_neg), ctypes.c_double(g.my_mu_min), ctypes.c_double(g.my_mu_max),
            self.weight.ctypes.data_as(_dp), mb.ctypes.data_as(_dp), bw.ctypes.data_as(_dp)))

    def options(self, record=False, samplerun=True, always_switch=False, npt=False, wl_factor=0.0,
                log_unbiased_norm=0.0):
        """Switch on the rest of a translation-only mc_cycle (two lattices): mc_update_wl_bins after every
        move (``record``; ``samplerun`` keeps the weights fixed) and a lattice-switch attempt after every move
        (``always_switch``, mc_moves.F90:243-248)."""
        self.em._chk(self.L.mw_sweep_options(int(record), int(samplerun), int(always_switch), int(npt),
                                             ctypes.c_double(self.grid.av_binwidth), ctypes.c_double(wl_factor),
                                             ctypes.c_double(log_unbiased_norm), ctypes.c_double(self.pressure)))

    # -- the run-control options beyond the 'mw' defaults ---------------------------------
    def leshift(self, ref_enthalpy=None):
        """leshift (userparams.f90:41): the lattices' reference enthalpies are taken out of the order parameter
        (mc_moves.F90:860,1371,1526,1584,2256,2401) and of the switch acceptance (:1567,1572).  ``ref_enthalpy``:
        the pair (Hartree), e.g. input_ref_enthalpy; None switches it off.  main.f90:146-150 takes the starting
        configuration's model_energy (+ pressure*volume under npt): :meth:`starting_enthalpy`."""
        self.ref_enthalpy = None if ref_enthalpy is None else (float(ref_enthalpy[0]), float(ref_enthalpy[1]))
        r1, r2 = (0.0, 0.0) if self.ref_enthalpy is None else self.ref_enthalpy
        self.dref = r1 - r2
        self.em._chk(self.L.mw_sweep_leshift(ctypes.c_double(r1), ctypes.c_double(r2)))

    def minu(self, on=True):
        """The reference's ``-DMINU`` build as a run option (mc_moves.F90:1119-1140,1168-1170,1385-1401,1426-1429): an
        accepted translation or volume move also takes the walker to the lattice of lower enthalpy (E + PV, less
        ref_enthalpy under :meth:`leshift`), with the switch's terms in the acceptance."""
        self.em._chk(self.L.mw_sweep_minu(int(bool(on))))

    def starting_enthalpy(self, walker=1, npt=False):
        """ref_enthalpy as main.f90:146-147 forms it from a walker's current full-box energies."""
        b = (walker - 1) * 2
        e, v = self.em.model_energy, self.em.volume
        return tuple(float(e[b + l] + (self.pressure * v[b + l] if npt else 0.0)) for l in range(2))

    def beta_dh(self):
        """beta (H_ref(2) - H_ref(1)), the correction mc_compute_deltaG_from_hist adds back (mc_moves.F90:2586)."""
        return -self.beta * self.dref

    def swetnam(self, on, wl_alpha=1.0, orig_wl_factor=0.0):
        """wl_swetnam (mc_moves.F90:1636-1653): every walker recomputes its own increment after each recorded move from
        the r.m.s. deviation of its histogram from flat; per-walker increments and visit totals: :meth:`factors`."""
        g = self.grid
        self.em._chk(self.L.mw_sweep_swetnam(int(bool(on)), ctypes.c_double(wl_alpha), ctypes.c_double(orig_wl_factor),
                                             ctypes.c_double(g.mu_min), ctypes.c_double(g.mu_max)))

    def dd(self, on, eq_mc_cycles=0):
        """parallel_strategy = 'dd' (mc_moves.F90:181-210,243-248): until cycle eq_mc_cycles a walker outside its window
        carries no weight and attempts no lattice switch; one still outside at eq_mc_cycles raises the flag
        :meth:`check_flags` reports.  Windows: :meth:`set_windows`; increments are per walker (:meth:`set_factors`)."""
        self.em._chk(self.L.mw_sweep_dd(int(bool(on)), int(eq_mc_cycles)))

    def set_windows(self, windows, first_walker=1):
        """One window (MuGrid.window) per walker from ``first_walker`` on; None: everybody back on the grid's own."""
        _ip = ctypes.POINTER(ctypes.c_int)
        if windows is None:
            self.em._chk(self.L.mw_sweep_windows(1, self.nwalkers, None, None, None, None))
            return
        sb = np.ascontiguousarray([w["start_bin"] for w in windows], dtype=np.int32)
        eb = np.ascontiguousarray([w["end_bin"] for w in windows], dtype=np.int32)
        lo = np.ascontiguousarray([w["mu_min"] for w in windows], dtype=np.float64)
        hi = np.ascontiguousarray([w["mu_max"] for w in windows], dtype=np.float64)
        self.em._chk(self.L.mw_sweep_windows(first_walker, len(windows), sb.ctypes.data_as(_ip), eb.ctypes.data_as(_ip),
                                             lo.ctypes.data_as(_dp), hi.ctypes.data_as(_dp)))

    def set_factors(self, wl_factor=None, sumhist=None, first_walker=1):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (wl_factor, sumhist)]
        count = len(next(a for a in arrs if a is not None))
        self.em._chk(self.L.mw_sweep_set_factors(first_walker, count, *[None if a is None else a.ctypes.data_as(_dp) for a in arrs]))

    def factors(self, first_walker=1, count=None):
        """(wl_factor, Swetnam visit total, walker_in_window) per walker."""
        count = self.nwalkers - first_walker + 1 if count is None else count
        f, sh, iw = np.zeros(count), np.zeros(count), np.zeros(count, dtype=np.int32)
        self.em._chk(self.L.mw_sweep_get_factors(first_walker, count, f.ctypes.data_as(_dp), sh.ctypes.data_as(_dp),
                                                 iw.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return f, sh, iw.astype(bool)

    def set_steps(self, max_trans_bohr=None, dv_max_bohr=None, first_walker=1):
        """Per-walker step sizes in bohr (None: everybody back on the common ones)."""
        if max_trans_bohr is None or dv_max_bohr is None:
            self.em._chk(self.L.mw_sweep_steps(1, self.nwalkers, None, None))
            return
        a = np.ascontiguousarray(max_trans_bohr, dtype=np.float64)
        b = np.ascontiguousarray(dv_max_bohr, dtype=np.float64)
        self.em._chk(self.L.mw_sweep_steps(first_walker, len(a), a.ctypes.data_as(_dp), b.ctypes.data_as(_dp)))

    def counters(self, first_walker=1, count=None):
        """(accepted translations, attempted volume moves, accepted volume moves) per walker since configuration."""
        count = self.nwalkers - first_walker + 1 if count is None else count
        _lp = ctypes.POINTER(ctypes.c_longlong)
        a, v, w = (np.zeros(count, dtype=np.int64) for _ in range(3))
        self.em._chk(self.L.mw_sweep_get_counters(first_walker, count, a.ctypes.data_as(_lp), v.ctypes.data_as(_lp), w.ctypes.data_as(_lp)))
        return a, v, w

    def tables_range(self, first_walker=1, count=None):
        """(weight, histogram, unbiased_hist), each count x nbins."""
        count = self.nwalkers - first_walker + 1 if count is None else count
        nb = self.grid.nbins
        w, h, u = np.zeros((count, nb)), np.zeros((count, nb)), np.zeros((count, nb))
        self.em._chk(self.L.mw_sweep_get_tables_range(first_walker, count, w.ctypes.data_as(_dp), h.ctypes.data_as(_dp), u.ctypes.data_as(_dp)))
        return w, h, u

    def set_tables_range(self, first_walker, weight=None, histogram=None, unbiased_hist=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (weight, histogram, unbiased_hist)]
        count = len(next(a for a in arrs if a is not None))
        self.em._chk(self.L.mw_sweep_set_tables_range(first_walker, count, *[None if a is None else a.ctypes.data_as(_dp) for a in arrs]))

    def moves(self, trans_prob=0.5, vol_prob=0.0, dv_max_ang=0.924):
        """Move mix of mc_cycle (mc_moves.F90:157-166): transP = trans/(trans + vol); volume moves change one
        symmetric cell element by at most dv_max."""
        from .lattice import ANG_TO_BOHR
        self.transP = trans_prob / (trans_prob + vol_prob)
        self.dv_max = dv_max_ang * ANG_TO_BOHR
        self.em._chk(self.L.mw_sweep_moves(ctypes.c_double(self.transP), ctypes.c_double(self.dv_max)))

    def volume_moves(self, walker):
        a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
        self.em._chk(self.L.mw_sweep_get_volume_moves(walker, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def check_flags(self):
        """Fail loudly if a volume move of ANY walker outgrew the image-vector table since the last call (such a
        move is rejected and undone on the device, so the walker's state is consistent, but its chain has left
        mc_volume's), or if a 'dd' walker was still outside its window at eq_mc_cycles (the reference stops there,
        mc_moves.F90:187-201)."""
        self.em._chk(self.L.mw_sweep_check_flags(1, self.nwalkers))

    def last_launch(self):
        """What the last launch looked like: dict(nlat, ahead, residency, volume_moves, lds_bytes, row_stride) -- residency 0: a
        walker's data in global memory, 1: positions in LDS, 2: positions and list rows in LDS; ahead: moves of a chain in flight."""
        v = [ctypes.c_int(0) for _ in range(6)]
        self.em._chk(self.L.mw_sweep_last_launch(*[ctypes.byref(x) for x in v]))
        return dict(zip(("nlat", "ahead", "residency", "volume_moves", "lds_bytes", "row_stride"), (x.value for x in v)))

    def sync_cells(self):
        """Bring the host's hmatrix / volume / image vectors / grid descriptors up to date after device-side
        volume moves (call before rebuilding neighbour lists)."""
        nb = self.em.num_lattices
        h = np.zeros((nb, 3, 3))
        self.em._chk(self.L.mw_sweep_sync_cells(1, nb, h.ctypes.data_as(_dp)))
        self.em.hmatrix[:] = h
        # |det hmatrix| as util_determinant expands it (util.f90:16-41), all boxes at once: numpy's stacked LAPACK determinant
        # took 2.4 ms for a farm's 16 384 boxes -- with an NPT farm calling this before every list rebuild, time the GPU idled
        m = h.reshape(nb, 9)
        det = m[:, 0] * (m[:, 4] * m[:, 8] - m[:, 7] * m[:, 5])
        det = det - m[:, 3] * (m[:, 1] * m[:, 8] - m[:, 7] * m[:, 2])
        det = det + m[:, 6] * (m[:, 1] * m[:, 5] - m[:, 4] * m[:, 2])
        self.em.volume[:] = np.abs(det)
        self.em._stale[:] = [False] * nb                # positions on the device are the authoritative ones here
        return h

    # -- chain synchronisation (mc_check_chain_synchronisation, mc_moves.F90:2217-2416) -------------------------
    def set_reference(self):
        """Remember the reference configuration (ref_hmatrix / ref_ljr of init.f90:90,106) from the host's current
        hmatrix / ljr.  The reference keeps rescaling ref_ljr in every volume move; since that preserves fractional
        coordinates, the fractional reference positions are kept instead and ref_ljr = hmatrix . s_ref at any time."""
        self._ref_h = np.array(self.em.hmatrix, dtype=np.float64)
        self._s_ref = np.einsum("bnd,bdk->bnk", self.em.ljr, np.linalg.inv(self._ref_h))

    def chain_synchronise(self):
        """Re-impose lattice 2 of every walker from its lattice 1 (rare: every latt_sync_int = 10^4 cycles, so this is
        host arithmetic over bulk transfers): cell 2 = ref cell 2 + (cell 1 - ref cell 1), molecule positions =
        reference fractional position + lattice 1's fractional displacement; then image vectors, full-box energies (on
        the device) and ls_mu."""
        if self.nlat != 2:
            raise ValueError("chain synchronisation needs two lattices")
        if not hasattr(self, "_s_ref"):
            raise ValueError("call set_reference() at the start of the run")
        em, nb, n = self.em, self.em.num_lattices, self.em.nwater
        h = self.sync_cells()                                              # current cells, host mirrors refreshed
        x = np.zeros((nb, n, 3))
        em._chk(self.L.mw_download_positions_range(1, nb, x.ctypes.data_as(_dp)))
        h1, h2_old, r1, r2 = h[0::2], h[1::2], self._ref_h[0::2], self._ref_h[1::2]
        h2_new = r2 + (h1 - r1)                                            # :2261-2277
        s1 = np.einsum("bnd,bdk->bnk", x[0::2], np.linalg.inv(h1))         # svect(:,1)
        ref_ljr2 = np.einsum("bnk,bkd->bnd", self._s_ref[1::2], h2_old)    # ref_ljr(:,2) as the volume moves left it
        ref_s2 = np.einsum("bnd,bdk->bnk", ref_ljr2, np.linalg.inv(h2_new))   # ref_svect(:,2) in the NEW cell 2
        s2 = ref_s2 + (s1 - self._s_ref[0::2])                             # :2339
        x2 = np.einsum("bnk,bkd->bnd", s2, h2_new)                         # :2341
        self._s_ref[1::2] = ref_s2                                         # ref_ljr(:,2) itself is unchanged: new fractional coords
        x[1::2] = x2
        em._chk(self.L.mw_upload_positions_range(1, nb, np.ascontiguousarray(x).ctypes.data_as(_dp)))
        for w in range(self.nwalkers):
            em.hmatrix[2 * w + 1] = h2_new[w]
            em.volume[2 * w + 1] = abs(np.linalg.det(h2_new[w]))
            em.compute_ivects(2 * w + 2)                                   # :2385-2390
            em._stale[2 * w + 1] = False
        em.ljr[:] = x
        e = em.model_energy_batch(1, nb)                                   # :2395-2396
        v = em.volume
        for w in range(self.nwalkers):                                     # :2398-2400
            mu = e[2 * w] + self.pressure * v[2 * w] - e[2 * w + 1] - self.pressure * v[2 * w + 1]
            mu = mu - self.dref                                            # leshift, :2401
            mu = mu * self.beta - n * math.log(v[2 * w] / v[2 * w + 1])
            self.set_state(w + 1, self.state(w + 1)["ls"], mu)

    def tables(self, walker):
        """(weight, histogram, unbiased_hist) of one walker."""
        nb = self.grid.nbins
        w, h, u = np.zeros(nb), np.zeros(nb), np.zeros(nb)
        self.em._chk(self.L.mw_sweep_get_tables(walker, w.ctypes.data_as(_dp), h.ctypes.data_as(_dp), u.ctypes.data_as(_dp)))
        return w, h, u

    def set_tables(self, walker, weight=None, histogram=None, unbiased_hist=None):
        args = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (weight, histogram, unbiased_hist)]
        self.em._chk(self.L.mw_sweep_set_tables(walker, *[None if a is None else a.ctypes.data_as(_dp) for a in args]))

    def switches(self, walker):
        v = ctypes.c_longlong(0)
        self.em._chk(self.L.mw_sweep_get_switches(walker, ctypes.byref(v)))
        return v.value

    def synchronise(self, comms=None, regauge=False):
        """The mpi_sync_int block of mc_cycle (mc_moves.F90:258-276) for a farm: every walker is a 'rank' of the
        reference; the increments of all walkers of this GPU are summed on the host, then over the GPUs by
        ``comms`` (WalkerComms, one all-reduce), and every walker receives the synchronised tables.

        ``regauge=False`` is the reference's arithmetic to the letter: a rank's weight increment includes the window
        minimum that mc_update_wl_bins subtracts after every update (:1682-1685), so R ranks subtract a minimum m
        R times where one shared table would lose it once.  The synchronised table then sits at about -(R-1) m, every
        rank adds that back on its next update, the sum comes out at +(R-1)^2 m, and so on: a uniform offset that
        grows by the factor -(R-1) per synchronisation once no bin is left at weight 0 (the reference carries a
        disabled "negative growth of eta" check for it, comms_mpi.f90:258-263).  Eight ranks live with that for a
        while; eight thousand walkers per GPU lose all precision within a few synchronisations.
        ``regauge=True`` (farm.run's opt-in ``regauge``): the device keeps each walker's accumulated minimum, the sum is taken
        over weight + that (the increments proper), and the window minimum is subtracted once from the result --
        one shared table in the reference's own gauge; identical to the reference for a single walker."""
        comms = self.local_comms() if comms is None else comms
        nb, nw = self.grid.nbins, self.nwalkers
        # a rank's contribution = sum over its walkers of (table - last), taken on the device (the tables stay there:
        # 3 x nbins numbers cross PCIe instead of 3 x nwalkers x nbins); WalkerComms is handed `last + that sum`
        last = [np.ascontiguousarray(a, dtype=np.float64) for a in (comms.eta_last_sync, comms.hist_last_sync, comms.uhist_last_sync)]
        sums = [np.zeros(nb) for _ in range(3)]
        self.em._chk(self.L.mw_sweep_reduce_tables(1, nw, *[a.ctypes.data_as(_dp) for a in last],
                                                    *[a.ctypes.data_as(_dp) for a in sums], int(bool(regauge)), int(bool(regauge))))
        summed = [last[t] + sums[t] for t in range(3)]
        comms.sync(summed[0], summed[1], summed[2])
        if regauge:
            summed[0] -= summed[0][self.grid.start_bin - 1:self.grid.end_bin].min()
            comms.set_weights(summed[0])
        self.em._chk(self.L.mw_sweep_broadcast_tables(1, nw, *[np.ascontiguousarray(a).ctypes.data_as(_dp) for a in summed]))
        return summed

    def local_comms(self):
        """The exchange object of a single-process farm (no process group: the 'all-reduce' is the sum over this
        GPU's walkers alone).  Holds the last-synchronised tables, so use either this or one WalkerComms throughout."""
        if getattr(self, "_local_comms", None) is None:
            from .comms import WalkerComms
            self._local_comms = WalkerComms(self.grid.nbins, samplerun=True)
        return self._local_comms

    # -- the table operations of mc_check_flatness (mc_moves.F90:1936-2185) for a farm ---------------------------
    def allreduce_hist(self, comms=None):
        """comms_allreduce_hist (comms_mpi.f90:461-494) with every walker a 'rank': the increments of this GPU's
        walkers since the last synchronisation are summed on the host, then over the GPUs; every walker receives
        the global histogram, which is also returned."""
        nb, nw = self.grid.nbins, self.nwalkers
        comms = self.local_comms() if comms is None else comms
        last = np.ascontiguousarray(comms.hist_last_sync, dtype=np.float64)
        inc = np.zeros(nb)
        self.em._chk(self.L.mw_sweep_reduce_tables(1, nw, None, last.ctypes.data_as(_dp), None,
                                                    None, inc.ctypes.data_as(_dp), None, 0, 0))
        total = last + inc                                 # this GPU's contribution, as WalkerComms expects it
        comms.allreduce_hist(total)
        self.em._chk(self.L.mw_sweep_broadcast_tables(1, nw, None, np.ascontiguousarray(total).ctypes.data_as(_dp), None))
        return total

    def reset_histogram(self, comms=None):
        """histogram = 0 on every walker, and the synchronisation baseline with it (comms_set_histogram,
        comms_mpi.f90:533-548; mc_moves.F90:1976-1977,2105-2106)."""
        nb, nw = self.grid.nbins, self.nwalkers
        self.em._chk(self.L.mw_sweep_broadcast_tables(1, nw, None, np.zeros(nb).ctypes.data_as(_dp), None))
        (self.local_comms() if comms is None else comms).set_histogram(np.zeros(nb))

    def shift_weights(self):
        """weight(:) -= weight(nbins/2 + 1) on every walker (mc_moves.F90:2062-2066)."""
        nb, nw = self.grid.nbins, self.nwalkers
        w = np.zeros((nw, nb))
        self.em._chk(self.L.mw_sweep_get_tables_range(1, nw, w.ctypes.data_as(_dp), None, None))
        w -= w[:, nb // 2][:, None]
        self.em._chk(self.L.mw_sweep_set_tables_range(1, nw, np.ascontiguousarray(w).ctypes.data_as(_dp), None, None))

    def initial_mu(self, walker):
        """ls_mu as main.f90:170-174 / mc_init (mc_moves.F90:857-862) form it."""
        if self.nlat == 1:
            return 0.0
        b = (walker - 1) * 2
        e, v, n = self.em.model_energy, self.em.volume, self.em.nwater
        mu = (e[b] + self.pressure * v[b]) - (e[b + 1] + self.pressure * v[b + 1])
        mu = mu - self.dref                                                # leshift, :860
        return mu * self.beta - n * math.log(v[b] / v[b + 1])

    def initial_mus(self):
        """initial_mu of every walker in one go (main.f90:170-174 / mc_moves.F90:857-862, vectorised)."""
        if self.nlat == 1:
            return np.zeros(self.nwalkers)
        e, v, n = self.em.model_energy, self.em.volume, self.em.nwater
        mu = (e[0::2] + self.pressure * v[0::2]) - (e[1::2] + self.pressure * v[1::2])
        mu = mu - self.dref                                                # leshift, :860
        lg = np.array([math.log(r) for r in (v[0::2] / v[1::2])])          # (libm's log, as initial_mu: numpy's may differ in the last bit)
        return mu * self.beta - n * lg

    def set_states(self, ls, ls_mu=None, first_walker=1):
        """Active lattice and order parameter of `len(ls)` consecutive walkers in two transfers (ls_mu None: initial_mus)."""
        ls = np.ascontiguousarray(np.broadcast_to(ls, (self.nwalkers - first_walker + 1,)) if np.ndim(ls) == 0 else ls, dtype=np.int32)
        mu = self.initial_mus()[first_walker - 1:first_walker - 1 + len(ls)] if ls_mu is None else ls_mu
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        self.em._chk(self.L.mw_sweep_set_states_range(first_walker, len(ls), ls.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), mu.ctypes.data_as(_dp)))

    def set_state(self, walker, ls=1, ls_mu=None):
        mu = self.initial_mu(walker) if ls_mu is None else ls_mu
        self.em._chk(self.L.mw_sweep_set_state(walker, ls, ctypes.c_double(mu)))

    def state(self, walker):
        ls, mu, acc = ctypes.c_int(0), ctypes.c_double(0.0), ctypes.c_longlong(0)
        e = (ctypes.c_double * 2)()
        self.em._chk(self.L.mw_sweep_get_state(walker, ctypes.byref(ls), ctypes.byref(mu), e, ctypes.byref(acc)))
        return dict(ls=ls.value, ls_mu=mu.value, model_energy=[e[k] for k in range(self.nlat)], accepted=acc.value)

    def sweep(self, nmoves, seed, move0=0, first_walker=1, count=None, log=False):
        count = self.nwalkers - first_walker + 1 if count is None else count
        if log:
            out = np.zeros((count, nmoves, 8))
            self.em._chk(self.L.mw_sweep_translation(first_walker, count, nmoves, ctypes.c_ulonglong(seed),
                                                     ctypes.c_ulonglong(move0), out.ctypes.data_as(_dp)))
            return out
        self.em._chk(self.L.mw_sweep_translation(first_walker, count, nmoves, ctypes.c_ulonglong(seed),
                                                 ctypes.c_ulonglong(move0), None))
        return None

    def sweep_launch(self, nmoves, seed, move0=0, first_walker=1, count=None):
        count = self.nwalkers - first_walker + 1 if count is None else count
        self.em._chk(self.L.mw_sweep_translation_launch(first_walker, count, nmoves, ctypes.c_ulonglong(seed),
                                                        ctypes.c_ulonglong(move0), 0))

    def positions(self, ils):
        x = np.zeros((self.em.nwater, 3))
        self.em._chk(self.L.mw_download_positions(ils, x.ctypes.data_as(_dp)))
        return x
