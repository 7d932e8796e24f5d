"""ORACLE / TEST INFRASTRUCTURE -- not product code.

ctypes loaders for
  * ``COracle``  -- oracle/libmw_oracle.so, the C restatement (mw_oracle.c), and
  * ``RefOracle`` -- oracle/_ref/libmw_ref.so, the reference's own Fortran hot path
    compiled from /root/reference by oracle/Makefile (build container only; the
    prebuilt .so travels to the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (mc_water_ls_mw_amd) never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
C_LIB = os.path.join(HERE, "libmw_oracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libmw_ref.so")
MAXNEIGH = 50          # molint.F90:79
MAX_IVECT = 4096

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_lp = ctypes.POINTER(ctypes.c_longlong)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def build(ref=True):
    """Compile the C restatement and, when /root/reference is present (build container only), everything
    oracle/Makefile derives from the reference: libmw_ref.so, the drop-in driver pair and the whole-program
    builds (mc_water_ref, _ref_scrub, _ref_rng, _hip, _hip_rccl).  The last two link libmw_hip.so / libmw_comms.so: build those first."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference"):
        with open(os.devnull, "w") as null:     # flang prints literal-widening warnings for the reference sources
            subprocess.check_call(["make", "-s", "-C", HERE, "ref", "dropin", "fullprog"], stderr=null)


class COracle:
    """The C restatement.  Stateless functions over explicit arrays."""

    def __init__(self):
        if not os.path.exists(C_LIB):
            build(ref=False)
        L = ctypes.CDLL(C_LIB)
        L.mwo_model_energy.restype = ctypes.c_double
        L.mwo_local_energy.restype = ctypes.c_double
        self.L = L

    def constants(self):
        out = np.zeros(8)
        self.L.mwo_constants(_d(out))
        return out

    def ivects(self, h):
        h = np.ascontiguousarray(h, dtype=np.float64)
        iv = np.zeros((MAX_IVECT, 3))
        n = self.L.mwo_compute_ivects(_d(h), _d(iv), MAX_IVECT)
        if n < 0:
            raise RuntimeError("too many image vectors")
        return np.ascontiguousarray(iv[:n])

    def neighbours(self, xyz, ivect, maxneigh=MAXNEIGH):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        n = len(xyz)
        nn = np.zeros(n, dtype=np.int32)
        jn = np.zeros((n, maxneigh), dtype=np.int32)
        vn = np.zeros((n, maxneigh), dtype=np.int32)
        rc = self.L.mwo_compute_neighbours(n, _d(xyz), _d(ivect), len(ivect), maxneigh, _i(nn), _i(jn), _i(vn))
        if rc < 0:
            raise RuntimeError("neighbour list overflow (nn > maxneigh)")
        return nn, jn, vn

    def model_energy(self, xyz, ivect, nn, jn, vn, counts=False):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        c = np.zeros(2, dtype=np.int64)
        e = self.L.mwo_model_energy(len(xyz), _d(xyz), _d(ivect), jn.shape[1], _i(nn), _i(jn), _i(vn),
                                    c.ctypes.data_as(_lp))
        return (e, c) if counts else e

    def local_energy(self, imol, xyz, ivect, nn, jn, vn, counts=False):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        c = np.zeros(2, dtype=np.int64)
        e = self.L.mwo_local_energy(int(imol), len(xyz), _d(xyz), _d(ivect), jn.shape[1], _i(nn), _i(jn), _i(vn),
                                    c.ctypes.data_as(_lp))
        return (e, c) if counts else e

    def local_energy_all(self, xyz, ivect, nn, jn, vn, counts=False):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        e = np.zeros(len(xyz))
        c = np.zeros(2, dtype=np.int64)
        self.L.mwo_local_energy_all(len(xyz), _d(xyz), _d(ivect), jn.shape[1], _i(nn), _i(jn), _i(vn), _d(e),
                                    c.ctypes.data_as(_lp))
        return (e, c) if counts else e

    def trial_moves(self, imol, trial, xyz, ivect, nn, jn, vn):
        xyz = np.array(xyz, dtype=np.float64, order="C")
        imol = np.ascontiguousarray(imol, dtype=np.int32)
        trial = np.ascontiguousarray(trial, dtype=np.float64)
        eo = np.zeros(len(imol))
        en = np.zeros(len(imol))
        self.L.mwo_trial_moves(len(imol), _i(imol), _d(trial), len(xyz), _d(xyz), _d(ivect), jn.shape[1],
                               _i(nn), _i(jn), _i(vn), _d(eo), _d(en))
        return eo, en


class _Eta(ctypes.Structure):
    _fields_ = [("nbins", ctypes.c_int), ("eta_interp", ctypes.c_int), ("start_bin", ctypes.c_int), ("end_bin", ctypes.c_int),
                ("r_pos", ctypes.c_double), ("a_pos", ctypes.c_double), ("r_neg", ctypes.c_double), ("a_neg", ctypes.c_double),
                ("mu_lo", ctypes.c_double), ("mu_hi", ctypes.c_double),
                ("weight", _dp), ("mu_bin", _dp), ("binwidth", _dp)]


class _CycleOpts(ctypes.Structure):
    _fields_ = [("record", ctypes.c_int), ("samplerun", ctypes.c_int), ("always_switch", ctypes.c_int), ("npt", ctypes.c_int),
                ("av_binwidth", ctypes.c_double), ("wl_factor", ctypes.c_double), ("log_unbiased_norm", ctypes.c_double),
                ("pressure", ctypes.c_double), ("volume", ctypes.c_double * 2)]


class SweepOracle:
    """mc_water_translation restated on the CPU (oracle/mw_oracle.c, mwo_sweep_translation)."""

    def __init__(self):
        self.C = COracle()
        self.L = self.C.L
        self.L.mwo_eta_weight.restype = ctypes.c_double

    def uniforms(self, seed, walker, move):
        u = np.zeros(6)
        self.L.mwo_move_uniforms(ctypes.c_uint64(seed), ctypes.c_uint32(walker), ctypes.c_uint64(move), _d(u))
        return u

    def mu_grid(self, nbins, mu_min, mu_max):
        mb, bw, gp = np.zeros(nbins), np.zeros(nbins), np.zeros(4)
        self.L.mwo_mu_grid(nbins, ctypes.c_double(mu_min), ctypes.c_double(mu_max), _d(mb), _d(bw), _d(gp))
        return mb, bw, gp

    def _eta(self, grid, weight, eta_interp):
        self._keep = (np.ascontiguousarray(weight, dtype=np.float64), np.ascontiguousarray(grid.mu_bin),
                      np.ascontiguousarray(grid.binwidth))
        return _Eta(grid.nbins, int(eta_interp), grid.start_bin, grid.end_bin, grid.r_pos, grid.a_pos, grid.r_neg,
                    grid.a_neg, grid.my_mu_min, grid.my_mu_max, _d(self._keep[0]), _d(self._keep[1]), _d(self._keep[2]))

    def eta_weight(self, grid, weight, eta_interp, mu):
        e = self._eta(grid, weight, eta_interp)
        return self.L.mwo_eta_weight(ctypes.byref(e), ctypes.c_double(mu))

    def mu_to_bin(self, grid, mu):
        e = self._eta(grid, np.zeros(grid.nbins), True)
        return self.L.mwo_mu_to_bin(ctypes.byref(e), ctypes.c_double(mu))

    def sweep(self, nmoves, seed, walker, move0, hs, xs, beta, max_trans, grid=None, weight=None, eta_interp=True,
              ls=1, ls_mu=0.0, model_energy=None, maxneigh=MAXNEIGH, lists=None):
        """hs, xs: lists (1 or 2 lattices).  Returns dict(xyz, ls, ls_mu, model_energy, accepted, log, lists)."""
        nlat, n = len(xs), len(xs[0])
        xyz = np.ascontiguousarray(np.stack(xs), dtype=np.float64).copy()
        h = np.ascontiguousarray(np.stack(hs), dtype=np.float64)
        ivs = [self.C.ivects(hh) for hh in hs]
        ivstride = max(len(v) for v in ivs)
        iv = np.zeros((nlat, ivstride, 3))
        for l, v in enumerate(ivs):
            iv[l, :len(v)] = v
        if lists is None:
            lists = [self.C.neighbours(xs[l], ivs[l], maxneigh) for l in range(nlat)]
        nn = np.ascontiguousarray(np.stack([t[0] for t in lists]))
        jn = np.ascontiguousarray(np.stack([t[1] for t in lists]))
        vn = np.ascontiguousarray(np.stack([t[2] for t in lists]))
        if model_energy is None:
            model_energy = [self.C.model_energy(xs[l], ivs[l], *lists[l]) for l in range(nlat)]
        me = np.ascontiguousarray(model_energy, dtype=np.float64).copy()
        eta = self._eta(grid, weight if weight is not None else np.zeros(grid.nbins), eta_interp) if grid is not None \
            else _Eta(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, None, None)
        lsv, mu, acc = ctypes.c_int(ls), ctypes.c_double(ls_mu), ctypes.c_longlong(0)
        log = np.zeros((nmoves, 8))
        self.L.mwo_sweep_translation(nmoves, ctypes.c_uint64(seed), ctypes.c_uint32(walker), ctypes.c_uint64(move0),
                                     nlat, n, _d(xyz), _d(h), _d(iv), ivstride, maxneigh, _i(nn), _i(jn), _i(vn),
                                     ctypes.c_double(beta), ctypes.c_double(max_trans), ctypes.byref(eta),
                                     ctypes.byref(lsv), ctypes.byref(mu), _d(me), ctypes.byref(acc), _d(log))
        return dict(xyz=xyz, ls=lsv.value, ls_mu=mu.value, model_energy=me, accepted=acc.value, log=log, lists=lists)

    def full(self, st, nmoves, seed, walker, move0, transP, dv_max, beta, max_trans, grid, weight, histogram,
             unbiased_hist, eta_interp=True, record=True, samplerun=True, always_switch=True, npt=True, wl_factor=0.0,
             log_unbiased_norm=0.0, pressure=0.0):
        """mwo_sweep_full on a FullSweepState (updated in place); weight/histogram/unbiased_hist arrays are updated in place."""
        nn = np.ascontiguousarray(np.stack([t[0] for t in st.lists]))
        jn = np.ascontiguousarray(np.stack([t[1] for t in st.lists]))
        vn = np.ascontiguousarray(np.stack([t[2] for t in st.lists]))
        mb, bw = np.ascontiguousarray(grid.mu_bin), np.ascontiguousarray(grid.binwidth)
        eta = _Eta(grid.nbins, int(eta_interp), grid.start_bin, grid.end_bin, grid.r_pos, grid.a_pos, grid.r_neg,
                   grid.a_neg, grid.my_mu_min, grid.my_mu_max, _d(weight), _d(mb), _d(bw))
        vol = (ctypes.c_double * 2)(*[float(st.volume[l]) if l < st.nlat else 0.0 for l in range(2)])
        opt = _CycleOpts(int(record), int(samplerun), int(always_switch), int(npt), grid.av_binwidth, wl_factor,
                         log_unbiased_norm, pressure, vol)
        lsv, mu = ctypes.c_int(st.ls), ctypes.c_double(st.ls_mu)
        acc, sw = ctypes.c_longlong(0), ctypes.c_longlong(0)
        nvol = np.zeros(2, dtype=np.int64)
        log = np.zeros((nmoves, 8))
        self.L.mwo_sweep_full_ref(nmoves, ctypes.c_uint64(seed), ctypes.c_uint32(walker), ctypes.c_uint64(move0),
                              ctypes.c_double(transP), ctypes.c_double(dv_max), st.nlat, st.n, _d(st.xyz), _d(st.ref_xyz), _d(st.h),
                              _d(st.volume), _d(st.ivect), st.ivstride, _i(st.nivect), st.maxneigh, _i(nn), _i(jn), _i(vn),
                              ctypes.c_double(beta), ctypes.c_double(max_trans), ctypes.byref(eta), ctypes.byref(opt),
                              _d(histogram), _d(unbiased_hist), _d(weight), ctypes.byref(lsv), ctypes.byref(mu),
                              _d(st.model_energy), ctypes.byref(acc), ctypes.byref(sw), nn.ctypes.data_as(_lp) if False else nvol.ctypes.data_as(_lp), _d(log))
        st.ls, st.ls_mu = lsv.value, mu.value
        st.accepted += acc.value
        st.switches += sw.value
        st.nvol += nvol
        return log

    def chain_sync(self, st, beta, pressure):
        """mwo_chain_sync on a two-lattice FullSweepState (in place)."""
        nn = np.ascontiguousarray(np.stack([t[0] for t in st.lists]))
        jn = np.ascontiguousarray(np.stack([t[1] for t in st.lists]))
        vn = np.ascontiguousarray(np.stack([t[2] for t in st.lists]))
        mu = ctypes.c_double(st.ls_mu)
        rc = self.L.mwo_chain_sync(st.n, _d(st.xyz), _d(st.ref_xyz), _d(st.h), _d(st.ref_h), _d(st.volume), _d(st.ivect),
                                   st.ivstride, _i(st.nivect), st.maxneigh, _i(nn), _i(jn), _i(vn), ctypes.c_double(beta),
                                   ctypes.c_double(pressure), ctypes.byref(mu), _d(st.model_energy))
        assert rc == 0
        st.ls_mu = mu.value

    # run options the reference keeps in module variables (they apply to every later call until changed)
    def set_leshift(self, ref1=0.0, ref2=0.0):
        self.L.mwo_set_leshift(ctypes.c_double(ref1), ctypes.c_double(ref2))

    def set_minu(self, on):
        """the reference compiled with -DMINU (mc_moves.F90:1119-1140,1385-1401): accepted moves end in the lattice of lower enthalpy"""
        self.L.mwo_set_minu(int(on))

    def set_swetnam(self, on, alpha=1.0, orig_wl_factor=0.0, mu_min=0.0, mu_max=0.0, sumhist=0.0):
        self.L.mwo_set_swetnam(int(on), ctypes.c_double(alpha), ctypes.c_double(orig_wl_factor), ctypes.c_double(mu_min),
                               ctypes.c_double(mu_max), ctypes.c_double(sumhist))

    def get_swetnam(self):
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self.L.mwo_get_swetnam(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def set_dd(self, on, eq_cycles=0, in_window=False):
        self.L.mwo_set_dd(int(on), int(eq_cycles), int(in_window))

    def get_dd(self):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        self.L.mwo_get_dd(ctypes.byref(a), ctypes.byref(b))
        return bool(a.value), bool(b.value)

    def cycle(self, nmoves, seed, walker, move0, hs, xs, beta, max_trans, grid, weight, histogram, unbiased_hist,
              eta_interp=True, ls=1, ls_mu=0.0, model_energy=None, lists=None, record=True, samplerun=True,
              always_switch=True, npt=False, wl_factor=0.0, log_unbiased_norm=0.0, pressure=0.0, maxneigh=MAXNEIGH):
        """mwo_sweep_cycle: translations + mc_update_wl_bins + lattice-switch attempts.  weight / histogram /
        unbiased_hist are copied; the updated tables come back in the result."""
        nlat, n = len(xs), len(xs[0])
        xyz = np.ascontiguousarray(np.stack(xs), dtype=np.float64).copy()
        h = np.ascontiguousarray(np.stack(hs), dtype=np.float64)
        ivs = [self.C.ivects(hh) for hh in hs]
        ivstride = max(len(v) for v in ivs)
        iv = np.zeros((nlat, ivstride, 3))
        for l, v in enumerate(ivs):
            iv[l, :len(v)] = v
        if lists is None:
            lists = [self.C.neighbours(xs[l], ivs[l], maxneigh) for l in range(nlat)]
        nn = np.ascontiguousarray(np.stack([t[0] for t in lists]))
        jn = np.ascontiguousarray(np.stack([t[1] for t in lists]))
        vn = np.ascontiguousarray(np.stack([t[2] for t in lists]))
        if model_energy is None:
            model_energy = [self.C.model_energy(xs[l], ivs[l], *lists[l]) for l in range(nlat)]
        me = np.ascontiguousarray(model_energy, dtype=np.float64).copy()
        w = np.ascontiguousarray(weight, dtype=np.float64).copy()
        hi = np.ascontiguousarray(histogram, dtype=np.float64).copy()
        uh = np.ascontiguousarray(unbiased_hist, dtype=np.float64).copy()
        mb, bw = np.ascontiguousarray(grid.mu_bin), np.ascontiguousarray(grid.binwidth)
        eta = _Eta(grid.nbins, int(eta_interp), grid.start_bin, grid.end_bin, grid.r_pos, grid.a_pos, grid.r_neg,
                   grid.a_neg, grid.my_mu_min, grid.my_mu_max, _d(w), _d(mb), _d(bw))      # eta reads the live weights
        vol = (ctypes.c_double * 2)(*[abs(np.linalg.det(hh)) for hh in hs])
        opt = _CycleOpts(int(record), int(samplerun), int(always_switch), int(npt), grid.av_binwidth, wl_factor,
                         log_unbiased_norm, pressure, vol)
        lsv, mu, acc, sw = ctypes.c_int(ls), ctypes.c_double(ls_mu), ctypes.c_longlong(0), ctypes.c_longlong(0)
        log = np.zeros((nmoves, 8))
        self.L.mwo_sweep_cycle(nmoves, ctypes.c_uint64(seed), ctypes.c_uint32(walker), ctypes.c_uint64(move0),
                               nlat, n, _d(xyz), _d(h), _d(iv), ivstride, maxneigh, _i(nn), _i(jn), _i(vn),
                               ctypes.c_double(beta), ctypes.c_double(max_trans), ctypes.byref(eta), ctypes.byref(opt),
                               _d(hi), _d(uh), _d(w), ctypes.byref(lsv), ctypes.byref(mu), _d(me),
                               ctypes.byref(acc), ctypes.byref(sw), _d(log))
        return dict(xyz=xyz, ls=lsv.value, ls_mu=mu.value, model_energy=me, accepted=acc.value, switches=sw.value,
                    log=log, lists=lists, weight=w, histogram=hi, unbiased_hist=uh)


class FullSweepState:
    """Mutable walker state for SweepOracle.full (cells change under volume moves)."""

    def __init__(self, C, hs, xs, maxneigh=MAXNEIGH):
        self.nlat, self.n, self.maxneigh = len(xs), len(xs[0]), maxneigh
        self.xyz = np.ascontiguousarray(np.stack(xs), dtype=np.float64).copy()
        self.h = np.ascontiguousarray(np.stack(hs), dtype=np.float64).copy()
        self.volume = np.array([abs(np.linalg.det(hh)) for hh in hs])
        self.ivstride = 64
        self.ivect = np.zeros((self.nlat, self.ivstride, 3))
        self.nivect = np.zeros(self.nlat, dtype=np.int32)
        for l in range(self.nlat):
            v = C.ivects(self.h[l])
            self.ivect[l, :len(v)] = v
            self.nivect[l] = len(v)
        self.rebuild_lists(C)
        self.model_energy = np.array([C.model_energy(self.xyz[l], self.iv(l), *self.lists[l]) for l in range(self.nlat)])
        self.ls, self.ls_mu = 1, 0.0
        self.accepted = self.switches = 0
        self.nvol = np.zeros(2, dtype=np.int64)
        self.ref_xyz = self.xyz.copy()             # ref_ljr (init.f90:106), rescaled by volume moves
        self.ref_h = self.h.copy()                 # ref_hmatrix (init.f90:90), never changes

    def iv(self, l):
        return np.ascontiguousarray(self.ivect[l, :self.nivect[l]])

    def rebuild_lists(self, C):
        self.lists = [C.neighbours(self.xyz[l], self.iv(l), self.maxneigh) for l in range(self.nlat)]


class RefOracle:
    """The reference's compiled Fortran (module energy), one system at a time.

    The Fortran modules hold global state, so this is a singleton per process:
    ``load(h_list, xyz_list)`` calls energy_init for ``len(h_list)`` lattices.
    """

    _lib = None

    @staticmethod
    def available():
        return os.path.exists(REF_LIB)

    def __init__(self):
        if RefOracle._lib is None:
            if not os.path.exists(REF_LIB):
                raise FileNotFoundError(REF_LIB + " (run `make -C oracle ref` in the build container)")
            L = ctypes.CDLL(REF_LIB)
            L.ref_model_energy.restype = ctypes.c_double
            L.ref_local_energy.restype = ctypes.c_double
            L.ref_time_model_energy.restype = ctypes.c_double
            L.ref_time_local_energy.restype = ctypes.c_double
            RefOracle._lib = L
            RefOracle._loaded = False
        self.L = RefOracle._lib
        self.n = 0
        self.nlat = 0

    def load(self, h_list, xyz_list):
        if RefOracle._loaded:
            self.L.ref_finalize()
        h = np.ascontiguousarray(np.stack(h_list), dtype=np.float64)
        x = np.ascontiguousarray(np.stack(xyz_list), dtype=np.float64)
        self.nlat, self.n = x.shape[0], x.shape[1]
        self.L.ref_init(self.n, self.nlat, _d(h), _d(x))
        RefOracle._loaded = True

    def constants(self):
        out = np.zeros(8)
        self.L.ref_constants(_d(out))
        return out

    def set_positions(self, ils, xyz):
        x = np.ascontiguousarray(xyz, dtype=np.float64)
        self.L.ref_set_positions(ils, _d(x))

    def set_position(self, ils, imol, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        self.L.ref_set_position(ils, int(imol), _d(r))

    def set_cell(self, ils, h):
        h = np.ascontiguousarray(h, dtype=np.float64)
        self.L.ref_set_cell(ils, _d(h))

    def ivects(self, ils):
        iv = np.zeros((MAX_IVECT, 3))
        n = self.L.ref_compute_ivects(ils, _d(iv), MAX_IVECT)
        return np.ascontiguousarray(iv[:n])

    def compute_neighbours(self, ils):
        self.L.ref_compute_neighbours(ils)

    def neighbours(self, ils):
        nn = np.zeros(self.n, dtype=np.int32)
        jn = np.zeros((self.n, MAXNEIGH), dtype=np.int32)
        vn = np.zeros((self.n, MAXNEIGH), dtype=np.int32)
        self.L.ref_get_neighbours(ils, _i(nn), _i(jn), _i(vn))
        return nn, jn, vn

    def model_energy(self, ils):
        return self.L.ref_model_energy(ils)

    def local_energy(self, imol, ils):
        return self.L.ref_local_energy(int(imol), ils)

    def local_energy_all(self, ils):
        e = np.zeros(self.n)
        self.L.ref_local_energy_all(ils, _d(e))
        return e

    def trial_moves(self, ils, imol, trial):
        imol = np.ascontiguousarray(imol, dtype=np.int32)
        trial = np.ascontiguousarray(trial, dtype=np.float64)
        eo = np.zeros(len(imol))
        en = np.zeros(len(imol))
        self.L.ref_trial_moves(ils, len(imol), _i(imol), _d(trial), _d(eo), _d(en))
        return eo, en

    def time_model_energy(self, ils, nrep):
        return self.L.ref_time_model_energy(ils, nrep)

    def time_local_energy(self, ils, nrep, imol):
        imol = np.ascontiguousarray(imol, dtype=np.int32)
        return self.L.ref_time_local_energy(ils, nrep, len(imol), _i(imol))
