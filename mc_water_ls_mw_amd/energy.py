"""Host-side mirror of the reference's Fortran ``module energy`` (molint.F90:10-37)
over the C ABI of ``libmw_hip.so`` (include/mw_energy.h).

:class:`EnergyModule` keeps the reference's public names and argument meaning --
``energy_init``, ``energy_deinit``, ``compute_ivects(ils)``,
``compute_neighbours(ils)``, ``compute_model_energy(ils)`` (result in
``model_energy[ils-1]``), ``compute_local_real_energy(imol, ils)`` -- with 1-based
``ils`` / ``imol`` exactly as the Fortran callers pass them, and it holds the
implicit inputs the Fortran module reads from ``module model``
(``ljr``, ``hmatrix``, ``volume``; data_structures.f90:39-46) as host arrays the
caller mutates freely, like mc_moves.F90 does.  It is the same thin layer as
the ISO_C_BINDING module in ``fortran/energy_hip.F90`` and follows the same
host<->device coherence protocol (DESIGN.md "Coherence").

There is no CPU path: if ``libmw_hip.so`` is missing or no gfx950 device is
usable, construction / ``energy_init`` raises.
"""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libmw_hip.so")
MAXNEIGH = 50  # molint.F90:79

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_llp = ctypes.POINTER(ctypes.c_longlong)
_fp = ctypes.POINTER(ctypes.c_float)

#: every symbol include/mw_energy.h declares (tests check the .so exports them all)
ABI_SYMBOLS = (
    "mw_init", "mw_finalize", "mw_is_initialised", "mw_last_error", "mw_constants",
    "mw_set_cell", "mw_get_ivects", "mw_upload_positions", "mw_download_positions", "mw_patch_position",
    "mw_upload_positions_range", "mw_download_positions_range", "mw_set_cells_range",
    "mw_build_neighbours", "mw_build_neighbours_batch", "mw_get_neighbours",
    "mw_model_energy", "mw_model_energy_of", "mw_model_energy_batch", "mw_model_energy_counts",
    "mw_model_energy_counts_total", "mw_neighbour_total",
    "mw_local_energy", "mw_local_energy_patched", "mw_local_energy_post", "mw_local_energy_collect",
    "mw_local_energy_batch", "mw_delta_energy_batch",
    "mw_moves_upload", "mw_moves_launch", "mw_moves_fetch", "mw_moves_counts",
    "mw_model_energy_launch", "mw_step_launch", "mw_model_energy_fetch", "mw_build_neighbours_launch", "mw_sync",
    "mw_timer_start", "mw_timer_stop", "mw_timer_elapsed_ms", "mw_device_info",
    "mw_sweep_configure", "mw_set_model_energy", "mw_sweep_set_state", "mw_sweep_set_states_range", "mw_sweep_get_state",
    "mw_sweep_translation", "mw_sweep_translation_launch", "mw_sweep_lds_bytes", "mw_sweep_last_launch",
    "mw_sweep_options", "mw_sweep_get_tables", "mw_sweep_set_tables", "mw_sweep_get_switches", "mw_sweep_get_shifts_range",
    "mw_sweep_reduce_tables", "mw_sweep_broadcast_tables",
    "mw_sweep_get_tables_range", "mw_sweep_set_tables_range",
    "mw_sweep_moves", "mw_sweep_get_volume_moves", "mw_sweep_sync_cells", "mw_sweep_check_flags",
    "mw_sweep_leshift", "mw_sweep_minu", "mw_sweep_swetnam", "mw_sweep_dd", "mw_sweep_windows", "mw_sweep_set_factors", "mw_sweep_get_factors", "mw_sweep_steps", "mw_sweep_get_counters",
)


class MwError(RuntimeError):
    """A C-ABI call returned nonzero; the text is mw_last_error()."""


_lib = None


def load_library(path=LIB_PATH):
    """dlopen libmw_hip.so.  Raises if it has not been built -- there is no fallback.
    (MW_HIP_LIB: another build of the same library, for A/B measurements inside one GPU session.)"""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MW_HIP_LIB", path)
    if not os.path.exists(path):
        raise MwError(f"{path} not found: build it with `python -m mc_water_ls_mw_amd.build` "
                      "(the mW engine has no CPU fallback)")
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(path)
    L.mw_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own
    libamdhip64.so (same SONAME as /opt/rocm's); if libmw_hip.so were loaded first it
    would bind the system copy, a later ``import torch`` would bring in a second
    runtime, and whichever initialises second sees no device.  So when torch is
    installed, its copy is loaded (globally) before ours, whatever the import order.
    A process without torch -- the Fortran host -- simply uses /opt/rocm's."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class EnergyModule:
    """One process-wide engine context (like the Fortran module, it is a singleton)."""

    def __init__(self, nwater, num_lattices=1, maxneigh=MAXNEIGH, device=0):
        self.L = load_library()
        self.nwater = int(nwater)
        self.num_lattices = int(num_lattices)
        self.maxneigh = int(maxneigh)
        self.device = int(device)
        # module model (data_structures.f90): the host owns these and mutates them at will
        self.ljr = np.zeros((self.num_lattices, self.nwater, 3), dtype=np.float64)   # ljr(:,1,imol,ils)
        self.hmatrix = np.zeros((self.num_lattices, 3, 3), dtype=np.float64)          # [ils][k] = hmatrix(:,k,ils)
        self.volume = np.zeros(self.num_lattices, dtype=np.float64)
        # module energy public variables (molint.F90:33-37)
        self.model_energy = np.zeros(self.num_lattices, dtype=np.float64)
        self.nivect = np.zeros(self.num_lattices, dtype=np.int32)
        self._last_imol = [0] * self.num_lattices    # molecule queried last per lattice (coherence protocol)
        self._stale = [True] * self.num_lattices     # set by compute_ivects: bulk position change pending
        self._live = False
        self.warnings = []
        # the single call is latency-bound: everything that can be prepared once is
        self._local = self.L.mw_local_energy_patched
        self._e_out = [ctypes.c_double(0.0) for _ in range(self.num_lattices)]    # per lattice: distinct lattices may be
        self._e_ref = [ctypes.byref(v) for v in self._e_out]                      # queried from distinct host threads
        self._ljr_seen, self._ljr_addr = None, 0

    # -- plumbing ------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise MwError(self.L.mw_last_error().decode())

    def _ils(self, ils):
        if not (1 <= ils <= self.num_lattices):
            raise MwError(f"lattice index {ils} outside 1..{self.num_lattices}")
        return ils - 1

    # -- energy_init / energy_deinit (molint.F90:91-171) ------------------------------
    def energy_init(self):
        """Allocate, set volume(ils) = |det h|, then ivects + neighbours + energy per lattice."""
        self.setup_boxes()                                                          # :125,134 + the mirrored positions
        # :147-148 for every lattice: all lists in one batch of launches, then all energies in one launch
        self.build_neighbours_batch(1, self.num_lattices)
        self.model_energy_batch(1, self.num_lattices)

    def setup_boxes(self):
        """mw_init + volume(ils) = |det h| + compute_ivects + the mirrored positions of EVERY box, in a handful of
        transfers whatever the number of boxes (farms hold thousands)."""
        self._chk(self.L.mw_init(self.device, self.nwater, self.num_lattices, self.maxneigh))
        self._live = True
        self.volume[:] = np.abs(np.linalg.det(self.hmatrix))                        # :125
        self.set_cells_range(1, self.num_lattices)                                  # :134
        self.upload_range(1, self.num_lattices)

    def set_cells_range(self, first_ils=1, count=None):
        """compute_ivects for ``count`` consecutive lattices in one call (one transfer per device array)."""
        count = self.num_lattices - first_ils + 1 if count is None else count
        self._ils(first_ils), self._ils(first_ils + count - 1)
        h = np.ascontiguousarray(self.hmatrix[first_ils - 1:first_ils - 1 + count], dtype=np.float64)
        n = np.zeros(count, dtype=np.int32)
        self._chk(self.L.mw_set_cells_range(first_ils, count, _d(h), _i(n)))
        self.nivect[first_ils - 1:first_ils - 1 + count] = n
        for b in range(first_ils - 1, first_ils - 1 + count):
            self._stale[b] = True

    def upload_range(self, first_ils=1, count=None):
        """Mirror ljr of ``count`` consecutive lattices in one transfer."""
        count = self.num_lattices - first_ils + 1 if count is None else count
        self._ils(first_ils), self._ils(first_ils + count - 1)
        x = np.ascontiguousarray(self.ljr[first_ils - 1:first_ils - 1 + count], dtype=np.float64)
        self._chk(self.L.mw_upload_positions_range(first_ils, count, _d(x)))
        for b in range(first_ils - 1, first_ils - 1 + count):
            self._last_imol[b] = 0
            self._stale[b] = False

    def energy_deinit(self):
        if self._live:
            self._chk(self.L.mw_finalize())
            self._live = False

    close = energy_deinit

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.energy_deinit()

    # -- compute_ivects (molint.F90:174-217) ------------------------------------------
    def compute_ivects(self, ils):
        b = self._ils(ils)
        h = np.ascontiguousarray(self.hmatrix[b], dtype=np.float64)
        n = ctypes.c_int(0)
        self._chk(self.L.mw_set_cell(ils, _d(h), ctypes.byref(n)))
        self.nivect[b] = n.value
        # a cell change comes with a bulk position change -- also when a volume move is REJECTED: the host
        # rescales every position back and calls nothing but compute_ivects (mc_moves.F90:1410-1514)
        self._stale[b] = True

    def ivect(self, ils):
        self._ils(ils)
        n = ctypes.c_int(0)
        self._chk(self.L.mw_get_ivects(ils, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, 3))
        self._chk(self.L.mw_get_ivects(ils, _d(out), n.value, ctypes.byref(n)))
        return out

    # -- compute_neighbours (molint.F90:501-559) --------------------------------------
    def _upload(self, ils):
        x = np.ascontiguousarray(self.ljr[ils - 1], dtype=np.float64)
        self._chk(self.L.mw_upload_positions(ils, _d(x)))
        self._last_imol[ils - 1] = 0
        self._stale[ils - 1] = False

    def compute_neighbours(self, ils):
        self._ils(ils)
        self.compute_ivects(ils)                                                    # :518
        self._upload(ils)
        mn, mx = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(self.L.mw_build_neighbours(ils, ctypes.byref(mn), ctypes.byref(mx)))
        if mn.value < 16:                                                           # :552-554
            nn, _, _ = self.neighbours(ils)
            for imol in np.nonzero(nn < 16)[0]:
                msg = "WARNING: Molecule %5d has only %5d neighbours" % (imol + 1, nn[imol])
                self.warnings.append(msg)
                print(msg, file=sys.stderr)
        return mn.value, mx.value

    def neighbours(self, ils):
        """(nn, jn, vn) in the reference's layout and 1-based numbering."""
        self._ils(ils)
        nn = np.zeros(self.nwater, dtype=np.int32)
        jn = np.zeros((self.nwater, self.maxneigh), dtype=np.int32)
        vn = np.zeros((self.nwater, self.maxneigh), dtype=np.int32)
        self._chk(self.L.mw_get_neighbours(ils, _i(nn), _i(jn), _i(vn)))
        return nn, jn, vn

    # -- compute_model_energy (molint.F90:407-499) ------------------------------------
    def compute_model_energy(self, ils):
        """Full-box energy of lattice ils -> model_energy[ils-1].  All positions are
        re-mirrored first: callers reach this after bulk changes (volume move
        mc_moves.F90:1314-1357, chain sync :2331-2396, restart :842-852)."""
        b = self._ils(ils)
        self._upload(ils)
        e = ctypes.c_double(0.0)
        self._chk(self.L.mw_model_energy(ils, ctypes.byref(e)))
        self.model_energy[b] = e.value
        return e.value

    def model_energy_counts(self, ils):
        p, t = ctypes.c_longlong(0), ctypes.c_longlong(0)
        self._chk(self.L.mw_model_energy_counts(ils, ctypes.byref(p), ctypes.byref(t)))
        return p.value, t.value

    def model_energy_counts_total(self, first_ils=1, count=None):
        count = self.num_lattices - first_ils + 1 if count is None else count
        p, t = ctypes.c_longlong(0), ctypes.c_longlong(0)
        self._chk(self.L.mw_model_energy_counts_total(first_ils, count, ctypes.byref(p), ctypes.byref(t)))
        return p.value, t.value

    def neighbour_total(self, first_ils=1, count=None):
        count = self.num_lattices - first_ils + 1 if count is None else count
        t = ctypes.c_longlong(0)
        self._chk(self.L.mw_neighbour_total(first_ils, count, ctypes.byref(t)))
        return t.value

    # -- compute_local_real_energy (molint.F90:220-404) -------------------------------
    def compute_local_real_energy(self, imol, ils):
        """Local energy of molecule imol in lattice ils from the HOST's current ljr.

        The caller moves molecules without telling us (trial move
        mc_moves.F90:1079, silent revert :1186), so the host position of imol and
        of the molecule queried just before it travel with the call."""
        b = self._ils(ils)
        if not (1 <= imol <= self.nwater):
            raise MwError(f"molecule index {imol} outside 1..{self.nwater}")
        if self._stale[b]:
            self._upload(ils)
        prev = self._last_imol[b]
        e, eref = self._e_out[b], self._e_ref[b]
        lj = self.ljr
        if lj is not self._ljr_seen:           # (re)bound by the caller: its address, once, if it is laid out as we pass it
            ok = lj.dtype == np.float64 and lj.flags.c_contiguous and lj.shape == (self.num_lattices, self.nwater, 3)
            self._ljr_seen, self._ljr_addr = lj, (lj.ctypes.data if ok else 0)
        base = self._ljr_addr
        if base:                                # the two positions straight out of the caller's array (24 bytes per molecule)
            row = base + 24 * (b * self.nwater - 1)
            if prev >= 1 and prev != imol:
                rc = self._local(ils, imol, ctypes.c_void_p(row + 24 * imol), prev, ctypes.c_void_p(row + 24 * prev), eref)
            else:
                rc = self._local(ils, imol, ctypes.c_void_p(row + 24 * imol), 0, None, eref)
            if rc:
                self._chk(rc)
        else:
            r = np.ascontiguousarray(lj[b, imol - 1], dtype=np.float64)
            if prev >= 1 and prev != imol:
                rp = np.ascontiguousarray(lj[b, prev - 1], dtype=np.float64)
                self._chk(self._local(ils, imol, _d(r), prev, _d(rp), eref))
            else:
                self._chk(self._local(ils, imol, _d(r), 0, None, eref))
        self._last_imol[b] = imol
        return e.value

    # -- batched single-move path -----------------------------------------------------
    def local_energy_batch(self, ils, imol, trial_xyz=None):
        """Local energies of many (ils[m], imol[m]) from the MIRRORED positions
        (call ``sync_positions`` first if the host changed ljr); with ``trial_xyz``
        molecule m is evaluated at that position instead."""
        ils = np.ascontiguousarray(np.broadcast_to(ils, np.shape(imol)), dtype=np.int32)
        imol = np.ascontiguousarray(imol, dtype=np.int32)
        out = np.zeros(len(imol))
        t = None if trial_xyz is None else np.ascontiguousarray(trial_xyz, dtype=np.float64)
        self._chk(self.L.mw_local_energy_batch(len(imol), _i(ils), _i(imol), None if t is None else _d(t), _d(out)))
        return out

    def delta_energy_batch(self, ils, imol, trial_xyz):
        """(e_old, e_new) per trial move: the two compute_local_real_energy calls of
        mc_water_translation (mc_moves.F90:1010,1083) for many moves in one launch."""
        ils = np.ascontiguousarray(np.broadcast_to(ils, np.shape(imol)), dtype=np.int32)
        imol = np.ascontiguousarray(imol, dtype=np.int32)
        t = np.ascontiguousarray(trial_xyz, dtype=np.float64)
        eo, en = np.zeros(len(imol)), np.zeros(len(imol))
        self._chk(self.L.mw_delta_energy_batch(len(imol), _i(ils), _i(imol), _d(t), _d(eo), _d(en)))
        return eo, en

    def sync_positions(self, ils=None):
        for b in ([ils] if ils else range(1, self.num_lattices + 1)):
            self._upload(b)

    def model_energy_batch(self, first_ils=1, count=None):
        """Full-box energies of ``count`` boxes in one launch (mirrored positions)."""
        count = self.num_lattices - first_ils + 1 if count is None else count
        out = np.zeros(count)
        self._chk(self.L.mw_model_energy_batch(first_ils, count, _d(out)))
        self.model_energy[first_ils - 1:first_ils - 1 + count] = out
        return out

    def build_neighbours_batch(self, first_ils=1, count=None):
        count = self.num_lattices - first_ils + 1 if count is None else count
        mn, mx = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(self.L.mw_build_neighbours_batch(first_ils, count, ctypes.byref(mn), ctypes.byref(mx)))
        return mn.value, mx.value

    # -- device-resident pipeline used by bench.py ------------------------------------
    def moves_upload(self, ils, imol, trial_xyz):
        ils = np.ascontiguousarray(np.broadcast_to(ils, np.shape(imol)), dtype=np.int32)
        imol = np.ascontiguousarray(imol, dtype=np.int32)
        t = np.ascontiguousarray(trial_xyz, dtype=np.float64)
        self._chk(self.L.mw_moves_upload(len(imol), _i(ils), _i(imol), _d(t)))
        self._nmoves = len(imol)

    def moves_launch(self):
        self._chk(self.L.mw_moves_launch())

    def moves_fetch(self):
        eo, en = np.zeros(self._nmoves), np.zeros(self._nmoves)
        self._chk(self.L.mw_moves_fetch(_d(eo), _d(en)))
        return eo, en

    def moves_counts(self):
        """(interactions_old, slots_old, interactions_new, slots_new) summed over the staged moves."""
        out = (ctypes.c_longlong * 4)()
        self._chk(self.L.mw_moves_counts(out))
        return tuple(int(v) for v in out)

    def model_energy_launch(self, first_ils, count):
        self._chk(self.L.mw_model_energy_launch(first_ils, count))

    def step_launch(self, first_ils, count, timer_slot=-1):
        """model_energy_launch + moves_launch in one host call; timer_slot >= 0: event timers timer_slot (full-box
        kernel) and timer_slot + 1 (move kernels)."""
        self._chk(self.L.mw_step_launch(first_ils, count, timer_slot))

    def model_energy_fetch(self, first_ils, count):
        out = np.zeros(count)
        self._chk(self.L.mw_model_energy_fetch(first_ils, count, _d(out)))
        return out

    def build_neighbours_launch(self, first_ils, count):
        self._chk(self.L.mw_build_neighbours_launch(first_ils, count))

    def sync(self):
        self._chk(self.L.mw_sync())

    def timer_start(self, slot=0):
        self._chk(self.L.mw_timer_start(slot))

    def timer_stop(self, slot=0):
        self._chk(self.L.mw_timer_stop(slot))

    def timer_ms(self, slot=0):
        ms = ctypes.c_float(0.0)
        self._chk(self.L.mw_timer_elapsed_ms(slot, ctypes.byref(ms)))
        return ms.value

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cu, mem = ctypes.c_int(0), ctypes.c_longlong(0)
        self._chk(self.L.mw_device_info(name, 256, ctypes.byref(cu), ctypes.byref(mem)))
        return name.value.decode(), cu.value, mem.value

    def constants(self):
        out = np.zeros(8)
        self._chk(self.L.mw_constants(_d(out)))
        return out


def load_boxes(h_list, xyz_list, maxneigh=MAXNEIGH, device=0):
    """Convenience: an initialised EnergyModule holding the given boxes (bohr)."""
    em = EnergyModule(len(xyz_list[0]), len(xyz_list), maxneigh=maxneigh, device=device)
    for b, (h, x) in enumerate(zip(h_list, xyz_list)):
        em.hmatrix[b] = h
        em.ljr[b] = x
    em.energy_init()
    return em
