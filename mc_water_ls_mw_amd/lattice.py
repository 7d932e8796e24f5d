"""Synthetic ice boxes and the xmol input format for the mW energy path.

The reference ships only 48-molecule cells; every larger
configuration BASELINE.json names (96, 1536, 4096, 32768) is synthesised here.
Conventions follow the reference host (init.f90:38-125, data_structures.f90):
lengths in bohr, positions ``xyz[i, :]`` = ``ljr(:,1,i+1,ils)``, cell ``h[k, :]``
= cell vector k = column ``hmatrix(:,k+1,ils)``.  A C-contiguous ``h`` is therefore
byte-identical to the column-major Fortran ``hmatrix(:,:,ils)``.  Positions are
never wrapped into the cell.
"""
from __future__ import annotations

import numpy as np

# constants.f90:42-43
BOHR_TO_ANG = 0.5291772108
ANG_TO_BOHR = 1.0 / 0.5291772108

#: nearest O-O distance of the shipped cells (measured 2.724-2.739 A), SURVEY.md 8(d)
D_OO_ANG = 2.73


def read_xmol(path):
    """Read an ``inputNNN.xmol`` as init.f90:38-125 does.

    Line 1: N.  Line 2: nine cell numbers, read column-major into hmatrix, i.e.
    three consecutive cell vectors.  Then N lines ``O x y z`` in Angstrom.
    Returns ``(h_bohr (3,3), xyz_bohr (N,3))``.
    """
    with open(path) as fh:
        lines = fh.read().split("\n")
    n = int(lines[0].split()[0])
    h = np.array(lines[1].split()[:9], dtype=np.float64).reshape(3, 3)
    xyz = np.array([ln.split()[1:4] for ln in lines[2:2 + n]], dtype=np.float64)
    if xyz.shape != (n, 3):
        raise ValueError(f"{path}: expected {n} atom lines")
    # init.f90:89,103: multiply by ang_to_bohr
    return h * ANG_TO_BOHR, xyz * ANG_TO_BOHR


def write_xmol(path, h_bohr, xyz_bohr):
    """Inverse of :func:`read_xmol` (Angstrom on disk)."""
    h = np.asarray(h_bohr) * BOHR_TO_ANG
    xyz = np.asarray(xyz_bohr) * BOHR_TO_ANG
    with open(path, "w") as fh:
        fh.write(f"{len(xyz):7d}\n")
        fh.write(" ".join(f"{v:.12f}" for v in h.reshape(-1)) + "\n")
        for r in xyz:
            fh.write(f"O {r[0]:.12f} {r[1]:.12f} {r[2]:.12f}\n")


def replicate(h, xyz, reps):
    """Supercell: atoms ``r + i*h1 + j*h2 + k*h3``, replica loops i (slowest), j, k.

    Within each replica the atom order of the parent cell is kept, so atom
    ``m`` of replica ``(i,j,k)`` is ``((i*nb + j)*nc + k)*N + m``.
    """
    na, nb, nc = reps
    h = np.asarray(h, dtype=np.float64)
    xyz = np.asarray(xyz, dtype=np.float64)
    out = []
    for i in range(na):
        for j in range(nb):
            for k in range(nc):
                out.append(xyz + (i * h[0] + j * h[1] + k * h[2]))
    hh = h * np.array([[na], [nb], [nc]], dtype=np.float64)
    return np.ascontiguousarray(hh), np.ascontiguousarray(np.concatenate(out, axis=0))


def ice_ic_cell(d_oo_ang=D_OO_ANG):
    """8-atom cubic cell of cubic ice Ic (diamond lattice of oxygens)."""
    a = 4.0 * d_oo_ang / np.sqrt(3.0)
    fcc = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]], dtype=np.float64)
    frac = np.concatenate([fcc, fcc + 0.25], axis=0)
    h = np.eye(3) * a
    return h * ANG_TO_BOHR, (frac @ h) * ANG_TO_BOHR


def ice_ih_cell(d_oo_ang=D_OO_ANG):
    """8-atom orthorhombic cell of hexagonal ice Ih (lonsdaleite lattice of oxygens).

    a = d*sqrt(8/3), b = sqrt(3)*a, c = 8d/3; ideal tetrahedral geometry (u = 3/8).
    """
    a = d_oo_ang * np.sqrt(8.0 / 3.0)
    b = np.sqrt(3.0) * a
    c = 8.0 * d_oo_ang / 3.0
    u = 3.0 / 8.0
    p1 = np.array([0.0, a / np.sqrt(3.0)])         # hex (1/3, 2/3)
    p2 = np.array([a / 2.0, a / (2.0 * np.sqrt(3.0))])  # hex (2/3, 1/3)
    base = np.array([
        [p1[0], p1[1], 0.0],
        [p2[0], p2[1], c / 2.0],
        [p1[0], p1[1], u * c],
        [p2[0], p2[1], c / 2.0 + u * c],
    ])
    centre = np.array([a / 2.0, b / 2.0, 0.0])
    xyz = np.concatenate([base, base + centre], axis=0)
    h = np.diag([a, b, c])
    return h * ANG_TO_BOHR, xyz * ANG_TO_BOHR


def ice_box(kind, reps, sigma_ang=0.0, seed=20250228):
    """Synthetic ice box: ``kind`` in {"ih","ic"}, ``reps`` replicas of the 8-atom
    cell, optional Gaussian thermal displacement (Angstrom) with a fixed seed."""
    h, xyz = {"ih": ice_ih_cell, "ic": ice_ic_cell}[kind]()
    h, xyz = replicate(h, xyz, reps)
    if sigma_ang > 0.0:
        xyz = thermalise(xyz, sigma_ang, seed)
    return h, xyz


def thermalise(xyz_bohr, sigma_ang, seed):
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(xyz_bohr + rng.normal(0.0, sigma_ang * ANG_TO_BOHR, size=np.shape(xyz_bohr)))


def trial_moves(xyz_bohr, nmoves, max_trans_ang=1.1, seed=1):
    """Trial translations as mc_water_translation draws them (mc_moves.F90:1001-1039):
    molecule uniform in 1..N; direction = normalised (2u-1)^3; length = max_trans*(2u-1).

    Returns ``(imol int32 (nmoves,) 1-based, trial_xyz (nmoves,3) bohr)``.
    """
    rng = np.random.default_rng(seed)
    n = len(xyz_bohr)
    imol = np.minimum((rng.random(nmoves) * n).astype(np.int64) + 1, n).astype(np.int32)
    v = 2.0 * rng.random((nmoves, 3)) - 1.0
    v /= np.sqrt((v * v).sum(axis=1))[:, None]
    r = 2.0 * rng.random(nmoves) - 1.0
    disp = v * (max_trans_ang * ANG_TO_BOHR * r)[:, None]
    trial = np.asarray(xyz_bohr)[imol - 1] + disp
    return imol, np.ascontiguousarray(trial)
