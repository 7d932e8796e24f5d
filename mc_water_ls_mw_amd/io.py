"""The reference's on-disk formats, so that a device-resident farm and the reference program can
hand runs to each other: per-rank checkpoint files and the multicanonical weight / histogram tables.
(``inputNNN.xmol`` is in :mod:`mc_water_ls_mw_amd.lattice`.)

Checkpoint (``checkpointRRR.dat.{1,2}``, written alternately, mc_moves.F90:324-390; the loader takes the readable one with
the larger cycle number, :393-501): Fortran *unformatted sequential* records -- each record is
``int32 nbytes | payload | int32 nbytes`` -- in this order::

    nwater (i4) | mc_cycle_num (i4) | mc_max_trans, mc_dv_max (2 f8, bohr) | wl_factor (f8) | histogram(nbins) |
    weight(nbins) | wl_invt_active (logical*4) | [unbiased_hist(nbins), only if samplerun] |
    hmatrix(3,3,L) | ref_ljr(3,1,N,L) | ljr(3,1,N,L) | ls (i4)

Arrays are column-major, i.e. byte-identical to C arrays ``[L][3][3]`` (row k = cell vector k) and ``[L][N][3]``.

Tables (``eta_weights.dat``, ``histogram.dat``, ``unbiased_histogram.dat``; written :1823-1849, weights read :738-770):
a header line ``#Current energy increment = <E20.12>`` and one ``mu_bin value`` pair per bin.
"""
from __future__ import annotations

import os
import struct

import numpy as np


def _records(path):
    data = open(path, "rb").read()
    recs, off = [], 0
    while off < len(data):
        (n,) = struct.unpack_from("<i", data, off)
        (m,) = struct.unpack_from("<i", data, off + 4 + n)
        if n != m:
            raise ValueError(f"{path}: record markers disagree at byte {off}")
        recs.append(data[off + 4:off + 4 + n])
        off += 8 + n
    return recs


def read_checkpoint(path):
    """Parse one checkpoint file.  ``samplerun`` (whether the unbiased histogram is present) and the number of
    lattices are inferred from the record count and sizes."""
    r = _records(path)
    if len(r) not in (11, 12):
        raise ValueError(f"{path}: {len(r)} records, expected 11 or 12")
    samplerun = len(r) == 12
    nwater, cycle = struct.unpack("<i", r[0])[0], struct.unpack("<i", r[1])[0]
    max_trans, dv_max = struct.unpack("<2d", r[2])
    k = 8 if samplerun else 7
    hm = np.frombuffer(r[k], dtype="<f8")
    nlat = hm.size // 9
    out = dict(nwater=nwater, cycle=cycle, mc_max_trans=max_trans, mc_dv_max=dv_max,
               wl_factor=struct.unpack("<d", r[3])[0],
               histogram=np.frombuffer(r[4], dtype="<f8").copy(), weight=np.frombuffer(r[5], dtype="<f8").copy(),
               wl_invt_active=bool(struct.unpack("<i", r[6])[0]), samplerun=samplerun,
               unbiased_hist=np.frombuffer(r[7], dtype="<f8").copy() if samplerun else None,
               hmatrix=hm.reshape(nlat, 3, 3).copy(),
               ref_ljr=np.frombuffer(r[k + 1], dtype="<f8").reshape(nlat, nwater, 3).copy(),
               ljr=np.frombuffer(r[k + 2], dtype="<f8").reshape(nlat, nwater, 3).copy(),
               ls=struct.unpack("<i", r[k + 3])[0])
    return out


def write_checkpoint(path, c):
    """Write a dict of the shape :func:`read_checkpoint` returns; the reference can restart from the file."""
    def rec(b):
        return struct.pack("<i", len(b)) + b + struct.pack("<i", len(b))

    def arr(a):
        return np.ascontiguousarray(a, dtype="<f8").tobytes()

    parts = [rec(struct.pack("<i", int(c["nwater"]))), rec(struct.pack("<i", int(c["cycle"]))),
             rec(struct.pack("<2d", float(c["mc_max_trans"]), float(c["mc_dv_max"]))),
             rec(struct.pack("<d", float(c["wl_factor"]))), rec(arr(c["histogram"])), rec(arr(c["weight"])),
             rec(struct.pack("<i", 1 if c.get("wl_invt_active") else 0))]
    if c.get("samplerun"):
        parts.append(rec(arr(c["unbiased_hist"])))
    parts += [rec(arr(c["hmatrix"])), rec(arr(c["ref_ljr"])), rec(arr(c["ljr"])), rec(struct.pack("<i", int(c["ls"])))]
    with open(path, "wb") as fh:
        fh.write(b"".join(parts))


def latest_checkpoint(directory, rank=0):
    """The file mc_checkpoint_load would pick: the readable one of .1/.2 with the larger cycle (mc_moves.F90:412-437)."""
    best = None
    for fn in (1, 2):
        p = os.path.join(directory, f"checkpoint{rank:03d}.dat.{fn}")
        try:
            c = read_checkpoint(p)
        except (OSError, ValueError, struct.error):
            continue
        if best is None or c["cycle"] > best[1]["cycle"]:
            best = (p, c)
    if best is None:
        raise FileNotFoundError(f"no valid checkpoint for rank {rank} in {directory}")
    return best


def read_table(path):
    """``eta_weights.dat`` / ``histogram.dat``: returns (wl_factor from the header or None, mu_bin, values)."""
    wl, mu, val = None, [], []
    for ln in open(path):
        if ln.startswith("#"):
            try:
                wl = float(ln.split("=")[1].replace("D", "E"))
            except (IndexError, ValueError):
                pass
            continue
        f = ln.split()
        if len(f) >= 2:
            mu.append(float(f[0].replace("D", "E")))
            val.append(float(f[1].replace("D", "E")))
    return wl, np.array(mu), np.array(val)


def fortran_e(x, width=20, digits=12):
    """A real as Fortran's Ew.d edit descriptor prints it: 0.ddddE+xx, right-justified."""
    x = float(x)
    if x == 0.0:
        body = "0." + "0" * digits + "E+00"
    else:
        mant, exp = f"{abs(x):.{digits - 1}E}".split("E")        # d.ddd E xx  ->  0.dddd E xx+1
        body = ("-" if x < 0 else "") + "0." + mant.replace(".", "") + f"E{int(exp) + 1:+03d}"
    return body.rjust(width)


def write_table(path, wl_factor, mu_bin, values):
    """Same layout as mc_moves.F90:1826-1841: E20.12 header, list-directed pairs."""
    with open(path, "w") as fh:
        fh.write(f"#Current energy increment = {fortran_e(wl_factor)}\n")
        for m, v in zip(mu_bin, values):
            fh.write(f"  {float(m)!r}        {float(v)!r}\n")


def append_wlf(directory, rows, replace=False):
    """``wlf.dat``, the history of the Wang-Landau increment: one ``(I10,E20.12)`` line per (cycle, wl_factor)
    (mc_moves.F90:2070-2082 when the histogram turns flat, :2152-2161 in 1/t mode)."""
    with open(os.path.join(directory, "wlf.dat"), "w" if replace else "a") as fh:
        for cycle, f in rows:
            fh.write(f"{int(cycle):10d}{fortran_e(f)}\n")


def read_wlf(directory):
    rows = []
    for ln in open(os.path.join(directory, "wlf.dat")):
        f = ln.split()
        if len(f) == 2:
            rows.append((int(f[0]), float(f[1].replace("D", "E"))))
    return rows


def write_tagged_tables(directory, tag, wl_factor, mu_bin, weight, histogram):
    """``eta_weights.dat_<tag>`` and ``histogram.dat_<tag>`` (mc_moves.F90:2084-2100 with tag = the increment as
    F20.12; :2163-2177 with tag = the cycle as I20.20)."""
    write_table(os.path.join(directory, "eta_weights.dat_" + tag), wl_factor, mu_bin, weight)
    write_table(os.path.join(directory, "histogram.dat_" + tag), wl_factor, mu_bin, histogram)
