"""CPU: the C-ABI library loads, exports every symbol include/mw_energy.h declares,
and fails loudly -- never silently -- when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _lib():
    from mc_water_ls_mw_amd import build
    from mc_water_ls_mw_amd.energy import load_library
    build.build()
    return load_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mw_energy.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mw_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = _lib()
    names = _declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(L, name), f"libmw_hip.so does not export {name}"
    from mc_water_ls_mw_amd.energy import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == names


def test_constants_without_a_device():
    L = _lib()
    out = np.zeros(8)
    assert L.mw_constants(out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
    assert np.array_equal(out, np.load(GOLDEN + "/constants.npz")["constants"])


def test_calls_before_init_fail_with_a_message():
    L = _lib()
    if L.mw_is_initialised():
        pytest.skip("engine is live in this process")
    e = ctypes.c_double(0)
    assert L.mw_model_energy(1, ctypes.byref(e)) != 0
    assert b"not initialised" in L.mw_last_error()
    assert L.mw_sync() != 0


def test_no_cpu_fallback_without_gpu():
    """Without a gfx950 device mw_init must fail and say so; the Python mirror raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device behaviour is checked on the CPU box")
    L = _lib()
    assert L.mw_init(0, 48, 1, 50) != 0
    msg = L.mw_last_error().decode()
    assert "no HIP device" in msg or "no CPU fallback" in msg
    from mc_water_ls_mw_amd.energy import EnergyModule, MwError
    em = EnergyModule(48, 1)
    with pytest.raises(MwError):
        em.energy_init()


def test_argument_validation_needs_no_device():
    L = _lib()
    if L.mw_is_initialised():
        pytest.skip("engine is live in this process")
    assert L.mw_init(0, 0, 1, 50) != 0 and b"positive" in L.mw_last_error()
    assert L.mw_init(0, 48, 1, 65) != 0 and b"maxneigh" in L.mw_last_error()


def test_product_never_imports_the_oracle():
    """The oracle is the checker, never the thing shipped."""
    pkg = os.path.join(ROOT, "mc_water_ls_mw_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".F90", ".f90", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "mw_oracle" not in text and "libmw_ref" not in text, f


def test_comms_library_exports_every_declared_symbol():
    """libmw_comms.so (include/mw_comms.h): loads without a GPU, exports the whole interface, and refuses to work
    before mw_comms_init."""
    from mc_water_ls_mw_amd import build
    L = ctypes.CDLL(build.build_comms())
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mw_comms.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mw_comms_[a-z_0-9]+)\s*\(", text)))
    assert len(names) == 13
    for name in names:
        assert hasattr(L, name), f"libmw_comms.so does not export {name}"
    L.mw_comms_last_error.restype = ctypes.c_char_p
    assert L.mw_comms_barrier() != 0 and b"mw_comms_init first" in L.mw_comms_last_error()
    assert L.mw_comms_finalize() == 0 and L.mw_comms_abort() == 0     # nothing to tear down is not an error


def test_comms_bootstrap_skips_an_id_file_whose_writer_is_gone(tmp_path):
    """The RCCL id travels through a file (mw_comms.h).  A file left by a job that died must not be taken for this job's:
    ranks other than 0 only accept a record whose writer -- pid and process start time in the record -- is alive.  (No GPU
    is touched before the id is in hand, so this runs on the CPU; a record that IS accepted gets as far as the GPU probe.)"""
    import struct
    import subprocess
    import sys
    script = """
import ctypes, os, sys
L = ctypes.CDLL(sys.argv[1]); L.mw_comms_last_error.restype = ctypes.c_char_p
r, s = ctypes.c_int(-1), ctypes.c_int(-1)
rc = L.mw_comms_init(ctypes.byref(r), ctypes.byref(s))
print(rc, L.mw_comms_last_error().decode())
"""
    lib = os.path.join(ROOT, "mc_water_ls_mw_amd", "libmw_comms.so")
    idf = tmp_path / "id"
    env = dict(os.environ, MW_COMMS_ID_FILE=str(idf), MW_COMMS_RANK="1", MW_COMMS_SIZE="2", MW_COMMS_TIMEOUT="1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")

    def start_time(pid):
        return int(open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[19])

    def record(world, pid, start):
        return b"MWCOMMS1" + struct.pack("<iiQ", world, pid, start) + bytes(128)

    def run():
        p = subprocess.run([sys.executable, "-c", script, lib], env=env, capture_output=True, text=True, timeout=120)
        return p.stdout.strip()
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    idf.write_bytes(record(2, dead.pid, 12345))                       # a dead job's leftover
    out = run()
    assert out.startswith("1 ") and "its writer is gone" in out
    idf.write_bytes(record(3, os.getpid(), start_time(os.getpid())))    # alive, but another job's shape
    out = run()
    assert out.startswith("1 ") and "another world size" in out
    idf.write_bytes(b"\0" * 40)                                        # the old bare-id format / a torn write
    assert "not an id record" in run()
    idf.write_bytes(record(2, os.getpid(), start_time(os.getpid())))    # a live rank 0: accepted -- the next stop is the GPU
    out = run()
    assert "waited" not in out and ("GPU" in out or "hip" in out or out.startswith("0 "))
    env.pop("MW_COMMS_ID_FILE")
    env.pop("MASTER_PORT", None)
    out = run()                                                         # no way to tell two jobs on one host apart: refused
    assert out.startswith("1 ") and "MASTER_PORT or MW_COMMS_ID_FILE" in out
