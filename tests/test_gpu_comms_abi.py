"""GPU: the exchange layer for the Fortran host -- libmw_comms.so (include/mw_comms.h, RCCL) and the `module comms`
replacement mc_water_ls_mw_amd/fortran/comms_rccl.f90.

One GPU here, so one rank: the collectives are EXECUTED on RCCL (bootstrap file, communicator, staging, stream) and
must be the identity; the N-rank arithmetic (delta scheme, window joins) is the same as WalkerComms', which the
gloo tests cover.  Then the whole reference program linked with BOTH replacement modules (HIP energy + RCCL comms,
oracle/_ref/mc_water_hip_rccl) runs a weight-generation job with a table synchronisation every 20 cycles and must
reproduce the run of the serial-comms build."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import test_gpu_full_program as fp

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import ctypes, json, os, sys
import numpy as np
L = ctypes.CDLL(os.path.join(sys.argv[1], "mc_water_ls_mw_amd", "libmw_comms.so"))
L.mw_comms_last_error.restype = ctypes.c_char_p
dp = ctypes.POINTER(ctypes.c_double)
out = {}
def ok(rc):
    assert rc == 0, L.mw_comms_last_error().decode()
x = np.ones(4)
assert L.mw_comms_allreduce_sum(x.ctypes.data_as(dp), 4) != 0            # before init: fails with a message
out["before_init"] = L.mw_comms_last_error().decode()
r, s = ctypes.c_int(-1), ctypes.c_int(-1)
ok(L.mw_comms_init(ctypes.byref(r), ctypes.byref(s)))
out["rank"], out["size"], out["id_file_gone_once_everybody_joined"] = r.value, s.value, not os.path.exists(os.environ["MW_COMMS_ID_FILE"])
rng = np.random.default_rng(5)
a, b, c = rng.normal(size=101), rng.normal(size=101), rng.normal(size=101)
a0, b0, c0 = a.copy(), b.copy(), c.copy()
ok(L.mw_comms_allreduce_sum(a.ctypes.data_as(dp), 101)); out["sum"] = bool(np.array_equal(a, a0))
ok(L.mw_comms_allreduce3(a.ctypes.data_as(dp), b.ctypes.data_as(dp), c.ctypes.data_as(dp), 101))
out["sum3"] = bool(np.array_equal(a, a0) and np.array_equal(b, b0) and np.array_equal(c, c0))
ok(L.mw_comms_allreduce3(a.ctypes.data_as(dp), b.ctypes.data_as(dp), None, 101)); out["sum2"] = bool(np.array_equal(b, b0))
ok(L.mw_comms_allreduce_max(c.ctypes.data_as(dp), 101)); out["max"] = bool(np.array_equal(c, c0))
g = np.zeros(101)
ok(L.mw_comms_allgather(a.ctypes.data_as(dp), g.ctypes.data_as(dp), 101)); out["gather"] = bool(np.array_equal(g, a0))
big = rng.normal(size=200000)                                             # grows the staging buffers
big0 = big.copy()
ok(L.mw_comms_allreduce_sum(big.ctypes.data_as(dp), len(big))); out["big"] = bool(np.array_equal(big, big0))
buf = (ctypes.c_char * 7)(*b"abcdefg")
ok(L.mw_comms_bcast(buf, 7, 0)); out["bcast"] = bytes(buf).decode()
out["bad_root"] = L.mw_comms_bcast(buf, 7, 3) != 0
ok(L.mw_comms_barrier())
ok(L.mw_comms_finalize())
out["id_file_removed"] = not os.path.exists(os.environ["MW_COMMS_ID_FILE"])
out["after_finalize"] = L.mw_comms_barrier() != 0
print(json.dumps(out))
"""


def test_collectives_execute_on_rccl_with_one_rank(tmp_path):
    env = dict(os.environ, MW_COMMS_ID_FILE=str(tmp_path / "id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MW_COMMS_RANK", "MW_COMMS_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert "mw_comms_init first" in out.pop("before_init")
    assert out.pop("rank") == 0 and out.pop("size") == 1 and out.pop("bcast") == "abcdefg"
    assert all(out.values()), out


HIP_RCCL = os.path.join(ROOT, "oracle", "_ref", "mc_water_hip_rccl")

GEN_WEIGHTS = fp.LATTICE_SWITCH.replace("samplerun        = .true.", "samplerun        = .false.\nwl_factor        = 0.005") \
    .replace("list_update_int  = 10", "list_update_int  = 10\nmpi_sync_int     = 20\nflat_chk_int     = 100\nmonitor_int      = 100") \
    .replace("max_mc_cycles    = 400", "max_mc_cycles    = 300")


@pytest.mark.skipif(not (os.path.exists(HIP_RCCL) and os.path.exists(fp.HIP)), reason="oracle/_ref/mc_water_hip_rccl not built")
def test_reference_program_on_rccl_comms_reproduces_the_serial_build(tmp_path):
    from mc_water_ls_mw_amd import io as mwio
    for name in ("serial", "rccl"):
        fp._prepare(str(tmp_path / name), GEN_WEIGHTS, True)
        os.remove(str(tmp_path / name / "eta_weights.dat"))                # weight generation starts from nothing
    a, _ = fp._run(fp.HIP, str(tmp_path / "serial"))
    env = dict(os.environ, MW_COMMS_ID_FILE=str(tmp_path / "id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([HIP_RCCL, "ice.input"], cwd=str(tmp_path / "rccl"), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    therm = [f for f in os.listdir(tmp_path / "rccl") if f.endswith("_therm.dat")][0]
    b = open(tmp_path / "rccl" / therm).read().splitlines()
    fp._compare(b, a)
    log = open(tmp_path / "rccl" / "mc.log").read() if os.path.exists(tmp_path / "rccl" / "mc.log") else ""
    assert "Number of MPI tasks" not in log or " 1" in [ln for ln in log.splitlines() if "Number of MPI tasks" in ln][0]
    fa, mu_a, wa = mwio.read_table(str(tmp_path / "serial" / "eta_weights.dat"))
    fb, mu_b, wb = mwio.read_table(str(tmp_path / "rccl" / "eta_weights.dat"))
    assert fa == fb and np.array_equal(mu_a, mu_b) and wa.max() > 0
    assert np.allclose(wb, wa, rtol=1e-9, atol=1e-11)                      # (w - last) + last may differ from w in the last bit
