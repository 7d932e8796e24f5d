"""CPU: what the compiler made of the Monte Carlo driver (k_sweep).  Its design point is a register budget: four wavefronts per
SIMD (<= 128 VGPRs) for the builds that take one move at a time -- thousands of walkers, bound by instruction issue -- and three
(<= 168) for two lattices with look-ahead, the handful of walkers whose speed is one chain's.  Which side of the budget the
allocator lands on moves with unrelated code in the same translation unit, so the budget is pinned (`amdgpu_waves_per_eu`) and
this test holds the code object to it.  A few spilled registers are a cost, not an error: round 3 recorded a GPU fault of a
spilling build and forbade spills; round 4 ran the whole driver suite on a build capped at 80 registers (13-44 spilled per
instantiation, tools/variants.py spill6): 107 tests green -- the fault of that day was k_cell_pairs' (DESIGN.md 3.3).  What is
held here is the performance guard: no more than a handful of spilled vector registers in any build."""
import os
import re
import struct
import subprocess

import pytest

from conftest import ROOT

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _gfx950_code_object(tmp_path):
    lib = os.path.join(ROOT, "mc_water_ls_mw_amd", "libmw_hip.so")
    data = open(lib, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in libmw_hip.so"
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl]
        off += tl
        if b"gfx950" in triple:
            p = tmp_path / "dev.co"
            p.write_bytes(data[i + o:i + o + s])
            return str(p)
    raise AssertionError("libmw_hip.so holds no gfx950 code object")


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf not in this image")
def test_sweep_kernels_keep_their_register_budget(tmp_path):
    from mc_water_ls_mw_amd import build as mwbuild
    mwbuild.build()
    notes = subprocess.run([READELF, "--notes", _gfx950_code_object(tmp_path)], capture_output=True, text=True, check=True).stdout
    seen = 0
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        m = re.match(r"_ZN2mw7k_sweepILi([12])ELi([12468])ELb([01])ELb([01])ELb([01])E", name)
        if not m:
            continue
        seen += 1
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))     # noqa: E731
        three = m.group(1) == "2" and m.group(2) != "1"                           # two lattices + look-ahead
        # (the builds that also carry mc_volume AND the moment path of the walkers in LDS -- scalar registers run out first there and
        #  take vector lanes with them -- are allowed twice the handful: measured with 13 spilled, the NPT farm gained 17 % from the path)
        withvol_lds = m.group(5) == "1" and m.group(4) == "1"
        assert get("vgpr_spill_count") <= (16 if withvol_lds else 8), (name, get("vgpr_spill_count"))
        assert get("vgpr_count") <= (168 if three else 128), (name, get("vgpr_count"))
    assert seen == 40          # lattices x residency x with / without volume moves, + look-ahead 2 / 4 for walkers in global memory
                               # and for walkers entirely or partly in LDS, + 8 for one-lattice walkers in global memory


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf not in this image")
def test_eight_small_walkers_share_a_compute_unit(tmp_path):
    """The reference's own 48-molecule Ic/Ih pair (examples/ice1_gen_weights: nbins = 101), translations only: a walker's
    static + dynamic LDS must stay within 160 KiB / 8 for the row lengths a replica farm reaches (longest row of 16 384 boxes
    after 100 cycles at 200 K: 26) -- at 20.6 KiB a compute unit took seven walkers instead of eight and the farm lost 9 %
    (profiles/r03e_*)."""
    from mc_water_ls_mw_amd import build as mwbuild
    from mc_water_ls_mw_amd.energy import load_library
    mwbuild.build()
    L = load_library()
    notes = subprocess.run([READELF, "--notes", _gfx950_code_object(tmp_path)], capture_output=True, text=True, check=True).stdout
    static = static_vol = None
    for blk in notes.split("- .agpr_count")[1:]:
        if re.search(r"\.name:\s+_ZN2mw7k_sweepILi2ELi1ELb1ELb1ELb0EE", blk):
            static = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        if re.search(r"\.name:\s+_ZN2mw7k_sweepILi2ELi1ELb1ELb1ELb1EE", blk):
            static_vol = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
    assert static is not None and static_vol is not None
    for stride in (22, 26, 28):          # the reference's examples are NPT: the build with volume moves keeps eight walkers per CU too
        dyn = L.mw_sweep_lds_bytes(2, 48, 101, stride, 1, 0, 0)
        assert 0 < dyn and static_vol + dyn <= 20480, (stride, static_vol, dyn)
    for stride in (22, 26, 30):
        dyn = L.mw_sweep_lds_bytes(2, 48, 101, stride, 0, 0, 0)
        assert 0 < dyn and static + dyn <= 20480, (stride, static, dyn)
    assert static + L.mw_sweep_lds_bytes(2, 48, 101, 26, 0, 1, 0) <= 20480          # a sample run carries the unbiased histogram too
    assert L.mw_sweep_lds_bytes(2, 48, 101, 40, 0, 0, 0) == -1 and L.mw_sweep_lds_bytes(2, 100, 101, 20, 0, 0, 0) == -1
