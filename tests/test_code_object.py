"""CPU: what the compiler made of the Monte Carlo driver (k_sweep).  Its design point is a register budget -- four
wavefronts per SIMD (<= 128 VGPRs; three, <= 168, only for two lattices + look-ahead + mc_volume) --
and NO register spills to scratch: a build of this kernel that spilled vector registers faulted on the GPU (round 3), and
which side of the budget the allocator lands on moves with unrelated code in the same translation unit."""
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
        m = re.match(r"_ZN2mw7k_sweepILi([12])ELi([124])ELb([01])ELb([01])ELb([01])E", name)
        if not m:
            continue
        seen += 1
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))     # noqa: E731
        three = m.group(5) == "1" and m.group(1) == "2" and m.group(2) != "1"     # volume moves + two lattices + look-ahead
        assert get("vgpr_spill_count") == 0 and get("private_segment_fixed_size") == 0, name
        assert get("vgpr_count") <= (168 if three else 128), (name, get("vgpr_count"))
    assert seen == 36          # lattices x residency x with / without volume moves, + look-ahead 2 / 4 for walkers in global memory
                               # and for walkers entirely or partly in LDS


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
