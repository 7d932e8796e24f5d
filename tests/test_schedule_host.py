"""CPU: the host-side schedule layer of the 'dd' strategy (WindowSchedules, mc_water_ls_mw_amd/schedule.py) against the
oracle's per-rank flatness_step (oracle/schedule.py, pinned to the reference program by tests/test_options_pin.py), on
synthetic histograms; and the window bookkeeping farm.run derives from MuGrid.window."""
import numpy as np
import pytest

from mc_water_ls_mw_amd.schedule import WangLandauSchedule, WindowSchedules
from mc_water_ls_mw_amd.sweep import MuGrid
from oracle import schedule as osch


class FakeFarm:
    """Tables of `count` walkers held on the host, with the two calls WindowSchedules makes."""

    def __init__(self, hist):
        self.hist = np.array(hist, dtype=float)
        self.writes = 0

    def tables_range(self, first, count):
        return None, self.hist[first - 1:first - 1 + count].copy(), None

    def set_tables_range(self, first, histogram=None):
        self.hist[first - 1:first - 1 + len(histogram)] = histogram
        self.writes += 1


@pytest.mark.parametrize("schedule,kw,okw", [
    (0, dict(wl_flattol=0.3), dict(flattol=0.3)),
    (1, dict(wl_minhist=5), dict(minhist=5)),
    (2, dict(wl_flattol=0.4), dict(flattol=0.4)),
    (0, dict(wl_flattol=0.5, wl_useinvt=True), dict(flattol=0.5, useinvt=True)),
])
def test_window_schedules_follow_the_oracle_rank_by_rank(schedule, kw, okw):
    grid = MuGrid(101, -400.0, 400.0)
    size, overlap, nwater = 4, 2, 48
    ws = WindowSchedules(grid, size, overlap, 0, size, 0.05, wl_schedule=schedule, **kw)
    states = [osch.new_state(0.05, schedule=schedule, **okw) for _ in range(size)]
    rng = np.random.default_rng(3)
    farm = FakeFarm(np.zeros((size, 101)))
    ohist = np.zeros((size, 101))
    for step in range(1, 13):
        cycle = 8 * step
        for k, w in enumerate(ws.windows):                 # visits inside the window only, flatter for some ranks
            add = rng.poisson(6 + 3 * k, 101).astype(float) + (4.0 if k % 2 else 0.0)
            add[:w["start_bin"] - 1] = 0.0
            add[w["end_bin"]:] = 0.0
            farm.hist[k] += add
            ohist[k] += add
        events = ws.check_flatness(cycle, nwater, farm)
        expect = []
        for k, w in enumerate(ws.windows):
            row = list(ohist[k])
            what = osch.flatness_step(states[k], cycle, nwater, row, [0.0] * 101, start_bin=w["start_bin"],
                                      end_bin=w["end_bin"], dd=True)
            ohist[k] = row
            if what not in ("none", "checked"):
                expect.append((k + 1, what))
        assert events == expect
        assert np.array_equal(farm.hist, ohist)
        assert list(ws.wl_factors) == [s["wl_factor"] for s in states]
        assert [s.invt_active for s in ws.scheds] == [s["invt_active"] for s in states]
        assert list(ws.move_factors(cycle + 1, nwater)) == [osch.cycle_factor(s, cycle + 1, nwater, 101) for s in states]
    assert any(s["wl_factor"] < 0.05 for s in states)      # something did happen


def test_a_histogram_that_is_zero_outside_the_window_never_triggers_the_first_reset():
    """mc_check_flatness takes minval over the WHOLE histogram for the first-cycle reset (mc_moves.F90:1972) -- in 'dd'
    the bins outside the window stay empty, so only a negative wl_minhist ever fires it; kept as the reference has it."""
    s = WangLandauSchedule(101, 0.05, wl_schedule=1, wl_minhist=20, start_bin=1, end_bin=52)
    h = np.zeros(101)
    h[:52] = 1000.0
    assert s.check_window(8, 48, h) == ("halved", True) and s.histogram_reset is False
    s2 = WangLandauSchedule(101, 0.05, wl_schedule=1, wl_minhist=-1, start_bin=1, end_bin=52)
    assert s2.check_window(8, 48, h) == ("first reset", True) and s2.histogram_reset is True


def test_swetnam_switches_the_flatness_test_off():
    s = WangLandauSchedule(101, 0.05, wl_schedule=1, wl_minhist=0, wl_swetnam=True)
    s.firstcycle = False
    assert s.check_window(8, 48, np.full(101, 50.0)) == ("swetnam", False) and s.wl_factor == 0.05


def test_windows_tile_the_grid_with_the_overlap():
    grid = MuGrid(101, -400.0, 400.0)
    for size in (2, 3, 4, 8):
        ws = [grid.window(r, size, 2) for r in range(size)]
        bpw = 101 // size
        assert ws[0]["start_bin"] == 1 and ws[-1]["end_bin"] == 101
        for r in range(1, size):
            assert ws[r]["start_bin"] == r * bpw - 2 and ws[r - 1]["end_bin"] == r * bpw + 2
            assert ws[r]["mu_min"] < ws[r - 1]["mu_max"]
        assert [w["ls"] for w in ws][0] in (1, None) and [w["ls"] for w in ws][-1] in (2, None)
