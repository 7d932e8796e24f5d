"""TEST INFRASTRUCTURE -- CPU restatement of the reference's Wang-Landau schedule layer for ONE rank (serial comms),
pinned against the reference program itself by tests/test_schedule_pin.py (its checkpoint holds wl_factor,
histogram, weights and wl_invt_active; wlf.dat the increment history; the log the delta G lines).

    flatness_step   mc_check_flatness                mc_moves.F90:1936-2185
    cycle_factor    1/t clamp in mc_update_wl_bins   mc_moves.F90:1655-1657
    unbiased_norm   log_unbiased_norm in mc_init     mc_moves.F90:778-806
    delta_g         mc_compute_deltaG_from_hist      mc_moves.F90:2546-2586

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math

TINY = 2.2250738585072014e-308


def new_state(wl_factor, schedule=0, flattol=0.05, minhist=20, useinvt=False):
    return dict(wl_factor=float(wl_factor), schedule=schedule, flattol=flattol, minhist=minhist, useinvt=useinvt,
                firstcycle=True, histogram_reset=False, invt_active=False, wlf=[])


def cycle_factor(st, cycle, nwater, nbins):
    if st["invt_active"]:
        st["wl_factor"] = min(st["wl_factor"], float(nbins) / float(cycle * nwater))          # :1657
    return st["wl_factor"]


def flatness_step(st, cycle, nwater, histogram, weight, start_bin=1, end_bin=None, dd=False):
    """In place on the lists/arrays `histogram` and `weight` (one rank).  Returns what happened.  dd: the window
    decomposition's branch of a flat histogram (:2114-2126): no weight shift, no wlf.dat, only reset + halving."""
    nbins = len(histogram)
    end_bin = nbins if end_bin is None else end_bin
    if sum(float(h) for h in histogram) < TINY:                                               # :1962
        return "none"
    mini = int(round_half_away(min(float(h) for h in histogram)))                             # :1972
    if st["firstcycle"] and not st["histogram_reset"] and mini > st["minhist"]:               # :1973-1980
        st["histogram_reset"] = True
        for k in range(nbins):
            histogram[k] = 0.0
        return "first reset"
    av, count = 0.0, 0
    for k in range(start_bin - 1, end_bin):                                                   # :1983-1989
        av += float(histogram[k])
        count += 1
    av /= float(count)
    what = "checked"
    if not st["invt_active"]:                                                                 # :2018
        flat = True
        if st["schedule"] == 0:                                                               # :2024-2031
            for k in range(start_bin - 1, end_bin):
                if abs(float(histogram[k]) - av) / av > st["flattol"]:
                    flat = False
        elif st["schedule"] == 1:                                                             # :2033-2039
            if int(round_half_away(min(float(histogram[k]) for k in range(start_bin - 1, end_bin)))) < st["minhist"]:
                flat = False
        else:                                                                                 # :2041-2048
            for k in range(start_bin - 1, end_bin):
                if float(histogram[k]) < (1.0 - st["flattol"]) * av:
                    flat = False
        if flat:
            if not dd:
                mid = float(weight[nbins // 2])                                               # :2063  weight(nbins/2+1)
                for k in range(nbins):
                    weight[k] = float(weight[k]) - mid
                st["wlf"] += [(cycle, st["wl_factor"]), (cycle, 0.5 * st["wl_factor"])]       # :2080-2081
            for k in range(nbins):
                histogram[k] = 0.0                                                            # :2105
            st["wl_factor"] *= 0.5                                                            # :2107
            st["firstcycle"] = False
            what = "halved"
        wl_invt = float(nbins) / float(cycle * nwater)                                        # :2136
        if st["wl_factor"] < wl_invt and st["wl_factor"] > TINY and st["useinvt"]:            # :2137-2143
            st["invt_active"] = True
            st["wl_factor"] = wl_invt
    else:
        what = "invt"
    return what


def round_half_away(x):
    """Fortran nint()."""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def unbiased_norm(weight, av_binwidth, max_mc_cycles, eq_mc_cycles, nranks, nwater):
    nbins = len(weight)
    hits = (float(max_mc_cycles) - float(eq_mc_cycles)) * float(nranks * nwater) / float(nbins)   # :781-782
    incr = hits * av_binwidth
    lun = math.log(incr) + float(weight[0])                                                   # :789
    for k in range(1, nbins):
        if lun > float(weight[k]) + math.log(incr):                                           # :796-804
            lun = lun + math.log(1.0 + incr * math.exp(float(weight[k]) - lun))
        else:
            lun = math.log(incr) + float(weight[k]) + math.log(1.0 + math.exp(lun - float(weight[k])) / incr)
    return lun


def delta_g(unbiased_hist, binwidth):
    """log(pA / pB): lattice 1 = bins 1..nbins/2, lattice 2 = the rest (:2546-2584)."""
    nbins = len(unbiased_hist)
    pnorm = 0.0
    for i in range(nbins):
        pnorm += float(unbiased_hist[i]) * float(binwidth[i])
    pa = pb = 0.0
    for i in range(nbins // 2):
        pa += float(unbiased_hist[i]) / pnorm * float(binwidth[i])
    for i in range(nbins // 2, nbins):
        pb += float(unbiased_hist[i]) / pnorm * float(binwidth[i])
    return math.log(pa / pb)
