import os, sys, subprocess, shutil, re
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_full_program as T
nl = T.SINGLE_BOX.replace("file_output_int  = 50", "file_output_int  = 1").replace("max_mc_cycles    = 600", "max_mc_cycles    = 30")
nl = nl.replace("eq_adjust_mc     = .true.", "eq_adjust_mc     = .true.\nmonitor_int      = 1")
base = "/tmp/dbgfull"
shutil.rmtree(base, ignore_errors=True)
for name in ("ref", "hip"):
    T._prepare(os.path.join(base, name), nl, False)
a, oa = T._run(T.REF, os.path.join(base, "ref"))
b, ob = T._run(T.HIP, os.path.join(base, "hip"))
def drifts(d):
    out = []
    for f in ("node000.log", "mc.log"):
        p = os.path.join(base, d, f)
        if os.path.exists(p):
            for ln in open(p):
                m = re.match(r"#\s+1\s+(-?\d+\.\d+)\s+(-?\d+\.\d+)\s+(-?\d+\.\d+)", ln)
                if m: out.append(tuple(float(x) for x in m.groups()))
    return out
da, db = drifts("ref"), drifts("hip")
print(len(da), len(db))
for k, (x, y) in enumerate(zip(da, db)):
    print(k + 1, "ref stored/computed/drift", x, "| hip", y)
