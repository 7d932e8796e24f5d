import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden
from mc_water_ls_mw_amd.energy import load_boxes
for name in ["ic48","ih48_t020","ih1536_t012","ih4096_t015"]:
    z = load_golden(name)
    t=time.time(); em = load_boxes([z["h"]],[z["xyz"]]); dt=time.time()-t
    e = em.model_energy[0]; r = float(z["model_energy"])
    print(name, "init %.3fs"%dt, "E", e, r, (e-r)/r, "counts", em.model_energy_counts(1), "nn eq", np.array_equal(em.neighbours(1)[0], z["nn"]))
    loc = em.local_energy_batch(1, np.arange(1, int(z["n"])+1))
    print("  local max rel", np.abs((loc - z["local"])/z["local"]).max())
    eo,en = em.delta_energy_batch(1, z["trial_imol"], z["trial_xyz"])
    print("  trial max abs", np.abs(eo-z["trial_old"]).max(), np.abs(en-z["trial_new"]).max())
    print("  single", em.compute_local_real_energy(3,1), z["local"][2])
    em.energy_deinit()
