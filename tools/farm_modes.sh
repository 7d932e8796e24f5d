# Smoke of the farm CLI in its run modes on the GPU box (prints one line per mode): bash tools/farm_modes.sh
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/modes
run() { name=$1; shift; timeout -k 10 200 python -m mc_water_ls_mw_amd.farm "$@" > gpurun_out/modes/$name.json 2> gpurun_out/modes/$name.err; rc=$?; python - "$name" $rc <<'PY'
import json,sys
name,rc=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open(f"gpurun_out/modes/{name}.json").read().strip().split("\n")[-1])
    print(name, "rc",rc, {k:d.get(k) for k in ("moves_per_s","acceptance","switches_per_walker","weight_max","wl_factor","delta_g","volume_moves_walker1","joined_weight_range")})
except Exception as e:
    print(name,"rc",rc,"NO JSON",e); print(open(f"gpurun_out/modes/{name}.err").read()[-800:])
PY
}
W=256
run dd --walkers 8 --cycles 60 --strategy dd --eq-cycles 5 --leshift --flat-chk 20
run dd_npt --walkers 8 --cycles 60 --strategy dd --eq-cycles 5 --leshift --npt --flat-chk 20
run leshift --walkers $W --cycles 50 --leshift --npt
run swetnam --walkers $W --cycles 50 --wl-swetnam --wl-alpha 0.01 --npt
run minu --walkers $W --cycles 50 --minu --npt
run regauge --walkers $W --cycles 100 --regauge --npt --flat-chk 25 --wl-schedule 1
run eqadj --walkers $W --cycles 60 --eq-adjust --eq-cycles 40 --monitor 10 --npt
rm -rf /tmp/chk && mkdir -p /tmp/chk
run chk --walkers 64 --cycles 40 --npt --chkpt 20 --outdir /tmp/chk
run restart --walkers 64 --cycles 20 --npt --chkpt 20 --outdir /tmp/chk --restart
run invt --walkers $W --cycles 60 --wl-useinvt --flat-chk 10 --wl-schedule 2 --npt
