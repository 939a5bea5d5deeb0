# Round 5, optional (~8 GPU-minutes): DESIGN 7.1 (d) -- the AGC_EXTRA-delayed fetch of stage 2 only for lanes that use it, as an experiment build (-DAB_MASKED_DELAY; bit-exact
# on the host: tests/test_host_demod.py / test_host_wave64.py with AIRBAND_HOST_DEFINES=-DAB_MASKED_DELAY, 8 000 + 800 fuzz seeds).  Parity of the build on the GPU, then
# stage-2 time (two interleaved runs each) and FETCH_SIZE / WRITE_SIZE per kernel for product and experiment.
#   gpurun --timeout 1500 -- 'bash scripts/r05_masked_delay.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/masked; rm -rf $O; mkdir -p $O
AIRBAND_EXTRA_DEFINES=-DAB_MASKED_DELAY AIRBAND_BUILD_TAG=masked_delay timeout 900 python rtlsdr-airband_amd/_build.py > $O/build_exp.log 2>&1 || tail -5 $O/build_exp.log
EXP=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_masked_delay.so
ls $EXP || exit 1
AIRBAND_HIP_LIB=$EXP AIRBAND_FUZZ_SEEDS_GPU=400 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -k "stage2 or random_plans_on_the_gpu or end_to_end or full_slot or golden or opening_timer" -p no:cacheprovider > $O/parity_exp.log 2>&1; tail -3 $O/parity_exp.log
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 60"
for rep in 1 2; do
  timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_product_$rep.json
  AIRBAND_HIP_LIB=$EXP timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_masked_$rep.json
done
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/masked"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["ms_per_step"], j.get("stage_ms"), j.get("verified_dongles"))
    except Exception as e: print(f, "unreadable", e)
PY
P="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 3 --warmup 1"
for which in product masked; do
  [ $which = masked ] && export AIRBAND_HIP_LIB=$EXP
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$which -- python bench.py $P > $O/pmc_fetch_$which.log 2>&1
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$which -- python bench.py $P > $O/pmc_write_$which.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/masked"
for d in sorted(glob.glob(O+"/pmc_*_*")):
    if not os.path.isdir(d): continue
    agg=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    scale = 2*1024 if "fetch" in d else 1024   # the guide's gfx950 correction for FETCH_SIZE (x2), KiB units
    print(os.path.basename(d), {k:"%.2f GB"%(sum(v)/len(v)*scale/1e9) for k,v in agg.items() if "demod" in k or "tone" in k or "back" in k or "channelizer" in k})
PY
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
