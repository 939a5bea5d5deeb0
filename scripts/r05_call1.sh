# Round 5, first GPU call: the multi-part / fabric mixer tests (shim through the stand-in librccl, two-process and one-thread exchanges), the new
# alignment case, a bench line of the current build on this box, and the masked-delay experiment (prebuilt: libairband_hip_exp_masked_delay.so).
#   gpurun --timeout 1500 -- 'bash scripts/r05_call1.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fabric.py tests/test_dropin_shim.py tests/test_abi.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider \
  -k "fabric or dropin or abi or 300_byte or do_not_start_on_16 or mixer_exchange or two_ranks or three_ranks or served or shard or failed or harness or waterfall or configs0 or end_of_file or classes" > $O/tests.log 2>&1; tail -15 $O/tests.log
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
EXP=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_masked_delay.so
for rep in 1 2; do
  timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_product_$rep.json
  AIRBAND_HIP_LIB=$EXP timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_masked_$rep.json
done
AIRBAND_HIP_LIB=$EXP AIRBAND_FUZZ_SEEDS_GPU=200 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -k "stage2 or random_plans_on_the_gpu or end_to_end or golden or opening_timer" -p no:cacheprovider > $O/parity_masked.log 2>&1; tail -3 $O/parity_masked.log
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_call1"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["ms_per_step"], j.get("stage_ms"), j.get("verified_dongles"), j.get("build_info"))
    except Exception as e: print(f, "unreadable", e)
PY
P="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 3 --warmup 1"
for which in product masked; do
  [ $which = masked ] && export AIRBAND_HIP_LIB=$EXP
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$which -- python bench.py $P > $O/pmc_fetch_$which.log 2>&1
done
unset AIRBAND_HIP_LIB
python - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_call1"
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
du -sh $O
