# Round 6, call 11: the tone kernel skips channels without audio in the batch (product against the previous commit's library, _base); regrouped workgroups NOT in step
# (-DAB_REGROUP_FREE) against the lockstep form and slot order; the pipelined mode's own five-per-CU hold.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c11; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_dropin_shim.py tests/test_gpu_fabric.py -m gpu -x -q -n 4 > $O/suite.log 2>&1; tail -3 $O/suite.log
AIRBAND_HIP_LIB=$L/libairband_hip_exp_rgfree.so AIRBAND_HIP_REGROUP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -n 4 > $O/suite_rgfree.log 2>&1; tail -3 $O/suite_rgfree.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 8 --steps 40"
for r in 1 2 3; do
  AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so timeout 300 python bench.py $N 2>/dev/null | tail -1 > $O/base_cfg3_$r.json
  timeout 300 python bench.py $N 2>$O/err_new_$r.log | tail -1 > $O/new_cfg3_$r.json
  timeout 300 python bench.py $N --regroup 1 2>/dev/null | tail -1 > $O/new_rg1_cfg3_$r.json
  AIRBAND_HIP_LIB=$L/libairband_hip_exp_rgfree.so timeout 300 python bench.py $N --regroup 1 2>$O/err_rgfree_$r.log | tail -1 > $O/rgfree_rg1_cfg3_$r.json
done
for w in "cfg4" "cfg2 --dongles 65536"; do
  t=$(echo $w | tr -d ' -'); 
  timeout 300 python bench.py $N --workload $w --regroup 0 2>/dev/null | tail -1 > $O/new_rg0_$t.json
  timeout 300 python bench.py $N --workload $w --regroup 1 2>/dev/null | tail -1 > $O/new_rg1_$t.json
  AIRBAND_HIP_LIB=$L/libairband_hip_exp_rgfree.so timeout 300 python bench.py $N --workload $w --regroup 1 2>/dev/null | tail -1 > $O/rgfree_rg1_$t.json
done
timeout 300 python bench.py $N --key-on-s 0.15 2>/dev/null | tail -1 > $O/new_duty10.json
AIRBAND_HIP_LIB=$L/libairband_hip_exp_rgfree.so timeout 300 python bench.py $N --key-on-s 0.15 --regroup 1 2>/dev/null | tail -1 > $O/rgfree_rg1_duty10.json
AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so timeout 300 python bench.py $N --key-on-s 0.15 2>/dev/null | tail -1 > $O/base_duty10.json
for r in 1 2; do timeout 300 python bench.py $N --pipelined 2>/dev/null | tail -1 > $O/new_pipelined_$r.json; done
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py $N --verify 0 --steps 8 --warmup 2 > $O/kt_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py $N --verify 0 --steps 8 --warmup 2 > $O/kt.log 2>&1
AIRBAND_HIP_LIB=$L/libairband_hip_exp_rgfree.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_rgfree -- python bench.py $N --verify 0 --steps 8 --warmup 2 --regroup 1 > $O/kt_rgfree.log 2>&1
AIRBAND_HIP_LIB=$L/libairband_hip_exp_rgfree.so timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_rgfree -- python bench.py $N --verify 0 --steps 3 --warmup 1 --dongles 32768 --regroup 1 > $O/pmc_fetch_rgfree.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c11"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["schedule"][:12], d["config"].get("stage2_regrouped"), d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
