# Round 6, call 1: cache policy of the channelizer's LDS-DMA transfers (VERDICT r05 item 3 i).  Interleaved A/B on one box:
#   base  = product (aux 0 everywhere)     nt = read-once pieces of a step nt (aux 2), the two overlap pieces default     ntall = every piece nt
#   AIRBAND_EXTRA_DEFINES="-DAB_DMA_NT_AUX=2" AIRBAND_BUILD_TAG=nt python rtlsdr-airband_amd/_build.py
#   AIRBAND_EXTRA_DEFINES="-DAB_DMA_NT_AUX=2 -DAB_DMA_EDGE_AUX=2" AIRBAND_BUILD_TAG=ntall python rtlsdr-airband_amd/_build.py
#   gpurun --timeout 1500 -- 'bash scripts/r06_call1.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_nt; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
for round in 1 2 3; do
  for l in base nt ntall; do
    lib=$L/libairband_hip.so; [ $l != base ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N 2>$O/err_${l}_$round.log | tail -1 > $O/${l}_cfg3_$round.json
    [ $round -lt 3 ] && AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/${l}_am65536_$round.json
  done
done
# the traffic counters of the winner candidates (FETCH_SIZE pass only: does nt change what HBM sees?)
for l in base nt; do
  lib=$L/libairband_hip.so; [ $l != base ] && lib=$L/libairband_hip_exp_$l.so
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$l -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 3 --warmup 1 --dongles 32768 > $O/pmc_$l.log 2>&1
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_nt"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
