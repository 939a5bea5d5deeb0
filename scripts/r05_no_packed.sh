# Round 5: the build without packed-f32 instructions (-Xclang -target-feature -Xclang -packed-fp32-ops: libairband_hip_exp_nopk.so).
#  1. what it costs: product / nopk / product / nopk bench lines of the default workload
#  2. exchange-kernel victims on the nopk build beside three PRODUCT bench.py loops in other processes (the arm that gave 63 / 70 events per ~1 000 runs)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/no_packed; rm -rf $O; mkdir -p $O
NOPK=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_nopk.so
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 30 --warmup 3"
i=0
for l in product nopk product nopk; do
  i=$((i+1)); lib=""; [ $l = nopk ] && lib=$NOPK
  AIRBAND_HIP_LIB=$lib timeout 60 python bench.py $N 2>/dev/null | tail -1 > $O/${l}_$i.json
  python - $O/${l}_$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
SECS=${1:-70}
apids=""
for a in 1 2 3; do
  (for k in 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --dongles 4096 --steps 400 --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 0 > $O/aggr$a.$k.txt 2>&1; done) &
  apids="$apids $!"
done
sleep 15
pids=""
for p in 1 2 3 4 5 6; do AIRBAND_HIP_LIB=$NOPK R05_VICTIM=fft_wave64 timeout $((SECS + 300)) python scripts/r05_fuzz_repro.py arm.$p $SECS $((p * 5000 + 40000)) $O > $O/arm.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
for q in $apids; do pkill -P $q 2>/dev/null; kill $q 2>/dev/null; done; sleep 3
grep -h "^EVENT {" $O/arm.*.log | cut -c1-300 | head -4
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/arm.jsonl")]
print("nopk victims beside product aggressors: victims %d configs %d runs %d launches %d events %d" % (len(rows), sum(r["configs"] for r in rows), sum(r["runs"] for r in rows), sum(r["launches"] for r in rows), sum(r["events"] for r in rows)))
PY
