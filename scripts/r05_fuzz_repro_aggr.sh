# Round 5: victims (scripts/r05_fuzz_repro.py: the exchange kernel at fft >= 4096, every run compared with the first) beside aggressors that run the matrix-core channelizers.
#   gpurun --timeout 900 -- 'bash scripts/r05_fuzz_repro_aggr.sh 120 6 6 i8'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-120}; V=${2:-6}; A=${3:-6}; KIND=${4:-any}
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz_repro_$KIND; rm -rf $O; mkdir -p $O
python -c "import torch"
pids=""
for p in $(seq 1 $A); do timeout $((SECS + 300)) python scripts/r05_aggressor.py $((SECS + 20)) $((p * 7000 + 300000)) $KIND > $O/aggr.$p.log 2>&1 & done
for p in $(seq 1 $V); do
  timeout $((SECS + 300)) python scripts/r05_fuzz_repro.py arm.$p $SECS $((p * 5000)) $O > $O/arm.$p.log 2>&1 &
  pids="$pids $!"
done
wait $pids
wait
grep -h EVENT $O/arm.*.log | cut -c1-1500 | head -20
tail -q -n 1 $O/aggr.*.log | head -12
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/arm.jsonl")]
print("$KIND: victims %d configs %d runs %d launches %d events %d" % (len(rows), sum(r["configs"] for r in rows), sum(r["runs"] for r in rows), sum(r["launches"] for r in rows), sum(r["events"] for r in rows)))
PY
du -sh $O
