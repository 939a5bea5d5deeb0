# Round 5: more statistics on "victims beside int8 matrix-core aggressors", and two more arms: heavy aggressors (bench.py loops: thousands of dongles per launch) and aggressors
# INSIDE the victims' own processes (a thread with its own handles and streams).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-120}
bash scripts/r05_fuzz_repro_aggr.sh $SECS 6 6 i8
mv gpurun_out/fuzz_repro_i8 gpurun_out/fuzz_repro_i8_second
# heavy aggressors
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz_repro_heavy; rm -rf $O; mkdir -p $O
for a in 1 2 3; do (for k in 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --dongles 4096 --steps 400 --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 0 > $O/aggr$a.$k.txt 2>&1; done) & done
sleep 15
pids=""
for p in 1 2 3 4 5 6; do timeout $((SECS + 300)) python scripts/r05_fuzz_repro.py arm.$p $SECS $((p * 5000 + 40000)) $O > $O/arm.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
kill %1 %2 %3 2>/dev/null; sleep 2
grep -h EVENT $O/arm.*.log | cut -c1-600 | head
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/arm.jsonl")]
print("heavy: victims %d configs %d runs %d launches %d events %d" % (len(rows), sum(r["configs"] for r in rows), sum(r["runs"] for r in rows), sum(r["launches"] for r in rows), sum(r["events"] for r in rows)))
PY
# aggressors inside the victims' processes
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz_repro_inproc; rm -rf $O; mkdir -p $O
pids=""
for p in 1 2 3 4 5 6; do R05_INPROCESS_AGGRESSOR=1 timeout $((SECS + 300)) python scripts/r05_fuzz_repro.py arm.$p $SECS $((p * 5000 + 80000)) $O > $O/arm.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
grep -h EVENT $O/arm.*.log | cut -c1-600 | head
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/arm.jsonl")]
print("inproc: victims %d configs %d runs %d launches %d events %d aggressor batches %d" % (len(rows), sum(r["configs"] for r in rows), sum(r["runs"] for r in rows), sum(r["launches"] for r in rows), sum(r["events"] for r in rows), sum(r.get("aggressor_batches", 0) for r in rows)))
PY
