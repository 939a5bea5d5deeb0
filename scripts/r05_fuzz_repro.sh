# Round 5: scripts/r05_fuzz_repro.py in P processes.    gpurun --timeout 900 -- 'bash scripts/r05_fuzz_repro.sh 150 12'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-150}; P=${2:-12}
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz_repro; rm -rf $O; mkdir -p $O
python -c "import torch" 
pids=""
for p in $(seq 1 $P); do
  timeout $((SECS + 300)) python scripts/r05_fuzz_repro.py arm.$p $SECS $((p * 5000)) $O > $O/arm.$p.log 2>&1 &
  pids="$pids $!"
done
wait $pids
grep -h EVENT $O/arm.*.log | cut -c1-1500 | head -20
grep -L '"launches"' $O/arm.*.log | head -3 | while read f; do echo "== $f"; tail -5 $f; done
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/arm.jsonl")]
print("processes %d configs %d runs %d launches %d events %d" % (len(rows), sum(r["configs"] for r in rows), sum(r["runs"] for r in rows), sum(r["launches"] for r in rows), sum(r["events"] for r in rows)))
PY
du -sh $O
