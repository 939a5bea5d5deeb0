# Round 5: the MAIN path (int8 matrix-core channelizer + stage 2) beside heavy aggressors in other processes, checked WITHOUT the oracle and without the signal generator in the
# loop: handles whose 1 024 dongles replay dongle 0's bytes, every dongle's audio rows compared with dongle 0's on the GPU after every batch (scripts/r05_exchange_stress.py, big shape).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-80}
O=$GRAFT_REPO_ROOT/gpurun_out/main_path_shared; rm -rf $O; mkdir -p $O
python -c "import torch"
apids=""
for a in 1 2 3; do
  (for k in 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --dongles 4096 --steps 400 --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 0 > $O/aggr$a.$k.txt 2>&1; done) &
  apids="$apids $!"
done
sleep 15
pids=""
for p in 1 2 3 4; do R05_MAIN_PATH=1 timeout $((SECS + 240)) python scripts/r05_exchange_stress.py shared $SECS u8 1024 $O/shared.jsonl > $O/shared.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
for q in $apids; do pkill -P $q 2>/dev/null; kill $q 2>/dev/null; done; sleep 3
grep -h EVENT $O/shared.*.log | cut -c1-500 | head -12
grep -L '"batches"' $O/shared.*.log | head -3 | while read f; do echo "== $f"; tail -5 $f; done
# the same four workers ALONE
pids=""
for p in 1 2 3 4; do R05_MAIN_PATH=1 timeout $((SECS + 240)) python scripts/r05_exchange_stress.py alone 40 u8 1024 $O/alone.jsonl > $O/alone.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.jsonl")):
    rows = [json.loads(l) for l in open(f)]
    print(os.path.basename(f), "processes", len(rows), "batches", sum(r["batches"] for r in rows), "hop transforms %.3g" % sum(r["hop_transforms"] for r in rows), "events", sum(r["events"] for r in rows))
PY
