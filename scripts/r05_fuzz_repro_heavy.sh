# Round 5: victims beside HEAVY aggressors (bench.py loops, 4 096 dongles per launch, in three other processes).  arg 2: library the AGGRESSORS load ("" = product; an experiment
# build, e.g. -DAB_ABL_NO_DMA: the int8 channelizer without its global_load_lds transfers); arg 3: the victims' channelizer (fft_wave64 | dft_mfma_i8 | dft_mfma_f32); arg 4: tag
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-100}; ALIB=$2; VICTIM=${3:-fft_wave64}; TAG=${4:-heavy}
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz_repro_$TAG; rm -rf $O; mkdir -p $O
apids=""
for a in 1 2 3; do
  (for k in 1 2 3 4 5 6 7 8; do AIRBAND_HIP_LIB=$ALIB timeout 300 python bench.py --dongles 4096 --steps 400 --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 0 > $O/aggr$a.$k.txt 2>&1; done) &
  apids="$apids $!"
done
sleep 15
pids=""
for p in 1 2 3 4 5 6; do R05_VICTIM=$VICTIM timeout $((SECS + 300)) python scripts/r05_fuzz_repro.py arm.$p $SECS $((p * 5000 + 40000)) $O > $O/arm.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
for q in $apids; do pkill -P $q 2>/dev/null; kill $q 2>/dev/null; done; sleep 3
grep -h "^EVENT {" $O/arm.*.log | cut -c1-300 | head -4
grep -c . $O/aggr1.1.txt | head -1; tail -c 300 $O/aggr1.1.txt | head -3
python - <<PY
import json
rows = [json.loads(l) for l in open("$O/arm.jsonl")]
print("$TAG (aggressor library '$ALIB', victims $VICTIM): victims %d configs %d runs %d launches %d events %d" % (len(rows), sum(r["configs"] for r in rows), sum(r["runs"] for r in rows), sum(r["launches"] for r in rows), sum(r["events"] for r in rows)))
PY
