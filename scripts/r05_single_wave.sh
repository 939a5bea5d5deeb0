# Round 5: do the wrong results under multi-process sharing need SEVERAL wavefronts per workgroup?  (1) the exchange kernel as one-wavefront workgroups beside heavy aggressors;
# (2) three bench.py processes at once with the tone kernel as one-wavefront workgroups (64 sampled dongles against the oracle each), and the same with the default 256 threads.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import torch"
AIRBAND_HIP_FFT_THREADS=64 bash scripts/r05_fuzz_repro_heavy.sh 90 '' fft_wave64 heavy_fft64 2>&1 | tail -2
for mode in 64 256; do
  O=$GRAFT_REPO_ROOT/gpurun_out/shared_bench_tone$mode; rm -rf $O; mkdir -p $O
  for round in 1 2; do
    pids=""
    for p in 1 2 3; do
      AIRBAND_HIP_TONE_THREADS=$mode timeout 600 python bench.py --dongles 4096 --steps $((100 + 37 * p)) --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 64 2>$O/err.$round.$p.txt | tail -n 1 > $O/bench.$round.$p.json &
      pids="$pids $!"
    done
    wait $pids
  done
  python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench.*.json")):
    try:
        j = json.load(open(f)); print("tone threads $mode", f.split("/")[-1], "ms/step", j["ms_per_step"], "verified", j.get("verified_dongles"), str(j.get("verify"))[:110])
    except Exception as e:
        print(f, "unreadable", e)
PY
done
