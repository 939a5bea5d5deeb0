# fft 512 as two 256-sample pieces on two waves (AIRBAND_DFT_PIECE=256): parity subset, then bench lines with and without
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; rm -rf $O; mkdir -p $O
AIRBAND_DFT_PIECE=256 timeout 900 python -m pytest tests -x -q -m gpu -k "$TESTS" 2>&1 | tail -5
N="--no-cpu-baseline --no-traffic --verify 8 --steps 30"
for e in 512 256; do
  export AIRBAND_DFT_PIECE=$e
  for w in "" "--workload cfg2 --dongles 65536" "--workload cfg2"; do
    timeout 300 python bench.py $N $w 2>$O/err.log | tail -1 > $O/b.json
    python -c "import json; j=json.load(open('$O/b.json')); print('RESULT piece=$e [$w]', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, 'frac', j['roofline']['frac'], 'verified', j.get('verified_dongles'))" || tail -5 $O/err.log
  done
done
