# fft 1024 / 2048 on the multi-wave channelizer: parity of the format matrix, then bench lines
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "$TESTS" 2>&1 | tail -8
N="--no-cpu-baseline --no-traffic --verify 8 --steps 20"
for w in "--fft-log 10" "--fft-log 11" ""; do
  timeout 300 python bench.py $N $w 2>$O/err.log | tail -1 > $O/b.json
  python -c "import json; j=json.load(open('$O/b.json')); print('RESULT [$w]', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, 'frac', j['roofline']['frac'], 'verified', j.get('verified_dongles'), j['config']['channelizer'])" || tail -5 $O/err.log
done
