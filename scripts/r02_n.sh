# channelizer A/B over libraries: parity subset on the product, then bench lines per library
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; rm -rf $O; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest tests -x -q -m gpu -k "$TESTS" 2>&1 | tail -5; fi
N="--no-cpu-baseline --no-traffic --verify 8 --steps 30"
for v in $VARIANTS; do
  [ "$v" = "base" ] && v=""
  export AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so
  for w in "" "--workload cfg2 --dongles 65536" "--sample-rate 2400000" "--sample-format s16 --ring 1" "--fft-log 10"; do
    timeout 300 python bench.py $N $w 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; j=json.load(open('$O/b.json')); print('RESULT $v [$w]', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, 'frac', j['roofline']['frac'], 'verified', j.get('verified_dongles'))"
  done
done
