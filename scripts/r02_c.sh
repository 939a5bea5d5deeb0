set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -x > $O/pytest.log 2>&1; tail -15 $O/pytest.log
B="--steps 30 --no-cpu-baseline --no-traffic --verify 4"
for v in "" $VARIANTS; do
  AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench$v.json
  python -c "import json,sys; j=json.load(open('$O/bench$v.json')); print('RESULT $v', j['ms_per_step'], j['stage_ms'], j.get('verified_dongles'), j['config']['build_defines'])"
done
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 6 --warmup 2 > $O/kt_serial.log 2>&1
cat $O/kt_serial/*/*kernel_stats.csv | head -9 | cut -c1-160
