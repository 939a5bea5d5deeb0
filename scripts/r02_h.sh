set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; rm -rf $O; mkdir -p $O
for v in $VARIANTS; do
  [ "$v" = "base" ] && v=""
  export AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so
  timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-traffic --verify 4 2>/dev/null | tail -1 > $O/bench$v.json
  python -c "import json; j=json.load(open('$O/bench$v.json')); print('RESULT $v', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, j.get('verified_dongles'), j['config']['build_defines'])"
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$v -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 6 --warmup 2 > /dev/null 2>&1
  grep -h "demod_kernel\|tone_kernel\|back_kernel" $O/kt$v/*/*kernel_stats.csv | cut -d, -f1,4 | sed 's/"void airband:://; s/(airband::DemodArgs[^"]*"//; s/"airband:://'
done
