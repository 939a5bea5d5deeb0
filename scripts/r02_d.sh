set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; rm -rf $O; mkdir -p $O
B="--steps 30 --no-cpu-baseline --verify 4"
for v in $VARIANTS; do
  [ "$v" = "base" ] && v=""
  AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so timeout 400 python bench.py $B 2>/dev/null | tail -1 > $O/bench$v.json
  python - <<PY
import json
j=json.load(open('$O/bench$v.json'))
print('RESULT $v', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, j.get('verified_dongles'), j['config']['build_defines'])
td=j['roofline'].get('traffic_detail') or {}
print('   chan traffic GB', (j['roofline']['traffic'] or 0)/1e9, {k:round(v/1e9,2) for k,v in td.items() if k.endswith('_bytes')})
for k,v in (td.get('other_kernels') or {}).items(): print('   ',k,{a:round(b/1e9,2) for a,b in v.items()})
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -k "scale" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
