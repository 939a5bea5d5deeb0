set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=10 -k "scale or stage2 or full_slot or end_to_end" > $O/pytest.log 2>&1; tail -25 $O/pytest.log
B="--steps 30 --no-cpu-baseline --no-traffic --verify 4"
for v in "" _exp_pf3 _exp_nopf3 _exp_nopf4 _exp_pf4c4 _exp_pf3c4; do
  AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench$v.json
  python -c "import json,sys; j=json.load(open('$O/bench$v.json')); print('$v', j['ms_per_step'], j['stage_ms'], j.get('verified_dongles'), j['config']['build_defines'])"
done
