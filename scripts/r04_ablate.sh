# round 4: what the matrix-core channelizer costs without its HBM reads / matrix pipe / stores / LDS reads (experiment builds, wrong results by construction)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/abl; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 20 --warmup 3"
for round in 1 2; do
for t in base nodma nomfma nostore nolds compute memory stream; do
  lib=$L/libairband_hip_exp_$t.so; [ $t = base ] && lib=$L/libairband_hip.so
  AIRBAND_HIP_LIB=$lib timeout 200 python bench.py $N 2>/dev/null | tail -n 1 > $O/${t}_$round.json
  python -c "
import json; d=json.load(open('$O/${t}_$round.json')); print('$t', $round, d['ms_per_step'], d['stage_ms'])"
done; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "zero_copy" > $O/parity.log 2>&1; tail -3 $O/parity.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40 --sample-rate 2000000 2>$O/b2000.err | tail -n 1 > $O/bench_cfg3_2000k.json; cut -c1-300 $O/bench_cfg3_2000k.json; tail -2 $O/b2000.err
