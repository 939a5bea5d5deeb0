# round 4: pipelined mode with the channelizer held to one wave per SIMD (walking kernel, grid per CU) against the plain pipelined and sequential schedules
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/cores; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pipelined" > $O/parity.log 2>&1; tail -3 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { tag=$1; shift; timeout 300 env "$@" python bench.py $N ${FLAGS} 2>/dev/null | tail -n 1 > $O/$tag.json; python -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['ms_per_step'], {k:round(v,2) for k,v in d['stage_ms'].items()}, d.get('verified_dongles'))"; }
for round in 1 2; do
FLAGS="" run seq_$round AIRBAND_HIP_CHAN_GRID=0
FLAGS="--pipelined" run pipe_plain_$round AIRBAND_HIP_CHAN_GRID=0
FLAGS="--pipelined" run pipe_walk1024_$round AIRBAND_HIP_CHAN_GRID=1024
FLAGS="--pipelined" run pipe_walk2048_$round AIRBAND_HIP_CHAN_GRID=2048
FLAGS="--pipelined" run pipe_walk768_$round AIRBAND_HIP_CHAN_GRID=768
FLAGS="" run seq_walk1024_$round AIRBAND_HIP_CHAN_GRID=1024
done
