# round 4, fifth GPU call: the whole GPU suite on the ABI-2 library and shim v2, then the default bench line
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c5; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_suite.log 2>&1; tail -40 $O/gpu_suite.log
timeout 900 python bench.py 2>$O/bench_cfg3.err | tail -n 1 > $O/bench_cfg3.json; cut -c1-400 $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err
