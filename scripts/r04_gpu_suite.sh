# the whole GPU suite + smoke, as the driver runs them at round end
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/suite; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; tail -16 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
