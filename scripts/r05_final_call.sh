# Round 5, last GPU call: the build without packed-f32 instructions.  The GPU suite in FOUR pytest workers on the one GPU (several processes of this library
# side by side -- the load that used to produce wrong transforms in the fuzz tests), then the round's measurements, most important first, inside the time left.
# arg 1: seconds the whole call may take
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TOTAL=${1:-520}; T00=$(date +%s)
mkdir -p gpurun_out/final_suite
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider -n 4 > gpurun_out/final_suite/gpu_suite.log 2>&1; tail -n 3 gpurun_out/final_suite/gpu_suite.log
LIMIT=$(( TOTAL - ( $(date +%s) - T00 ) - 25 )) bash scripts/profile_round.sh 2>&1 | grep -v "^+" | tail -n 40
cp gpurun_out/final_suite/gpu_suite.log gpurun_out/final/gpu_suite.log
