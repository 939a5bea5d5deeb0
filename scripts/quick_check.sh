# A short box for small changes of the library: `bash scripts/quick_check.sh [<pytest -k expression> [<seconds>]]` -- by default the golden streams, the
# end-to-end streams, mixers and device_enable through the C ABI (14 tests, ~20 s).
#   gpurun --timeout 80 -- 'bash scripts/quick_check.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/qc; rm -rf $O; mkdir -p $O
K=${1:-"golden or end_to_end_stream or mixers or device_enable or ragged"}
timeout ${2:-72} python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/check.log 2>&1
tail -4 $O/check.log
