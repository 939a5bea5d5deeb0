# A short box for host-side changes of the library: the golden streams, the end-to-end streams, mixers and device_enable through the C ABI.
#   gpurun --timeout 80 -- 'bash scripts/quick_check.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/qc; rm -rf $O; mkdir -p $O
timeout 72 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "golden or end_to_end_stream or mixers or device_enable or ragged" > $O/check.log 2>&1
tail -4 $O/check.log
