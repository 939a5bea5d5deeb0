set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_dropin_shim.py tests/test_gpu_parity.py -m gpu -x -q -k "dropin or end_of_file or two_device or device_enable or harness" > $O/new.log 2>&1; tail -15 $O/new.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_tone50.so timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "replicated and mixed" > $O/tone50_replica.log 2>&1; tail -2 $O/tone50_replica.log
