# Round 6: fuzz campaigns on the round's build -- channelizer configurations (random sample format / fft size / rate / channel count: the new CF32 variants among them), submit
# chunking (every third seed pipelined: the five-per-CU hold), random stage-2 plans in slot order and regrouped, mixer wirings.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_fuzz; rm -rf $O; mkdir -p $O
AIRBAND_FUZZ_SEEDS_STAGE1=${S1:-600} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -k random_channelizer_configurations -q -n 8 -p no:cacheprovider > $O/fuzz_stage1.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_stage1.log | cut -c1-420 | head -20; tail -n 1 $O/fuzz_stage1.log | cut -c1-200
AIRBAND_FUZZ_SEEDS_CHUNKS=${S2:-300} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -k how_the_bytes_arrive -q -n 8 -p no:cacheprovider > $O/fuzz_chunks.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_chunks.log | cut -c1-420 | head -20; tail -n 1 $O/fuzz_chunks.log | cut -c1-200
AIRBAND_FUZZ_SEEDS_GPU=${S3:-300} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -k random_plans_on_the_gpu -q -n 8 -p no:cacheprovider > $O/fuzz_plans.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_plans.log | cut -c1-420 | head -20; tail -n 1 $O/fuzz_plans.log | cut -c1-200
AIRBAND_HIP_REGROUP=1 AIRBAND_FUZZ_SEEDS_GPU=${S3:-300} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -k random_plans_on_the_gpu -q -n 8 -p no:cacheprovider > $O/fuzz_plans_regrouped.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_plans_regrouped.log | cut -c1-420 | head -20; tail -n 1 $O/fuzz_plans_regrouped.log | cut -c1-200
AIRBAND_FUZZ_SEEDS_MIXERS=${S4:-100} timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -k random_mixer_wirings -q -n 8 -p no:cacheprovider > $O/fuzz_mixers.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_mixers.log | cut -c1-420 | head -20; tail -n 1 $O/fuzz_mixers.log | cut -c1-200
