# GPU fuzz campaign over channelizer configurations (tests/test_gpu_parity.py::test_random_channelizer_configurations)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz1; rm -rf $O; mkdir -p $O
SEEDS=${1:-400}
timeout 600 python -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || tail -5 $O/build.log
AIRBAND_FUZZ_SEEDS_STAGE1=$SEEDS timeout 1500 python -m pytest tests/test_gpu_parity.py -k random_channelizer_configurations -q -n 12 -p no:cacheprovider > $O/fuzz_stage1.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_stage1.log | cut -c1-420 | head -60
tail -3 $O/fuzz_stage1.log | cut -c1-300
