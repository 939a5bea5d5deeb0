# GPU fuzz campaign: random plans of every kind on made-up stage-1 output, stage 2 on the GPU against the oracle (tests/test_gpu_parity.py::test_random_plans_on_the_gpu)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz; rm -rf $O; mkdir -p $O
SEEDS=${1:-600}
timeout 600 python -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || tail -5 $O/build.log  # once, before the workers start
AIRBAND_FUZZ_SEEDS_GPU=$SEEDS timeout 1500 python -m pytest tests/test_gpu_parity.py -k random_plans_on_the_gpu -q -n 12 -p no:cacheprovider > $O/fuzz_gpu.log 2>&1
tail -30 $O/fuzz_gpu.log | cut -c1-400
