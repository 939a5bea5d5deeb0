# GPU fuzz: results independent of how the bytes are submitted (tests/test_gpu_parity.py::test_results_do_not_depend_on_how_the_bytes_arrive)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz2; rm -rf $O; mkdir -p $O
SEEDS=${1:-300}
timeout 600 python -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || tail -5 $O/build.log
REPEAT=${2:-1}
for r in $(seq 1 $REPEAT); do
AIRBAND_FUZZ_SEEDS_CHUNKS=$SEEDS timeout 1200 python -m pytest tests/test_gpu_parity.py -k how_the_bytes_arrive -q -n 12 -p no:cacheprovider > $O/fuzz_chunks_$r.log 2>&1
grep -E "^E +AssertionError" $O/fuzz_chunks_$r.log | cut -c1-3000 | head -12
tail -2 $O/fuzz_chunks_$r.log | cut -c1-300
done
