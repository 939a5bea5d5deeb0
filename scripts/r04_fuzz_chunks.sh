# GPU fuzz: results independent of how the bytes are submitted (tests/test_gpu_parity.py::test_results_do_not_depend_on_how_the_bytes_arrive)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz2; rm -rf $O; mkdir -p $O
SEEDS=${1:-300}
timeout 600 python -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || tail -5 $O/build.log
AIRBAND_FUZZ_SEEDS_CHUNKS=$SEEDS timeout 1200 python -m pytest tests/test_gpu_parity.py -k how_the_bytes_arrive -q -n 12 -p no:cacheprovider > $O/fuzz_chunks.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_chunks.log | cut -c1-420 | head -40
grep -E "^E  " $O/fuzz_chunks.log | sort | uniq -c | sort -rn | head -20 | cut -c1-300
tail -3 $O/fuzz_chunks.log | cut -c1-300
