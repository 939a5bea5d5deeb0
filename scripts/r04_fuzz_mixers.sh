# GPU fuzz: random mixer wirings against the reference's summation order (tests/test_gpu_parity.py::test_random_mixer_wirings)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz3; rm -rf $O; mkdir -p $O
SEEDS=${1:-300}
timeout 600 python -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || tail -5 $O/build.log
AIRBAND_FUZZ_SEEDS_MIXERS=$SEEDS timeout 1200 python -m pytest tests/test_gpu_parity.py -k random_mixer_wirings -q -n 12 -p no:cacheprovider > $O/fuzz_mixers.log 2>&1
grep -E "^(FAILED|ERROR)" $O/fuzz_mixers.log | cut -c1-300 | head -30
grep -E "^E  +(Assertion|assert|rtlsdr)" $O/fuzz_mixers.log | cut -c1-260 | sort | uniq -c | sort -rn | head -20
tail -3 $O/fuzz_mixers.log | cut -c1-300
