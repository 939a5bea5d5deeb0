# Round 6, final GPU call on the committed tree: smoke, the GPU suite as the driver runs it (one process, -x), the chunking fuzz, then the round's measurements (scripts/profile_round.sh).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final_suite
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/final_suite/gpu_suite.log 2>&1; tail -n 3 gpurun_out/final_suite/gpu_suite.log
AIRBAND_FUZZ_SEEDS_CHUNKS=300 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -k how_the_bytes_arrive -q -n 8 -p no:cacheprovider > gpurun_out/final_suite/fuzz_chunks.log 2>&1; tail -n 1 gpurun_out/final_suite/fuzz_chunks.log
R=r06 bash scripts/profile_round.sh 2>&1 | grep -v "^+" | tail -n 60
cp gpurun_out/final_suite/*.log gpurun_out/final/
