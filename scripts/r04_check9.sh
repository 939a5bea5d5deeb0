set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c9; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "later_handle or other_formats or ragged or random_channelizer" -p no:cacheprovider > $O/f32.log 2>&1; tail -3 $O/f32.log
AIRBAND_FUZZ_CHUNKS_PIPE=0 bash scripts/r04_fuzz_chunks.sh 360 4
