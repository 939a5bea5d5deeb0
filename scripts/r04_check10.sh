set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c10; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wavefront_fft.py tests/test_gpu_parity.py -q -m gpu -k "wavefront or fft_wave64 or force_fft or other_formats or afc" -p no:cacheprovider > $O/fft.log 2>&1; tail -3 $O/fft.log
AIRBAND_FUZZ_CHUNKS_PIPE=0 bash scripts/r04_fuzz_chunks.sh 360 ${1:-4}
