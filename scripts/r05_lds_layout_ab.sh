# Round 5, optional (~12 GPU-minutes): more statistics on round 4's rare event (profiles/r04_experiments.md I) -- the chunking campaign under 12 processes per GPU,
# N repeats on the product library (exchange buffers in front of the samples) and N on an experiment build with the old layout (-DAB_FFT_XB_BEHIND).
#   gpurun --timeout 1500 -- 'bash scripts/r05_lds_layout_ab.sh 4'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-4}
O=$GRAFT_REPO_ROOT/gpurun_out/lds_ab; rm -rf $O; mkdir -p $O
AIRBAND_EXTRA_DEFINES=-DAB_FFT_XB_BEHIND AIRBAND_BUILD_TAG=xb_behind timeout 900 python rtlsdr-airband_amd/_build.py > $O/build_exp.log 2>&1 || tail -5 $O/build_exp.log
ls rtlsdr-airband_amd/libairband_hip_exp_xb_behind.so || exit 1
for which in product old_layout; do
  [ $which = old_layout ] && export AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_xb_behind.so
  AIRBAND_FUZZ_CHUNKS_PIPE=0 bash scripts/r04_fuzz_chunks.sh 360 $N > $O/$which.log 2>&1
  grep -E "passed|failed" $O/$which.log; grep -E "^E +AssertionError" $O/$which.log | cut -c1-600
done
