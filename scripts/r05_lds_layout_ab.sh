# Round 5, optional (~6 GPU-minutes per variant at N = 4): more statistics on round 4's rare event (profiles/r04_experiments.md I) -- the chunking campaign under 12 processes
# per GPU, N repeats each on: the product library; an experiment build with the old LDS layout (-DAB_FFT_XB_BEHIND); one whose wavefront-level exchanges also wait for the LDS
# counter (-DAB_WAVE_SYNC_WAITS: if the events stop, they sit between a wavefront's LDS write and read; if not, elsewhere).
#   gpurun --timeout 1800 -- 'bash scripts/r05_lds_layout_ab.sh 4'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-4}
O=$GRAFT_REPO_ROOT/gpurun_out/lds_ab; rm -rf $O; mkdir -p $O
AIRBAND_EXTRA_DEFINES=-DAB_FFT_XB_BEHIND AIRBAND_BUILD_TAG=xb_behind timeout 900 python rtlsdr-airband_amd/_build.py > $O/build_exp.log 2>&1 || tail -5 $O/build_exp.log
AIRBAND_EXTRA_DEFINES=-DAB_WAVE_SYNC_WAITS AIRBAND_BUILD_TAG=sync_waits timeout 900 python rtlsdr-airband_amd/_build.py > $O/build_exp2.log 2>&1 || tail -5 $O/build_exp2.log
ls rtlsdr-airband_amd/libairband_hip_exp_xb_behind.so rtlsdr-airband_amd/libairband_hip_exp_sync_waits.so || exit 1
for which in product old_layout sync_waits; do
  [ $which = old_layout ] && export AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_xb_behind.so
  [ $which = sync_waits ] && export AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_sync_waits.so
  AIRBAND_FUZZ_CHUNKS_PIPE=0 bash scripts/r04_fuzz_chunks.sh 360 $N > $O/$which.log 2>&1
  grep -E "passed|failed" $O/$which.log; grep -E "^E +AssertionError" $O/$which.log | cut -c1-600
done
