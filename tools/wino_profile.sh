#!/bin/bash
# Run ON THE GPU BOX: the resource profile of a Winograd variant of the 3x3 GRU convolutions measured on the production
# kernel.  tools/conv_variant.sh builds the diagnostic libraries first (abl_taps0/4/5, abl_nodma).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04}_winograd_resource_profile.txt
cd $ROOT
: > $OUT
for lib in libanystereo_hip abl_taps5 abl_taps4 abl_taps0 abl_nodma; do
  echo "== $lib" >> $OUT
  for rep in 1 2; do
    ANYSTEREO_ALLOW_STALE_LIB=1 ANYSTEREO_LIB=$ROOT/any-stereo_amd/anystereo/lib/$lib.so timeout -k 10 300 python3 tools/kbench.py gru04_zr gru04_q head_conv1 enc_conv gru08_zr_bs --reps 50 >> $OUT 2>&1
  done
done
cat $OUT
