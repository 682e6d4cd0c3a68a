#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
for sg in 0 8 16 32 64; do
  echo "== AS_CONV_XCD_STAGGER=$sg"
  AS_CONV_XCD_STAGGER=$sg python $ROOT/tools/kbench.py gru04_zr gru04_q head_conv1 enc_conv enc_c2d2 gru08_zr_bs --graph --reps 30 2>&1 | grep launch
done
done
