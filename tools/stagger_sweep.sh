#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
for sg in 0 64 128 192 256 384; do
  echo "== AS_CONV_LEAN_OFFSET=$sg"
  AS_CONV_LEAN_OFFSET=$sg python $ROOT/tools/kbench.py gru04_zr head_conv1 --graph --reps 30 2>&1 | grep launch
done
done
