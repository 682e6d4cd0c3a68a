#!/bin/bash
# Run ON THE GPU BOX: micro-time single kernels (tools/kbench.py) under several library builds.
#   tools/ab_kbench.sh "<kernels>" <name|cur> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
kernels=$1; shift
for v in "$@"; do
  if [ "$v" = cur ]; then unset ANYSTEREO_LIB; else export ANYSTEREO_LIB=$ROOT/any-stereo_amd/anystereo/lib/$v.so; fi
  echo "== $v"
  python $ROOT/tools/kbench.py $kernels --reps 30 2>&1 | grep "us/launch"
done
