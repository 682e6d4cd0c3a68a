#!/bin/bash
# liif_mlp_bwd_kernel with / without its activation stores and scatter atomics (diagnostic variants built by tools/variant.sh)
set -o pipefail
L=$PWD/any-stereo_amd/anystereo/lib
python tools/kbench_liif_bwd.py 10 2>&1 | grep liif_mlp_bwd || exit 1
for v in bwd_nohd bwd_noat bwd_nohd_noat; do
  ANYSTEREO_LIB=$L/$v.so ANYSTEREO_ALLOW_STALE_LIB=1 python tools/kbench_liif_bwd.py 10 2>&1 | grep liif_mlp_bwd || exit 1
done
