#!/bin/bash
# Build a DIAGNOSTIC library any-stereo_amd/anystereo/lib/<name>.so from the working tree with extra defines for conv.hip
# (the other objects are the product build's: run `python any-stereo_amd/build.py` first):
#   tools/conv_variant.sh stamps -DAS_CONV_STAMPS          (read with tools/conv_stamps.py on the GPU box)
#   tools/conv_variant.sh now    -DAS_ABL_NO_W             (timing-only ablations: ANYSTEREO_LIB=... tools/kbench.py gru_zr)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
tmp=$(mktemp -d)
python3 any-stereo_amd/build.py > /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-pass-failed "$@" -c any-stereo_amd/csrc/conv.hip -o $tmp/conv.o 2>/dev/null
objs=$tmp/conv.o
for o in any-stereo_amd/build/*.o; do
  [ "$(basename $o)" = "conv.o" ] || objs="$objs $o"
done
# the loader checks as_source_hash() against the tree: a diagnostic build carries its name instead (run with ANYSTEREO_ALLOW_STALE_LIB=1)
echo "extern \"C\" const char* as_source_hash(void) { return \"variant:$name\"; }" > $tmp/stamp.cpp
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $tmp/stamp.cpp -o any-stereo_amd/anystereo/lib/$name.so
rm -rf $tmp
echo built any-stereo_amd/anystereo/lib/$name.so
