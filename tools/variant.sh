#!/bin/bash
# Build a DIAGNOSTIC library any-stereo_amd/anystereo/lib/<name>.so from the working tree with extra defines for ONE source file
# (the other objects are the product build's):
#   tools/variant.sh lookup.hip lk_traffic -DAS_LK_TRAFFIC_ONLY
# Run with ANYSTEREO_LIB=$PWD/any-stereo_amd/anystereo/lib/<name>.so ANYSTEREO_ALLOW_STALE_LIB=1 (tools/ab_kbench.sh does both).
set -e
cd "$(dirname "$0")/.."
file=$1; name=$2; shift; shift
base=${file%.hip}
tmp=$(mktemp -d)
python3 any-stereo_amd/build.py > /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-pass-failed "$@" -c any-stereo_amd/csrc/$file -o $tmp/$base.o 2>/dev/null
objs=$tmp/$base.o
for o in any-stereo_amd/build/*.o; do
  [ "$(basename $o)" = "$base.o" ] || objs="$objs $o"
done
echo "extern \"C\" const char* as_source_hash(void) { return \"variant:$name\"; }" > $tmp/stamp.cpp
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $tmp/stamp.cpp -o any-stereo_amd/anystereo/lib/$name.so
rm -rf $tmp
echo built any-stereo_amd/anystereo/lib/$name.so
