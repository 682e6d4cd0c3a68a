#!/bin/bash
# Build libanystereo_hip.so of another git revision's csrc into any-stereo_amd/anystereo/lib/<name>.so for A/B timing:
#   tools/build_variant.sh <git-rev> <name>      then     ANYSTEREO_LIB=$PWD/any-stereo_amd/anystereo/lib/<name>.so python bench.py
set -e
rev=$1; name=$2
tmp=$(mktemp -d)
mkdir -p $tmp/any-stereo_amd/csrc $tmp/include
for f in $(git ls-tree --name-only $rev any-stereo_amd/csrc/); do git show $rev:$f > $tmp/$f; done
git show $rev:include/anystereo_hip.h > $tmp/include/anystereo_hip.h
objs=""
for f in $tmp/any-stereo_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c $f -o ${f%.hip}.o &
  objs="$objs ${f%.hip}.o"
done
wait
# the loader checks as_source_hash() against the tree: a variant carries its revision instead (run with ANYSTEREO_ALLOW_STALE_LIB=1)
echo "extern \"C\" const char* as_source_hash(void) { return \"variant:$rev\"; }" > $tmp/stamp.cpp
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $tmp/stamp.cpp -o any-stereo_amd/anystereo/lib/$name.so
rm -rf $tmp
echo built any-stereo_amd/anystereo/lib/$name.so
