#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of tools/kbench.py.   tools/prof_kbench.sh <tag> <kbench args...>
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $ROOT/tools/kbench.py "$@" > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep -v "at::native\|elementwise" $f | cut -d, -f1-4 | cut -c1-160 | head -${TOPN:-8}; else tail -5 $out.log; fi
rm -rf $out $out.log
