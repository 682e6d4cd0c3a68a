#!/bin/bash
# Run ON THE GPU BOX: per-phase kernel breakdown of the last forward of tools/time_model.py <args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/_trm -- python $ROOT/tools/time_model.py "$@" > /dev/null 2>&1
python $ROOT/tools/pass_breakdown.py $OUT/_trm $OUT/model_pass_breakdown.json
rm -rf $OUT/_trm
