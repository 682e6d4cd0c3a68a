#!/bin/bash
# Run ON THE GPU BOX: steady-state breakdown of one eager forward pass -> gpurun_out/<tag>_pass_breakdown.json
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/_tr -- python $ROOT/bench.py --no-cpu-baseline --no-batched --no-graph --steps 2 --warmup 2 > /dev/null 2>&1
python $ROOT/tools/pass_breakdown.py $OUT/_tr $OUT/${TAG}_pass_breakdown.json
rm -rf $OUT/_tr
