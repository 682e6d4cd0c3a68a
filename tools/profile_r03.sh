#!/bin/bash
# Run ON THE GPU BOX: round-3 profile set -> gpurun_out/r03_* (kernel stats in-graph / serial, PMC traffic, pass timeline)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
bash $ROOT/tools/profile_round.sh r03 > $OUT/r03_profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/_tl -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
python3 $ROOT/tools/pass_timeline.py $OUT/_tl $OUT/r03_pass_timeline.json > $OUT/r03_pass_timeline.txt 2>&1
rm -rf $OUT/_tl
