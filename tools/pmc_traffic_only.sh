#!/bin/bash
# Run ON THE GPU BOX: only the two HBM-traffic PMC passes of tools/profile_round.sh -> gpurun_out/<tag>_pmc_traffic.json
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/_pf -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --no-graph --steps 1 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/_pw -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --no-graph --steps 1 --warmup 1 > /dev/null 2>&1
python3 $ROOT/tools/summarize_pmc.py $OUT/_pf $OUT/_pw $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/_pf $OUT/_pw
