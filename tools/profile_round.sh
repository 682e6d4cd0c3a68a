#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + the two HBM-traffic PMC passes of bench.py,
# each in its own rocprofv3 run (the pool refuses PMC combined with API tracing).  Results -> gpurun_out/.
#   tools/profile_round.sh r01
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_kt -- python $ROOT/bench.py --no-cpu-baseline --no-batched > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $OUT/_kt/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/_pf -- python $ROOT/bench.py --no-cpu-baseline --no-batched --no-graph --steps 1 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/_pw -- python $ROOT/bench.py --no-cpu-baseline --no-batched --no-graph --steps 1 --warmup 1 > /dev/null 2>&1
python $ROOT/tools/summarize_pmc.py $OUT/_pf $OUT/_pw $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/_kt $OUT/_pf $OUT/_pw
