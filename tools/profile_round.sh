#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + the two HBM-traffic PMC passes of bench.py, each in its own
# rocprofv3 run (the pool refuses PMC combined with API tracing).  Results -> gpurun_out/ (copy the summaries to profiles/).
#   tools/profile_round.sh r02
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time of the default bench command (timed region = hipGraph replays; the stats cover the whole process)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_kt -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $OUT/_kt/*/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/_kt
# 1b. the same kernels one at a time (one stream, eager): the averages bench.py's HIP events report (kernel_times_us, roofline.avg_us)
#     are durations of kernels running ALONE; inside the replayed graph they share the chip with the other streams' kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_kt -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --serial-loop --steps 3 --warmup 1 > /dev/null 2>&1
cp $OUT/_kt/*/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats_serial.csv
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), eager launches so every dispatch is sampled
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/_pf -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --no-graph --steps 1 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/_pw -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --no-graph --steps 1 --warmup 1 > /dev/null 2>&1
python3 $ROOT/tools/summarize_pmc.py $OUT/_pf $OUT/_pw $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/_kt $OUT/_pf $OUT/_pw
