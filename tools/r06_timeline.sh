#!/bin/bash
# Run ON THE GPU BOX: timeline of one graph replay of the default forward (rocprofv3 kernel trace) + a short bench line.
#   tools/r06_timeline.sh <tag>
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/_tr -- python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-extras --steps 5 --warmup 2 > $OUT/${TAG}_under_rocprof.json 2> $OUT/${TAG}_under_rocprof.err
python3 $ROOT/tools/pass_timeline.py $OUT/_tr $OUT/${TAG}_pass_timeline.json > $OUT/${TAG}_pass_timeline.txt 2>&1
rm -rf $OUT/_tr
cd $ROOT
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --no-batched --steps 20 --warmup 3 > $OUT/${TAG}_line.json 2> $OUT/${TAG}_line.err
python3 -c "
import json
d=json.load(open('$OUT/${TAG}_line.json'))
print(d['value'], d['ms_per_step'], d['ms_per_gru_iter'])
print({k: v['avg'] for k, v in d['kernel_times_us'].items()})"
head -5 $OUT/${TAG}_pass_timeline.txt
