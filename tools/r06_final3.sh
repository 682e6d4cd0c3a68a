#!/bin/bash
# round-6 final run 3 (GPU box): training profile (rocprofv3 per-kernel totals of the timed steps) + the 1-rank launcher form
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_train.sh r06 > gpurun_out/r06_profile_train.log 2>&1
echo "profile_train rc=$?"; tail -42 gpurun_out/r06_profile_train.log | head -45
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-batched > gpurun_out/r06_bench_line_torchrun_1rank.json 2> gpurun_out/r06_bench_line_torchrun_1rank.err
echo "torchrun rc=$?"; python3 -c "
import json
d=json.load(open('gpurun_out/r06_bench_line_torchrun_1rank.json'))
print(d['value'], d['train_mode'])"
