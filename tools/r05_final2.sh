#!/bin/bash
# round-5 final run 2 (GPU box): rocprofv3 kernel stats (in-graph + serial) and PMC traffic of the default bench command, then the
# default bench line itself (which reads the PMC file just written if it is committed), the training line, the training profile
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
echo "profile_round rc=$?"; ls -la gpurun_out/r05_bench_kernel_stats.csv gpurun_out/r05_pmc_traffic.json
cp gpurun_out/r05_pmc_traffic.json profiles/r05_pmc_traffic.json 2>/dev/null
timeout -k 10 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
echo "bench rc=$?"; python -c "
import json
d=json.load(open('gpurun_out/r05_bench_line.json'))
print(d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['roofline'], d['host'], d.get('train_mode',{}).get('ms_per_step'))"
