#!/bin/bash
# round-6 final run 2 (GPU box): rocprofv3 kernel stats (in-graph + serial) and PMC traffic of the default bench command, the pass
# phases from in-graph markers (coarse + the stages of two GRU iterations), then the default bench line itself
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
echo "profile_round rc=$?"; ls -la gpurun_out/r06_bench_kernel_stats.csv gpurun_out/r06_pmc_traffic.json
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json 2>/dev/null
timeout -k 10 300 python3 tools/pass_phases.py --reps 7 > gpurun_out/r06_pass_timeline.json 2> gpurun_out/r06_pass_timeline.err
timeout -k 10 300 python3 tools/pass_phases.py --reps 7 --fine > gpurun_out/r06_pass_timeline_fine.json 2> gpurun_out/r06_pass_timeline_fine.txt
echo "phases rc=$?"; cut -c1-400 gpurun_out/r06_pass_timeline.json
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
echo "bench rc=$?"; python3 -c "
import json
d=json.load(open('gpurun_out/r06_bench_line.json'))
print(d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['value_spread']['pairs_per_s'], d['pass_phases'] and {k: d['pass_phases'][k] for k in ('pre_loop_wall_ms','loop_ms','post_loop_ms')})
print(d['roofline']); print(d['host'])
tm = d.get('train_mode') or {}
print(tm.get('ms_per_step'), tm.get('value'), (tm.get('reduced_precision') or {}).get('ms_per_step'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['stage_split_s'])"
