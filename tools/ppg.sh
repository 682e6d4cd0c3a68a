for n in 1 2 4; do python bench.py --no-cpu-baseline --pairs-per-gpu $n > gpurun_out/ppg_$n.json 2> gpurun_out/ppg_$n.err; tail -c 300 gpurun_out/ppg_$n.err | tail -2; done
