#!/bin/bash
# Run ON THE GPU BOX: which convolution problems of the cfg-4 training step MIOpen serves with its naive reference solvers
# (naive_conv_ab_nonpacked_*): one eager training step under MIOpen's logging, the "ConvDirectNaive*" choices with their problems.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export MIOPEN_ENABLE_LOGGING=1 MIOPEN_ENABLE_LOGGING_CMD=1 MIOPEN_LOG_LEVEL=6
timeout -k 10 500 python - > gpurun_out/miopen_log.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "any-stereo_amd"))
import torch
from anystereo.harness.synthetic import fill_module_deterministic
from anystereo.harness.train import Trainer, synthetic_train_batch
from anystereo.models import __models__, default_args
args = default_args("continuous_IGEVStereo")
m = __models__["continuous_IGEVStereo"](args)
fill_module_deterministic(m, base_seed=1)
tr = Trainer(m.to("cuda:0"), train_iters=2, max_disp=args.max_disp, graph=False)
batch = synthetic_train_batch(4, 160, 320, seed=0, device="cuda:0")
tr.step(batch)
torch.cuda.synchronize()
print("STEP DONE", flush=True)
PY
echo "rc=$?"
grep -c . gpurun_out/miopen_log.txt
# every distinct problem whose chosen solver is a naive one (forward / data gradient / weight gradient), with the find-time estimate
grep -o "SetValues\] [^,]*, content inserted: ConvDirectNaiveConv[A-Za-z]*:[0-9.e+-]*" gpurun_out/miopen_log.txt | sort | uniq -c | sort -rn > gpurun_out/r05_naive_convs.txt
grep -c "kernel_name = naive_conv" gpurun_out/miopen_log.txt >> gpurun_out/r05_naive_convs.txt
rm -f gpurun_out/miopen_log.txt
cat gpurun_out/r05_naive_convs.txt
