#!/bin/bash
# AS_CONV_XCD=1 (channel tiles of a pixel tile on one XCD) vs 2 (+ a contiguous band of pixel tiles per XCD): parity of the conv
# tests under mode 2, stand-alone kernel times, FETCH_SIZE of the loop's convolutions, and the bench line's headline.
set -o pipefail
mkdir -p gpurun_out
K="gru04_zr gru04_q head_conv1 enc_conv enc_c2d2 gru08_zr_bs gru08_q cnet_l1_bsbs"
AS_CONV_XCD=2 timeout -k 10 500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv or gru or update_block or whole_model or model_options or resamplers" 2>&1 | tail -3 || exit 1
for r in 1 2; do
  for m in 1 2; do
    echo "== AS_CONV_XCD=$m"
    AS_CONV_XCD=$m python tools/kbench.py $K --reps 30 --graph 2>&1 | grep "us/launch" | tr '\n' ';'; echo
  done
done
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
  rm -rf /tmp/pf_$m
  AS_CONV_XCD=$m timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$m -- python3 $GRAFT_REPO_ROOT/tools/kbench.py gru04_zr gru04_q head_conv1 enc_conv --reps 5 > /dev/null 2>&1
  python3 - /tmp/pf_$m $m <<'P'
import csv, glob, os, sys
from collections import defaultdict
fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    if "conv_split" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"][r["Kernel_Name"].index("<"):r["Kernel_Name"].index(">") + 1]
        acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in acc:
    print(f"AS_CONV_XCD={sys.argv[2]} FETCH_SIZE {k}: {acc[k] / n[k] / 1024:.1f} MB raw per launch ({n[k]} launches; x2 per the guide = {acc[k] / n[k] / 512:.1f} MB)")
P
done
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for m in 1 2; do
    AS_CONV_XCD=$m timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/xcd_$m.json 2> gpurun_out/xcd_$m.err || { echo bench failed; tail -5 gpurun_out/xcd_$m.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/xcd_$m.json')); print('AS_CONV_XCD=$m', d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['value_spread']['pairs_per_s'])"
  done
done
