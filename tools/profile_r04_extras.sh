#!/bin/bash
# Run ON THE GPU BOX: round-4 evidence that is not part of tools/profile_round.sh —
#   * rocprofv3 kernel stats of the ISOLATED launches the bench line quotes (lookup_convc1, lookup, corr_build, geo_pyramid) and of the
#     direct vs Winograd gate convolutions + the Winograd input transforms (tools/kbench.py, eager launches, one stream)
#   * SQ / TCC counters of conv_wino_kernel and of the direct kernel on gru04 z|r (separate --pmc passes, tools/pmc_kbench.sh)
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_kx -- python3 $ROOT/tools/kbench.py lookup_convc1 lookup corr_build geo_pyramid gru04_zr gru04_zr_wino gru04_q gru04_q_wino wino_tr384 wino_tr128 --reps 40 > $OUT/${TAG}_kbench_isolated.txt 2>&1
cp $OUT/_kx/*/*kernel_stats.csv $OUT/${TAG}_kbench_isolated_kernel_stats.csv
rm -rf $OUT/_kx
{
  echo "== conv_wino_kernel on gru04 z|r (tools/pmc_kbench.sh, rocprofv3 --pmc, separate passes)"
  bash $ROOT/tools/pmc_kbench.sh conv_wino_kernel "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;FETCH_SIZE;WRITE_SIZE" gru04_zr_wino --reps 10
  echo "== conv_split_kernel (direct) on gru04 z|r"
  bash $ROOT/tools/pmc_kbench.sh conv_split_kernel "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;FETCH_SIZE;WRITE_SIZE" gru04_zr --reps 10
  echo "== wino_transform_kernel (384 channels)"
  bash $ROOT/tools/pmc_kbench.sh wino_transform_kernel "GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE" wino_tr384 --reps 10
} > $OUT/${TAG}_pmc_winograd_vs_direct.txt 2>&1
cat $OUT/${TAG}_kbench_isolated.txt | grep "us/launch"
cat $OUT/${TAG}_pmc_winograd_vs_direct.txt
