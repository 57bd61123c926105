#!/bin/bash
# SQ counters of the half-width F(2x4) workgroups (wino24n_kernel) beside the shipped full-width kernels (wino24_kernel, wino24h_kernel) on the
# narrow layer shapes: three rocprofv3 --pmc passes of tools/wino24n_ab.py, per-kernel means (tools/pmc_sq.py).   bash tools/wino24n_pmc.sh <tag>
set -o pipefail
tag=${1:-rXX}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_w24n_sq$i -o pmc -- python3 tools/wino24n_ab.py 2 > gpurun_out/${tag}_w24n_sq$i.log 2>&1 || exit 4
  python tools/pmc_sq.py gpurun_out/${tag}_w24n_sq$i wino24 >> gpurun_out/${tag}_w24n_counters.txt
done
echo done
