#!/bin/bash
# BASELINE.json configs[4] on ONE GPU (512x512, bs32 per GPU, bf16): the evidence profile_round.sh collects for configs[1], at this workload
# (run through gpurun from the repo root):  bash tools/profile_config5.sh <tag>
# Writes gpurun_out/<tag>_c5_*: un-profiled bench line, single-stream kernel stats, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate
# runs), two-stream trace gaps, the per-layer table.  tools/make_profiles_c5.py turns them into profiles/<round>_*_512.*
set -o pipefail
tag=${1:-rXX}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
W="--dtype bf16 --size 512 --batch 32 --no-cpu-baseline --no-parity --also \"\""
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --dtype bf16 --also "" --no-cpu-baseline --size 512 --batch 32 > gpurun_out/${tag}_c5_bench.json 2> gpurun_out/${tag}_c5_bench.err || exit 1
echo "bench done"
export CLAMD_WGRAD_STREAM=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c5_prof -o $tag -- python3 bench.py --steps 4 --warmup 2 --dtype bf16 --size 512 --batch 32 --no-cpu-baseline --no-parity --also "" > gpurun_out/${tag}_c5_bench_under_rocprof.json 2> gpurun_out/${tag}_c5_prof.err || exit 2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_c5_pmc_$c -o pmc -- python3 bench.py --steps 2 --warmup 1 --dtype bf16 --size 512 --batch 32 --no-cpu-baseline --no-parity --also "" --no-kernel-timing > gpurun_out/${tag}_c5_pmc_$c.json 2> gpurun_out/${tag}_c5_pmc_$c.err || exit 3
done
unset CLAMD_WGRAD_STREAM
echo "pmc done"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_c5_gaps -o g -- python3 bench.py --steps 3 --warmup 1 --dtype bf16 --size 512 --batch 32 --no-cpu-baseline --no-parity --also "" --no-kernel-timing > gpurun_out/${tag}_c5_gaps.json 2> gpurun_out/${tag}_c5_gaps.err || exit 6
python tools/trace_gaps.py gpurun_out/${tag}_c5_gaps/g_kernel_trace.csv 2 24 > gpurun_out/${tag}_c5_trace_gaps.txt
python tools/layer_table.py bf16 512 32 > gpurun_out/${tag}_c5_layers.txt 2>/dev/null
echo "all done"
