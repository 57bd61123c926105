#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh <tag>
# Writes gpurun_out/<tag>_*: un-profiled default bench, config-5 bench, per-dtype rocprofv3 kernel stats, per-dtype PMC
# traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs), SQ counter passes of the fp32 step, the held-CU rehearsal.
set -o pipefail
tag=${1:-rXX}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --also bf16x3,bf16 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 1
# BASELINE.json configs[4] on one GPU (512x512, bs32 per GPU, bf16)
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --dtype bf16 --also "" --no-cpu-baseline --size 512 --batch 32 > gpurun_out/${tag}_bench_config5_bf16.json 2> gpurun_out/${tag}_bench_config5_bf16.err || exit 1
echo "bench done"
# Kernel traces with every kernel on ONE stream (CLAMD_WGRAD_STREAM=0): the same kernels with the same arguments as the
# two-stream step, but a kernel's begin-to-end time is its own (under the overlap it includes the time it shares the chip with
# a kernel of the other stream), so the averages agree with the HIP-event timings bench.py takes live.  The fp32 trace is
# repeated with the overlap on (<tag>_prof_fp32_overlap) to show the concurrency itself.
export CLAMD_WGRAD_STREAM=0
for dt in fp32 bf16x3 bf16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$dt -o $tag -- python3 bench.py --steps 6 --warmup 2 --dtype $dt --no-cpu-baseline --no-parity --also "" > gpurun_out/${tag}_bench_${dt}_under_rocprof.json 2> gpurun_out/${tag}_prof_$dt.err || exit 2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_${dt}_$c -o pmc -- python3 bench.py --steps 2 --warmup 1 --dtype $dt --no-cpu-baseline --no-parity --also "" --no-kernel-timing > gpurun_out/${tag}_pmc_${dt}_$c.json 2> gpurun_out/${tag}_pmc_${dt}_$c.err || exit 3
  done
  echo "done $dt"
done
unset CLAMD_WGRAD_STREAM
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_fp32_overlap -o $tag -- python3 bench.py --steps 6 --warmup 2 --dtype fp32 --no-cpu-baseline --no-parity --also "" > gpurun_out/${tag}_bench_fp32_overlap_under_rocprof.json 2> gpurun_out/${tag}_prof_fp32_overlap.err || exit 2
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_sq${i}_fp32 -o pmc -- python3 bench.py --steps 1 --warmup 1 --dtype fp32 --no-cpu-baseline --no-parity --also "" --no-kernel-timing > gpurun_out/${tag}_sq${i}_fp32.json 2> gpurun_out/${tag}_sq${i}_fp32.err || exit 4
done
# the same counter sets for the bf16 step (north_star: >= 40 % MFMA utilisation on the 3x3 convolutions at bf16, evidenced by counters)
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_sq${i}_bf16 -o pmc -- python3 bench.py --steps 1 --warmup 1 --dtype bf16 --no-cpu-baseline --no-parity --also "" --no-kernel-timing > gpurun_out/${tag}_sq${i}_bf16.json 2> gpurun_out/${tag}_sq${i}_bf16.err || exit 4
done
echo "sq done"
# where the matrix pipes idle: two-stream kernel traces of the shipped step
for dt in fp32 bf16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_gaps_$dt -o g -- python3 bench.py --steps 4 --warmup 1 --dtype $dt --no-cpu-baseline --no-parity --also "" --no-kernel-timing > gpurun_out/${tag}_gaps_$dt.json 2> gpurun_out/${tag}_gaps_$dt.err || exit 6
  python tools/trace_gaps.py gpurun_out/${tag}_gaps_$dt/g_kernel_trace.csv 2 24 > gpurun_out/${tag}_trace_gaps_$dt.txt
done
for a in "bf16 8" "fp32 8" "bf16x3 8" "bf16 16" "fp32 16"; do GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/cu_steal.py $a 2>/dev/null | grep '^{' >> gpurun_out/${tag}_cu_steal.jsonl || exit 5; done
python tools/layer_table.py fp32 > gpurun_out/${tag}_layers_fp32.txt 2>/dev/null
python tools/layer_table.py bf16 > gpurun_out/${tag}_layers_bf16.txt 2>/dev/null
python tools/layer_table.py bf16x3 > gpurun_out/${tag}_layers_bf16x3.txt 2>/dev/null
echo "all done"
