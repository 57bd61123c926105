#!/bin/bash
# Runs a list of measurement / test steps on the GPU box in ONE gpurun call (getting a box costs minutes, so steps are batched):
#     bash tools/gpu_batch.sh <tag> "<seconds> <command ...>" "<seconds> <command ...>" ...
# Each step runs under `timeout -k 10 <seconds>` with its output in gpurun_out/<tag>_<n>.log.  An ordinary failure (a failing test) does not
# stop the chain; a step that hits its time limit or is killed does -- nothing further is started on a GPU that may be hung.
tag=$1; shift
mkdir -p gpurun_out
n=0
for step in "$@"; do
    n=$((n + 1))
    secs=${step%% *}; cmd=${step#* }
    log=gpurun_out/${tag}_${n}.log
    echo "== [$n] $cmd" | tee "$log"
    timeout -k 10 "$secs" bash -c "$cmd" >> "$log" 2>&1
    rc=$?
    echo "== [$n] rc=$rc" | tee -a "$log"
    tail -n 4 "$log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n hit its limit: stopping the batch"; exit $rc; fi
done
exit 0
