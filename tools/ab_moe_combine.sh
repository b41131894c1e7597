#!/bin/bash
# A/B of the batch MoE combine fusion on one box: 8x7B Q4_K_M, steps of 8 / 64 / 512 tokens, with / without / with.
set -e
cd "$(dirname "$0")/.."
for run in a b c; do
  for p in 8 64 512; do
    if [ $run = b ]; then export GGML_MI355_NO_MOE_COMBINE=1; else unset GGML_MI355_NO_MOE_COMBINE; fi
    timeout -k 10 500 tools/run_llama_bench.sh 8x7b Q4_K_M -p $p -n 0 -r 3 2>>gpurun_out/comb.err | grep -o '"pp": [0-9]*, "pp_avg_ts": [0-9.]*' | sed "s/^/$run /" >> gpurun_out/comb.txt
  done
done
cat gpurun_out/comb.txt
