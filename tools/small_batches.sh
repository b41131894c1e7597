#!/bin/bash
# pp<N> of one random model for a list of step sizes, one driver run each; prints "N t/s".
# usage: tools/small_batches.sh <model> <ftype> "8 16 32" [reps]
model=$1; ftype=$2; reps=${4:-5}
for p in $3; do
    out=$(timeout -k 10 300 tools/run_llama_bench.sh $model $ftype -p $p -n 0 -r $reps 2>/dev/null | grep -o '"pp_avg_ts": [0-9.]*') || exit 1
    echo "$p ${out#*: }"
done
