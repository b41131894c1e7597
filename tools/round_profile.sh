#!/bin/bash
# One GPU call that refreshes the evidence under gpurun_out/: bench line, rocprofv3 kernel stats of
# bench.py, and kernel stats of end-to-end decode (tg) and prefill (pp) of the random 8B Q4_K_M model.
# usage (on the GPU box, repo root): tools/round_profile.sh <tag>
tag=${1:-r01}
ROOT=$PWD
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
cat gpurun_out/${tag}_bench.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${tag}_prof_bench -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-other-configs --no-graph > $ROOT/gpurun_out/${tag}_bench_prof.json 2>$ROOT/gpurun_out/${tag}_bench_prof.err) || exit 1
timeout -k 10 400 tools/run_llama_bench.sh 8b Q4_K_M -p 512 -n 128 -r 3 > gpurun_out/${tag}_llama8b.log 2>&1 || { tail -5 gpurun_out/${tag}_llama8b.log; exit 1; }
tail -4 gpurun_out/${tag}_llama8b.log
timeout -k 10 300 tools/prof_llama.sh gpurun_out/${tag}_prof_tg 8b Q4_K_M -p 0 -n 64 -r 1 > gpurun_out/${tag}_prof_tg.log 2>&1 || exit 1
timeout -k 10 300 tools/prof_llama.sh gpurun_out/${tag}_prof_pp 8b Q4_K_M -p 512 -n 0 -r 2 > gpurun_out/${tag}_prof_pp.log 2>&1 || exit 1
echo done
