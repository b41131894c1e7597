#!/bin/bash
# HBM traffic of the bench's dominant kernel from the PMC counters, in a run of its own (no trace domains beside
# --pmc): rocprofv3 --pmc FETCH_SIZE over an eager bench.py run.  FETCH_SIZE is reported in KiB and, on gfx950,
# counts 128-B requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section): the summary script
# doubles it.  usage (GPU box, repo root): tools/pmc_traffic.sh <outdir under gpurun_out>
out=${1:-pmc_bench}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/$out -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e --no-other-configs > $ROOT/gpurun_out/$out.log 2>&1
ls $ROOT/gpurun_out/$out/*/ | head
