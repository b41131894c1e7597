#!/bin/bash
# SQ counters of the kernels whose name contains <pattern>, two --pmc passes in runs of their own (no trace domains).
# usage (GPU box, repo root): tools/pmc_kernel.sh <pattern> <tag> <python script and its arguments...>
ROOT=$PWD
pat=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_a -- python3 "$@" > $ROOT/gpurun_out/pmc_${tag}_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_b -- python3 "$@" > $ROOT/gpurun_out/pmc_${tag}_b.log 2>&1 || exit 1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for p in ("a", "b"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$pat" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items()): print("%-28s per launch %14.0f  (%d samples)" % (k, v / n, n))
PY
