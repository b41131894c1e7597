#!/bin/bash
# per-dispatch durations of the MUL_MAT_ID GEMM launches of one 8x7B 512-token step, grouped by kernel and grid (gate / up: 112 row tiles, down: 32)
cd "$(dirname "$0")/.."
gguf=/tmp/random-8x7b-Q4_K_M.gguf
[ -f $gguf ] || ./oracle/_ref/make-random-gguf $gguf 8x7b Q4_K_M || exit 1
export GGML_BACKEND_PATH=$PWD/llm_amd/lib/libggml-mi355.so
export TMPDIR=/tmp
rm -rf /tmp/moetrace
echo "== MI355_MMDQ_DBG=$MI355_MMDQ_DBG"; timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/moetrace -- ./oracle/_ref/mi355-llama-bench -m $gguf -p 512 -n 0 -r 1 > gpurun_out/trace_run.txt 2>&1 || exit 1
f=$(find /tmp/moetrace -name "*kernel_trace.csv" | head -1)
head -1 $f > gpurun_out/trace_header.txt
python3 - "$f" <<'PY' >> gpurun_out/trace_moe.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "mmdq_ws_kernel" not in name and "mmdq_kernel" not in name: continue
    key = (name.split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(acc.items()):
    v.sort()
    print(k, "n", len(v), "min %.1f med %.1f max %.1f" % (v[0], v[len(v) // 2], v[-1]))
PY
cat gpurun_out/trace_moe.txt
