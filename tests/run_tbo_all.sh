#!/bin/bash
# run the reference's test-backend-ops (mode test) for every op the plugin claims; summary per op
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rc=0
for op in MUL_MAT MUL_MAT_ID FLASH_ATTN_EXT ADD MUL SUB DIV RMS_NORM ROPE GLU CPY CONT DUP GET_ROWS SCALE SOFT_MAX; do
    timeout -k 10 900 tests/run_backend_ops.sh test -o $op > gpurun_out/tbo_$op.log 2>&1
    r=$?
    ok=$(grep -c "OK$\|OK.\[0m$" gpurun_out/tbo_$op.log)
    ns=$(grep -c "not supported" gpurun_out/tbo_$op.log)
    fail=$(grep -c "FAIL" gpurun_out/tbo_$op.log)
    echo "$op rc=$r ok_lines=$ok not_supported=$ns fail_lines=$fail"
    [ $r -ne 0 ] && rc=1 && grep "FAIL\|error\|abort" gpurun_out/tbo_$op.log | head -8
    if [ $r -eq 134 ] || [ $r -eq 139 ] || [ $r -eq 124 ] || [ $r -eq 137 ]; then echo "stopping after abort/timeout"; exit 1; fi
done
exit $rc
