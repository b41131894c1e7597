#!/bin/bash
# rocprofv3 kernel stats of an end-to-end run. usage: prof_llama.sh <outdir> <model> <ftype> [driver args]
cd "$(dirname "$0")/.."
ROOT=$PWD
out=$1; model=$2; ftype=$3; shift 3
gguf=/tmp/random-$model-$ftype.gguf
[ -f $gguf ] || ./oracle/_ref/make-random-gguf $gguf $model $ftype || exit 1
export GGML_BACKEND_PATH=$ROOT/llm_amd/lib/libggml-mi355.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out -- $ROOT/oracle/_ref/mi355-llama-bench -m $gguf "$@"
