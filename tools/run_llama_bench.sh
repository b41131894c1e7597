#!/bin/bash
# end-to-end tg/pp through the reference's libllama with the MI355 plugin.
# usage: run_llama_bench.sh <8b|70b|tiny> <ftype> [driver args...]
cd "$(dirname "$0")/.."
model=$1; ftype=$2; shift 2
gguf=/tmp/random-$model-$ftype.gguf
[ -f $gguf ] || ./oracle/_ref/make-random-gguf $gguf $model $ftype || exit 1
# MI355_NO_PLUGIN=1: ggml-cpu alone.  (With the plugin loaded, "-ngl 0" is not a pure CPU run: the scheduler hands the MUL_MAT /
# MUL_MAT_ID of batches >= 32 tokens to the device through offload_op.)
[ -n "$MI355_NO_PLUGIN" ] || export GGML_BACKEND_PATH=$PWD/llm_amd/lib/libggml-mi355.so
exec ./oracle/_ref/mi355-llama-bench -m $gguf "$@"
