#!/bin/bash
# Runs the REFERENCE's own conformance harness (oracle/_ref/test-backend-ops, built from the
# reference's sources by oracle/Makefile) against the MI355 plugin.  usage: run_backend_ops.sh <mode> [args...]
cd "$(dirname "$0")/.."
export GGML_BACKEND_PATH=$PWD/llm_amd/lib/libggml-mi355.so
exec ./oracle/_ref/test-backend-ops "$@"
