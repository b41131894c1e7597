#!/bin/bash
# A/B of rows-per-wave for the ncols=1 mat-vec
S="q4_K 4096 4096 1 q4_K 14336 4096 1 q4_K 4096 14336 1 q6_K 4096 14336 1 q6_K 128256 4096 1 q4_K 28672 8192 1 q8_0 14336 4096 1 q4_0 14336 4096 1 q5_K 14336 4096 1"
for r in 1 2 4; do echo "== RPW $r"; MI355_MMVQ_RPW=$r timeout -k 10 120 ./llm_amd/lib/mmvq_bench $S || exit 1; done
