#!/bin/bash
# usage (on the GPU box, from the repo root): tools/trace/run.sh <k1|k2|k0b> <tag> [ENV=VAL ...] -- [bench args]
# runs bench.py on the trace build and prints what the per-workgroup records say:
#   k1   TBZ_WAVE_TRACE  every K1g workgroup: lifetime, header / build / rounds, wave trips and lane iterations
#   k2   TBZ_K2_TRACE    the first 8192 workgroups of tbz_k2_lz77_dual: batches, barrier waits of both waves
#   k0b  TBZ_VAL_TRACE   tbz_k0b_validate: lifetime and symbol-loop trips per tile
kind=$1; tag=$2; shift 2
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
case $kind in k1) var=TBZ_WAVE_TRACE;; k2) var=TBZ_K2_TRACE;; k0b) var=TBZ_VAL_TRACE;; *) echo "k1|k2|k0b"; exit 2;; esac
mkdir -p gpurun_out
env "${envs[@]}" $var=gpurun_out/trace_$tag.bin timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify \
    --lib "$PWD/3bz_amd/lib3bz_trace.so" "$@" > gpurun_out/trace_$tag.log 2>&1 || { tail -3 gpurun_out/trace_$tag.log; exit 1; }
echo "== $kind $tag ${envs[*]} $*"
python tools/trace/${kind}_trace.py gpurun_out/trace_$tag.bin
