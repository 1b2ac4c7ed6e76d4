#!/bin/bash
# every config's bench line into gpurun_out/<tag>_bench.jsonl (run on the GPU box)
tag=${1:-r02}
out=gpurun_out/${tag}_bench.jsonl
mkdir -p gpurun_out
: > $out
for c in 2 1 3 3u 4 2b nf 5 5f; do
  steps=10; warm=2
  case $c in 2b|nf|5|5f) steps=3; warm=1;; esac
  echo "== config $c" >&2
  timeout -k 10 900 python bench.py --config $c --steps $steps --warmup $warm 2> gpurun_out/${tag}_bench_$c.err | tail -1 >> $out || echo "{\"config\": \"$c\", \"failed\": true}" >> $out
done
cat $out | cut -c1-400
