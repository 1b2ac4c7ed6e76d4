#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_config.sh <tag> <config> [extra bench args]
# rocprofv3 --kernel-trace --stats of one bench config -> gpurun_out/<tag>_kernel_stats_<config>.txt
TAG=$1; CFG=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the workload is generated (worker processes fork) by a run WITHOUT the profiler, whose library initialises the GPU before Python starts
python3 $R/bench.py --config $CFG --corpus-cache /tmp/tbz_corpus_$(id -u) --gen-only "$@" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_$CFG -o s -- python3 $R/bench.py --config $CFG --corpus-cache /tmp/tbz_corpus_$(id -u) --steps 3 --warmup 1 --no-cpu-baseline --no-h2h "$@" > $O/${TAG}_stats_$CFG.log 2>&1 || { echo "stats pass failed"; tail -3 $O/${TAG}_stats_$CFG.log; exit 1; }
tail -1 $O/${TAG}_stats_$CFG.log | cut -c1-200
python3 $R/tools/prof_summary.py $(ls $O/${TAG}_stats_$CFG/*results.db $O/${TAG}_stats_$CFG/*/*results.db 2>/dev/null | head -1) $O/${TAG}_kernel_stats_$CFG.txt
rm -rf $O/${TAG}_stats_$CFG
