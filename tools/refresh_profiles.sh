#!/bin/bash
# usage (on the GPU box, from the repo root): tools/refresh_profiles.sh <tag> [commit the working tree was built from]
# Writes gpurun_out/<tag>_{bench.jsonl,kernel_stats.txt,pmc_raw.json,pmc_traffic.json}; copy them into profiles/.
TAG=$1
COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
# the headline line as a driver gets it: the plain default command (every other config attached as other_configs).  The
# profiler passes below read their workload from a private cache, generated here WITHOUT the profiler (the generators fork)
python3 $R/bench.py > $O/${TAG}_bench.jsonl 2> $O/${TAG}_bench.err || { echo "bench failed"; tail -5 $O/${TAG}_bench.err; exit 1; }
python3 $R/bench.py --corpus-cache /tmp/tbz_corpus_$(id -u) --gen-only || exit 1
tail -c 600 $O/${TAG}_bench.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o s -- python3 $R/bench.py --corpus-cache /tmp/tbz_corpus_$(id -u) --steps 5 --warmup 1 --no-cpu-baseline --no-h2h > $O/${TAG}_stats.log 2>&1 || { echo "stats pass failed"; tail -3 $O/${TAG}_stats.log; exit 1; }
python3 $R/tools/prof_summary.py $(ls $O/${TAG}_stats/*results.db $O/${TAG}_stats/*/*results.db 2>/dev/null | head -1) $O/${TAG}_kernel_stats.txt > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}_pmc_$C -o p -- python3 $R/bench.py --corpus-cache /tmp/tbz_corpus_$(id -u) --steps 2 --warmup 1 --no-cpu-baseline --no-h2h > $O/${TAG}_pmc_$C.log 2>&1 || { echo "pmc $C failed"; tail -3 $O/${TAG}_pmc_$C.log; exit 1; }
done
F=$(ls $O/${TAG}_pmc_FETCH_SIZE/*results.db $O/${TAG}_pmc_FETCH_SIZE/*/*results.db 2>/dev/null | head -1)
W=$(ls $O/${TAG}_pmc_WRITE_SIZE/*results.db $O/${TAG}_pmc_WRITE_SIZE/*/*results.db 2>/dev/null | head -1)
python3 $R/tools/pmc_traffic.py $F $W $O/${TAG}_pmc_raw.json $O/${TAG}_pmc_traffic.json 561175286 1073741824 $COMMIT $(date -u +%Y-%m-%d) | tail -2
cat $O/${TAG}_kernel_stats.txt
