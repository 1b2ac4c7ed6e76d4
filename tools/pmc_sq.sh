#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <size-mib> [config] [commit] — the three SQ counter passes only (instruction mix, waits, LDS)
# -> gpurun_out/pmc_<tag>.txt (every counter per kernel) and gpurun_out/pmc_issue_<tag>.json (the issue roof: bench.py reads
#    profiles/pmc_issue.json)
TAG=$1; SZ=${2:-1024}; CFG=${3:-2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config $CFG --size-mib $SZ --corpus-cache /tmp/tbz_corpus_$(id -u) --gen-only || exit 1  # (forks: before the profiler)
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
         "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python3 $R/bench.py --config $CFG --corpus-cache /tmp/tbz_corpus_$(id -u) --steps 1 --warmup 1 --size-mib $SZ --no-cpu-baseline --no-h2h > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.log; }
done
python3 $R/tools/pmc_summary.py "$R/gpurun_out/pmc_${TAG}_*/*.db" > $R/gpurun_out/pmc_${TAG}.txt 2>&1
python3 $R/tools/pmc_issue.py "$R/gpurun_out/pmc_${TAG}_*/*.db" $R/gpurun_out/pmc_issue_${TAG}.json $CFG ${4:-unknown} $(date -u +%Y-%m-%d)
