#!/bin/bash
# usage: tools/pmc_run.sh <tag> <size-mib> ; runs from the repo root on the GPU box.
# Separate rocprofv3 --pmc passes (8 SQ slots / 4 TCC slots per pass; FETCH_SIZE and WRITE_SIZE do not fit together),
# each with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
TAG=$1; SZ=${2:-256}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --size-mib $SZ --corpus-cache /tmp/tbz_corpus_$(id -u) --gen-only || exit 1  # (forks: before the profiler)
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_INSTS_BRANCH" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python3 $R/bench.py --corpus-cache /tmp/tbz_corpus_$(id -u) --steps 1 --warmup 1 --size-mib $SZ --no-cpu-baseline --no-h2h > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.log; }
done
ls $R/gpurun_out/ | head -30
