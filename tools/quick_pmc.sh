#!/bin/bash
# dev helper, runs on the GPU box: PMC passes of `python3 $3...` for kernels matching $1 -> gpurun_out/qp_$2.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; K=$1; T=${2:-x}; shift; shift; O=/tmp/qp_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc --output-format csv -- python3 "$@" > $R/gpurun_out/qp_$T.p$i.log 2>&1
done
cd $R
python3 tools/pmc_kernel.py "$K" $O/p1/pmc_counter_collection.csv $O/p2/pmc_counter_collection.csv $O/p3/pmc_counter_collection.csv $O/p4/pmc_counter_collection.csv > gpurun_out/qp_$T.txt 2>&1
cat gpurun_out/qp_$T.txt
