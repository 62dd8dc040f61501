#!/bin/bash
# Stall / occupancy counters per kernel of one serial scan: six separate rocprofv3 --pmc passes (never combined with other traces).
# usage (GPU box): bash tools/pmc_stalls.sh <tag>   -> gpurun_out/<tag>/pmc_stalls.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-stalls}; o=gpurun_out/$tag; mkdir -p $o
pass() { n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $o/p$n -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-h2d --streams 1 --pipelined-geometry > /dev/null 2> $o/p$n.err || echo "pass $n failed"; }
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
pass 3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass 4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
pass 5 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass 6 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
python3 tools/pmc_kernels2.py $o/p1 $o/p2 $o/p3 $o/p4 $o/p5 $o/p6 > $o/pmc_stalls.txt 2> $o/agg.err
rm -rf $o/p1 $o/p2 $o/p3 $o/p4 $o/p5 $o/p6
# the budget table + the JSON bench.py attaches to its line (`roofline.vector_pipe_us_per_scan`) for this build
python3 tools/vector_pipe_budget.py $o/pmc_stalls.txt $o/vector_pipe_budget.txt $o/vector_pipe_budget.json && cp $o/vector_pipe_budget.json profiles/vector_pipe_budget.json
head -50 $o/vector_pipe_budget.txt
