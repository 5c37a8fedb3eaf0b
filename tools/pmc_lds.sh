#!/bin/bash
# development helper: LDS counters of the fused kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_lds -o lds --output-format csv -- python3 $R/tools/time_stages.py --reps 2 || exit 1
