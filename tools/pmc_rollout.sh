#!/usr/bin/env bash
# SQ counter passes over one rollout workload (run on the GPU box):  tools/pmc_rollout.sh OUTDIR "quick_rollout args"
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
    --kernel-trace -d "$out/pmc1" -o p1 --output-format csv -- python3 tools/quick_rollout.py "$@" > "$out/pmc1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
    --kernel-trace -d "$out/pmc2" -o p2 --output-format csv -- python3 tools/quick_rollout.py "$@" > "$out/pmc2.log" 2>&1
find "$out" -name "*.csv" | head
