#!/bin/bash
# SQ counters of every kernel of one vo_frames_batch_dev call (tools/batch_frames.py): tools/pmc_frames.sh <tag> [F]
# two --pmc passes (the counters do not fit one); table by tools/summarize_sq.py
set -e
TAG=${1:-sqf}
F=${2:-200}
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python3 $R/tools/batch_frames.py $F > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/b -- python3 $R/tools/batch_frames.py $F > $OUT/b.log 2>&1
cd $R
python3 $R/tools/summarize_sq.py $OUT
