#!/bin/bash
# tools/valu_busy_calib.sh — calibrate the hardware's own VALU-busy counter (SQ_ACTIVE_INST_VALU) on loops of known
# cost (tools/valu_microbench.hip), so that the trace kernels' "VALU issue fraction" has a confirmation that does not
# go through SQ_INSTS_VALU x a cost table.  Run on the GPU box through gpurun.
ROOT=$(dirname $(dirname $(readlink -f $0)))
OUT=$ROOT/gpurun_out/calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 $ROOT/tools/valu_microbench.hip -o $OUT/valu_microbench || exit 1
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "SQ_INST_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_INST_CYCLES_SALU\|SQ_ACTIVE_INST_SCA" $OUT/avail.txt | sort | uniq -c
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/pmc -- $OUT/valu_microbench > $OUT/microbench.json 2> $OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
if grep -q SQ_INST_CYCLES_VALU $OUT/avail.txt; then
  rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $OUT/valu_microbench > $OUT/microbench2.json 2> $OUT/pmc2.err || { tail -5 $OUT/pmc2.err; exit 1; }
fi
python3 $ROOT/tools/valu_busy_calib.py $OUT
