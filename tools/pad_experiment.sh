#!/bin/bash
# tools/pad_experiment.sh — is C2's sample kernel bound by VALU issue?  Variants of the library with PT_EXP_PAD = 4 / 32 /
# 64 / 128 extra `v_or_b32 x, x, x` per loop iteration of pt_samples_q (tools/build_variant.sh padN -DPT_EXP_PAD=N; every
# variant carries the asm statement, so they share one code generation) are timed (rocprofv3 --kernel-trace) and counted
# (SQ_INSTS_VALU); the instruction's stand-alone cost comes from tools/valu_microbench.hip.
# Run on the GPU box through gpurun; tools/pad_experiment.py reads the results.
ROOT=$(dirname $(dirname $(readlink -f $0)))
OUT=$ROOT/gpurun_out/pad
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 $ROOT/tools/valu_microbench.hip -o $OUT/valu_microbench || exit 1
$OUT/valu_microbench > $OUT/microbench.json || exit 1
for n in 4 32 64 128; do
  lib=$ROOT/opencl-raytracing_amd/variants/pad$n.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$n -- python3 $ROOT/tools/quick_bench.py --lib=$lib c2:64 > $OUT/t$n.log 2> $OUT/t$n.err || { tail -3 $OUT/t$n.err; exit 1; }
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc$n -- python3 $ROOT/tools/quick_bench.py --lib=$lib c2:64 > $OUT/p$n.log 2> $OUT/p$n.err || { tail -3 $OUT/p$n.err; exit 1; }
  grep -h kernel $OUT/t$n.log
done
python3 $ROOT/tools/pad_experiment.py $OUT
