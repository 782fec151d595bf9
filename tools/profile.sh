#!/bin/bash
# rocprofv3 passes over the default bench workload (run on the GPU box through gpurun).
# Kernel trace and every --pmc group are SEPARATE runs (MI355X_MICROARCH.md: TCC slots; gpurun
# refuses --pmc combined with other trace domains).  Outputs land in gpurun_out/prof/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
STEPS=${STEPS:-10}
CMD="python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-caller-levels --no-reference-binning ${BENCH_ARGS:-}"
# the kernel trace on a LONGER run of the same command: the shader clock needs about a second of work to reach its sustained level
# (profiles/r04_clock_ramp.txt), and the ALU-bound blend kernels of a 30-launch run average 7 % above what bench.py's own
# HIP events measure in its timed region; counters are per-launch counts and cycles, which do not depend on the clock
TRACE_CMD="python3 $R/bench.py --steps ${TRACE_STEPS:-150} --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE_CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- $CMD > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- $CMD > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
