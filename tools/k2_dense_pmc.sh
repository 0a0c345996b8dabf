#!/bin/bash
# PMC passes over tools/k2_dense_probe.py (one mode at a time so that a kernel name maps to one mode).  Output under gpurun_out/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in ${MODES:-flag98 max}; do
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmcd_${mode}_a -- python3 $GRAFT_REPO_ROOT/tools/k2_dense_probe.py --modes $mode --iters 2 ${PROBE_ARGS:-} > $OUT/pmcd_${mode}_a.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmcd_${mode}_b -- python3 $GRAFT_REPO_ROOT/tools/k2_dense_probe.py --modes $mode --iters 2 ${PROBE_ARGS:-} > $OUT/pmcd_${mode}_b.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmcd_${mode}_c -- python3 $GRAFT_REPO_ROOT/tools/k2_dense_probe.py --modes $mode --iters 2 ${PROBE_ARGS:-} > $OUT/pmcd_${mode}_c.log 2>&1 || echo "pass c failed"
done
ls $OUT | grep pmcd
