#!/bin/bash
# PMC counter passes for one kbench selection.  usage: KB=k2 bash tools/gpu_pmc.sh
set -u
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -c . $OUT/counters_list.txt
run_pass() { # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --rows ${ROWS:-1000000} --only ${KB:-k2} --iters 2 ${KBARGS:-} > $OUT/pmc_$name.log 2>&1
  echo "pass $name rc=$?"
}
run_pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
run_pass b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run_pass c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
find $OUT -name "*counter_collection.csv" | head
