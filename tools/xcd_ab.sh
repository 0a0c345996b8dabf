#!/bin/bash
# A/B: workgroup -> row-range mapping of the fused wave kernels.  In the product workgroup b takes piece b of the table;
# build_exp/libdyd_xcd.so is k12_fused.hip with -DK12_XCD_REMAP (every XCD streams one contiguous eighth).  WORKLOADS="c3 c5" by default.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for w in ${WORKLOADS:-c3 c5}; do
    for v in product xcd; do
      lib=$GRAFT_REPO_ROOT/deal-yolo-daya_amd/libdyd_gfx950.so
      [ $v = xcd ] && lib=$GRAFT_REPO_ROOT/build_exp/libdyd_xcd.so
      DYD_LIB_PATH=$lib timeout -k 10 200 python bench.py --workload $w --host-rows 0 --cpu-sample 0 --pipeline 0 --steps 40 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w', '$v', round(d['ms_per_step'], 4), round(d['roofline']['frac'], 4))" || exit 1
    done
  done
done
