#!/bin/bash
# A/B: workgroup -> row-range mapping of the sparse fused kernel.  the product gives every XCD one contiguous eighth of the table;
# build_exp/libdyd_noxcd.so is k12_fused.hip with -DK12_NO_XCD_REMAP (workgroup b takes piece b)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in product plain; do
    lib=$GRAFT_REPO_ROOT/deal-yolo-daya_amd/libdyd_gfx950.so
    [ $v = plain ] && lib=$GRAFT_REPO_ROOT/build_exp/libdyd_noxcd.so
    DYD_LIB_PATH=$lib timeout -k 10 200 python bench.py --host-rows 0 --cpu-sample 0 --pipeline 0 --steps 60 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'], 4), round(d['roofline']['frac'], 4), d['config']['high_rows_rank0'])" || exit 1
  done
done
