#!/bin/bash
# A/B: image rows per wave of the sparse fused wave kernel (KW_ROWS, k12_wave.h): builds of 8 / 16 / 32 under build_exp/ 
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for r in 16 8 32; do
    lib=$GRAFT_REPO_ROOT/deal-yolo-daya_amd/libdyd_gfx950.so
    [ $r != 16 ] && lib=$GRAFT_REPO_ROOT/build_exp/libdyd_r$r.so
    for w in c3 c2; do
      DYD_LIB_PATH=$lib timeout -k 10 200 python bench.py --workload $w --host-rows 0 --cpu-sample 0 --pipeline 0 --steps 60 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rows/wave $r', '$w', round(d['ms_per_step'], 4), round(d['roofline']['frac'], 4))" || exit 1
    done
  done
done
