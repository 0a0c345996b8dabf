#!/bin/bash
# A/B: worker threads of the native pass against the cgroup's CPU share (default = the quota, 16 on a one-GPU box)
cd $GRAFT_REPO_ROOT
cat /sys/fs/cgroup/cpu.max
for t in 16 24 32 12; do
  DYD_HOST_THREADS=$t timeout -k 10 300 python tools/e2e_bench.py --rows 1000000 --reps 2 --check 0 2>/dev/null | grep rows_per_s | grep -v text_dtype | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('threads $t', d['seconds'], d['phases_s']['s_pipeline'], d['phases_s']['s_strings'], d['phases_s']['s_frame_in'])" || exit 1
done
