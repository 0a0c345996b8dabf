#!/bin/bash
# One gpurun call: parity tests -> smoke -> per-kernel micro-bench -> host-inclusive step bench -> bench.py -> rocprofv3 stats
# (PMC traffic passes: tools/gpu_profile.sh; summaries into profiles/: tools/collect_profiles.py <tag>).
# A step killed by its timeout (rc >= 124) ends the script: never start GPU work after a hang.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "== $name"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc"; tail -n 12 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name timed out; stopping"; exit $rc; fi
  return $rc
}
step pytest_gpu 500 python -m pytest tests -m gpu -q -x || exit 1
step smoke 200 python __graft_entry__.py smoke || exit 1
step kbench 300 python tools/kbench.py --rows 1000000 --only k1,k2,k12,k3,k4,k5,k6,k7
step step_bench 600 python tools/step_bench.py --rows 20000
step bench 400 python bench.py
cd /tmp
step_rocprof() {
  echo "== rocprof"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_bench" -- python3 "$GRAFT_REPO_ROOT/bench.py" --cpu-sample 0 > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log" 2>&1
  echo "== rocprof rc=$?"; tail -n 5 "$GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log"
}
step_rocprof
find "$GRAFT_REPO_ROOT/gpurun_out/prof_bench" -name "*stats*" | head
