#!/bin/bash
# K8 + K6 alone (dyd_split_ids_seeded_dev on configs[2]'s 165 M records): plain timing, then the same command under the kernel
# trace.  Writes under gpurun_out/; the stats csv is copied to profiles/ by hand (tools/collect_profiles.py knows only bench.py).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/k8_bench.py --reps 7 ${K8ARGS:-} > $OUT/k8_bench.log 2> $OUT/k8_bench.err; rc=$?
echo "k8_bench rc=$rc"; tail -n 1 $OUT/k8_bench.log
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp
rm -rf $OUT/prof_k8
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k8 -- python3 $GRAFT_REPO_ROOT/tools/k8_bench.py --reps 7 ${K8ARGS:-} > $OUT/rocprof_k8.log 2>&1
echo "rocprof rc=$?"
rm -rf $OUT/prof_k8_prev; ls -t $(find $OUT/prof_k8 -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $OUT/k8_kernel_stats.csv
head -30 $OUT/k8_kernel_stats.csv | cut -c1-200
