#!/bin/bash
# bench.py (default command: 10M resident rows) + its rocprofv3 evidence.  Writes everything under gpurun_out/;
# tools/collect_profiles.py <tag> turns it into the committed summaries under profiles/.
#   usage on the GPU box:  bash tools/gpu_profile.sh        (BENCH_ARGS="--workload c2" for another workload)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py ${BENCH_ARGS:-} > $OUT/bench.log 2> $OUT/bench.err; rc=$?
echo "bench rc=$rc"; tail -n 1 $OUT/bench.log | cut -c1-600
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp
# the same command under the kernel trace (stats = per-kernel calls / average duration).  --host-rows 0 --cpu-sample 0: the
# host-inclusive leg launches the SAME fused kernel on 16 small shares of a 1M-row table, which would pull the kernel's average
# duration away from the 10M-row launches the roofline is quoted on; everything on the device stays (ramp, warm-up, the K timed
# steps and the full-pipeline stages: every k12_wave_kernel call is a 10M-row launch)
rm -rf $OUT/prof_bench $OUT/pmc_*
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --host-rows 0 --cpu-sample 0 --strong-rows 0 ${BENCH_ARGS:-} > $OUT/rocprof_bench.log 2>&1
echo "rocprof stats rc=$?"
# counters: their own runs, kernel trace off, one TCC counter per pass (FETCH_SIZE and WRITE_SIZE do not fit together); the
# device-resident part of the command only (the host-side legs launch the same kernels on other table sizes)
SHORT="--steps 3 --warmup 1 --ramp-ms 0 --cpu-sample 0 --host-rows 0 --pipeline 0 --dense 0 --strong-rows 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py $SHORT ${BENCH_ARGS:-} > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_SQ -- python3 $GRAFT_REPO_ROOT/bench.py $SHORT ${BENCH_ARGS:-} > $OUT/pmc_SQ.log 2>&1
echo "pmc SQ rc=$?"
# the dense table of configs[4] (the `dense` object of the default line draws the same table): FETCH_SIZE / WRITE_SIZE of its launch
if [ -z "${BENCH_ARGS:-}" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_c5_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 $SHORT > $OUT/pmc_c5_$c.log 2>&1
    echo "pmc c5 $c rc=$?"
  done
fi
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_SQ2 -- python3 $GRAFT_REPO_ROOT/bench.py $SHORT ${BENCH_ARGS:-} > $OUT/pmc_SQ2.log 2>&1
echo "pmc SQ2 rc=$?"
ls $OUT
