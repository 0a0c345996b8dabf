#!/bin/bash
# Host C++ (host_json.cpp, host_csv.cpp) under AddressSanitizer + UBSan on the CPU: builds build_exp/libdyd_asan.so
# (device objects as built by `make`, host objects instrumented) and runs the host test modules against it.
set -eu
cd "$(dirname "$0")/.."
make -C deal-yolo-daya_amd/csrc -s
mkdir -p build_exp
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for f in host_json host_csv; do
  $HIPCC -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -pthread \
    -c deal-yolo-daya_amd/csrc/$f.cpp -o build_exp/${f}_asan.o
done
OBJS=$(ls deal-yolo-daya_amd/csrc/*.o | grep -v host_)
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -fsanitize=address,undefined -o build_exp/libdyd_asan.so $OBJS build_exp/host_json_asan.o build_exp/host_csv_asan.o
ASAN=$($HIPCC -print-file-name=libclang_rt.asan-x86_64.so)
rm -f build_exp/asan.log.* build_exp/ubsan.log.*
# the reports go to files: a sanitizer abort inside a ctypes call ends the interpreter before pytest can show anything
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=$PWD/build_exp/asan.log \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$PWD/build_exp/ubsan.log \
  DYD_LIB_PATH=$PWD/build_exp/libdyd_asan.so python -m pytest tests/test_native_json_cpu.py tests/test_fastcsv_cpu.py \
  tests/test_merge_cpu.py tests/test_host_steps_cpu.py tests/test_yolo_host_cpu.py tests/test_label_replace_cpu.py tests/test_split_cpu.py \
  tests/test_pycells_cpu.py -x -q \
  || { head -n 12 build_exp/asan.log.* build_exp/ubsan.log.* 2>/dev/null; exit 1; }
