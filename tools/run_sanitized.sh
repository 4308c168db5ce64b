#!/bin/bash
# Host-side sanitizer passes (AddressSanitizer + UndefinedBehaviorSanitizer), the counterpart of the reference's
# sanitizer targets (src/Makefile:214-238).  `make -C c-kzg-4844_amd sanitize` first.
#   bash tools/run_sanitized.sh cpu   the host arithmetic (field, curve, pairing, GLV, safegcd) through the sanitized
#                                     host shim + the ABI tests on the sanitized product (no GPU needed)
#   bash tools/run_sanitized.sh tsan  on the GPU box (`make -C c-kzg-4844_amd tsan` first): ThreadSanitizer over the host
#                                     side while 6 threads hammer one async-loaded KZGSettings (tools/stress_gpu.py) and
#                                     the async / fan-out tests run
#   bash tools/run_sanitized.sh gpu   on the GPU box: the consensus-spec vectors, the malformed-input fuzz suite and the
#                                     round-3 verification tests through the sanitized libckzg_hip_san.so
# Output: gpurun_out/sanitize_<mode>.log ; exit code 1 if a sanitizer report appears.
cd "$(dirname "$0")/.."
mode=${1:-cpu}
mkdir -p gpurun_out
LOG=gpurun_out/sanitize_$mode.log
: > $LOG
# GCC's runtimes for both libraries (c-kzg-4844_amd/Makefile says why not clang's on a GPU box)
GCC_RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
CLANG_RT=$GCC_RT
export ASAN_OPTIONS=detect_leaks=0:use_sigaltstack=0:abort_on_error=0:detect_odr_violation=0   # use_sigaltstack=0: the HIP runtime's threads trip ASan's alternate-stack teardown
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
if [ "$mode" = tsan ]; then
  export TSAN_OPTIONS="report_signal_unsafe=0:suppressions=$PWD/tools/tsan.supp:history_size=4:exitcode=0"
  # clang's own TSan runtime here (it has no HSA interceptors, unlike its ASan runtime, and copes with kernels that
  # randomise 32 bits of every mapping, which GCC 11's libtsan refuses to start on)
  TSAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
  echo "== ThreadSanitizer: stress (6 threads, 20 s) + async / fan-out tests" >> $LOG
  # (GCC 11's libtsan does not know the address-space layout of kernels with 32 bits of mmap entropy: no ASLR here)
  NOASLR="setarch $(uname -m) -R"
  LD_PRELOAD=$TSAN_RT CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_tsan.so CKZG_TESTS_NO_AUTOBUILD=1 \
    timeout 900 $NOASLR python tools/stress_gpu.py 20 6 >> $LOG 2>&1
  echo "rc=$?" >> $LOG
  LD_PRELOAD=$TSAN_RT CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_tsan.so CKZG_TESTS_NO_AUTOBUILD=1 \
    timeout 1200 $NOASLR python -m pytest tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider -k "async or fan_out or pipelined" >> $LOG 2>&1
  echo "rc=$?" >> $LOG
  echo "== ThreadSanitizer: the combiner (concurrent one-blob callers sharing batch launches; native + Python threads)" >> $LOG
  LD_PRELOAD=$TSAN_RT CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_tsan.so CKZG_TESTS_NO_AUTOBUILD=1 \
    timeout 1500 $NOASLR python -m pytest tests/test_gpu_coalesce.py -m gpu -q -x -p no:cacheprovider >> $LOG 2>&1
  echo "rc=$?" >> $LOG
  tail -15 $LOG
  if grep -q "FATAL: ThreadSanitizer" $LOG; then echo "TSAN DID NOT START"; exit 3; fi
  if grep -q "WARNING: ThreadSanitizer" $LOG; then echo "TSAN REPORTS FOUND"; grep -A12 "WARNING: ThreadSanitizer" $LOG | head -120; exit 1; fi
  grep -q "rc=[1-9]" $LOG && exit 2
  echo "sanitizers: clean"
  exit 0
fi
if [ "$mode" = cpu ]; then
  echo "== host shim under gcc ASan+UBSan: tests/test_host_arith.py tests/test_fk20_edge_builder.py" >> $LOG
  LD_PRELOAD="$GCC_RT" CKZG_SHIM_SO=c-kzg-4844_amd/csrc/libhost_shim_san.so CKZG_TESTS_NO_AUTOBUILD=1 \
    timeout 3000 python -m pytest tests/test_host_arith.py tests/test_fk20_edge_builder.py -q -x -p no:cacheprovider >> $LOG 2>&1
  echo "rc=$?" >> $LOG
  echo "== product (host half sanitized) under ASan+UBSan: tests/test_abi_exports.py" >> $LOG
  LD_PRELOAD="$CLANG_RT" CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_san.so CKZG_TESTS_NO_AUTOBUILD=1 \
    timeout 600 python -m pytest tests/test_abi_exports.py -q -x -p no:cacheprovider >> $LOG 2>&1
  echo "rc=$?" >> $LOG
else
  echo "== product (host half sanitized) on the GPU: vectors, fuzz, round-3 verification forms" >> $LOG
  LD_PRELOAD="$CLANG_RT" CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_san.so CKZG_TESTS_NO_AUTOBUILD=1 HSA_XNACK=0 \
    timeout 1500 python -m pytest tests/test_gpu_vectors.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider >> $LOG 2>&1
  echo "rc=$?" >> $LOG
fi
tail -15 $LOG
if grep -qE "ERROR: AddressSanitizer|runtime error:" $LOG; then echo "SANITIZER REPORTS FOUND"; grep -E "ERROR: AddressSanitizer|runtime error:" $LOG | sort | uniq -c | head -20; exit 1; fi
grep -q "rc=[1-9]" $LOG && exit 2
echo "sanitizers: clean"
