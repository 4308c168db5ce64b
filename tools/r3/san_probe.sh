#!/bin/bash
# where does the sanitized product stop on the GPU box?  incremental steps, unbuffered
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CLANG_RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:use_sigaltstack=0:abort_on_error=0:detect_odr_violation=0
export CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_san.so CKZG_TESTS_NO_AUTOBUILD=1
echo "-- step 1: python under the ASan runtime"; LD_PRELOAD="$CLANG_RT" python -c "print('py ok')"; echo "rc=$?"
echo "-- step 2: HIP runtime"; LD_PRELOAD="$CLANG_RT" python -c "
import ctypes as C
rt=C.CDLL('/opt/rocm/lib/libamdhip64.so'); n=C.c_int(); print('hipGetDeviceCount', rt.hipGetDeviceCount(C.byref(n)), n.value)"; echo "rc=$?"
echo "-- step 3: load + commit through the sanitized library"; LD_PRELOAD="$CLANG_RT" python -X faulthandler -c "
import sys; sys.path.insert(0,'tests')
from kzg_ctypes import Kzg, HIP_SO
print('lib', HIP_SO)
api=Kzg(HIP_SO,'',precompute=0,options={'commit_wbits':8,'proof_wbits':0})
print('loaded')
print(api.blob_to_kzg_commitment(bytes(131072)).hex())
api.close(); print('closed')"; echo "rc=$?"
echo "-- step 4: pytest"; LD_PRELOAD="$CLANG_RT" timeout 900 python -X faulthandler -m pytest tests/test_gpu_vectors.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15; echo "rc=${PIPESTATUS[0]}"
