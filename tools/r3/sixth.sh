#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_sixth
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 900 bash tools/r3/san_probe.sh > $O/san_probe.log 2>&1
head -40 $O/san_probe.log | cut -c1-300
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitize.out 2>&1; echo "sanitize rc=$?" >> $O/sanitize.out
tail -25 $O/sanitize.out | cut -c1-300; cp gpurun_out/sanitize_gpu.log $O/ 2>/dev/null
