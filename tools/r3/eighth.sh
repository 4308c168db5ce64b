#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_eighth
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
# back-to-back 238 GB loads: the second and third start while the driver is still scrubbing the VRAM the previous one released
for i in 1 2 3; do timeout 300 python tools/bench_load.py 16 16 13 async >> $O/load.log 2>> $O/load.err; done
cat $O/load.log | cut -c1-420
