#!/bin/bash
# final collection of round 3: full GPU suite, default bench + rocprofv3 passes of the headline, kernel stats of the rows
export TMPDIR=/tmp
O=gpurun_out/r3_final
rm -rf $O && mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -14 $O/pytest.log
bash tools/profile_bench.sh > $O/prof_bench.log 2>&1
bash tools/profile_rows.sh > $O/prof_rows.log 2>&1
tail -c 600 $O/prof_bench.log
