#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r02_dense
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/time_full.py 4 128 48 48 bf16x3 > $OUT/time_full.log 2>&1
python3 tools/trace_last_call.py $OUT/trace 400 > $OUT/timeline.txt
rm -rf $OUT/trace
tail -3 $OUT/time_full.log
