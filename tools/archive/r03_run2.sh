#!/bin/bash
# round 3, GPU call 2: whole -m gpu suite without -x (prints kept); kernel traces of the two dense workloads
mkdir -p gpurun_out/r03_run2
python -m pytest tests -m gpu -q -s > gpurun_out/r03_run2/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_run2/rc.txt
cd /tmp && export TMPDIR=/tmp
for w in full ffhq; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_run2/trace_$w -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 2 --streams 1 --preroll-s 0 > $GRAFT_REPO_ROOT/gpurun_out/r03_run2/trace_$w.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_run2/trace_$w.err
done
cd $GRAFT_REPO_ROOT
find gpurun_out/r03_run2 -name "*kernel_trace.csv" -size +30M -delete
tail -5 gpurun_out/r03_run2/pytest.log; cat gpurun_out/r03_run2/rc.txt; ls -la gpurun_out/r03_run2/*/* | head
