cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_p1; mkdir -p $o
rocprofv3 --kernel-trace --stats -f csv -d $o/crit -o t -- python3 bench.py --mode train --criterion --no-roofline --steps 10 > $o/crit.log 2>&1
f=$(find $o/crit -name '*kernel_stats.csv' | head -1); head -40 $f | cut -c1-200
find $o -name '*kernel_trace.csv' -delete
rocprofv3 --kernel-trace --stats -f csv -d $o/hpe -o t -- python3 tools/bench_head_pe.py > $o/hpe.log 2>&1
tail -5 $o/hpe.log
f=$(find $o/hpe -name '*kernel_stats.csv' | head -1); head -25 $f | cut -c1-200
find $o -name '*kernel_trace.csv' -delete
