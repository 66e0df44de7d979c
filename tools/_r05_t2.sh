cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t2; mkdir -p $o
timeout 2400 python3 -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $o/pytest.log
for e in 1 0; do
  GD4D_TMP_OFFSETS_EXACT=$e timeout 900 python3 -m pytest tests/test_full_size_gpu.py -x -q -s -m gpu 2>&1 | grep -E "per-layer|passed|failed" | sed "s/^/exact=$e: /"
done | tee $o/flipped.txt
for rep in 1 2; do for e in 1 0; do
  GD4D_TMP_OFFSETS_EXACT=$e timeout 300 python3 bench.py --no-cpu-baseline --no-roofline --no-nhwc-figure --inflight 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('exact=$e', round(d['value'],1), round(d['ms_per_step'],4), d['ms_per_step_min'], d['ms_per_step_max'], d['windows'])"
done; done | tee $o/ab_exact.txt
timeout 600 python3 bench.py > $o/bench.json 2> $o/bench.err; tail -c 1500 $o/bench.json; tail -3 $o/bench.err
