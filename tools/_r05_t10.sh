cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t10; mkdir -p $o
timeout 900 python3 -m pytest tests/test_modules_gpu.py -q -m gpu -k "any_num_points" > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest.log | tail -12
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $o/cumask_probe.txt
import torch, collections
from graph_detr4d_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
full = (1 << cus) - 1
lo = (1 << (cus // 2)) - 1
even = sum(1 << i for i in range(0, cus, 2))
for name, mask in (('all', full), ('lo half', lo), ('hi half', full ^ lo), ('even', even), ('first 32', (1 << 32) - 1), ('bits 0-7', 0xff)):
    st = ops.masked_stream(dev, mask)
    out = torch.full((4096,), -1, device=dev, dtype=torch.int32)
    with torch.cuda.stream(st):
        _lib.check(lib.gd4d_xcd_placement_probe(out.data_ptr(), 4096, st.cuda_stream), 'probe')
    st.synchronize()
    c = collections.Counter(out.cpu().tolist())
    print(name, 'XCC histogram of 4096 workgroups:', sorted(c.items()), 'pattern j%8:', bool((out.cpu() == out.cpu()[:8].repeat(512)).all()))
PY
for rep in 1 2; do for sp in "" halves interleaved; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-roofline --no-nhwc-figure --min-seconds 0.5 ${sp:+--cu-split $sp} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['requests_in_flight']; print('split=$sp', 'batch1', round(d['value'],1), 'inflight2', round(r['value'],1), round(r['ms_per_step'],3))"
done; done | tee $o/cusplit.txt
timeout 300 python3 tools/bench_hungarian.py 40 100 2>&1 | grep problems
timeout 900 python3 -m pytest tests/test_head_loss_gpu.py tests/test_end_to_end_gpu.py -q -m gpu > $o/pytest2.log 2>&1; echo "pytest2 rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest2.log | tail -12
timeout 600 python3 bench.py --mode train --criterion --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train criterion', round(d['ms_per_step'],3), d['config']['launch'])"
