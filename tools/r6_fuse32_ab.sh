timeout 900 python -m pytest tests/test_gpu_round6.py -q -k "nt2 or fused_dx" 2>&1 | tail -6
for m in 0 1 0 1; do LC_FUSE_DX=$m timeout 600 python bench.py --workload c4 --no-cpu-baseline --no-secondary --no-cli-corpus --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('c4 fuse_dx=$m', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['breakdown_ms_per_step'], d['config']['last_loss_per_label'])
"; done
