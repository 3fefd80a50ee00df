python tools/r6_fuse32_check.py 2>&1 | grep -v amdgpu.ids
for m in 0 1; do LC_FUSE_DX=$m timeout 900 python bench.py --workload c4 --no-cpu-baseline --no-secondary --no-cli-corpus --steps 120 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('c4 fuse_dx=$m 125 steps', d['ms_per_step'], 'loss/label', d['config']['last_loss_per_label'], 'fallbacks', d['config']['persist_fallbacks'])
"; done
