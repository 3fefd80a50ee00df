for w in c2 c3 c2x3 c3x3; do for ov in 1 0; do LC_OVERLAP_WGRAD=$ov timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-cli-corpus 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w overlap=$ov', d['ms_per_step'], d.get('breakdown_ms_per_step'))"; done; done
