# Round 6: the data-parallel code path on RCCL with ONE rank (all this pool can give): per-layer buckets forced on
# (LC_DP_BUCKETS=2), so the async all-reduces, the waits in front of each recurrence and finish() all run - what they cost
# when there is nobody to talk to is the fixed overhead of the N > 1 step.
for w in c4 c5; do
  for b in 0 2; do
    LC_DP_BUCKETS=$b timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload $w --no-cpu-baseline --no-secondary --no-cli-corpus --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w RCCL 1 rank, LC_DP_BUCKETS=$b:', d['ms_per_step'], 'ms;', 'allreduce', d.get('allreduce'), 'ranks', d['config'].get('rccl_ranks'), d['config'].get('collective_backend'), 'bucket ranges', d['config'].get('dp_bucket_ranges_per_step'))
"
  done
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-cli-corpus --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w bare python (no process group):', d['ms_per_step'], 'ms')
"
done
