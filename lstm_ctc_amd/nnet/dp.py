"""Utterance-batch data parallelism (new functionality — the reference is single-GPU).

One process per GPU; parameters and optimizer slots are replicated (same seed or broadcast).  The ONLY
exchange step of a train step is one all-reduce(sum) over the flat fp32 gradient buffer, done BEFORE the L2 /
global-norm clip / update, so N ranks x B utterances give exactly the update of one batch of N*B utterances
(nnet/graph.py:190 clips the total gradient of a SUM loss).  The logged running means need the per-step
(size, eval_loss, eval) triple summed as well.  Backend: "nccl" (= RCCL over xGMI) on GPUs; the same code
runs on "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist

from .. import ops


def world_size(pg):
    return dist.get_world_size(pg) if pg is not None else 1


def rank(pg):
    return dist.get_rank(pg) if pg is not None else 0


def allreduce_sum_(flat, pg):
    """In-place sum of a flat tensor over the group (no-op without a group).  RCCL ("nccl") reduces device buffers in
    place over xGMI; on a gloo group (CPU tests, or two ranks sharing one GPU in the GPU tests) a device tensor is
    staged through the host."""
    if pg is not None:
        if flat.is_cuda and dist.get_backend(pg) == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=pg)
            flat.copy_(host)
        else:
            # bench.py's live profile: the compute stream waits for the collective, so the bracket is its duration
            ev = ops._prof_begin() if flat.is_cuda else None
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=pg)
            ops._prof_end("allreduce", float(flat.numel() * flat.element_size()), ev)
    return flat


class GradientBuckets:
    """Per-layer buckets of the flat gradient, all-reduced while the backward pass is still running.

    The backward walks the layers top-down; when the BPTT of layer i has been enqueued, every gradient of layer i + 1 is
    final, so its contiguous range of the flat buffer can go out.  ``issue(lo, hi)`` starts ``all_reduce(sum)`` of
    ``flat[lo:hi]`` on the collective's own stream (it waits for the work already enqueued on the current stream - i.e. for
    that BPTT - and then runs beside the layer's weight-gradient GEMMs); ``wait()`` makes the current stream wait for every
    bucket issued so far - ``Model.backward`` calls it in front of the next persistent recurrence, which needs every CU of
    the GPU and must not find a collective's kernel resident.  ``finish()`` reduces whatever has not been reduced yet
    (biases, layer 0, the head, batch-norm parameters) and returns.  N ranks x B utterances still see exactly the sum
    gradient: each element is all-reduced exactly once.  On a gloo group the buckets are reduced synchronously (CPU tests)."""

    def __init__(self, flat, pg):
        self.flat, self.pg = flat, pg
        self.done, self.pending, self._inflight_bytes = [], [], 0
        self.async_ok = pg is not None and dist.get_backend(pg) != "gloo"

    def issue(self, lo, hi):
        if self.pg is None or hi <= lo:
            return
        assert all(hi <= a or lo >= b for a, b in self.done), "overlapping gradient buckets"
        self.done.append((lo, hi))
        if self.async_ok:
            self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            self._inflight_bytes += (hi - lo) * self.flat.element_size()
        else:
            allreduce_sum_(self.flat[lo:hi], self.pg)

    def wait(self):
        if not self.pending:
            return
        # bracket = how long the compute stream stalls for buckets still in flight (0 when they hid under the GEMMs)
        ev = ops._prof_begin() if self.flat.is_cuda else None
        for w in self.pending:
            w.wait()
        ops._prof_end("allreduce", float(self._inflight_bytes), ev)
        self.pending, self._inflight_bytes = [], 0

    def finish(self):
        self.wait()
        pos = 0
        for lo, hi in sorted(self.done) + [(self.flat.numel(), self.flat.numel())]:
            if lo > pos:
                allreduce_sum_(self.flat[pos:lo], self.pg)
            pos = max(pos, hi)
        self.done = []


def broadcast_(flat, pg, src=0):
    if pg is not None and dist.get_world_size(pg) > 1:
        if flat.is_cuda and dist.get_backend(pg) == "gloo":      # tests: ranks sharing one GPU; staged through the host
            host = flat.cpu()
            dist.broadcast(host, src=src, group=pg)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src, group=pg)
    return flat


def reduce_triple(size, eval_loss, batch_eval, pg, device):
    """Sum of (size, eval_loss, eval) over ranks, in float64 (funcs.py:48-54 consumes it)."""
    if pg is None or dist.get_world_size(pg) <= 1:
        return size, eval_loss, batch_eval
    if dist.get_backend(pg) == "gloo":
        device = "cpu"
    t = torch.tensor([float(size), float(eval_loss), float(batch_eval or 0.0)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return int(round(t[0].item())), t[1].item(), (t[2].item() if batch_eval is not None else None)
