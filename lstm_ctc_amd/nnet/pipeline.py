"""Batching pipelines — mirror of mobvoi/lstm_ctc ``nnet/pipeline.py``.

``create_pipeline_sequence_batch`` (24-63): consecutive runs of ``batch_size`` utterances, padded to the
longest: features with 0.0, labels with -1 (int64); the last batch may be smaller.  This dict IS the input
contract of the hot path.  ``create_pipeline_sequential`` (66-86): one utterance per step
with its file name, for inference.

How a batch is made (the reference: ``TFRecordDataset.map(_parse, num_parallel_calls)`` + ``padded_batch``,
tfrecord.py:122-123, pipeline.py:35-61, all inside TF's C++ runtime): two native calls per batch, each fanned out
over ``num_parallel_calls`` std::threads - ``lc_batch_open`` (read, CRC check, counts of every utterance) and, the padded
shape being known, ``lc_batch_decode`` straight into the utterance's rows of the batch buffer: no per-utterance array, no
collate copy, no Python thread pool around sub-millisecond calls (that was measured: the GIL hand-offs made 8 threads
slower than one).  The buffer is TIME-MAJOR ``[T, B, D]`` - the layout the GPU path consumes - in page-locked
memory when a GPU is present, and ``batch["nnet_input"]`` is its ``[B, T, D]`` transposed VIEW: consumers that follow the
reference's contract index it as ever, and ``CTCGraph`` recognises the view and uploads the buffer as it is (one DMA, no
transposing kernel).  ``batch_threads`` batches are assembled concurrently and ``prefetch`` finished ones wait in a queue.

Data parallelism (new): with ``world_size`` > 1 rank r takes every world_size-th batch, so the global batch
of a step is world_size consecutive batches.  All ranks see the same number of steps (a ragged tail is
dropped) so the collectives line up.
"""
import ctypes
import queue
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class _HostBuffers:
    """A ring of reusable host staging buffers (page-locked when CUDA is up: pinning costs milliseconds per call, and a
    pageable source turns the H2D copy into a synchronous two-hop one).  A slot comes up for reuse only after `depth`
    further batches - by which time an upload of it has long finished (the step that consumed it has synchronised) - AND
    only if nobody holds a view of it any more.  Ownership is explicit: every hand-out goes through a fresh LEASE object (a
    ctypes array over the slot's memory; numpy keeps it as the base of every view derived from the hand-out), and the ring
    keeps a weak reference to it - the lease dies with the last view, whatever the interpreter's reference-count conventions
    are.  A slot whose lease is still alive belongs to a consumer that kept the batch (``list(pipe)``, a cached CV set: the
    reference's padded_batch hands out arrays the consumer owns, pipeline.py:35-61): it is left alone and THIS batch gets a
    plain pageable array, so the page-locked memory of a ring never exceeds its `depth` buffers."""

    def __init__(self, depth):
        self.depth, self.slots, self.leases, self.n = depth, [None] * depth, [None] * depth, 0
        self.pageable_handouts = 0
        self.lock = threading.Lock()
        try:
            import torch
            self.torch = torch if torch.cuda.is_available() else None
        except Exception:                       # CPU-only tooling
            self.torch = None

    def take(self, nfloats):
        with self.lock:
            i = self.n % self.depth
            self.n += 1
            buf, lease = self.slots[i], self.leases[i]
            if lease is not None and lease() is not None:          # a consumer still holds the batch made in this slot
                self.pageable_handouts += 1
                if self.pageable_handouts == 1:
                    # once: a consumer that keeps batches for `depth` or more iterations (a deeper prefetch queue, a cached CV
                    # list) gets pageable memory from here on - its uploads become synchronous two-hop copies
                    from . import tflog
                    tflog.info("WARNING: a batch is still held %d hand-outs after it was made: staging buffers are handed "
                               "out as pageable memory while that lasts (uploads of retained batches are not pinned)" % self.depth)
                return np.empty(nfloats, np.float32)
            if buf is None or buf.size < nfloats:
                want = int(nfloats * 1.25) + 1024      # head-room: batches of a length-sorted list grow slowly
                if self.torch is not None:             # (the numpy array keeps the pinned tensor alive through .base)
                    buf = self.torch.empty(want, dtype=self.torch.float32, pin_memory=True).numpy()
                else:
                    buf = np.empty(want, np.float32)
                self.slots[i] = buf
            token = (ctypes.c_float * max(nfloats, 1)).from_buffer(buf)      # holds `buf`; held by every view of the hand-out
            self.leases[i] = weakref.ref(token)
        return np.frombuffer(token, np.float32, count=nfloats)


class SequenceBatchPipeline:
    def __init__(self, dataset, input_dim, batch_size, rank=0, world_size=1, prefetch=4, batch_threads=2,
                 num_parallel_calls=None):
        self.dataset, self.input_dim, self.batch_size = dataset, input_dim, batch_size
        self.rank, self.world_size, self.prefetch = rank, world_size, max(1, prefetch)
        self.batch_threads = max(1, min(int(batch_threads or 1), 4))
        npc = num_parallel_calls if num_parallel_calls is not None else getattr(dataset, "num_parallel_calls", 8)
        self.num_parallel_calls = max(1, int(npc))

    # ------------------------------------------------------------------------------------------ one batch
    def _collate(self, items):
        """Padding of already-loaded utterance dicts (any dataset with ``load``): the generic, copying form."""
        B = len(items)
        T = max(int(it["sequence_length"]) for it in items)
        L = max([int(it.get("target_length", 0)) for it in items] + [0])
        x = np.zeros((B, T, self.input_dim), np.float32)                       # padding value 0
        y = np.full((B, L), -1, np.int64)                                      # padding value -1
        for b, it in enumerate(items):
            x[b, :it["nnet_input"].shape[0]] = it["nnet_input"]
            if "nnet_target" in it:
                y[b, :len(it["nnet_target"])] = it["nnet_target"]
        return {"nnet_input": x, "nnet_target": y,
                "sequence_length": np.asarray([it["sequence_length"] for it in items], np.int32),
                "target_length": np.asarray([it.get("target_length", 0) for it in items], np.int32)}

    def _assemble(self, paths, pool, buffers):
        ds = self.dataset
        if not hasattr(ds, "open_batch"):                                       # any dataset with ``load``
            return self._collate(list(pool.map(ds.load, paths)))
        nb = ds.open_batch(paths, self.num_parallel_calls)                      # read + CRC + counts, native threads
        B, D = len(paths), self.input_dim
        T = int(nb.frames.max()) if B else 0
        L = int(nb.labels.max()) if B and ds.has_label else 0
        flat = buffers.take(T * B * D)                                          # time-major staging buffer [T, B, D]
        y = np.empty((B, L), np.int64)
        nb.decode(flat, D, B * D, T, y if ds.has_label else None)               # copy + splice / subsample + padding
        if not ds.has_label:
            y = np.full((B, 0), -1, np.int64)
        return {"nnet_input": flat.reshape(T, B, D).transpose(1, 0, 2), "nnet_target": y,
                "sequence_length": nb.frames.copy(), "target_length": nb.labels.copy()}

    def _starts(self):
        n = len(self.dataset)
        starts = list(range(0, n, self.batch_size))
        if self.world_size > 1:
            usable = len(starts) // self.world_size * self.world_size
            starts = starts[self.rank:usable:self.world_size]
        return starts

    # ------------------------------------------------------------------------------------------ the stream of batches
    def __iter__(self):
        files = self.dataset.files
        starts = self._starts()
        out = queue.Queue(maxsize=self.prefetch)
        stop = object()
        # slots in flight: queued + being assembled + the one (two, with the device prefetch) the consumer still holds
        buffers = _HostBuffers(self.prefetch + 2 * self.batch_threads + 3)
        pool = ThreadPoolExecutor(max_workers=self.num_parallel_calls, thread_name_prefix="lc-decode")
        batchers = ThreadPoolExecutor(max_workers=self.batch_threads, thread_name_prefix="lc-batch")
        cancelled = threading.Event()

        def produce():
            try:
                pending = []
                for s in starts:                       # keep batch_threads batches in assembly, deliver in order
                    if cancelled.is_set():
                        return
                    pending.append(batchers.submit(self._assemble, files[s:s + self.batch_size], pool, buffers))
                    if len(pending) >= self.batch_threads:
                        out.put(pending.pop(0).result())
                for p in pending:
                    out.put(p.result())
            except BaseException as exc:               # surface loader errors in the consumer
                out.put(exc)
            finally:
                out.put(stop)

        threading.Thread(target=produce, daemon=True, name="lc-pipeline").start()
        try:
            while True:
                item = out.get()
                if item is stop:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            cancelled.set()
            while True:                                # unblock a producer stuck on a full queue
                try:
                    out.get_nowait()
                except queue.Empty:
                    break
            batchers.shutdown(wait=False)
            pool.shutdown(wait=False)


def create_pipeline_sequence_batch(dataset, input_dim, batch_size=64, batch_threads=8, num_epochs=1, rank=0,
                                   world_size=1, num_parallel_calls=None):
    """Returns (initializer, pipeline) like the reference; the initializer is a no-op callable."""
    return (lambda: None), SequenceBatchPipeline(dataset, input_dim, batch_size, rank, world_size,
                                                 batch_threads=batch_threads, num_parallel_calls=num_parallel_calls)


class SequentialPipeline:
    def __init__(self, filename, tfrecord):
        self.filename, self.tfrecord = filename, tfrecord

    def __len__(self):
        return len(self.filename)

    def __iter__(self):
        for name, path in zip(self.filename, self.tfrecord.files):
            item = self.tfrecord.load(path)
            yield {"filename": name, "nnet_input": item["nnet_input"], "sequence_length": item["sequence_length"]}


def create_pipeline_sequential(filename, tfrecord, num_epochs=1):
    return (lambda: None), SequentialPipeline(filename, tfrecord)
