"""Batching pipelines — mirror of mobvoi/lstm_ctc ``nnet/pipeline.py``.

``create_pipeline_sequence_batch`` (24-63): consecutive runs of ``batch_size`` utterances, padded to the
longest: features with 0.0, labels with -1 (int64); the last batch may be smaller.  This dict IS the input
contract of the hot path.  ``create_pipeline_sequential`` (66-86): one utterance per step
with its file name, for inference.

Data parallelism (new): with ``world_size`` > 1 rank r takes every world_size-th batch, so the global batch
of a step is world_size consecutive batches.  All ranks see the same number of steps (a ragged tail is
dropped) so the collectives line up.
"""
import numpy as np


class SequenceBatchPipeline:
    def __init__(self, dataset, input_dim, batch_size, rank=0, world_size=1, prefetch=4):
        self.dataset, self.input_dim, self.batch_size = dataset, input_dim, batch_size
        self.rank, self.world_size, self.prefetch = rank, world_size, prefetch

    def _collate(self, items):
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

    def _batches(self):
        n = len(self.dataset)
        starts = list(range(0, n, self.batch_size))
        if self.world_size > 1:
            usable = len(starts) // self.world_size * self.world_size
            starts = starts[self.rank:usable:self.world_size]
        files = self.dataset.files
        for s in starts:
            yield [self.dataset.load(p) for p in files[s:s + self.batch_size]]

    def __iter__(self):
        # a small background thread keeps file parsing off the GPU step's critical path
        import queue
        import threading
        q = queue.Queue(maxsize=self.prefetch)
        stop = object()

        def work():
            try:
                for items in self._batches():
                    q.put(self._collate(items))
            except BaseException as exc:      # surface loader errors in the consumer
                q.put(exc)
            q.put(stop)

        threading.Thread(target=work, daemon=True).start()
        while True:
            item = q.get()
            if item is stop:
                return
            if isinstance(item, BaseException):
                raise item
            yield item


def create_pipeline_sequence_batch(dataset, input_dim, batch_size=64, batch_threads=8, num_epochs=1, rank=0,
                                   world_size=1):
    """Returns (initializer, pipeline) like the reference; the initializer is a no-op callable."""
    return (lambda: None), SequenceBatchPipeline(dataset, input_dim, batch_size, rank, world_size)


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
