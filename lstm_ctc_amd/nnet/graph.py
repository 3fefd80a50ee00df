"""Train / validation / inference "graphs" — host-side mirror of mobvoi/lstm_ctc ``nnet/graph.py``.

The reference builds TF graphs whose nodes ``sess.run`` evaluates; here a graph is a small object that
owns the model and executes one batch per :meth:`Session.run`, returning the same keys the reference's
graph dict exposes (``size, eval_loss, loss, eval, sequence_length, logits, nnet_output, filename`` …),
so ``nnet.train`` / ``nnet.validate`` (funcs.py) read exactly like the reference's loops.

* validation graph .. ``create_graph_for_validation_ctc``  nnet/graph.py:51-162
* training graph .... ``create_graph_for_training_ctc``    nnet/graph.py:165-209
* inference graph ... ``create_graph_for_inference``       nnet/graph.py:212-241
* data parallelism .. NEW (no reference counterpart): one process per GPU, the flat fp32
  gradient buffer is summed with ONE RCCL all-reduce before the clip, so N ranks x B utterances
  reproduce a single batch of N*B (the clip at graph.py:190 acts on the total gradient of a SUM loss).
"""
import numpy as np
import torch

from .. import ops
from . import dp
from .model import Model


class OutOfRangeError(Exception):
    """End of the input pipeline (tf.errors.OutOfRangeError in the reference's run loops)."""


def flatten_labels(dense):
    """Dense [B,Lmax] int64 labels padded with -1 -> (flat int32, offsets[B+1], max_len): the
    tf.where / gather_nd / SparseTensor conversion of nnet/graph.py:76-104."""
    dense = np.asarray(dense)
    keep = dense != -1
    lens = keep.sum(axis=1).astype(np.int64)
    flat = dense[keep].astype(np.int32)            # row-major order == tf.where order
    offs = np.zeros(len(lens) + 1, np.int32)
    np.cumsum(lens, out=offs[1:])
    return flat, offs, int(lens.max()) if len(lens) else 0


def _label_smoothing_setup(nnet_config, device):
    """(weight, log q or None) of the KL regulariser, bilstm.py:255-269: uniform wins over prior (the elif there)."""
    u, pw = nnet_config.get("uniform_label_sm"), nnet_config.get("prior_label_sm")
    if u is not None and u > 0:
        return float(u), None
    if pw is not None and pw > 0 and nnet_config.get("prior_label_path") is not None:
        from .class_prior import get_class_prior
        return float(pw), torch.from_numpy(get_class_prior(nnet_config["prior_label_path"])).to(device)
    return 0.0, None


def _create_logits(nnet_type):
    def create_logits(nnet_input, sequence_length, nnet_config, model=None):
        """``create_logits(nnet_input[B,T,D] f32, sequence_length[B] i32, nnet_config) -> (logits[B,T,V], encoder,
        reg_loss)`` - the callable ``get_create_logits`` hands to the graph builders (nnet/graph.py:24-34,63-67;
        bodies nnet/bilstm.py:25-273, nnet/lstm.py:125-368).  Inputs are GPU tensors (or numpy arrays, uploaded);
        the stack is built from ``nnet_config`` (seed = its ``seed`` key) unless an existing ``model`` is passed;
        the Model that ran is left on ``create_logits.model`` (its ParamStore holds the TF-named variables).
        ``logits`` is a [B,T,V] view of the time-major result; ``encoder`` is the final-state concat for blstm
        (bilstm.py:206-208) and None for lstm; ``reg_loss`` is the list of (device scalar, weight) pairs of
        bilstm.py:255-269 (empty for lstm and when both smoothing weights are 0)."""
        cfg = dict(nnet_config, nnet_type=nnet_type)
        if model is None:
            dev = nnet_input.device if torch.is_tensor(nnet_input) else torch.device("cuda")
            model = Model(cfg, dev, seed=cfg.get("seed"))
        dev = model.device
        x = torch.as_tensor(np.asarray(nnet_input) if not torch.is_tensor(nnet_input) else nnet_input)
        x = x.to(dev, torch.float32).permute(1, 0, 2).contiguous()                 # time-major [T,B,D]
        seq = torch.as_tensor(np.asarray(sequence_length) if not torch.is_tensor(sequence_length)
                              else sequence_length).to(dev, torch.int32)
        ops.lstm_status(dev).zero_()
        tbv = model.forward(x, seq)
        if int(ops.lstm_status(dev).item()) != 0:      # a persistent recurrence could not complete: NaN outputs
            with ops.force_launch_train():
                ops.lstm_status(dev).zero_()
                tbv = model.forward(x, seq)
            if int(ops.lstm_status(dev).item()) != 0:
                raise RuntimeError("LSTM recurrence failed on the launch train as well")
        reg_loss = []
        encoder = None
        if nnet_type == "blstm":
            encoder = model.encoder()
            w, logq = _label_smoothing_setup(cfg, dev)
            if w > 0:
                T_, B_, V_ = tbv.shape
                reg_loss.append((ops.label_smoothing(tbv.view(T_ * B_, V_), w, logq, None), w))
        create_logits.model = model
        return tbv.permute(1, 0, 2), encoder, reg_loss
    create_logits.model = None
    create_logits.__name__ = "create_logits_" + nnet_type
    return create_logits


create_logits_blstm = _create_logits("blstm")
create_logits_lstm = _create_logits("lstm")
# nnet/lstm.py:26-122, by intent like 'lstm': as shipped it returns the bare logits tensor where graph.py:63 unpacks a triple;
# here it returns (logits, None, []) - a stack of plain LSTM cells (no peepholes / projection / dropout, forget bias 0) and an
# affine head with sigma = 1 / sqrt(num_neurons)
create_logits_cudnnlstm = _create_logits("cudnnlstm")


def get_create_logits(string):
    """nnet/graph.py:24-34: 'blstm' | 'cudnnlstm' | 'lstm' -> the function, anything else -> None."""
    return {"blstm": create_logits_blstm, "cudnnlstm": create_logits_cudnnlstm,
            "lstm": create_logits_lstm}.get(string) if string else None


def get_optimizer(string):
    """nnet/graph.py:37-48 — the three optimizers the reference knows."""
    return string if string in ("adam", "sgd", "momentum") else None


class _FallbackLatch:
    """Bookkeeping of persistent-recurrence failures.  A failed launch costs its bounded waits (seconds) plus a second
    run of the step, so a cause that does not go away - a shared GPU, a resident collective, anything that keeps 256
    workgroups from being co-resident - must not be paid on every step: after LATCH_AFTER consecutive failures the
    graph stays on the launch train for the rest of the run.  Every failure is logged."""
    LATCH_AFTER = 2

    def __init__(self):
        self.consecutive, self.latched = 0, False

    def good(self):
        self.consecutive = 0

    def failed(self, status, total):
        from . import tflog
        self.consecutive += 1
        self.latched = self.consecutive >= self.LATCH_AFTER
        tflog.info("persistent LSTM launch did not complete (status %d, fallback #%d); re-running the step with the "
                   "per-step launch train%s" % (status, total, (" - and staying on it for the rest of this run (%d "
                                                                "failures in a row)" % self.consecutive)
                                                if self.latched else ""))


class CTCGraph:
    """Validation (and, with ``learn_rate``, training) graph over a batch pipeline."""

    def __init__(self, pipeline, nnet_config, learn_rate=None, clip_norm=5.0, optimizer="sgd",
                 l2_decay_weight=1e-5, device="cuda", seed=None, process_group=None):
        nnet_type = nnet_config.get("nnet_type")
        if get_create_logits(nnet_type) is None:
            raise ValueError("unsupported nnet_type: %s" % nnet_type)
        self.pipeline = pipeline
        self.model = Model(nnet_config, device, seed=seed)
        self.training = learn_rate is not None
        self.learn_rate = learn_rate
        self.clip_norm = clip_norm
        self.l2 = l2_decay_weight
        self.optimizer = optimizer
        if self.training and get_optimizer(optimizer) is None:
            raise ValueError("unsupported optimizer: %s" % optimizer)
        self.pg = process_group
        self.world = dp.world_size(process_group)
        self.global_step = 0
        self.opt_step = 0          # Adam's t: restarts with the process, like the reference (nnet-train.py:83)
        dev = self.model.device
        n = self.model.ps.n
        slots = {"sgd": 0, "momentum": 1, "adam": 2}.get(optimizer, 0) if self.training else 0
        self.opt_state = torch.zeros(max(slots * n, 1), dtype=torch.float32, device=dev)
        self.norm_out = torch.zeros(2, dtype=torch.float32, device=dev)
        # Dropout stream: every rank draws its own masks (rank r's utterance b must not share rank 0's noise), while
        # the INIT seed above stays common to all ranks so that the replicas start identical.
        self.rank = dp.rank(process_group)
        self.drop_seed = 0 if seed is None else int(seed)
        if self.world > 1:
            self.drop_seed = (self.drop_seed * 0x9E3779B1 + (self.rank + 1) * 0x85EBCA6B) & 0x7FFFFFFF
        self.persist_fallbacks = 0     # steps re-run on the launch train after a persistent launch failed
        self._fallback = _FallbackLatch()
        # per-layer gradient buckets, all-reduced beside the lower layers' weight-gradient GEMMs (dp.GradientBuckets);
        # LC_DP_BUCKETS=0: one all-reduce of the whole flat gradient after the backward
        import os
        self.dp_buckets = os.environ.get("LC_DP_BUCKETS", "1") != "0"
        self._force_buckets = os.environ.get("LC_DP_BUCKETS") == "2"      # also with a single rank (the RCCL test)
        self._buckets = None
        # label-smoothing regulariser (bilstm.py:255-269; blstm only): uniform wins over prior, like the elif there
        self.sm_weight, self.sm_logq = 0.0, None
        if nnet_type == "blstm":
            self.sm_weight, self.sm_logq = _label_smoothing_setup(nnet_config, dev)
        self.keys = ["nnet_input", "sequence_length", "logits", "raw_target", "nnet_target", "size", "eval_loss",
                     "loss", "eval", "global_step", "summary"] + (["lrate", "train"] if self.training else [])

    # the reference's graph is a dict; give the same key access
    def __contains__(self, key):
        return key in self.keys

    def __getitem__(self, key):
        if key not in self.keys:
            raise KeyError(key)
        return key

    # -------------------------------------------------------------------------------------------------
    def _upload(self, batch):
        dev = self.model.device
        a = batch["nnet_input"]
        if (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 3
                and a.transpose(1, 0, 2).flags.c_contiguous):
            # the batching pipeline's buffer: time-major (page-locked) memory seen through a [B,T,D] view - one DMA
            x = torch.from_numpy(a.transpose(1, 0, 2)).to(dev, non_blocking=True)
        else:
            x = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
            x = x.to(dev, non_blocking=True).permute(1, 0, 2).contiguous()       # [B,T,D] -> time-major [T,B,D]
        seq = np.ascontiguousarray(batch["sequence_length"], dtype=np.int32)
        flat, offs, maxlen = flatten_labels(batch["nnet_target"])
        d = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
        return x, d(seq), seq, d(flat), d(offs), flat, offs, maxlen

    def _validate_labels(self, flat):
        """tf.nn.ctc_loss raises InvalidArgument for labels outside [0, num_classes - 1) (the blank, V - 1, is not a label).
        Done in ``step`` for the batch being CONSUMED - not while batch k + 1 is staged, which would raise before step k has
        trained and been reported, and would run a collective on the copy stream.  Under data parallelism every rank must
        leave the step together (a rank that raised alone would strand the others in the gradient collectives), so the verdict
        is summed over the group first."""
        V = self.model.ps.V
        bad = bool(flat.size and (int(flat.min()) < 0 or int(flat.max()) >= V - 1))
        if self.pg is not None and self.world > 1:
            flag = torch.tensor([int(bad)], dtype=torch.int32, device=self.model.device)
            dp.allreduce_sum_(flag, self.pg)
            bad_anywhere = int(flag.item()) != 0
        else:
            bad_anywhere = bad
        if bad_anywhere:
            if bad:
                raise ValueError("nnet_target holds a label outside [0, %d): min %d, max %d (num_targets = %d, "
                                 "blank = %d)" % (V - 1, int(flat.min()), int(flat.max()), V, V - 1))
            raise ValueError("another rank's nnet_target holds a label outside [0, %d)" % (V - 1))

    def stage(self, batch):
        """Uploads ``batch`` on a side stream NOW, to be consumed by a later ``step(None, staged=...)``: called for batch
        k + 1 before step k is enqueued, the copy runs under step k's kernels instead of in front of step k + 1's.  Pure
        copies: nothing here can raise for batch k + 1's CONTENT or touch the process group (see _validate_labels)."""
        dev = self.model.device
        if getattr(self, "_h2d_stream", None) is None:
            self._h2d_stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(self._h2d_stream):
            up = self._upload(batch)
            ev = torch.cuda.Event()
            ev.record()
        return up, ev

    def step(self, batch, fetch_eval=True, fetch_logits=False, train=None, staged=None):
        """One sess.run of the graph on one batch (dict of numpy arrays in the pipeline contract of
        nnet/pipeline.py:35-61; or ``staged``, the result of an earlier ``stage(batch)``).  Returns a dict of host
        values."""
        if staged is not None:
            up, ev = staged
            cur = torch.cuda.current_stream(self.model.device)
            cur.wait_event(ev)
            for t in up:
                if torch.is_tensor(t):
                    t.record_stream(cur)          # allocated on the copy stream, consumed on this one
        else:
            up = self._upload(batch)
        x, seq_d, seq, flat_d, offs_d, flat, offs, maxlen = up
        self._validate_labels(flat)
        out = self.step_device(x, seq_d, flat_d, offs_d, maxlen, int(len(flat)), fetch_eval=fetch_eval,
                               fetch_logits=fetch_logits, train=train, flat_host=flat, offs_host=offs)
        out["sequence_length"] = seq
        return out

    def step_device(self, x, seq_d, flat_d, offs_d, maxlen, size, fetch_eval=False, fetch_logits=False, train=None,
                    flat_host=None, offs_host=None):
        """The same step on tensors already resident in HBM: x [T,B,D] time-major f32, seq_d [B] i32,
        labels flat i32 + offsets [B+1] i32.

        A persistent-recurrence launch that cannot complete (lstm_ctc_hip.h: bounded waits, sticky status word) leaves
        NaN outputs and makes the optimizer skip its update on the device; the word is read at the step's one sync
        point and the step is then re-run, in this process, with the per-step launch train."""
        train = self.training if train is None else train
        counters = (self.global_step, self.drop_seed, self.opt_step)
        args = (x, seq_d, flat_d, offs_d, maxlen, size, fetch_eval, fetch_logits, train, flat_host, offs_host)
        if self._fallback.latched:                     # co-residency is structurally unavailable: stay on the train
            with ops.force_launch_train():
                out, status = self._step_once(*args)
            if status != 0:
                raise RuntimeError("LSTM recurrence failed on the launch train (status %d)" % status)
            return out
        out, status = self._step_once(*args)
        if status == 0:
            self._fallback.good()
            return out
        self.global_step, self.drop_seed, self.opt_step = counters
        self.persist_fallbacks += 1
        self._fallback.failed(status, self.persist_fallbacks)
        with ops.force_launch_train():
            out, status = self._step_once(*args)
        if status != 0:
            raise RuntimeError("LSTM recurrence failed on the launch train as well (status %d)" % status)
        return out

    def _step_once(self, x, seq_d, flat_d, offs_d, maxlen, size, fetch_eval, fetch_logits, train, flat_host, offs_host):
        dev = self.model.device
        ops.lstm_status(dev).zero_()
        self.global_step += 1
        self.drop_seed = (self.drop_seed * 1664525 + 1013904223) & 0x7FFFFFFF
        logits = self.model.forward(x, seq_d, drop_seed=self.drop_seed)          # [T,B,V]
        loss_b, grad = ops.ctc_loss(logits, flat_d, offs_d, seq_d, maxlen, want_grad=train)
        out = {"size": size}
        tokens = out_len = None
        reg = None
        if self.sm_weight > 0:                                                   # graph.py:120-133: loss += reg
            T_, B_, V_ = logits.shape
            reg = ops.label_smoothing(logits.view(T_ * B_, V_), self.sm_weight, self.sm_logq,
                                      grad.view(T_ * B_, V_) if train else None)
        if fetch_eval:
            tokens, out_len = ops.ctc_greedy(logits, seq_d)
        bn_saved = None
        if train:
            bn_saved = dict(self.model.saved.get("bn") or {})
            self._buckets = (dp.GradientBuckets(self.model.ps.grad, self.pg)
                             if self.pg is not None and self.dp_buckets and (self.world > 1 or self._force_buckets) else None)
            self.model.backward(grad, buckets=self._buckets)
            self._apply_gradients()
        # one device->host sync per step, like the reference's sess.run
        eval_loss = float(loss_b.sum().item())                                   # graph.py:116 reduce_sum
        status = int(ops.lstm_status(dev).item())
        if status != 0:
            return None, status
        if train and bn_saved:
            self.model.update_moving_averages(bn_saved)   # batch-norm UPDATE_OPS (graph.py:194-196); only with use_bn
        out["eval_loss"] = eval_loss
        out["loss"] = eval_loss + (float(reg.item()) if reg is not None else 0.0)
        if fetch_eval:
            tok, n = tokens.cpu().numpy(), out_len.cpu().numpy()
            if flat_host is None:
                flat_host, offs_host = flat_d.cpu().numpy(), offs_d.cpu().numpy()
            out["eval"] = float(ops.edit_distance_host(tok, n, flat_host, offs_host).sum())   # graph.py:143-150
            out["decoded"] = (tok, n)
        if fetch_logits:
            out["logits"] = logits.permute(1, 0, 2).cpu().numpy()                # reference layout [B,T,V]
        if train:
            out["grad_norm"] = float(self.norm_out[0].item())
        return out, 0

    def _apply_gradients(self):
        """L2 + clip_by_global_norm + optimizer.apply_gradients (graph.py:183-200), after the DP all-reduce.  The
        update is guarded by the LSTM status word: a step whose recurrence failed leaves parameters and slots alone.
        (Under DP every rank must skip together: the status words are summed with the gradient's collective.)"""
        ps = self.model.ps
        guard = ops.lstm_status(ps.flat.device)
        if self.pg is not None and self.world > 1:
            dp.allreduce_sum_(guard, self.pg)
        if getattr(self, "_buckets", None) is not None:
            self.last_bucket_ranges = len(self._buckets.done)      # per-layer ranges that went out during the backward
            self._buckets.finish()               # the layers' buckets went out during the backward; this is the rest
            self._buckets = None
        else:
            dp.allreduce_sum_(ps.grad, self.pg)
        self.opt_step += 1
        ops.optimizer_step(ps.flat, ps.grad, ps.n_decay, self.l2, self.clip_norm, self.optimizer, self.learn_rate,
                           self.opt_step, self.opt_state, self.norm_out, guard=guard)

    # ------------------------------------------------------------------------------------------------- checkpoints
    def save(self, path):
        save_params(self.model.ps, path)

    def restore(self, path):
        load_params(self.model.ps, path)


class InferenceGraph:
    """create_graph_for_inference — nnet/graph.py:212-241: one utterance per run, softmax(smooth*logits)."""

    def __init__(self, pipeline, nnet_config, smooth_factor=1.0, device="cuda"):
        cfg = dict(nnet_config)
        self.pipeline = pipeline
        self.model = Model(cfg, device)
        self.smooth = smooth_factor
        self.persist_fallbacks = 0
        self._fallback = _FallbackLatch()
        self.keys = ["filename", "nnet_input", "sequence_length", "logits", "nnet_output"]

    def __getitem__(self, key):
        if key not in self.keys:
            raise KeyError(key)
        return key

    def forward_batch(self, feats_list):
        """feats_list: list of [T_i, D] float32 arrays -> list of (logits[T_i,V]) on the host, computed as ONE
        padded batch (result-identical to the reference's B=1 runs because padding is masked; SURVEY §8f NEXT-3)."""
        dev = self.model.device
        B = len(feats_list)
        T = max(f.shape[0] for f in feats_list)
        D = feats_list[0].shape[1]
        x = np.zeros((T, B, D), np.float32)
        for b, f in enumerate(feats_list):
            x[:f.shape[0], b] = f
        seq = np.asarray([f.shape[0] for f in feats_list], np.int32)
        xd, sd = torch.from_numpy(x).to(dev), torch.from_numpy(seq).to(dev)
        def run():
            ops.lstm_status(dev).zero_()
            out = self.model.forward(xd, sd)
            return out, int(ops.lstm_status(dev).item())

        if self._fallback.latched:
            with ops.force_launch_train():
                logits, status = run()
        else:
            logits, status = run()
            if status == 0:
                self._fallback.good()
            else:                                          # a persistent launch could not complete: launch train
                self.persist_fallbacks += 1
                self._fallback.failed(status, self.persist_fallbacks)
                with ops.force_launch_train():
                    logits, status = run()
        if status != 0:
            raise RuntimeError("LSTM recurrence failed on the launch train as well")
        return logits, seq

    def restore(self, path):
        load_params(self.model.ps, path)


def save_params(ps, path):
    """Single-file checkpoint at exactly ``path`` (the scripts pass an opaque prefix, scripts/train.sh:164,230):
    safetensors with the TF variable names and TF layouts (tf.trainable_variables only — no optimizer slots,
    like saver = tf.train.Saver(tf.trainable_variables()), bin/nnet-train.py:83)."""
    from safetensors.numpy import save_file
    tensors = {k: np.ascontiguousarray(v) for k, v in ps.export_tf().items()}
    save_file(tensors, path)


def load_params(ps, path):
    """``path`` is either this repository's single-file checkpoint (safetensors, TF variable names and layouts) or the
    PREFIX of a TensorFlow Saver checkpoint (``<path>.index`` + ``<path>.data-*``: what the reference's nnet-train.py
    wrote, bin/nnet-train.py:83,97) - a model trained there is continued or decoded here without TensorFlow."""
    import os
    from . import tf_checkpoint
    if not os.path.isfile(path) and tf_checkpoint.is_bundle(path):
        from . import tflog
        tflog.info("WARNING: reading a TensorFlow tensor-bundle checkpoint (%s.index / .data-*) without TensorFlow: this "
                   "reader is written from the format's definition and has only been tested against this repository's own "
                   "writer and hand-built streams, never against a file TensorFlow wrote - compare a validation loss with "
                   "the reference before relying on it" % path)
        ps.load_tf(tf_checkpoint.read_bundle(path))
        return
    from safetensors.numpy import load_file
    ps.load_tf(load_file(path))


# ----------------------------------------------------------------------------------------------------- reference-named factories
def create_graph_for_validation_ctc(pipeline, nnet_config, device="cuda", seed=None):
    return CTCGraph(pipeline, nnet_config, device=device, seed=seed)


def create_graph_for_training_ctc(pipeline, nnet_config, learn_rate, clip_norm=5.0, optimizer="sgd",
                                  l2_decay_weight=1e-5, device="cuda", seed=None, process_group=None):
    return CTCGraph(pipeline, nnet_config, learn_rate=learn_rate, clip_norm=clip_norm, optimizer=optimizer,
                    l2_decay_weight=l2_decay_weight, device=device, seed=seed, process_group=process_group)


def create_graph_for_inference(pipeline, nnet_config, smooth_factor=1.0, device="cuda"):
    return InferenceGraph(pipeline, nnet_config, smooth_factor=smooth_factor, device=device)


class Session:
    """Stands in for tf.Session in the run loops: ``run(nodes)`` executes the graph on the next batch of
    its pipeline and returns {key: value} for the requested keys; raises OutOfRangeError at end of data."""

    def __init__(self, graph):
        self.graph = graph
        self._it = None
        self._ahead = None           # the NEXT batch, already on its way to the device

    def _stage_next(self):
        try:
            batch = next(self._it)
        except StopIteration:
            return None
        return self.graph.stage(batch) if hasattr(self.graph, "stage") else (batch,)

    def run(self, nodes):
        if self._it is None:
            self._it = iter(self.graph.pipeline)
            self._ahead = self._stage_next()
        cur = self._ahead
        if cur is None:
            raise OutOfRangeError()
        self._ahead = self._stage_next()        # batch k + 1 starts its upload before step k is enqueued
        want = set(nodes.values()) if isinstance(nodes, dict) else set(nodes)
        kw = dict(fetch_eval=("eval" in want), fetch_logits=("logits" in want), train=("train" in want))
        res = self.graph.step(cur[0], **kw) if len(cur) == 1 else self.graph.step(None, staged=cur, **kw)
        res["train"] = None
        res["summary"] = None
        if isinstance(nodes, dict):
            return {k: res.get(v) for k, v in nodes.items()}
        return [res.get(v) for v in nodes]
