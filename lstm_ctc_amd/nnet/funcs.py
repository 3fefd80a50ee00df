"""Epoch run loops — mirror of mobvoi/lstm_ctc ``nnet/funcs.py`` (train 23-86, validate 89-152).

Same arithmetic (label-weighted running means of the per-label loss and token error rate), same log
lines, same exit behaviour: NaN running loss => ``tr_loss = nan`` + ``nan loss detected`` + exit 1.
Under data parallelism the per-step (eval_loss, eval, size) triple is first summed over ranks.
"""
import math
import os
import sys
import threading
import time

from . import tflog
from .graph import OutOfRangeError


class _Running:
    """loss += (batch_loss - loss) * size / processed   (funcs.py:48-54, python float64)."""

    def __init__(self, evaluate):
        self.step = 0
        self.processed = 0
        self.loss = 0.0
        self.acc = 0.0 if evaluate else None

    def update(self, size, eval_loss, batch_eval):
        if size > 0:
            self.processed += size
            self.loss += (eval_loss / size - self.loss) * size / self.processed
            if self.acc is not None:
                self.acc += (batch_eval / size - self.acc) * size / self.processed
        self.step += 1


def _reduce_triple(graph, size, eval_loss, batch_eval):
    from . import dp
    return dp.reduce_triple(size, eval_loss, batch_eval, getattr(graph, "pg", None), graph.model.device)


class _Throughput:
    """frames/sec of the run loop as the command line experiences it (loader + upload + GPU step + logging), counted
    from the end of step WARM on so that first-touch costs (allocations, code object loading) stay out.  One extra
    INFO line in front of the final tr_loss / cv_loss line; the recipes' greps are anchored on those and do not see it."""
    WARM = 3

    def __init__(self):
        self.steps = self.frames = 0
        self.t0 = None

    def update(self, seq_len):
        self.steps += 1
        if self.steps == self.WARM:
            self.t0 = time.perf_counter()
        elif self.steps > self.WARM and seq_len is not None:
            self.frames += int(sum(seq_len))

    def report(self, graph):
        if self.t0 is None or self.frames == 0:
            return
        dt = time.perf_counter() - self.t0
        world = getattr(graph, "world", 1) or 1
        tflog.info("throughput: steps = %d, frames = %d, seconds = %.3f, frames/sec = %.1f%s" % (
            self.steps - self.WARM, self.frames, dt, self.frames / dt,
            " (this rank; x %d ranks)" % world if world > 1 else ""))


class StepWatchdog:
    """Turns a hung step into the failure the recipes know how to handle.  The reference's loops can only end by
    finishing, by a NaN (exit 1, funcs.py:64-81) or by an exception; a runtime call that never returns - a wedged device, a
    collective whose peer died - would leave ``scripts/train*.sh`` waiting on this process for ever.  A daemon thread checks
    the time since the last ``kick()`` (one per completed ``sess.run``); after ``LC_STEP_TIMEOUT`` seconds (default 300; 0
    switches it off) without one it writes the ``FATAL:tensorflow:`` line and ends the PROCESS with status 1 -
    ``os._exit``, never a re-exec: a process that has initialised the GPU must not be replaced, and the main thread may be
    stuck inside a driver call that no Python exception can leave."""

    def __init__(self, timeout=None, tag="step", _exit=os._exit):
        if timeout is None:
            try:
                timeout = float(os.environ.get("LC_STEP_TIMEOUT", "300"))
            except ValueError:
                timeout = 300.0
        self.timeout, self.tag, self._exit = timeout, tag, _exit
        self._last = time.monotonic()
        self._steps = 0
        self._paused = False
        self._stop = threading.Event()
        self._thread = None

    def start(self):
        if self.timeout > 0 and self._thread is None:
            self._last = time.monotonic()
            self._thread = threading.Thread(target=self._watch, name="lc-step-watchdog", daemon=True)
            self._thread.start()
        return self

    def kick(self):
        self._steps += 1
        self._last = time.monotonic()

    def pause(self):
        """The watched part is over for now (nnet-forward: the device work of a batch is back on the host): what follows -
        writing to `ark:-` / a pipe whose reader may be slow for as long as it likes - is not this watchdog's business."""
        self._paused = True

    def resume(self):
        self._last = time.monotonic()
        self._paused = False

    def stop(self):
        self._stop.set()

    def _watch(self):
        poll = min(1.0, max(0.05, self.timeout / 4))
        while not self._stop.wait(poll):
            idle = 0.0 if self._paused else time.monotonic() - self._last
            if idle > self.timeout:
                tflog.fatal("no %s completed for %.0f s (LC_STEP_TIMEOUT = %g) after %d completed step(s): the device, the "
                            "input pipeline or a collective is not responding; exiting" % (self.tag, idle, self.timeout,
                                                                                           self._steps))
                self._exit(1)
                return

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()
        return False


def _loop(sess, graph, evaluate, report_interval, nodes, tag):
    run = _Running(evaluate)
    thr = _Throughput()
    dog = StepWatchdog(tag="%s step" % ("training" if tag == "tr_loss" else "validation")).start()
    try:
        while True:
            values = sess.run(nodes)
            dog.kick()
            thr.update(values.get("sequence_length"))
            size, eval_loss, batch_eval = _reduce_triple(graph, values["size"], values["eval_loss"],
                                                         values.get("eval") if evaluate else None)
            run.update(size, eval_loss, batch_eval)
            if report_interval and run.step % report_interval == 0:
                log = "step = %d, batch_size = %d, loss = %f" % (run.step, size, run.loss)
                if evaluate:
                    log += ", eval = %f" % run.acc
                tflog.info(log)
            if math.isnan(run.loss):
                raise ValueError
    except OutOfRangeError:
        tflog.info("done")
    except KeyboardInterrupt:
        tflog.fatal("interrupted by user")
        sys.exit(1)
    except ValueError:
        tflog.info("%s = %f" % (tag, run.loss))
        tflog.fatal("nan loss detected")
        sys.exit(1)
    finally:
        dog.stop()
    thr.report(graph)
    tflog.info("%s = %f" % (tag, run.loss))
    return run


def train(sess, graph, evaluate=False, report_interval=None):
    nodes = {"size": graph["size"], "train": graph["train"], "summary": graph["summary"], "loss": graph["loss"],
             "eval_loss": graph["eval_loss"], "sequence_length": graph["sequence_length"]}
    if evaluate:
        nodes["eval"] = graph["eval"]
    _loop(sess, graph, evaluate, report_interval, nodes, "tr_loss")
    return True


def validate(sess, graph, evaluate=False, report_interval=None):
    nodes = {"size": graph["size"], "loss": graph["loss"], "eval_loss": graph["eval_loss"],
             "sequence_length": graph["sequence_length"]}
    if evaluate:
        nodes["eval"] = graph["eval"]
    run = _loop(sess, graph, evaluate, report_interval, nodes, "cv_loss")
    if evaluate:
        tflog.info("cv_eval = %f" % run.acc)
    return True
