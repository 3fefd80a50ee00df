"""Label counts -> log prior with the blank rotated from index 0 to the end.

Same contract as mobvoi/lstm_ctc ``nnet/class_prior.py:21-47`` (pinned bit for bit by tests/test_ref_fixtures.py):
the counts file holds one bracketed, whitespace-separated row (``[ 10 5 0 85 ]``, EESEN's label.counts); only its
first line is read; the division and the logarithm are float32; classes whose probability is below 1e-10 get
-1e10; entry 0 (the blank in EESEN's numbering) moves to the end, where this code base keeps its blank."""
import numpy as np

PRIOR_CUTOFF = 1e-10
ZERO_LOG_PRIOR = -1e10


def read_label_counts(label_counts):
    """The numbers of the first line of ``label_counts`` (a path), brackets stripped; None for an empty file."""
    with open(label_counts) as handle:
        first = handle.readline()
    if not first:
        return None
    body = first.strip().lstrip('[').rstrip(']')
    return [float(tok) for tok in body.split()]


def get_class_prior(label_counts):
    counts = np.array(read_label_counts(label_counts), dtype=np.float32)
    prob = counts / counts.sum()
    with np.errstate(divide='ignore'):
        log_prior = np.log(prob)
    log_prior = np.where(prob < PRIOR_CUTOFF, np.float32(ZERO_LOG_PRIOR), log_prior).astype(np.float32)
    return np.roll(log_prior, -1)                # blank: first -> last
