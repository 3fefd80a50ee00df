"""Label counts -> log prior with the blank rotated from index 0 to the end —
mirrors mobvoi/lstm_ctc nnet/class_prior.py:30-47 (float32 arithmetic, -1e10 for zero counts)."""
import numpy as np

PRIOR_CUTOFF = 1e-10


def read_label_counts(label_counts):
    with open(label_counts) as fi:
        for line in fi:
            strs = line.strip().lstrip('[').rstrip(']').strip().split()
            return [float(k) for k in strs]


def get_class_prior(label_counts):
    a = read_label_counts(label_counts)
    dis = np.asarray(a, dtype=np.float32)
    dis = dis / np.sum(dis)
    with np.errstate(divide='ignore'):
        log_dis = np.log(dis)
    log_dis[dis < PRIOR_CUTOFF] = -1e10          # zero-probability classes
    return np.concatenate([log_dis[1:], log_dis[:1]])   # move the blank (index 0) to the end
