#!/usr/bin/env python3
"""Writes tests/golden/tf_host_arith_known_answers.json: the input / expected-output vectors that TensorFlow's OWN unit
tests hold for the host-arithmetic rows of the path (nnet/graph.py:37-48,183-200
clip + optimizers, nnet/graph.py:138-150 edit distance).  TensorFlow 1.8 is not installable here and is not vendored
under /root/reference, so the vectors are RECALLED from the upstream test files named per entry (same status as
ctc_tf_known_answers.json); each is arithmetic that can be checked by hand, and the hand check is in the `why` field.

  * clip_ops_test.py  ClipTest.testClipByGlobalNormClipped / NotClipped / Zero
  * adam_test.py      AdamOptimizerTest.testBasic: its expected values are produced by the test file's own numpy
                      recurrence `adam_update_numpy` over 3 steps - restated here verbatim in float64 and then
                      stored as numbers, so the fixture is data and the test does not re-derive it
  * momentum_test.py, gradient_descent_test.py   testBasic of each (the tests' own closed forms)
  * edit_distance_op_test.py  testEditDistanceNormalized / Unnormalized inputs (normalize=False distances are
                      the integers behind both) + the tf.edit_distance docstring example
"""
import json
import os

import numpy as np


def adam_update_numpy(param, g_t, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    # tensorflow/python/training/adam_test.py (r1.8), lines 36-47
    alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m_t = beta1 * m + (1 - beta1) * g_t
    v_t = beta2 * v + (1 - beta2) * g_t * g_t
    param_t = param - alpha_t * m_t / (np.sqrt(v_t) + epsilon)
    return param_t, m_t, v_t


def main():
    out = {"_source": "TensorFlow upstream unit tests (r1.8): clip_ops_test.py, adam_test.py, edit_distance_op_test.py. "
                      "Recalled external known answers, not produced by mobvoi/lstm_ctc; see make_tf_host_fixtures.py."}
    out["clip_by_global_norm"] = [
        {"name": "testClipByGlobalNormClipped", "tensors": [[-2.0, 0.0, 0.0, 4.0, 0.0, 0.0], [1.0, -2.0]],
         "clip_norm": 4.0, "global_norm": 5.0,
         "expected": [[-1.6, 0.0, 0.0, 3.2, 0.0, 0.0], [0.8, -1.6]],
         "why": "sqrt(4 + 16 + 1 + 4) = 5; every tensor scaled by 4 / 5"},
        {"name": "testClipByGlobalNormNotClipped", "tensors": [[-2.0, 0.0, 0.0, 4.0, 0.0, 0.0], [1.0, -2.0]],
         "clip_norm": 6.0, "global_norm": 5.0,
         "expected": [[-2.0, 0.0, 0.0, 4.0, 0.0, 0.0], [1.0, -2.0]],
         "why": "norm 5 < 6: unchanged"},
        {"name": "testClipByGlobalNormZero", "tensors": [[0.0, 0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0]],
         "clip_norm": 6.0, "global_norm": 0.0,
         "expected": [[0.0, 0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0]],
         "why": "all-zero gradient: norm 0, result 0 (no 0/0 NaN)"},
    ]
    var = [np.array([1.0, 2.0]), np.array([3.0, 4.0])]
    grads = [np.array([0.1, 0.1]), np.array([0.01, 0.01])]
    m = [np.zeros(2), np.zeros(2)]
    v = [np.zeros(2), np.zeros(2)]
    steps = []
    for t in range(1, 4):
        for i in range(2):
            var[i], m[i], v[i] = adam_update_numpy(var[i], grads[i], t, m[i], v[i])
        steps.append({"t": t, "var0": var[0].tolist(), "var1": var[1].tolist()})
    out["adam"] = {"name": "AdamOptimizerTest.testBasic", "lr": 0.001, "beta1": 0.9, "beta2": 0.999, "epsilon": 1e-8,
                   "var0": [1.0, 2.0], "var1": [3.0, 4.0], "grads0": [0.1, 0.1], "grads1": [0.01, 0.01],
                   "steps": steps,
                   "why": "constant gradient: m_t / (1 - b1^t) = g and v_t / (1 - b2^t) = g^2, so every step moves each "
                          "weight by ~lr = 1e-3 (var0[0]: 1.0 -> 0.999 -> 0.998 -> 0.997)"}
    # momentum_test.py MomentumOptimizerTest.testBasic (learning_rate 2.0, momentum 0.9, use_nesterov False): the test's own
    # closed forms - accum = momentum * accum + grad; var -= lr * accum
    out["momentum"] = {"name": "MomentumOptimizerTest.testBasic", "lr": 2.0, "momentum": 0.9,
                       "var0": [1.0, 2.0], "var1": [3.0, 4.0], "grads0": [0.1, 0.1], "grads1": [0.01, 0.01],
                       "steps": [
                           {"t": 1, "var0": [1.0 - 0.1 * 2.0, 2.0 - 0.1 * 2.0], "var1": [3.0 - 0.01 * 2.0, 4.0 - 0.01 * 2.0]},
                           {"t": 2, "var0": [1.0 - 0.1 * 2.0 - (0.9 * 0.1 + 0.1) * 2.0, 2.0 - 0.1 * 2.0 - (0.9 * 0.1 + 0.1) * 2.0],
                            "var1": [2.98 - (0.9 * 0.01 + 0.01) * 2.0, 3.98 - (0.9 * 0.01 + 0.01) * 2.0]}],
                       "why": "step 1: accumulator = gradient; step 2: accumulator = 0.9 * g + g = 0.19 (0.019)"}
    # gradient_descent_test.py GradientDescentOptimizerTest.testBasic (learning_rate 3.0)
    out["sgd"] = {"name": "GradientDescentOptimizerTest.testBasic", "lr": 3.0,
                  "var0": [1.0, 2.0], "var1": [3.0, 4.0], "grads0": [0.1, 0.1], "grads1": [0.01, 0.01],
                  "steps": [{"t": 1, "var0": [1.0 - 3.0 * 0.1, 2.0 - 3.0 * 0.1], "var1": [3.0 - 3.0 * 0.01, 4.0 - 3.0 * 0.01]}],
                  "why": "var -= lr * grad"}
    out["edit_distance"] = [
        {"name": "testEditDistanceNormalized (inputs; normalize=False distances)",
         "hyp": [[0, 1], [1, -1]], "truth": [[0], [1, 1]], "expected": [1, 1],
         "why": "normalized expected [1.0, 0.5] = [1/1, 1/2]"},
        {"name": "testEditDistanceUnnormalized", "hyp": [[10], [10, 11]], "truth": [[1, 2], [1, -1]],
         "expected": [2, 2], "why": "no common symbol: max(len) substitutions / insertions"},
        {"name": "tf.edit_distance docstring example (a=0, b=1, c=2), flattened over its [2,2] batch",
         "hyp": [[0], [], [1], []], "truth": [[], [0], [1, 2], [0]], "expected": [1, 1, 1, 1],
         "why": "normalized doc output [[inf, 1.0], [0.5, 1.0]]: distances 1 / 0, 1 / 1, 1 / 2, 1 / 1"},
    ]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_host_arith_known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
