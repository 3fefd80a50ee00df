"""GEMM timings on the c2 / c3 layer shapes (N = 320 / 512 models), next to torch.mm (hipBLASLt).  Development tool."""
import sys
import torch
sys.path.insert(0, ".")
from tools.probe import timeit
from lstm_ctc_amd import ops

for tag, N, V in (("c2", 320, 72), ("c3", 512, 72)):
    R = 32000
    shapes = [("NN zx   ", 0, 0, R, 4 * N, 2 * N), ("NT dY   ", 0, 1, R, 2 * N, 4 * N), ("TN dKx  ", 1, 0, 2 * N, 4 * N, R),
              ("NN proj ", 0, 0, R, N, N), ("NT dm'  ", 0, 1, R, N, N), ("TN dproj", 1, 0, N, N, R), ("TN dR   ", 1, 0, N, 4 * N, R),
              ("NN zx0  ", 0, 0, R, 4 * N, 40), ("TN dKx0 ", 1, 0, 40, 4 * N, R), ("NN head ", 0, 0, R, V, 2 * N)]
    tot = tot2 = 0.0
    for (name, ta, tb, M, Nn, K) in shapes:
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((Nn, K) if tb else (K, Nn), device="cuda")
        C = torch.empty((M, Nn), device="cuda")
        t = timeit(lambda: ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=C), iters=10)
        At, Bt = (A.t() if ta else A), (B.t() if tb else B)
        t2 = timeit(lambda: torch.mm(At, Bt, out=C), iters=10)
        fl = 2.0 * M * Nn * K
        tot += t; tot2 += t2
        print("%s %s M=%6d N=%5d K=%6d: mine %7.1f us %6.1f TF | torch.mm %7.1f us %6.1f TF" %
              (tag, name, M, Nn, K, t * 1e6, fl / t / 1e12, t2 * 1e6, fl / t2 / 1e12), flush=True)
    print("%s sum: mine %.1f us, torch.mm %.1f us" % (tag, tot * 1e6, tot2 * 1e6), flush=True)
