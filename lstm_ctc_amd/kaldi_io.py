"""Kaldi binary float-matrix table writer — the one piece of the reference's pyKaldiIO that the hot path's
output side needs (``pyKaldiIO.BaseFloatMatrixWriter`` as used by bin/nnet-forward.py:30-31,96,113).

Binary matrix entry (pyKaldiIO/kaldi_table.py:950-975, kaldi_matrix.py:280-287, io_funcs.py:86-100,231-254):
    ``<key> `` ``\\0B`` ``FM `` ``\\x04`` int32 rows ``\\x04`` int32 cols, then rows*cols float32, row-major.
Supported wspecifiers: ``ark:-``, ``ark:<file>``, ``ark,scp:<ark>,<scp>`` and the ``t`` (text) option.
"""
import struct
import sys

import numpy as np


class BaseFloatMatrixWriter:
    def __init__(self, wspecifier):
        spec, _, target = wspecifier.partition(":")
        opts = [o.strip() for o in spec.split(",")]
        if "ark" not in opts:
            raise ValueError("unsupported wspecifier (need ark): %s" % wspecifier)
        self.text = "t" in opts
        self.scp = None
        ark_path = target
        if "scp" in opts:
            first, second = [t.strip() for t in target.split(",")]
            ark_path, scp_path = (first, second) if opts.index("ark") < opts.index("scp") else (second, first)
            self.scp = open(scp_path, "w")
        self.ark_path = ark_path
        self.close_ark = ark_path != "-"
        self.ark = sys.stdout.buffer if ark_path == "-" else open(ark_path, "wb")

    def Write(self, key, matrix):
        m = np.ascontiguousarray(matrix, dtype=np.float32)
        assert m.ndim == 2
        self.ark.write((key + " ").encode())
        if self.scp is not None:
            self.scp.write("%s %s:%d\n" % (key, self.ark_path, self.ark.tell()))
        if self.text:                                 # kaldi_matrix.py:289-298: ' [' + rows of '%f ' + ']'
            if m.shape[0] == 0 or m.shape[1] == 0:
                self.ark.write(b" []\n")
            else:
                body = "".join("\n  " + "".join("%f " % v for v in row) for row in m.tolist())
                self.ark.write((" [" + body + "]\n").encode())
        else:
            self.ark.write(b"\0B" + b"FM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" +
                           struct.pack("<i", m.shape[1]))
            self.ark.write(m.astype("<f4").tobytes())
        return True

    def Close(self):
        self.ark.flush()
        if self.close_ark:
            self.ark.close()
        if self.scp is not None:
            self.scp.close()


def read_float_matrix_ark(path):
    """Minimal reader of the binary entries written above (tests / round trips)."""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        sp = data.index(b" ", pos)
        key = data[pos:sp].decode()
        pos = sp + 1
        assert data[pos:pos + 5] == b"\0BFM ", data[pos:pos + 5]
        pos += 5
        assert data[pos] == 4
        rows = struct.unpack("<i", data[pos + 1:pos + 5])[0]
        assert data[pos + 5] == 4
        cols = struct.unpack("<i", data[pos + 6:pos + 10])[0]
        pos += 10
        out[key] = np.frombuffer(data[pos:pos + 4 * rows * cols], "<f4").reshape(rows, cols).copy()
        pos += 4 * rows * cols
    return out
