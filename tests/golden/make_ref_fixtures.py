#!/usr/bin/env python3
"""Generates tests/golden/ref_host_fixtures.{json,npz} by RUNNING the reference's own host-side modules.

Build-container only: needs /root/reference (mobvoi/lstm_ctc).  The reference's Python never travels to the GPU
box - only the input/output vectors written here do.  Re-run with `python tests/golden/make_ref_fixtures.py`.

What runs and what it pins (last row):
  * nnet/config.py:40-63    parse_config       on the WSJ / Libri recipe config texts and on edge-case lines
  * nnet/class_prior.py:30-47 get_class_prior  on several label.counts lines
  * pyKaldiIO/kaldi_matrix.py:280-299 WriteFloatMatrixToStream (+ io_funcs.py:86-100,231-254) - the bytes of a
    binary and of a text float matrix; the archive entry is '<key> ' + those bytes (pyKaldiIO/kaldi_table.py:959-960;
    kaldi_table itself needs cStringIO and cannot be imported under Python 3).

The four modules are Python-2 sources that import cleanly under Python 3 as top-level modules.  Two py2-isms have
to be bridged to EXECUTE the matrix writer, both in the module namespace only (no reference file is touched):
`chr(4)` must yield a 1-byte bytes object for struct.pack('c', ...) and `xrange` is `range`.
The arithmetic of the hot path (TensorFlow 1.8) cannot be run here and stays unpinned by the reference.
"""
import json
import os
import re
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def recipe_config_text(script, extra):
    """The text the recipe's `(echo "k = $v" ...) > $nnet_config` block writes, with the script's own defaults."""
    src = open(os.path.join(REF, script)).read()
    env = dict(extra)
    for m in re.finditer(r"^([a-z_]+)=([^\s#`$]+)", src, flags=re.M):
        env.setdefault(m.group(1), m.group(2).strip('"'))
    block = re.search(r"nnet_config=\$dir/nnet.config\n(.*?)\) > \$nnet_config", src, flags=re.S).group(1)
    lines = []
    for m in re.finditer(r'echo "([^"]*)"', block):
        lines.append(re.sub(r"\$([a-z_]+)", lambda v: str(env[v.group(1)]), m.group(1)))
    return "\n".join(lines) + "\n"


class RecordingStream:
    """Stands in for KaldiOutputStream: collects what the writer emits (py2 str == bytes)."""

    def __init__(self):
        self.buf = bytearray()

    def Write(self, data):
        self.buf += data if isinstance(data, (bytes, bytearray)) else data.encode("latin-1")


def main():
    sys.path.insert(0, os.path.join(REF, "nnet"))
    sys.path.insert(0, os.path.join(REF, "pyKaldiIO"))
    import config as ref_config
    import class_prior as ref_prior
    import io_funcs as ref_io
    import kaldi_matrix as ref_mat
    ref_io.chr = lambda v: bytes([v])          # py2 chr() -> 1-char str == 1 byte
    ref_mat.xrange = range

    out_json = {"generator": "tests/golden/make_ref_fixtures.py", "reference": "mobvoi/lstm_ctc", "config": [],
                "class_prior": [], "matrix": []}
    arrays = {}

    # ---- parse_config
    texts = {
        "wsj_recipe (egs/wsj/run_wsj_phn.sh:226-243)":
            recipe_config_text("egs/wsj/run_wsj_phn.sh", dict(input_dim=120, num_targets=72, dir="exp/wsj",
                                                              prior_label_path="exp/wsj/label.counts")),
        "libri_recipe (egs/libri/run_libri_ph.sh:285-301, double space after prior_label_sm =)":
            recipe_config_text("egs/libri/run_libri_ph.sh", dict(input_dim=120, num_targets=44, dir="exp/libri",
                                                                 prior_label_path="exp/libri/label.counts")),
        "coercions":
            "a = 1\nb = -7\nc = 1e-3\nd = .5\ne = TRUE\nf = False\ng = yes\nh = 3.0\ni = 0x10\n"
            "k = nan\nl = inf\nm = path/to/file.txt\n",
        "comments_and_tokens":
            "# a full comment line\nkey = 5 # trailing comment\n  indented = 2.5\nlonely\nmulti = a b c\n"
            "x #y = 9\nuse_peepholes = true   \n",
    }
    for name, text in texts.items():
        with tempfile.NamedTemporaryFile("w", suffix=".config", delete=False) as f:
            f.write(text)
        parsed = ref_config.parse_config(f.name)
        os.unlink(f.name)
        # JSON has no nan/inf/int-vs-float distinction problems if the python type is recorded next to the repr
        out_json["config"].append({"name": name, "text": text,
                                   "expected": {k: [type(v).__name__, repr(v)] for k, v in parsed.items()}})

    # ---- get_class_prior
    count_lines = {
        "survey_example": " [ 10 5 0 85 ]\n",
        "single_class": "[ 7 ]\n",
        "no_brackets": "3 1 4 1 5 9 2 6\n",
        "uniform": "[ 2 2 2 2 2 ]\n",
        "tiny_and_huge": "[ 1e-12 1 1e12 3 ]\n",
        "second_line_ignored": "[ 1 2 3 ]\n[ 9 9 9 ]\n",
        "wsj_like_72": "[ " + " ".join(str((i * 7919) % 997 + (0 if i % 13 else 1000)) for i in range(72)) + " ]\n",
    }
    for name, line in count_lines.items():
        with tempfile.NamedTemporaryFile("w", suffix=".counts", delete=False) as f:
            f.write(line)
        with np.errstate(divide="ignore"):
            prior = ref_prior.get_class_prior(f.name)
        os.unlink(f.name)
        assert prior.dtype == np.float32
        out_json["class_prior"].append({"name": name, "text": line, "array": "prior_" + name})
        arrays["prior_" + name] = prior

    # ---- float-matrix writer
    rng = np.random.default_rng(20181)
    mats = {
        "one_by_one": np.array([[1.5]], np.float32),
        "posteriors_3x4": np.log(rng.dirichlet(np.ones(4), size=3)).astype(np.float32),
        "wide_2x44": rng.normal(size=(2, 44)).astype(np.float32),
        "specials": np.array([[0.0, -0.0, 1e-38, -1e10], [np.inf, -np.inf, 3.4e38, 1.0000001]], np.float32),
        "no_rows_0x5": np.zeros((0, 5), np.float32),
    }
    for name, m in mats.items():
        for binary in (True, False):
            s = RecordingStream()
            assert ref_mat.WriteFloatMatrixToStream(s, binary, m)
            key = "utt_%s" % name
            entry = ("%s " % key).encode("latin-1") + bytes(s.buf)          # kaldi_table.py:959 writes '%s ' % key first
            tag = "mat_%s_%s" % (name, "bin" if binary else "txt")
            arrays[tag] = np.frombuffer(entry, np.uint8).copy()
            out_json["matrix"].append({"name": name, "binary": binary, "key": key, "value": "matval_" + name,
                                       "entry_bytes": tag})
        arrays["matval_" + name] = m

    with open(os.path.join(HERE, "ref_host_fixtures.json"), "w") as f:
        json.dump(out_json, f, indent=1, sort_keys=True)
    np.savez(os.path.join(HERE, "ref_host_fixtures.npz"), **arrays)
    print("wrote", len(out_json["config"]), "configs,", len(out_json["class_prior"]), "priors,",
          len(out_json["matrix"]), "matrix entries")


if __name__ == "__main__":
    main()
