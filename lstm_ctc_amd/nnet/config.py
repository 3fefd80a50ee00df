"""``nnet.config`` reader with the reference's coercion rules (mobvoi/lstm_ctc nnet/config.py:40-63).

File format: one ``key = value`` pair per line.  Only the first and the last whitespace-separated
tokens matter (so ``prior_label_sm =  0`` with a double space parses), tokens beginning with ``#``
are ignored, and a line whose first character is ``#`` is a comment.  The value becomes an int if
``int()`` accepts it, else a float, else a bool for ``true``/``false`` (any case), else stays a str.

Deliberate leniency: blank lines are skipped (the reference dies with IndexError on them).
"""

_BOOLS = {"true": True, "false": False}


def _coerce(text):
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            pass
    return _BOOLS.get(text.lower(), text)


def parse_config(fn):
    """Returns the config dict for the file at ``fn``."""
    config = {}
    with open(fn, "r") as handle:
        for raw in handle:
            line = raw.strip()
            if line == "" or line[0] == "#":
                continue
            fields = [tok for tok in line.split() if tok[0] != "#"]
            if fields:
                config[fields[0]] = _coerce(fields[-1])
    return config
