#!/usr/bin/env python3
"""Random initialisation + CV pass + save.  Mirrors mobvoi/lstm_ctc bin/nnet-init.py (main 25-91, flags 107-128)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _here)
_spec = importlib.util.spec_from_file_location("nnet_validate_cli", os.path.join(_here, "nnet-validate.py"))
_val = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_val)

if __name__ == '__main__':
    args = _val.build_parser(True).parse_args()
    sys.stderr.write('INFO:tensorflow:' + ' '.join(sys.argv) + '\n')
    _val.run(args, init_only=True)
