"""The stderr log contract of the reference: tf.logging prefixes every line with
``INFO:tensorflow:`` / ``FATAL:tensorflow:`` and the recipe scripts grep those lines
(scripts/train.sh:145,156-157)."""
import sys


def info(msg, *args):
    sys.stderr.write("INFO:tensorflow:" + (msg % args if args else msg) + "\n")
    sys.stderr.flush()


def fatal(msg, *args):
    sys.stderr.write("FATAL:tensorflow:" + (msg % args if args else msg) + "\n")
    sys.stderr.flush()
