"""Shared plumbing of the bin/nnet-*.py command lines (flag parsing quirks, device / process-group setup)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def str2bool(v):
    """yes/true/t/y/1 and no/false/f/n/0, as every reference CLI does (bin/nnet-train.py:103-109)."""
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def setup_device():
    """One process per GPU.  Returns (device, process_group or None, rank, world_size)."""
    import torch
    if not torch.cuda.is_available():
        sys.stderr.write("FATAL:tensorflow:no GPU visible - lstm_ctc_amd has no CPU path\n")
        sys.exit(1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # tests only (tests/test_gpu_round6.py): LC_DP_TEST_SHARED_GPU=1 puts every rank on GPU 0 with gloo as the collective
    # backend (RCCL refuses two ranks on one device), so that an 8-rank launch can be rehearsed on a one-GPU box
    shared = world > 1 and os.environ.get("LC_DP_TEST_SHARED_GPU") == "1"
    if shared:
        local = 0
    elif torch.cuda.device_count() <= local:
        sys.stderr.write("FATAL:tensorflow:rank %d needs GPU %d but %d GPU(s) are visible (one process per GPU)\n"
                         % (rank, local, torch.cuda.device_count()))
        sys.exit(1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)      # nccl == RCCL on ROCm
        pg = torch.distributed.group.WORLD
    return device, pg, rank, world


def quiet_unless_rank0(rank):
    """Only rank 0 emits the machine-parsed log lines (scripts/train.sh greps one tr_loss / cv_loss line)."""
    if rank != 0:
        from lstm_ctc_amd.nnet import tflog
        tflog.info = lambda *a, **k: None


# ---- the command-line contract ---------------------------------------------------------------------------------------
# Flag names, types and defaults are what scripts/train*.sh and scripts/decode_*.sh pass (bin/nnet-train.py:113-151,
# nnet-validate.py:108-129, nnet-init.py:107-128, nnet-forward.py:129-151 of the reference) and must not move; the texts
# describe what the flag does HERE.
POSITIONAL = {
    'tfrecords_scp': ('<tfrecords.scp>', 'list of utterances: "<key> <rows> <cols> <has_label> <file.tfrecords>" per line'),
    'nnet_config': ('<nnet-config>', '"key = value" model description (nnet.parse_config)'),
    'nnet_in': ('<nnet-in>', 'checkpoint to start from (safetensors at exactly this path)'),
    'nnet_out': ('<nnet-out>', 'checkpoint to write (safetensors at exactly this path)'),
    'nnet_output': ('<nnet-output-wspecifier>', 'Kaldi table wspecifier (ark:... / ark,t:... / ark,scp:...) for the outputs'),
}
FLAGS = {
    '--objective': dict(type=str, default='xent', help='training criterion; only "ctc" is implemented (the recipes pass it)'),
    '--optimizer': dict(type=str, default='sgd', help='sgd | momentum (0.9) | adam, constant learning rate'),
    '--evaluate': dict(type=str2bool, default='false', help='also greedy-decode every batch and report the token error rate'),
    '--learn-rate': dict(type=float, default=0.0001, help='step size of the optimizer'),
    '--batch-size': dict(type=int, default=256, help='utterances per step (per GPU under torchrun)'),
    '--batch-threads': dict(type=int, default=8, help='batches assembled concurrently by the loader (capped at 4)'),
    '--seed': dict(type=int, default=777, help='seed of the file-list shuffle, the dropout stream and (nnet-train) the graph'),
    '--num-parallel-calls': dict(type=int, default=32, help='loader threads decoding TFRecords (native code, GIL released)'),
    '--report-interval': dict(type=int, default=100, help='log a progress line every this many steps / utterances'),
    '--shuffle': dict(type=str2bool, default='true', help='permute the utterance list (seeded) before batching'),
    '--clip-norm': dict(type=float, default=5.0, help='global-norm bound applied to the summed gradient'),
    '--apply-softmax': dict(type=str2bool, default='true', help='write softmax(smooth * logits) instead of logits'),
    '--apply-log': dict(type=str2bool, default='true', help='write log-posteriors (implies --apply-softmax)'),
    '--class-prior': dict(type=str, default=None, help='label.counts file; its log-prior is subtracted from the log-posteriors'),
    '--smooth-factor': dict(type=float, default=1.0, help='temperature multiplied into the logits before the softmax'),
    '--batch-utts': dict(type=int, default=16, help='utterances per padded GPU batch (new; results do not depend on it)'),
}


def build_cli(positional, flags):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for name in positional:
        metavar, text = POSITIONAL[name]
        parser.add_argument(name, metavar=metavar, type=str, help=text)
    for flag in flags:
        parser.add_argument(flag, metavar=flag.lstrip('-'), **FLAGS[flag])
    return parser
