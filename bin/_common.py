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
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=device)      # nccl == RCCL on ROCm
        pg = torch.distributed.group.WORLD
    return device, pg, rank, world


def quiet_unless_rank0(rank):
    """Only rank 0 emits the machine-parsed log lines (scripts/train.sh greps one tr_loss / cv_loss line)."""
    if rank != 0:
        from lstm_ctc_amd.nnet import tflog
        tflog.info = lambda *a, **k: None
