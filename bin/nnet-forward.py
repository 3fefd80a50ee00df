#!/usr/bin/env python3
"""Forward pass for WFST decoding: per utterance (log-)posteriors minus log-prior -> Kaldi ark.
Command line of mobvoi/lstm_ctc bin/nnet-forward.py (main 29-113, flags 129-151).  Utterances are pushed
through the GPU in padded batches of consecutive files (SURVEY.md §8f NEXT-3); the outputs are identical
to one-at-a-time runs because padding is masked, and are written in scp order."""
import argparse
import os
import sys

from _common import setup_device, str2bool


def main(args):
    import numpy as np
    import torch
    device, _, _, _ = setup_device()
    import lstm_ctc_amd.nnet as nnet
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.kaldi_io import BaseFloatMatrixWriter
    from lstm_ctc_amd.nnet import tflog
    writer = BaseFloatMatrixWriter(args.nnet_output)
    nnet_config = nnet.parse_config(args.nnet_config)
    nnet_config['is_training'] = False
    if args.apply_log:
        args.apply_softmax = True
    class_prior = None if args.class_prior is None else nnet.get_class_prior(args.class_prior)
    filename, tfrecord, _ = nnet.dataset_from_tfrecords(
        tfrecords_scp=args.tfrecords_scp, left_context=nnet_config.get('left_context'),
        right_context=nnet_config.get('right_context'), subsample=nnet_config.get('subsample'), shuffle=False)
    _, pipeline = nnet.create_pipeline_sequential(filename=filename, tfrecord=tfrecord)
    graph = nnet.create_graph_for_inference(pipeline=pipeline, nnet_config=nnet_config,
                                            smooth_factor=args.smooth_factor, device=device)
    graph.restore(args.nnet_in)
    prior_d = None if class_prior is None else torch.from_numpy(class_prior).to(device)
    try:
        processed = 0
        pending = []

        def flush():
            nonlocal processed
            if not pending:
                return
            logits, seq = graph.forward_batch([p["nnet_input"] for p in pending])     # [T,B,V] on device
            T, B, V = logits.shape
            out = ops.posteriors(logits.reshape(T * B, V), args.smooth_factor, args.apply_softmax, args.apply_log,
                                 prior_d).view(T, B, V).cpu().numpy()
            for b, p in enumerate(pending):
                key, _ = os.path.splitext(os.path.basename(p["filename"]))
                writer.Write(key, out[:seq[b], b])
                processed += 1
                if args.report_interval and processed % args.report_interval == 0:
                    tflog.info('processed = %d' % processed)
            pending.clear()

        for item in pipeline:
            pending.append(item)
            if len(pending) >= args.batch_utts:
                flush()
        flush()
        tflog.info('done')
    except KeyboardInterrupt:
        tflog.fatal('interrupted by user')
        sys.exit(1)
    writer.Close()


if __name__ == '__main__':
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('tfrecords_scp', metavar='<tfrecords-scp>', type=str, help='tfrecords-scp.')
    parser.add_argument('nnet_config', metavar='<nnet-config>', type=str, help='nnet-config.')
    parser.add_argument('nnet_in', metavar='<nnet-in>', type=str, help='nnet-in.')
    parser.add_argument('nnet_output', metavar='<nnet-output-wspecifier>', type=str,
                        help='wspecifier for nnet-output.')
    parser.add_argument('--apply-softmax', metavar='apply-softmax', type=str2bool, default='true',
                        help='whether to apply softmax.')
    parser.add_argument('--apply-log', metavar='apply-log', type=str2bool, default='true',
                        help='whether to apply log on top of softmax')
    parser.add_argument('--report-interval', metavar='report-interval', type=int, default=100,
                        help='progress report interval.')
    parser.add_argument('--class-prior', metavar='class-prior', type=str, default=None,
                        help='class prior to scale the softmax output')
    parser.add_argument('--smooth-factor', metavar='smooth factor', type=float, default=1.0,
                        help='smooth factor for softmax')
    parser.add_argument('--batch-utts', metavar='batch-utts', type=int, default=16,
                        help='utterances per padded GPU batch (new; results do not depend on it)')
    args = parser.parse_args()
    sys.stderr.write('INFO:tensorflow:' + ' '.join(sys.argv) + '\n')
    main(args)
