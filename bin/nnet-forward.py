#!/usr/bin/env python3
"""Forward pass for WFST decoding: per utterance (log-)posteriors minus log-prior -> Kaldi ark.
Command line of mobvoi/lstm_ctc bin/nnet-forward.py (main 29-113, flags 129-151).  Utterances are pushed
through the GPU in padded batches of consecutive files (SURVEY.md §8f NEXT-3); the outputs are identical
to one-at-a-time runs because padding is masked, and are written in scp order."""
import os
import sys

from _common import build_cli, setup_device


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
    from lstm_ctc_amd.nnet.funcs import StepWatchdog
    dog = StepWatchdog(tag="forward batch").start()     # a hung runtime call ends the process with status 1 (LC_STEP_TIMEOUT)
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
            dog.kick()                 # the device part of the batch is done ...
            dog.pause()                # ... and back-pressure from whoever reads the archive (copy-feats on a pipe) is not a hang
            for b, p in enumerate(pending):
                key, _ = os.path.splitext(os.path.basename(p["filename"]))
                writer.Write(key, out[:seq[b], b])
                processed += 1
                if args.report_interval and processed % args.report_interval == 0:
                    tflog.info('processed = %d' % processed)
            pending.clear()
            dog.resume()

        for item in pipeline:
            pending.append(item)
            if len(pending) >= args.batch_utts:
                flush()
        flush()
        tflog.info('done')
    except KeyboardInterrupt:
        tflog.fatal('interrupted by user')
        sys.exit(1)
    finally:
        dog.stop()
    writer.Close()


if __name__ == '__main__':
    args = build_cli(('tfrecords_scp', 'nnet_config', 'nnet_in', 'nnet_output'),
                     ('--apply-softmax', '--apply-log', '--report-interval', '--class-prior', '--smooth-factor',
                      '--batch-utts')).parse_args()
    sys.stderr.write('INFO:tensorflow:' + ' '.join(sys.argv) + '\n')
    main(args)
