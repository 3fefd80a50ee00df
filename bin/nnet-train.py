#!/usr/bin/env python3
"""One epoch of CTC training: <nnet-in> -> <nnet-out>.  Command line, log lines and exit codes of
mobvoi/lstm_ctc bin/nnet-train.py (flags 113-151, main 26-100), so scripts/train*.sh run unmodified.
Launch under ``python -m torch.distributed.run --nproc-per-node N`` for utterance-batch data parallelism."""
import argparse
import sys

from _common import setup_device, str2bool, quiet_unless_rank0


def main(args):
    try:
        device, pg, rank, world = setup_device()
        import lstm_ctc_amd.nnet as nnet
        from lstm_ctc_amd.nnet import tflog
        quiet_unless_rank0(rank)
        nnet_config = nnet.parse_config(args.nnet_config)
        nnet_config['is_training'] = True
        nnet_type = nnet_config.get('nnet_type')
        filename, tfrecord, input_dim = nnet.dataset_from_tfrecords(
            tfrecords_scp=args.tfrecords_scp, left_context=nnet_config.get('left_context'),
            right_context=nnet_config.get('right_context'), subsample=nnet_config.get('subsample'),
            shuffle=args.shuffle, seed=args.seed)
        if args.objective != 'ctc':
            tflog.fatal('unsupported objective: %s' % args.objective)
            sys.exit(1)
        if nnet_type not in ('blstm', 'lstm'):
            tflog.fatal('unsupported nnet_type: %s' % nnet_type)
            sys.exit(1)
        _, pipeline = nnet.create_pipeline_sequence_batch(dataset=tfrecord, input_dim=input_dim,
                                                          batch_size=args.batch_size, rank=rank, world_size=world)
        graph = nnet.create_graph_for_training_ctc(pipeline=pipeline, nnet_config=nnet_config,
                                                   learn_rate=args.learn_rate, clip_norm=args.clip_norm,
                                                   optimizer=args.optimizer, device=device, seed=args.seed,
                                                   process_group=pg)
        graph.restore(args.nnet_in)                       # trainable variables only; Adam slots restart
        sess = nnet.Session(graph)
        nnet.train(sess=sess, graph=graph, evaluate=args.evaluate, report_interval=args.report_interval)
        if rank == 0:
            tflog.info('saving nnet to "%s"' % args.nnet_out)
            graph.save(args.nnet_out)
    except KeyboardInterrupt:
        from lstm_ctc_amd.nnet import tflog
        tflog.fatal('interrupted by user')
        sys.exit(1)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('tfrecords_scp', metavar='<tfrecords.scp>', type=str, help='tfrecords.scp.')
    parser.add_argument('nnet_config', metavar='<nnet-config>', type=str, help='nnet-config.')
    parser.add_argument('nnet_in', metavar='<nnet-in>', type=str, help='nnet-in.')
    parser.add_argument('nnet_out', metavar='<nnet-out>', type=str, help='nnet-out.')
    parser.add_argument('--objective', metavar='objective', help='objective function.', type=str, default='xent')
    parser.add_argument('--optimizer', metavar='optimizer', help='optimizer to be used.', type=str, default='sgd')
    parser.add_argument('--evaluate', metavar='evaluate', type=str2bool, default='false',
                        help='whether to evaluate the model in addition to loss.')
    parser.add_argument('--learn-rate', metavar='learn-rate', type=float, help='learning rate.', default=0.0001)
    parser.add_argument('--batch-size', metavar='batch-size', type=int, help='batch size.', default=256)
    parser.add_argument('--batch-threads', metavar='batch-threads', type=int, help='batch threads.', default=8)
    parser.add_argument('--seed', metavar='seed', type=int, help='seed for shuffling training data.', default=777)
    parser.add_argument('--num-parallel-calls', metavar='num-parallel-calls', type=int, default=32,
                        help='num-parallel-calls.')
    parser.add_argument('--report-interval', metavar='report-interval', type=int, default=100,
                        help='progress report interval.')
    parser.add_argument('--shuffle', metavar='do shuffle in the training', type=str2bool, default='true',
                        help='whether to shuffle training data.')
    parser.add_argument('--clip-norm', metavar='gradient clip norm', type=float, help='gradient clip norm',
                        default=5.0)
    args = parser.parse_args()
    sys.stderr.write('INFO:tensorflow:' + ' '.join(sys.argv) + '\n')
    main(args)
