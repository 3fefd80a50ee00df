#!/usr/bin/env python3
"""One epoch of CTC training: <nnet-in> -> <nnet-out>.  Command line, log lines and exit codes of
mobvoi/lstm_ctc bin/nnet-train.py (flags 113-151, main 26-100), so scripts/train*.sh run unmodified.
Launch under ``python -m torch.distributed.run --nproc-per-node N`` for utterance-batch data parallelism."""
import sys

from _common import build_cli, setup_device, quiet_unless_rank0


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
            shuffle=args.shuffle, seed=args.seed, num_parallel_calls=args.num_parallel_calls)
        if args.objective != 'ctc':
            tflog.fatal('unsupported objective: %s' % args.objective)
            sys.exit(1)
        if nnet_type not in ('blstm', 'cudnnlstm', 'lstm'):
            tflog.fatal('unsupported nnet_type: %s' % nnet_type)
            sys.exit(1)
        _, pipeline = nnet.create_pipeline_sequence_batch(dataset=tfrecord, input_dim=input_dim,
                                                          batch_size=args.batch_size, batch_threads=args.batch_threads,
                                                          rank=rank, world_size=world)
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
    args = build_cli(('tfrecords_scp', 'nnet_config', 'nnet_in', 'nnet_out'),
                     ('--objective', '--optimizer', '--evaluate', '--learn-rate', '--batch-size', '--batch-threads',
                      '--seed', '--num-parallel-calls', '--report-interval', '--shuffle', '--clip-norm')).parse_args()
    sys.stderr.write('INFO:tensorflow:' + ' '.join(sys.argv) + '\n')
    main(args)
