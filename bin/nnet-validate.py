#!/usr/bin/env python3
"""CV pass of a saved model: logs ``cv_loss`` / ``cv_eval``.  Mirrors mobvoi/lstm_ctc bin/nnet-validate.py
(main 26-92, flags 108-129)."""
import sys

from _common import build_cli, setup_device, quiet_unless_rank0


def run(args, init_only):
    try:
        device, pg, rank, world = setup_device()
        import lstm_ctc_amd.nnet as nnet
        from lstm_ctc_amd.nnet import tflog
        quiet_unless_rank0(rank)
        nnet_config = nnet.parse_config(args.nnet_config)
        nnet_config['is_training'] = False
        nnet_type = nnet_config.get('nnet_type')
        filename, tfrecord, input_dim = nnet.dataset_from_tfrecords(
            tfrecords_scp=args.tfrecords_scp, left_context=nnet_config.get('left_context'),
            right_context=nnet_config.get('right_context'), subsample=nnet_config.get('subsample'), shuffle=False)
        if args.objective != 'ctc':
            tflog.fatal('unsupported objective: %s' % args.objective)
            sys.exit(1)
        if nnet_type not in ('blstm', 'cudnnlstm', 'lstm'):
            tflog.fatal('unsupported nnet_type: %s' % nnet_type)
            sys.exit(1)
        _, pipeline = nnet.create_pipeline_sequence_batch(dataset=tfrecord, input_dim=input_dim,
                                                          batch_size=args.batch_size, batch_threads=args.batch_threads,
                                                          rank=rank, world_size=world)
        # nnet-init sets no graph seed (bin/nnet-init.py:27-31): fresh random weights on every run
        graph = nnet.create_graph_for_validation_ctc(pipeline=pipeline, nnet_config=nnet_config, device=device,
                                                     seed=None if init_only else 123)
        if init_only and pg is not None:                   # every rank must score the same random model
            from lstm_ctc_amd.nnet import dp
            dp.broadcast_(graph.model.ps.flat, pg, src=0)
        graph.pg, graph.world = pg, world
        if not init_only:
            graph.restore(args.nnet_in)
        sess = nnet.Session(graph)
        nnet.validate(sess=sess, graph=graph, evaluate=args.evaluate, report_interval=args.report_interval)
        if init_only and rank == 0:
            tflog.info('saving nnet to "%s"' % args.nnet_out)
            graph.save(args.nnet_out)
    except KeyboardInterrupt:
        from lstm_ctc_amd.nnet import tflog
        tflog.fatal('interrupted by user')
        sys.exit(1)


def build_parser(init_only):
    return build_cli(('tfrecords_scp', 'nnet_config', 'nnet_out' if init_only else 'nnet_in'),
                     ('--objective', '--evaluate', '--batch-size', '--batch-threads', '--report-interval'))


if __name__ == '__main__':
    args = build_parser(False).parse_args()
    sys.stderr.write('INFO:tensorflow:' + ' '.join(sys.argv) + '\n')
    run(args, init_only=False)
