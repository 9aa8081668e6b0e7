"""CLI entry with the reference's flags (/root/reference/MIND_2020/run_v0.py:15-28):

    python -m pytorch_news_recommender_amd.run_v0 --model nrms_hip --dataset synthetic
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m pytorch_news_recommender_amd.run_v0 ...

``--dataset large|demo`` reads what the reference's preprocessing leaves under config.data_path
(idx_train_datas.pkl / idx_dev_datas.pkl, news_title.pkl or news_words.csv, dev_behaviors.csv:
data_handler.py:43-135, train_eval.py:36-39); ``--dataset synthetic`` fabricates a MIND-shaped corpus (no
data ships offline).  ``--batch_size`` (not a reference flag; its scripts hard-code 512 / 256, run_v0.py:44,
run_demo.py:28) lets BASELINE config 0 (batch 32) run through the same entry.
"""
import argparse
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import parallel
from .config import Config
from .data_handler import DeviceFeed, MyDataset, SyntheticMind, load_dataset, read_dev_labels
from .model import Model
from .train_eval import test, train


def build_parser():
    parser = argparse.ArgumentParser(description='MIND')
    parser.add_argument('--model', type=str, required=True, help='choose the proper model')
    parser.add_argument('--dataset', default='large', type=str, help='large | demo | synthetic')
    parser.add_argument('--test', default=False, type=bool, help='run the test dataset')
    parser.add_argument('--n_GPUs', type=int, default=1, help='number of GPUs (one process each: use torchrun)')
    parser.add_argument('--load', type=str, default=None, help='load the pretrained model ckpt file')
    parser.add_argument('--description', type=str, default=None, help='description of the experiment')
    parser.add_argument('--epochs', type=int, default=None)
    parser.add_argument('--max_batches', type=int, default=None)
    parser.add_argument('--synthetic_users', type=int, default=20480)
    parser.add_argument('--batch_size', type=int, default=512)
    parser.add_argument('--precision', type=str, default=None, help='fp32 | bf16x3 | bf16 | fp16 (HIP path)')
    parser.add_argument('--num_workers', type=int, default=6)
    parser.add_argument('--feed', type=str, default='device', help="device: batches assembled in HBM from a resident news "
                        "table (DeviceFeed); loader: the reference's DataLoader(MyDataset) with --num_workers processes")
    parser.add_argument('--data_path', type=str, default=None, help='overrides config.data_path (./data_processed/)')
    parser.add_argument('--save_path', type=str, default=None, help='overrides config.save_path (./save_model/)')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, local_rank, world = parallel.init_process_group()
    torch.manual_seed(422)
    torch.cuda.manual_seed_all(422)
    model_name = args.model + '_' + (args.description or time.strftime('%m-%d_%H'))
    config = Config(model_name)
    config.batch_size = args.batch_size
    if args.precision:
        config.precision = args.precision
    config.num_epochs = 6 if args.epochs is None else args.epochs
    config.mode = args.dataset
    if args.data_path:
        config.data_path = os.path.join(args.data_path, '')
    if args.save_path:
        config.save_path = os.path.join(args.save_path, '')
        config.log_path = os.path.join(args.save_path, 'logs', model_name)
    config.__nrms__()

    if args.dataset == 'synthetic':
        config.n_words_title = 30
        corpus = SyntheticMind(config, n_news=4000, seed=0)
        os.makedirs(config.data_path, exist_ok=True)
        emb = os.path.join(config.data_path, config.word_embedding_pretrained)
        if rank == 0 and not os.path.exists(emb):
            np.savez(emb, embeddings=corpus.embedding_table(config.word_embed_size))
        parallel.barrier()
        titles, absts = corpus.id2title_dict, corpus.id2abst_dict
        train_samples = corpus.train_samples(args.synthetic_users)
        dev_samples, dev_labels = corpus.eval_samples(1024)
    else:
        if args.dataset == 'demo':
            config.word_embedding_pretrained = 'demo_word_embedding.npz'       # run_demo.py:31
        titles = absts = None                          # MyDataset loads news_title.pkl / news_words.csv itself
        demo = args.dataset == 'demo'
        train_samples = load_dataset(config, 'small_train.pkl' if demo else config.train_data, config.data_path, _type=0)
        dev_samples = load_dataset(config, 'small_dev.pkl' if demo else config.dev_data, config.data_path, _type=1)[:100000]
        dev_labels = read_dev_labels(config)

    recommender = Model(config, args)
    if world > 1:
        recommender.model._rank_salt = rank * 0x632BE59BD9B4E019
        recommender.model.engine
        parallel.broadcast_parameters(recommender.model._flat)
        # every rank: the same number of equal-sized batches (the gradient all-reduce is collective and the loss is
        # the mean over world * batch_size users): equal shards, incomplete last batches dropped
        per_rank = len(train_samples) // world
        train_samples = train_samples[rank * per_rank:(rank + 1) * per_rank]
    if rank == 0:
        print(model_name, config.device, sum(p.numel() for p in recommender.parameters()), 'parameters')

    def loader(samples, typ, shuffle):
        if args.feed == 'device':
            return DeviceFeed(config, samples, type=typ, id2title_dict=titles, id2abst_dict=absts, batch_size=config.batch_size,
                              device=config.device, shuffle=shuffle, drop_last=(world > 1 and typ == 0), seed=422 + rank)
        return DataLoader(MyDataset(config, samples, type=typ, id2title_dict=titles, id2abst_dict=absts), batch_size=config.batch_size,
                          num_workers=args.num_workers, drop_last=(world > 1 and typ == 0), shuffle=shuffle,
                          pin_memory=True)

    if not args.test:
        hist = train(config, recommender, loader(train_samples, 0, True), loader(dev_samples, 1, False), dev_labels,
                     max_batches=args.max_batches, verbose=rank == 0)
        if rank == 0:
            print('final dev AUC:', hist['aucs'][-1] if hist['aucs'] else None)
        return hist
    else:
        # run_v0.py:93-111: the test set through the checkpoint named by --load (or the best one by file-name AUC)
        if args.dataset == 'synthetic':
            test_samples, shown = dev_samples, [len(y) for y in dev_labels]
        else:
            test_samples, shown = load_dataset(config, config.test_data, config.data_path, _type=1), None
        out = test(config, recommender, loader(test_samples, 1, False), shown, ckpt_file=args.load,
                   pick_best=args.load is None)
        print('saved to', out)
        return out


if __name__ == '__main__':
    main()
