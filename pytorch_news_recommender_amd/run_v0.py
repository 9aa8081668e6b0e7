"""CLI entry with the reference's flags (/root/reference/MIND_2020/run_v0.py:15-28):

    python -m pytorch_news_recommender_amd.run_v0 --model nrms_hip --dataset synthetic
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m pytorch_news_recommender_amd.run_v0 ...

``--dataset large|demo`` expects the reference's preprocessed files under config.data_path;
``--dataset synthetic`` fabricates a MIND-shaped corpus (no data ships offline).
"""
import argparse
import os
import pickle
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import parallel
from .config import Config
from .data_handler import MyDataset, SyntheticMind, load_dataset
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
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, local_rank, world = parallel.init_process_group()
    torch.manual_seed(422)
    torch.cuda.manual_seed_all(422)
    model_name = args.model + '_' + (args.description or time.strftime('%m-%d_%H'))
    config = Config(model_name)
    config.batch_size = 512
    config.num_epochs = 6 if args.epochs is None else args.epochs
    config.mode = args.dataset
    config.__nrms__()

    if args.dataset == 'synthetic':
        config.n_words_title = 30
        corpus = SyntheticMind(config, n_news=4000, seed=0)
        os.makedirs(config.data_path, exist_ok=True)
        emb = os.path.join(config.data_path, config.word_embedding_pretrained)
        if rank == 0 and not os.path.exists(emb):
            np.savez(emb, embeddings=corpus.embedding_table(config.word_embed_size))
        parallel.barrier()
        titles = corpus.id2title_dict
        train_samples = corpus.train_samples(args.synthetic_users)
        dev_samples, dev_labels = corpus.eval_samples(1024)
    else:
        with open(os.path.join(config.data_path, 'id2title_dict.pkl'), 'rb') as f:
            titles = pickle.load(f)
        train_samples = load_dataset(config, config.train_data, config.data_path, _type=0)
        dev_samples = load_dataset(config, config.dev_data, config.data_path, _type=1)[:100000]
        with open(os.path.join(config.data_path, 'dev_labels.pkl'), 'rb') as f:
            dev_labels = pickle.load(f)

    recommender = Model(config, args)
    if world > 1:
        recommender.model._rank_salt = rank * 0x632BE59BD9B4E019
        recommender.model.engine
        parallel.broadcast_parameters(recommender.model._flat)
        lo, hi = parallel.shard_rows(len(train_samples), rank, world)
        train_samples = train_samples[lo:hi]
    if rank == 0:
        print(model_name, config.device, sum(p.numel() for p in recommender.parameters()), 'parameters')

    def loader(samples, typ, shuffle):
        return DataLoader(MyDataset(config, samples, type=typ, id2title_dict=titles), batch_size=config.batch_size,
                          num_workers=6, drop_last=False, shuffle=shuffle, pin_memory=True)

    if not args.test:
        hist = train(config, recommender, loader(train_samples, 0, True), loader(dev_samples, 1, False), dev_labels,
                     max_batches=args.max_batches, verbose=rank == 0)
        if rank == 0:
            print('final dev AUC:', hist['aucs'][-1] if hist['aucs'] else None)
    else:
        recommender.load_state_dict(torch.load(os.path.join(config.save_path, args.load), weights_only=True))
        out = test(config, recommender, loader(dev_samples, 1, False), [len(y) for y in dev_labels])
        print('saved to', out)


if __name__ == '__main__':
    main()
