"""Throughput of the end-to-end training loop (device feed -> Model.train_step), without evaluations: the synthetic corpus of
run_v0.py, 20 480 users in batches of 512, fp16 mode.  Compares the lazy batch dict with an eagerly materialised one.
GPU box only.  Usage: python tools/bench_loop.py [epochs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pytorch_news_recommender_amd.config import Config
from pytorch_news_recommender_amd.data_handler import DeviceFeed, SyntheticMind
from pytorch_news_recommender_amd.model.nrms_hip import Model

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = Config("nrms_hip")
cfg.__nrms__()
cfg.n_words_title, cfg.batch_size, cfg.dropout, cfg.precision = 30, 512, 0.2, "fp16"
corpus = SyntheticMind(cfg, n_news=4000, seed=0)
table = torch.from_numpy(np.asarray(corpus.embedding_table(cfg.word_embed_size), dtype=np.float32))
model = Model(cfg, pretrained_word_embedding=table).cuda().train()
samples = corpus.train_samples(20480)
feed = DeviceFeed(cfg, samples, type=0, id2title_dict=corpus.id2title_dict, id2abst_dict=corpus.id2abst_dict,
                  batch_size=512, device="cuda", shuffle=True, drop_last=True)
for mode in ("lazy", "eager", "lazy"):
    for b in feed:                                   # warm-up epoch
        model.train_step(b if mode == "lazy" else dict(b.items()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(epochs):
        for b in feed:
            model.train_step(b if mode == "lazy" else dict(b.items()))
            n += len(b["browsed_titles"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-5s batch dict: %d users in %.3f s -> %.0f users/s (%.3f ms per 512-user step)" % (mode, n, dt, n / dt, dt / (n / 512) * 1e3))
