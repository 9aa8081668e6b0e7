#!/usr/bin/env python3
"""Exactly N identical train steps of the bench configuration (bench.py's model, batch and step; nothing else runs), for
profiler passes whose per-kernel totals are then divided by N.   python3 tools/train_steps.py --steps 4 [--precision fp16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_news_recommender_amd import synth
from pytorch_news_recommender_amd.config import Config
from pytorch_news_recommender_amd.model.nrms_hip import Model

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--users", type=int, default=512)
ap.add_argument("--precision", default="fp16")
ap.add_argument("--fp16-user-encoder", action="store_true")
ap.add_argument("--model", default="v0", choices=["v0", "v1"], help="v1: bench.py's nrms_v1 leg (20-word titles, 6 title heads, W_O)")
args = ap.parse_args()
if args.model == "v1":
    from pytorch_news_recommender_amd.model.nrms_v1_hip import Model
    shape = synth.Shape(n_words=synth.BENCH.n_words, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=args.users, history_len=50, n_candidates=5, n_words_title=20)
    cfg = Config("nrms_v1")
    cfg.__nrms__()
    cfg.num_attention_heads, cfg.title_heads_num = 10, 6
    cfg.dropout, cfg.learning_rate, cfg.precision = 0.2, 1e-3, args.precision
    cfg.fp16_v1_news_encoder = True        # the opt-in fused fp16 news encoder of nrms_v1 (csrc/fused16_v1*.hip): the kernels this profiles
    params = synth.make_params_v1(shape, seed=0)
    table = params["news_encoder.word_embedding.weight"]
else:
    shape = synth.BENCH
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.n_words_title, cfg.sample_size, cfg.batch_size = shape.n_words_title, shape.n_candidates - 1, args.users
    cfg.dropout, cfg.learning_rate, cfg.precision, cfg.fp16_user_encoder = 0.2, 1e-3, args.precision, args.fp16_user_encoder
    params = synth.make_params(shape, seed=0)
    table = params["news_encoder.word_embedding.0.weight"]
model = Model(cfg, pretrained_word_embedding=table)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to("cuda:0").train()
batch = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.make_batch(shape, seed=1, batch_size=args.users).items()}
for _ in range(args.steps):
    model.train_step(batch)
torch.cuda.synchronize()
