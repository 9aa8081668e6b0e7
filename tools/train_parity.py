"""Training-trajectory parity of the precision modes: the same model (bench dimensions), the same synthetic corpus, batches
and dropout-free steps in fp32 and in fp16 / bf16x3; prints the per-step loss gap and the dev AUC after training.
GPU box only.  Usage: python tools/train_parity.py [steps] [dropout] [v0|v1]   (v1: nrms_v1 -- 20-word titles, six title heads of 50, W_O)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pytorch_news_recommender_amd import train_eval
from pytorch_news_recommender_amd.config import Config
from pytorch_news_recommender_amd.data_handler import DeviceFeed, SyntheticMind

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
dropout = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
variant = sys.argv[3] if len(sys.argv) > 3 else "v0"
if variant == "v1":
    from pytorch_news_recommender_amd.model.nrms_v1_hip import Model
else:
    from pytorch_news_recommender_amd.model.nrms_hip import Model


def run(prec):
    cfg = Config("nrms_v1" if variant == "v1" else "nrms_hip")
    cfg.__nrms__()
    cfg.n_words_title, cfg.batch_size, cfg.dropout, cfg.precision, cfg.learning_rate = (20 if variant == "v1" else 30), 256, dropout, prec, 1e-3
    if variant == "v1":
        cfg.num_attention_heads, cfg.title_heads_num = 10, 6
    cfg.max_candidate_size = 40
    corpus = SyntheticMind(cfg, n_news=4000, seed=0)
    table = torch.from_numpy(np.asarray(corpus.embedding_table(cfg.word_embed_size), dtype=np.float32))
    torch.manual_seed(0)
    model = Model(cfg, pretrained_word_embedding=table).cuda().train()
    feed = DeviceFeed(cfg, corpus.train_samples(256 * 40), type=0, id2title_dict=corpus.id2title_dict,
                      id2abst_dict=corpus.id2abst_dict, batch_size=256, device="cuda", shuffle=True, drop_last=True, seed=3)
    dev_samples, dev_labels = corpus.eval_samples(1024, max_shown=30)
    dev = DeviceFeed(cfg, dev_samples, type=1, id2title_dict=corpus.id2title_dict, id2abst_dict=corpus.id2abst_dict,
                     batch_size=256, device="cuda")
    losses = []
    while len(losses) < steps:
        for b in feed:
            losses.append(model.train_step(b) / 256)
            if len(losses) >= steps:
                break
    losses = [float(v) for v in losses]
    auc = train_eval.evaluate(cfg, model, dev, dev_labels, verbose=False)
    same = {}
    if prec == "fp32":                              # the SAME trained weights scored in the other modes
        for other in ("fp16", "bf16x3"):
            model.config.precision = other
            same[other] = float(train_eval.evaluate(cfg, model, dev, dev_labels, verbose=False))
        model.config.precision = prec
    return np.array(losses), float(auc), same


ref_l, ref_auc, same = run("fp32")
print("fp32   : loss %.5f -> %.5f, dev AUC %.5f" % (ref_l[:8].mean(), ref_l[-8:].mean(), ref_auc))
print("         the same trained weights scored in fp16: AUC %.6f (gap %.1e), in bf16x3: %.6f (gap %.1e)" % (
    same["fp16"], abs(same["fp16"] - ref_auc), same["bf16x3"], abs(same["bf16x3"] - ref_auc)))
for prec in ("fp16", "bf16x3"):
    l, auc, _ = run(prec)
    d = np.abs(l - ref_l)
    print("%-7s: loss %.5f -> %.5f, dev AUC %.5f | vs fp32: |loss gap| first 20 steps max %.2e, all steps mean %.2e max %.2e; "
          "|AUC gap| %.2e" % (prec, l[:8].mean(), l[-8:].mean(), auc, d[:20].max(), d.mean(), d.max(), abs(auc - ref_auc)))
