"""Online data feed: the batch-dict contract of the reference's ``MyDataset``
(/root/reference/MIND_2020/data_handler.py:161-250) without its numpy-1 ``np.int`` (removed in
numpy 2) and without pandas/nltk at import time.

A *sample* is the tuple the reference indexes positionally (data_handler.py:206-231):
    sample[0] history news ids (1-based, 0 = pad)      sample[3] impression news ids
    sample[1] history category ids                      sample[4] impression category ids
    sample[2] history sub-category ids                  sample[5] impression sub-category ids
``id2title_dict[news_id - 1]`` is the padded word-id list of a title (data_handler.py:212).
Only the keys the NRMS path reads are always filled; abstract / category keys are emitted (zeros
unless the dictionaries are given) so the collated dict has the reference's full key set.

MIND itself is not available offline, so ``SyntheticMind`` fabricates a corpus + behaviours with
the same structure; ``load_dataset`` reads the reference's pickles when they exist.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset


class MyDataset(Dataset):
    def __init__(self, config, datas, type=0, id2title_dict=None, id2abst_dict=None):
        super().__init__()
        self.config = config
        self.data_type = type
        self.bacthes = datas                       # (sic) attribute name of the reference
        if id2title_dict is None:
            raise ValueError("id2title_dict is required (news index -> padded title word ids)")
        self.id2title_dict = id2title_dict
        self.id2abst_dict = id2abst_dict
        # training: 1 positive + sample_size negatives; evaluation: padded to max_candidate_size
        self.sample_size = config.sample_size + 1 if type < 1 else config.max_candidate_size   # :174-177

    def __len__(self):
        return len(self.bacthes)

    def __getitem__(self, index):
        cfg = self.config
        data = self.bacthes[index]
        H, L, A, S = cfg.history_len, cfg.n_words_title, cfg.n_words_abst, self.sample_size
        i64 = np.int64
        browsed_ids = np.zeros(H, dtype=i64)
        browsed_titles = np.zeros((H, L), dtype=i64)
        browsed_absts = np.zeros((H, A), dtype=i64)
        browsed_categ_ids = np.zeros(H, dtype=i64)
        browsed_subcateg_ids = np.zeros(H, dtype=i64)
        candidate_ids = np.zeros(S, dtype=i64)
        candidate_titles = np.zeros((S, L), dtype=i64)
        candidate_absts = np.zeros((S, A), dtype=i64)
        candidate_categ_ids = np.zeros(S, dtype=i64)
        candidate_subcateg_ids = np.zeros(S, dtype=i64)

        hist = list(data[0])[:H]
        x = len(hist)
        browsed_ids[:x] = hist
        browsed_mask = torch.zeros(H, dtype=torch.uint8)
        browsed_mask[:x] = 1
        if x:
            browsed_titles[:x] = np.asarray([self.id2title_dict[i - 1] for i in hist], dtype=i64)[:, :L]
            if self.id2abst_dict is not None:
                browsed_absts[:x] = np.asarray([self.id2abst_dict[i - 1] for i in hist], dtype=i64)[:, :A]
            if len(data) > 2 and data[1] is not None:
                browsed_categ_ids[:x] = np.asarray(data[1])[:x]
                browsed_subcateg_ids[:x] = np.asarray(data[2])[:x]

        imps = list(data[3])[:S]
        y = len(imps)
        candidate_ids[:y] = imps
        if y:
            candidate_titles[:y] = np.asarray([self.id2title_dict[i - 1] for i in imps], dtype=i64)[:, :L]
            if self.id2abst_dict is not None:
                candidate_absts[:y] = np.asarray([self.id2abst_dict[i - 1] for i in imps], dtype=i64)[:, :A]
            if len(data) > 5 and data[4] is not None:
                ss = min(len(data[4]), S)
                candidate_categ_ids[:ss] = np.asarray(data[4])[:ss]
                candidate_subcateg_ids[:ss] = np.asarray(data[5])[:ss]
        candidate_mask = torch.zeros(S, dtype=torch.uint8)
        candidate_mask[:y] = 1

        return {'browsed_lens': x,
                'browsed_ids': browsed_ids,
                'browsed_titles': browsed_titles,
                'browsed_absts': browsed_absts,
                'browsed_categ_ids': browsed_categ_ids,
                'browsed_subcateg_ids': browsed_subcateg_ids,
                'browsed_mask': browsed_mask,
                'candidate_ids': candidate_ids,
                'candidate_titles': candidate_titles,
                'candidate_absts': candidate_absts,
                'candidate_categ_ids': candidate_categ_ids,
                'candidate_subcateg_ids': candidate_subcateg_ids,
                'candidate_mask': candidate_mask}


def load_dataset(config, file, path, _type=0):
    """Index lists from the reference's preprocessed pickles (data_handler.py:43-110 caches them as
    ``idx_<file>``).  Only the cached, already-indexed form is read here: rebuilding it needs the
    MIND tsv files and the reference's offline ETL (data_processor.py), which is out of scope."""
    cache = os.path.join(path, 'idx_' + file)
    if not os.path.exists(cache):
        raise FileNotFoundError(
            "%s not found: run the reference's data_processor / load_dataset once to build it, or use "
            "--dataset synthetic" % cache)
    with open(cache, 'rb') as f:
        return pickle.load(f)


class SyntheticMind:
    """A MIND-shaped corpus + click behaviours with learnable structure: every news item and
    every user belongs to a latent topic; users click mostly in-topic news, and titles draw
    their words from topic-specific vocabularies -- so AUC rises above 0.5 when training works."""

    def __init__(self, config, n_news=2000, n_topics=8, seed=0, vocab=None):
        rng = np.random.default_rng(seed)
        self.config = config
        self.n_news = n_news
        V = int(vocab if vocab is not None else config.n_words)
        L = config.n_words_title
        self.topic = rng.integers(0, n_topics, size=n_news)
        span = (V - 1) // n_topics
        titles = np.zeros((n_news, L), dtype=np.int64)
        for i in range(n_news):
            n = int(rng.integers(min(5, L), L + 1))
            lo = 1 + self.topic[i] * span
            topical = rng.integers(lo, lo + span, size=n)
            noise = rng.integers(1, V, size=n)
            titles[i, :n] = np.where(rng.random(n) < 0.7, topical, noise)
        self.id2title_dict = {i: titles[i].tolist() for i in range(n_news)}
        self.n_topics = n_topics
        self.rng = rng
        self.by_topic = [np.flatnonzero(self.topic == t) + 1 for t in range(n_topics)]   # 1-based ids

    def embedding_table(self, d, seed=0):
        t = np.random.default_rng(seed).normal(0, 0.4, size=(self.config.n_words, d)).astype(np.float32)
        t[0] = 0
        return t

    def _pick(self, topic, n, p_in=0.8):
        out = []
        for _ in range(n):
            t = topic if self.rng.random() < p_in else int(self.rng.integers(0, self.n_topics))
            out.append(int(self.rng.choice(self.by_topic[t])))
        return out

    def train_samples(self, n_users):
        """[history, None, None, [positive] + negatives, None, None] per user (positive first, as
        the CE-with-label-0 loss of train_eval.py:116-117 expects)."""
        cfg = self.config
        samples = []
        for _ in range(n_users):
            t = int(self.rng.integers(0, self.n_topics))
            hist = self._pick(t, int(self.rng.integers(3, cfg.history_len + 1)))
            pos = self._pick(t, 1, p_in=1.0)
            neg = [int(x) for x in self.rng.integers(1, self.n_news + 1, size=cfg.sample_size)]
            samples.append([hist, None, None, pos + neg, None, None])
        return samples

    def eval_samples(self, n_imps, max_shown=40):
        """Impressions with 0/1 labels (>=1 of each), shown list shorter than max_candidate_size."""
        cfg = self.config
        samples, labels = [], []
        for _ in range(n_imps):
            t = int(self.rng.integers(0, self.n_topics))
            hist = self._pick(t, int(self.rng.integers(3, cfg.history_len + 1)))
            n = int(self.rng.integers(4, min(max_shown, cfg.max_candidate_size) + 1))
            npos = int(self.rng.integers(1, max(2, n // 4)))
            shown = self._pick(t, npos, p_in=1.0) + [int(x) for x in self.rng.integers(1, self.n_news + 1, size=n - npos)]
            y = [1] * npos + [0] * (n - npos)
            perm = self.rng.permutation(n)
            samples.append([hist, None, None, [shown[i] for i in perm], None, None])
            labels.append([y[i] for i in perm])
        return samples, labels
