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
the same structure; ``load_dataset`` / ``get_Words_Infos`` / ``read_dev_labels`` / ``get_Test_List`` read the
files the reference's own preprocessing writes under ``config.data_path`` when they exist (the user's own
data: nothing of the kind ships with the reference).
"""
from __future__ import annotations

import csv
import os
import pickle
from ast import literal_eval
from collections.abc import Mapping

import numpy as np
import torch
from torch.utils.data import Dataset


def _words_infos(config, title_pkl, abst_pkl, words_csv):
    """news index -> padded word-id list, for titles and abstracts: the pickled dicts when they exist, otherwise
    built from the headerless ``news_id,title,abstract`` csv (list literals) and cached as pickles, exactly as
    the reference does (data_handler.py:113-135)."""
    base = config.data_path
    tp, ap = os.path.join(base, title_pkl), os.path.join(base, abst_pkl)
    if os.path.exists(tp):
        with open(tp, 'rb') as f:
            title_dict = pickle.load(f)
        abst_dict = None
        if os.path.exists(ap):
            with open(ap, 'rb') as f:
                abst_dict = pickle.load(f)
        return title_dict, abst_dict
    src = os.path.join(base, words_csv)
    if not os.path.exists(src):
        raise FileNotFoundError("neither %s nor %s exists: run the reference's data_processor first, or pass "
                                "id2title_dict / use --dataset synthetic" % (tp, src))
    title_dict, abst_dict = {}, {}
    csv.field_size_limit(1 << 30)
    with open(src, newline='') as f:
        for i, row in enumerate(csv.reader(f)):          # columns: news_id, title, abstract (no header)
            title_dict[i] = literal_eval(row[1])
            abst_dict[i] = literal_eval(row[2])
    with open(tp, 'wb') as f:
        pickle.dump(title_dict, f)
    with open(ap, 'wb') as f:
        pickle.dump(abst_dict, f)
    return title_dict, abst_dict


def get_Words_Infos(config):
    """data_handler.py:113-135."""
    return _words_infos(config, 'news_title.pkl', 'news_abst.pkl', 'news_words.csv')


def get_Demo_Words_Infos(config):
    """data_handler.py:137-159."""
    return _words_infos(config, 'demo_news_title.pkl', 'demo_news_abst.pkl', 'demo_news_words.csv')


def read_dev_labels(config, file=None):
    """Per-impression 0/1 label lists from the ``y_true`` column (space-separated) of dev_behaviors.csv
    (train_eval.py:36-39; demo mode: small_dev_behaviors.csv, train_eval.py:156-158)."""
    if file is None:
        file = 'small_dev_behaviors.csv' if getattr(config, 'mode', 'large') == 'demo' else 'dev_behaviors.csv'
    path = os.path.join(config.data_path, file)
    csv.field_size_limit(1 << 30)
    with open(path, newline='') as f:
        return [[int(v) for v in row['y_true'].split(' ')] for row in csv.DictReader(f)]


def get_Test_List(config):
    """Number of shown candidates per test impression (train_eval.py:287-298): the cached
    test_imps_list.pkl, or counted from the 4th (impressions) column of test/behaviors.tsv and cached."""
    cache = os.path.join(config.data_path, 'test_imps_list.pkl')
    if os.path.exists(cache):
        with open(cache, 'rb') as f:
            return pickle.load(f)
    lens = []
    with open(os.path.join(config.test_path, 'behaviors.tsv')) as f:
        for line in f:
            cols = line.rstrip('\n').split('\t')
            lens.append(len(cols[-1].split(' ')))
    with open(cache, 'wb') as f:
        pickle.dump(lens, f)
    return lens


class MyDataset(Dataset):
    def __init__(self, config, datas, type=0, id2title_dict=None, id2abst_dict=None):
        super().__init__()
        self.config = config
        self.data_type = type
        self.bacthes = datas                       # (sic) attribute name of the reference
        if id2title_dict is None:
            # the reference's constructor (data_handler.py:162-170): the word dictionaries come from config.data_path
            if getattr(config, 'mode', 'large') == 'demo':
                id2title_dict, id2abst_dict = get_Demo_Words_Infos(config)
            else:
                id2title_dict, id2abst_dict = get_Words_Infos(config)
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
        # abstracts (same topical vocabulary, config.n_words_abst words) and category / sub-category ids for nrms_naml:
        # category = 1 + topic (0 is the padding slot), sub-category = a fixed refinement of it
        # (their own generator: the title / behaviour streams stay what they were before the abstracts existed)
        A = int(getattr(config, "n_words_abst", 40))
        absts = np.zeros((n_news, A), dtype=np.int64)
        rng2 = np.random.default_rng(seed + 1000003)
        for i in range(n_news):
            n = int(rng2.integers(min(8, A), A + 1))
            lo = 1 + self.topic[i] * span
            absts[i, :n] = np.where(rng2.random(n) < 0.6, rng2.integers(lo, lo + span, size=n), rng2.integers(1, V, size=n))
        self.id2abst_dict = {i: absts[i].tolist() for i in range(n_news)}
        n_cat, n_sub = int(getattr(config, "category_nums", 19)), int(getattr(config, "subcategory_nums", 294))
        self.category = 1 + self.topic % max(n_cat - 1, 1)
        self.subcategory = 1 + (self.topic * 7 + rng2.integers(0, 5, size=n_news)) % max(n_sub - 1, 1)
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
        """[history, its categories, its sub-categories, [positive] + negatives, their categories, their sub-categories] per
        user (data_handler.py:40; positive first, as the CE-with-label-0 loss of train_eval.py:116-117 expects)."""
        cfg = self.config
        samples = []
        for _ in range(n_users):
            t = int(self.rng.integers(0, self.n_topics))
            hist = self._pick(t, int(self.rng.integers(3, cfg.history_len + 1)))
            pos = self._pick(t, 1, p_in=1.0)
            neg = [int(x) for x in self.rng.integers(1, self.n_news + 1, size=cfg.sample_size)]
            imps = pos + neg
            samples.append([hist, self._cat(hist), self._sub(hist), imps, self._cat(imps), self._sub(imps)])
        return samples

    def _cat(self, news_ids):
        return [int(self.category[i - 1]) for i in news_ids]

    def _sub(self, news_ids):
        return [int(self.subcategory[i - 1]) for i in news_ids]

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
            imps = [shown[i] for i in perm]
            samples.append([hist, self._cat(hist), self._sub(hist), imps, self._cat(imps), self._sub(imps)])
            labels.append([y[i] for i in perm])
        return samples, labels


class LazyBatch(Mapping):
    """A read-only batch dict whose values are produced on first access and kept (keys, order and ``len`` are those of the
    eager dict; ``items()`` / ``values()`` materialise everything)."""

    def __init__(self, makers):
        self._makers, self._vals = makers, {}

    def __getitem__(self, key):
        if key not in self._vals:
            self._vals[key] = self._makers[key]()
        return self._vals[key]

    def __iter__(self):
        return iter(self._makers)

    def __len__(self):
        return len(self._makers)


class DeviceFeed:
    """The same batch dicts as ``DataLoader(MyDataset(config, samples, type), batch_size)`` -- same 13 keys, dtypes, padding
    and sample order -- assembled ON THE DEVICE: the news corpus (title and abstract word ids, 31 MB for MIND's 130 k news at
    30 words) lives in HBM once, the samples are packed into padded id arrays once, and a batch is two row gathers.
    ``MyDataset.__getitem__`` builds every sample in Python (55 dictionary lookups each, data_handler.py:185-250): eight
    loader workers deliver a few thousand users per second, the train step consumes a hundred and fifty thousand.

    Iterable like a DataLoader (``len()`` = batches per epoch); ``shuffle`` draws a fresh permutation per epoch from ``seed``.
    """

    def __init__(self, config, samples, type=0, id2title_dict=None, id2abst_dict=None, batch_size=None, device="cuda",
                 shuffle=False, drop_last=False, seed=0):
        self.config, self.data_type = config, type
        if id2title_dict is None:
            if getattr(config, 'mode', 'large') == 'demo':
                id2title_dict, id2abst_dict = get_Demo_Words_Infos(config)
            else:
                id2title_dict, id2abst_dict = get_Words_Infos(config)
        self.device = torch.device(device)
        self.batch_size = int(batch_size or config.batch_size)
        self.shuffle, self.drop_last, self.seed, self.epoch = bool(shuffle), bool(drop_last), int(seed), 0
        H, L, A = config.history_len, config.n_words_title, config.n_words_abst
        S = config.sample_size + 1 if type < 1 else config.max_candidate_size
        self.S = S

        def table(d, width):
            n = (max(d) + 1) if len(d) else 0                       # news index i lives in row i + 1 (0 = padding slot)
            t = np.zeros((n + 1, width), dtype=np.int64)
            for i, words in d.items():
                w = list(words)[:width]
                t[i + 1, :len(w)] = w
            return torch.from_numpy(t).to(self.device)

        self.titles = table(id2title_dict, L)
        self.absts = table(id2abst_dict, A) if id2abst_dict is not None else None
        n = len(samples)
        hist = np.zeros((n, H), dtype=np.int64)
        hcat, hsub = np.zeros((n, H), dtype=np.int64), np.zeros((n, H), dtype=np.int64)
        cand = np.zeros((n, S), dtype=np.int64)
        ccat, csub = np.zeros((n, S), dtype=np.int64), np.zeros((n, S), dtype=np.int64)
        hlen, clen = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        for k, data in enumerate(samples):
            h = list(data[0])[:H]
            x = len(h)
            hist[k, :x] = h
            hlen[k] = x
            if x and len(data) > 2 and data[1] is not None:
                hcat[k, :x] = np.asarray(data[1])[:x]
                hsub[k, :x] = np.asarray(data[2])[:x]
            c = list(data[3])[:S]
            y = len(c)
            cand[k, :y] = c
            clen[k] = y
            if y and len(data) > 5 and data[4] is not None:
                ss = min(len(data[4]), S)
                ccat[k, :ss] = np.asarray(data[4])[:ss]
                csub[k, :ss] = np.asarray(data[5])[:ss]
        dev = lambda a: torch.from_numpy(a).to(self.device)
        self.packed = dict(hist=dev(hist), hcat=dev(hcat), hsub=dev(hsub), cand=dev(cand), ccat=dev(ccat), csub=dev(csub),
                           hlen=dev(hlen), clen=dev(clen))
        self.n = n

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def batch(self, rows):
        """rows: int64 device tensor of sample indices -> the batch dict (device tensors).  The dict is LAZY: a value is
        gathered when it is first read (nrms_v0 reads 3 of the 13 keys, nrms_naml 9; gathering all of them cost 0.6 ms of GPU
        time per 512-user batch against a 3.3 ms train step)."""
        p, cfg = self.packed, self.config
        H, S, A = cfg.history_len, self.S, cfg.n_words_abst
        B = rows.shape[0]
        memo = {}

        def sel(name):                                                # rows of one packed array, gathered once
            if name not in memo:
                memo[name] = p[name].index_select(0, rows)
            return memo[name]

        def text(table, slots, width):
            if table is None:
                return torch.zeros(B, sel(slots).shape[1], width, dtype=torch.int64, device=self.device)
            return table.index_select(0, sel(slots).reshape(-1)).view(B, sel(slots).shape[1], -1)

        def mask(n_slots, lens):
            return (torch.arange(n_slots, device=self.device)[None, :] < sel(lens)[:, None]).to(torch.uint8)

        return LazyBatch({'browsed_lens': lambda: sel("hlen"),
                          'browsed_ids': lambda: sel("hist"),
                          'browsed_titles': lambda: text(self.titles, "hist", cfg.n_words_title),
                          'browsed_absts': lambda: text(self.absts, "hist", A),
                          'browsed_categ_ids': lambda: sel("hcat"),
                          'browsed_subcateg_ids': lambda: sel("hsub"),
                          'browsed_mask': lambda: mask(H, "hlen"),
                          'candidate_ids': lambda: sel("cand"),
                          'candidate_titles': lambda: text(self.titles, "cand", cfg.n_words_title),
                          'candidate_absts': lambda: text(self.absts, "cand", A),
                          'candidate_categ_ids': lambda: sel("ccat"),
                          'candidate_subcateg_ids': lambda: sel("csub"),
                          'candidate_mask': lambda: mask(S, "clen")})

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).to(self.device)
            self.epoch += 1
        else:
            order = torch.arange(self.n, device=self.device)
        for b in range(len(self)):
            yield self.batch(order[b * self.batch_size:(b + 1) * self.batch_size])
