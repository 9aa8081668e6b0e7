"""Stage timing of the evaluation path (SURVEY f-1) at the reference's dev batch (512 impressions x (50 + 300) slots).
Usage: python tools/bench_eval.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.data import DataLoader

from pytorch_news_recommender_amd import synth
from pytorch_news_recommender_amd.config import Config
from pytorch_news_recommender_amd.data_handler import MyDataset, SyntheticMind
from pytorch_news_recommender_amd.model.nrms_hip import Model


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda", 0)
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.n_words_title = synth.BENCH.n_words_title
    cfg.precision = "fp16"
    cfg.word_embed_size = synth.BENCH.word_embed_size
    cfg.max_candidate_size, cfg.history_len = 300, 50
    params = synth.make_params(synth.BENCH, seed=0)
    model = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.0.weight"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev).eval()
    corpus = SyntheticMind(cfg, n_news=4000, seed=3)
    samples, _ = corpus.eval_samples(4 * bs, max_shown=70)
    ds = MyDataset(cfg, samples, type=1, id2title_dict=corpus.id2title_dict)
    batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
               for b in DataLoader(ds, batch_size=bs, shuffle=False, num_workers=0)]
    eng = model.engine

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    with torch.no_grad():
        for name, dd, strip in (("all_slots", False, True), ("hash", True, True), ("cached", True, False)):
            model.dedup_inference = dd
            bb = [{k: v for k, v in b.items() if not strip or k not in ("browsed_ids", "candidate_ids")} for b in batches]

            def run():
                if name == "cached":
                    eng.news_cache_begin()
                for b in bb:
                    model(b)
                if name == "cached":
                    eng.news_cache_end()
            t = timed(run)
            print("%-10s %8.3f ms / batch of %d  -> %9.0f impressions/s" % (name, t / len(bb) * 1e3, bs, bs * len(bb) / t))
        # stages of the hash path on the first batch
        b = batches[0]
        flat = model._flat
        hist_ids, cand_ids, mask = b["browsed_titles"], b["candidate_titles"], b["candidate_mask"]
        B, H, L = hist_ids.shape
        Cn = cand_ids.shape[1]
        N = B * (H + Cn)
        ids = torch.cat([hist_ids.reshape(B * H, L), cand_ids.reshape(B * Cn, L)]).contiguous()
        st = {}
        st["unique_titles"] = timed(lambda: eng.unique_titles(ids))
        uniq, inverse = eng.unique_titles(ids)
        st["encode_unique(%d)" % uniq.shape[0]] = timed(lambda: eng.encode_titles(flat, uniq, tag="news_eval", trusted_ids=True))
        vec = eng.encode_titles(flat, uniq, tag="news_eval", trusted_ids=True)
        st["index_select"] = timed(lambda: vec.index_select(0, inverse))
        nv = vec.index_select(0, inverse)
        d = nv.shape[1]
        hist = nv[:B * H].view(B, H, d)
        cand = nv[B * H:].view(B, Cn, d)
        st["encode_users"] = timed(lambda: eng.encode_users(flat, hist, tag="user_eval"))
        user = eng.encode_users(flat, hist, tag="user_eval")
        m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
        st["click_scores"] = timed(lambda: eng.click_scores(cand, user, m8))
        st["encode_all(%d)" % N] = timed(lambda: eng.encode_titles(flat, ids, tag="news_eval", trusted_ids=True))
        for k, v in st.items():
            print("  %-24s %8.3f ms" % (k, v * 1e3))


if __name__ == "__main__":
    main()
