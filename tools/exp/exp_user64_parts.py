"""Where does the fused user-encoder forward spend its time?  Needs a library built with NRMS_HIPCC_EXTRA=-DNRMS_U64_EXPERIMENTS
(python -m pytorch_news_recommender_amd.build --force); runs the forward with parts switched off (results WRONG then)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, time
    from pytorch_news_recommender_amd import synth
    from tests.test_hip_parity import make_model
    shape = synth.Shape(n_words=1000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200, batch_size=512,
                        history_len=50, n_candidates=5, n_words_title=30)
    model = make_model(shape, synth.make_params(shape, seed=3), precision="bf16x3")
    e, flat = model.engine, model._flat
    x = (torch.randn(512, 50, 300) * 0.3).cuda()
    for _ in range(5):
        e.encode_users(flat, x, save=True, tag="dbg")
    e.timing_reset(); e.timing(True)
    for _ in range(20):
        e.encode_users(flat, x, save=True, tag="dbg")
    torch.cuda.synchronize()
    print("NRMS_U64_DBG=%s user64_fwd %.4f ms" % (os.environ.get("NRMS_U64_DBG", "0"), e.timing_read("user64_fwd")[0] / 20))
else:
    for dbg in (0, 1, 2, 4, 8, 16, 3, 7, 15, 31):
        env = dict(os.environ, NRMS_U64_DBG=str(dbg))
        subprocess.run([sys.executable, __file__, "child"], env=env)
