"""Demo entry (/root/reference/MIND_2020/run_demo.py:20-61: the MIND-small demo subset, a fixed small batch,
NRMS_V0).  The reference script cannot even import (it pulls NRMS_V0 from a package that does not export it);
this one runs the same plumbing on the HIP path: ``--dataset demo`` reads the reference's demo files
(small_train.pkl / small_dev.pkl / demo_news_words.csv / small_dev_behaviors.csv) when they exist, the default
fabricates a MIND-small-shaped corpus.  Batch 32 is BASELINE.json's config 0 (the reference hard-codes 256,
run_demo.py:28)."""
import sys

from .run_v0 import main


def demo_argv(extra=()):
    argv = ['--model', 'nrms_hip', '--dataset', 'synthetic', '--epochs', '2', '--synthetic_users', '4096',
            '--batch_size', '32', '--description', 'DEMO']
    return argv + list(extra)


if __name__ == '__main__':
    main(demo_argv(sys.argv[1:]))
