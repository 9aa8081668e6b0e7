"""Demo entry (/root/reference/MIND_2020/run_demo.py:20-61: MIND-small, batch 32, a few epochs).
The reference script cannot even import (it pulls NRMS_V0 from a package that does not export it);
this one runs the same plumbing on the HIP path with a synthetic MIND-small-shaped corpus."""
from .run_v0 import main

if __name__ == '__main__':
    main(['--model', 'nrms_hip', '--dataset', 'synthetic', '--epochs', '2', '--synthetic_users', '4096'])
