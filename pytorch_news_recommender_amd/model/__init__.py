"""Model dispatch with the reference's wrapper contract
(/root/reference/MIND_2020/model/__init__.py:13-38): ``Model(config, args)`` imports
``model.<args.model>`` and forwards ``batch -> scores``.  ``nrms_v0`` is served by the HIP
implementation (same parameter names, same outputs)."""
from importlib import import_module

import torch
import torch.nn as nn

ALIASES = {"nrms_v0": "nrms_hip", "nrms": "nrms_hip", "nrms_v1": "nrms_v1_hip", "nrms_naml": "nrms_naml_hip", "hierec": "hierec_hip", "gnn": "graph_hip", "graph": "graph_hip"}


class Model(nn.Module):
    def __init__(self, config, args):
        super(Model, self).__init__()
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.n_GPUs = getattr(args, "n_GPUs", 1)
        name = args.model.lower()
        module = import_module(__name__ + '.' + ALIASES.get(name, name))
        self.model = module.Model(config).to(self.device)

    def forward(self, batch):
        # the reference's data_parallel branch is unreachable (model/__init__.py:35: `not self.cpu`
        # tests a bound method); multi-GPU here is one process per GPU (see parallel.py)
        return self.model(batch)

    def train_step(self, batch, **kw):
        return self.model.train_step(batch, **kw)
