"""``ModelEmaV2`` with the interface the reference uses from timm (main_stl10.py:14,395,510,522,537,
557-558): ``ModelEmaV2(model, decay)``, ``.module`` (an eval-mode deep copy), ``.update(model)``,
``.state_dict()`` / ``.load_state_dict()`` with ``module.<param>`` keys.

timm is an un-pinned, un-vendored dependency of the reference; the published algorithm
(``ema_v.copy_(decay*ema_v + (1-decay)*model_v)`` over zipped state_dict values) is restated here.
When the model's parameters live in a FlatParams buffer (trainer.py) the update is one fused
kernel together with AdamW; this class then only owns the EMA copy."""
from __future__ import annotations

from copy import deepcopy

import torch
import torch.nn as nn


class ModelEmaV2(nn.Module):
    def __init__(self, model, decay=0.9999, device=None):
        super().__init__()
        for m in model.modules():  # packed bf16 shadows are caches, not state: do not deep-copy them
            m.__dict__.pop("_dm_pw", None)
            m.__dict__.pop("_dm_pack_table", None)
        self.module = deepcopy(model)
        self.module.eval()
        self.decay = decay
        self.device = device
        if self.device is not None:
            self.module.to(device=device)

    def _update(self, model, update_fn):
        with torch.no_grad():
            for ema_v, model_v in zip(self.module.state_dict().values(), model.state_dict().values()):
                if self.device is not None:
                    model_v = model_v.to(device=self.device)
                ema_v.copy_(update_fn(ema_v, model_v))

    def update(self, model):
        self._update(model, update_fn=lambda e, m: self.decay * e + (1.0 - self.decay) * m)

    def set(self, model):
        self._update(model, update_fn=lambda e, m: m)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
