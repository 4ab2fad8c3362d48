"""Drop-in for ``facenet_pytorch.InceptionResnetV1`` as the reference uses it:

    facenet_model = InceptionResnetV1(pretrained="vggface2").eval()       # server/model.py:19
    emb = facenet_model(face_tensor).detach().numpy().flatten()          # server/model.py:59

``__call__`` takes the ``(n, 3, H, W)`` float tensor that ``to_tensor`` produced and returns an
``(n, 512)`` L2-normalised tensor on the input's device.  No checkpoint can be downloaded in the
build environment, so ``pretrained`` only selects which packed weight blob the engine holds
(TRUELY_WEIGHTS, else seeded synthetic weights)."""
from __future__ import annotations

import torch

from .engine import Engine, default_engine


class InceptionResnetV1:
    def __init__(self, pretrained=None, classify=False, num_classes=None, dropout_prob=0.6, device=None,
                 engine: Engine | None = None):
        if classify:
            raise NotImplementedError("classify=True (logits head) is not on the reference's hot path")
        self.pretrained = pretrained
        self.engine = engine or default_engine()

    def eval(self):
        return self

    def to(self, device):
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward(x)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected (n, 3, H, W)")
        dev = x.device
        nhwc = x.to(self.engine.device, torch.float32).permute(0, 2, 3, 1).contiguous()
        emb = self.engine.facenet_embed(nhwc)
        return emb if dev.type == "cuda" else emb.to(dev)
