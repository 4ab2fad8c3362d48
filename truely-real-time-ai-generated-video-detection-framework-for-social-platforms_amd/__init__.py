"""truely_amd -- MI355X-native implementation of Truely's per-frame visual hot path
(server/model.py::run of the reference): MTCNN detect -> largest-face crop -> FaceNet
(InceptionResnetV1) embedding -> consecutive-frame cosine drift -> 0..100 score.

Importable as ``truely_amd`` (see truely_amd.py at the repository root; the directory name
required by the build contract is not a valid Python identifier)."""
from . import weights, synthetic  # noqa: F401  (pure numpy, usable without a GPU)

__all__ = ["weights", "synthetic", "Engine", "MTCNN", "InceptionResnetV1", "run", "analyze_video"]


def __getattr__(name):
    # GPU-facing pieces are imported lazily so `import truely_amd` works on a CPU-only box
    if name == "Engine":
        from .engine import Engine
        return Engine
    if name == "MTCNN":
        from .mtcnn import MTCNN
        return MTCNN
    if name == "InceptionResnetV1":
        from .inception_resnet_v1 import InceptionResnetV1
        return InceptionResnetV1
    if name in ("run", "analyze_video"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
