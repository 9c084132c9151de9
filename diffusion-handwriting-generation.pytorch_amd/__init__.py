"""MI355X-native reverse-diffusion handwriting sampler (drop-in for the reference's
``DiffusionModel.forward`` / ``infer`` sampling loop).  See DESIGN.md."""
from . import spec  # noqa: F401
