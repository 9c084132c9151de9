"""MI355X-native reverse-diffusion handwriting sampler: a drop-in for the reference's
``DiffusionModel.forward`` and the ``infer`` sampling loop.  See DESIGN.md / INTEGRATION.md."""
from . import spec  # noqa: F401
from .inference import get_alpha_set, get_beta_set, infer, sample  # noqa: F401
from .model import DiffusionModel, DiffusionWriter  # noqa: F401
from .tokenizer import Tokenizer, stroke_length  # noqa: F401
