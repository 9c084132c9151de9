"""MI355X-native reverse-diffusion handwriting sampler: a drop-in for the reference's
``DiffusionModel.forward`` and the ``infer`` sampling loop.  See DESIGN.md / INTEGRATION.md."""
from . import spec  # noqa: F401
from .checkpoint import find_checkpoint, load_model, read_config, read_state_dict  # noqa: F401
from .inference import get_alpha_set, get_beta_set, infer, infer_file, load_style, read_img, remove_whitespace, sample  # noqa: F401
from .vis import show_strokes, strokes_to_polylines  # noqa: F401
from .model import DiffusionModel, DiffusionWriter  # noqa: F401
from .style_extractor import StyleExtractor  # noqa: F401
from .tokenizer import Tokenizer, stroke_length  # noqa: F401
from . import train, train_model  # noqa: F401
