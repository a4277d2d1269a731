"""galerkin_transformer -- MI355X-native drop-in for the hot path of scaomath/galerkin-transformer.

Same import surface as the reference package (``from galerkin_transformer.model import *`` etc.);
the encoder layer and the spectral decoder execute in ``_lib/libgt_hip.so`` (hand-written HIP for
gfx950, C ABI in include/gt_hip.h).  There is no CPU fallback for those operators.
"""
from .layers import *          # noqa: F401,F403
from .layers import (FeedForward, Identity, SimpleAttention, SpectralConv1d, SpectralConv2d,  # noqa: F401
                     get_attention_dropout, push_attention_masks, set_attention_dropout)
from .model import *           # noqa: F401,F403
from .model import (FourierTransformer, FourierTransformer2D, FourierTransformer2DLite,  # noqa: F401
                    FourierTransformerEncoderLayer, PointwiseRegressor, SimpleTransformer,
                    SimpleTransformerEncoderLayer, SpectralRegressor)

from .ft import (BurgersDataset, DarcyDataset, UnitGaussianNormalizer, WeightedL2Loss,  # noqa: F401
                 WeightedL2Loss2d)
from .optim import FlatClipAdam  # noqa: F401
from ._hip import get_precision, set_precision  # noqa: F401
from .utils import get_num_params, get_seed  # noqa: F401

__version__ = "0.1.0"
