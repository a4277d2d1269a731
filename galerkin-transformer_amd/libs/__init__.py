"""``from libs import *`` as the reference's example scripts do (reference libs/__init__.py:1-5)."""
from galerkin_transformer.layers import *      # noqa: F401,F403
from galerkin_transformer.utils import *       # noqa: F401,F403
from galerkin_transformer.utils_ft import *    # noqa: F401,F403
from galerkin_transformer.ft import *          # noqa: F401,F403
from galerkin_transformer.model import *       # noqa: F401,F403
