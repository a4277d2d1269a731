"""``from libs.ns_lite import *`` as examples/ex4_navier_stokes_2+1d.py does (reference libs/ns_lite.py)."""
from galerkin_transformer.ns_lite import *      # noqa: F401,F403
from galerkin_transformer.ns_lite import NavierStokesDatasetLite, train_batch_ns, validate_epoch_ns  # noqa: F401
