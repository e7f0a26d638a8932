"""ranking_amd: the TF-Ranking loss-and-score hot path, rebuilt MI355X-first.

Python surface mirrors ``tensorflow_ranking`` (``losses``, ``metrics``,
``losses_impl``, ``metrics_impl``, ``utils``, ``keras.{losses,metrics,layers,
model,utils}``); the per-list work runs in hand-written gfx950 kernels behind
the C ABI in ``include/tfr_hip.h`` (``ranking_amd/csrc``).  There is no CPU
fallback: tensors must live on a HIP device and ``libtfr_hip.so`` must load.
"""
from . import _lib            # noqa: F401
from . import utils           # noqa: F401
from . import losses_impl     # noqa: F401
from . import metrics_impl    # noqa: F401
from . import losses          # noqa: F401
from . import metrics         # noqa: F401
from . import keras           # noqa: F401

__version__ = '0.1.0'
