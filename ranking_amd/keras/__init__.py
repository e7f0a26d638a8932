"""Mirror of ``tfr.keras`` for the hot path: losses, metrics, utils, layers, model."""
from . import utils      # noqa: F401
from . import losses     # noqa: F401
from . import metrics    # noqa: F401
from . import layers     # noqa: F401
from . import model      # noqa: F401
from . import pipeline  # noqa: F401,E402
