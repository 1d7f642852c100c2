"""MI355X-native (gfx950) forward/backward path for the Daft-Exprt acoustic model.

Public surface mirrors the reference (src/daft_exprt/model.py, loss.py): ``DaftExprt``, ``DaftExprtLoss``,
``HyperParams``.  The compute is in ``libdaft_exprt_hip.so`` (hand-written HIP, C ABI in include/daft_exprt_hip.h);
importing the model classes does not need a GPU, running them does.
"""
from .hparams import HyperParams  # noqa: F401
from .model import DaftExprt  # noqa: F401
from .loss import DaftExprtLoss  # noqa: F401
from .functional import manual_seed  # noqa: F401
from .ops import set_precision, get_precision, Runtime  # noqa: F401
