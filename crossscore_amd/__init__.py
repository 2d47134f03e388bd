"""crossscore_amd: the CrossScore cross-reference scoring forward on MI355X (gfx950).

Only what the hot path needs: the C-ABI library (csrc/ -> libcrossscore_hip.so), its ctypes binding, the
drop-in `CrossScoreNet` module, the config tree, synthetic weights/inputs and the batch-shard helpers.
"""
import os as _os

# The forward overlaps work on several HIP streams (encoder lanes, the decoder's K/V side stream, one stream per batch in flight).  The
# HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that land on one queue serialise.  Eight
# queues keep them apart (bench.py documents the measured effect).  Only a default: an explicit setting wins, and it only takes effect
# if this package is imported before the process makes its first HIP call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import Cfg, load_config, model_config  # noqa: F401,E402
from .model import CrossScoreNet, load_lightning_checkpoint  # noqa: F401,E402
