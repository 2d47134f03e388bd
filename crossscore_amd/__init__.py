"""crossscore_amd: the CrossScore cross-reference scoring forward on MI355X (gfx950).

Only what the hot path needs: the C-ABI library (csrc/ -> libcrossscore_hip.so), its ctypes binding, the
drop-in `CrossScoreNet` module, the config tree, synthetic weights/inputs and the batch-shard helpers.
"""
from .config import Cfg, load_config, model_config  # noqa: F401
from .model import CrossScoreNet, load_lightning_checkpoint  # noqa: F401
