"""crossscore_amd: the CrossScore cross-reference scoring forward on MI355X (gfx950).

Only what the hot path needs: the C-ABI library (csrc/ -> libcrossscore_hip.so), its ctypes binding, the
drop-in `CrossScoreNet` module, the config tree, synthetic weights/inputs and the batch-shard helpers.
"""
import os as _os


def configure_runtime(hw_queues: int = 8) -> bool:
    """Optional process set-up for drivers (bench.py, crossscore_amd.predict call it first thing): asks the HIP runtime for `hw_queues`
    hardware queues (GPU_MAX_HW_QUEUES; the runtime's default is 4) so that the streams of the encoder lanes / batches in flight have
    queues to spread over.  An explicit setting in the environment wins.  It only takes effect before the process makes its first
    HIP call; returns False (and changes nothing) when the GPU is already initialised.

    Importing the package no longer sets this (a library should not edit process-wide runtime configuration on import): overlap
    does not depend on it -- every stream the path uses is PROBED for real concurrency when it is created (cs_op_streams_overlap,
    pipeline.py, csrc/api.hip) and replaced until it overlaps, which works with 2, 4 or 8 queues."""
    try:
        import torch
        if torch.cuda.is_initialized():
            return False
    except Exception:
        pass
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    return True


from .config import Cfg, load_config, model_config  # noqa: F401,E402
from .model import CrossScoreNet, load_lightning_checkpoint  # noqa: F401,E402
