"""CPU restatement of the reference's score-map writers (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path).

  metric_map_write   utils/io/images.py:49-63      m*65535 ([0,1]) or (m+1)*32767 ([-1,1]), astype(int32), stored as 16-bit PNG
  gray2rgb           utils/misc/image.py:37-52     plt.Normalize(vmin, vmax) -> colormap("turbo") -> u8()
  u8                 utils/io/images.py:20-23      (x*255).astype(uint8)
Pinned by tests/test_predict_driver.py against matplotlib itself (the reference calls cm.get_cmap, removed in matplotlib 3.9; the
same table is matplotlib.colormaps["turbo"]): the oracle must reproduce matplotlib's Normalize + Colormap.__call__ byte for byte.
"""
import numpy as np


def gray16(score: np.ndarray, vrange) -> np.ndarray:
    m = np.asarray(score, np.float32)
    if list(vrange) == [0, 1]:
        m = m * 65535
    elif list(vrange) == [-1, 1]:
        m = (m + 1) * 32767
    else:
        raise ValueError("Invalid range for metric map writing. Must be '[0,1]' or '[-1,1]'")
    return np.clip(m.astype(np.int32), 0, 65535).astype(np.uint16)


def turbo_table() -> np.ndarray:
    import matplotlib

    return (np.asarray(matplotlib.colormaps["turbo"](np.arange(256)))[:, :3] * 255.0).astype(np.uint8)


def rgb(score: np.ndarray, vrange, table: np.ndarray) -> np.ndarray:
    vmin, vmax = np.float32(vrange[0]), np.float32(vrange[1])
    x = (np.asarray(score, np.float32) - vmin) / (vmax - vmin)   # Normalize.__call__ on a float32 array stays float32
    x = x * np.float32(256)
    idx = x.astype(np.int64)
    idx[x == 256] = 255
    idx[~(x >= 0)] = 0
    idx[x > 256] = 255
    return table[np.clip(idx, 0, 255)]
