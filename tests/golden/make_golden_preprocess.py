"""Golden vectors for the input stage (run in the build container; needs only torch): what the reference's transforms produce for
seeded uint8 images.  The reference applies torchvision T.Resize(antialias=True) to float CHW tensors, which is
torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=True); torchvision is absent here, so the
goldens call that function directly and apply the reference's /255, crop and (x - mean) / std around it.
usage: python tests/golden/make_golden_preprocess.py"""
import os
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
MEAN = torch.tensor((0.485, 0.456, 0.406))[:, None, None]
STD = torch.tensor((0.229, 0.224, 0.225))[:, None, None]


def out_size(h, w, short):
    return (short, int(short * w / h)) if h <= w else (int(short * h / w), short)


def run(name, h, w, short, crop, seed, compact=False):
    rng = np.random.Generator(np.random.PCG64(seed))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    # smooth structure as well as noise: low-frequency ramp in one channel
    img[..., 1] = ((np.arange(h)[:, None] * 3 + np.arange(w)[None, :] * 2) % 256).astype(np.uint8)
    x = torch.from_numpy(img.astype(np.float32) / np.float32(255.0)).permute(2, 0, 1)
    rs = out_size(h, w, short) if short > 0 else (h, w)
    if rs != (h, w):
        x = F.interpolate(x[None], size=rs, mode="bilinear", align_corners=False, antialias=True)[0]
    if crop is not None:
        y0, x0, ch, cw = crop
        x = x[:, y0:y0 + ch, x0:x0 + cw]
    x = x.sub(MEAN).div(STD)
    y = x.numpy()
    d = dict(h=h, w=w, short=short, rs=np.array(rs), crop=np.array(crop if crop is not None else (-1, -1, -1, -1)), seed=seed)
    if compact:  # large case: a few rows/columns and moments only
        d.update(rows=y[:, ::97, :], cols=y[:, :, ::101], mean=y.mean(axis=(1, 2), dtype=np.float64), abs_mean=np.abs(y).mean(dtype=np.float64))
    else:
        d.update(out=y)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, rs, y.shape, float(y.mean()))


if __name__ == "__main__":
    torch.set_num_threads(1)
    run("p0_down_45x60_s37", 45, 60, 37, None, 11)
    run("p1_down_120x90_s40_crop", 120, 90, 40, (0, 0, 28, 28), 12)
    run("p2_up_20x30_s28", 20, 30, 28, None, 13)
    run("p3_noresize_33x47", 33, 47, -1, (0, 0, 28, 42), 14)
    run("p4_540x720_s518", 540, 720, 518, None, 15, compact=True)
