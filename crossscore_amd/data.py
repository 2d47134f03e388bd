"""Predict-time data side of the drop-in: which files form a (query, N references) item, and the GPU input stage.

Mirrors, for the predict path only:
  SimpleReference.get_paths            dataloading/dataset/simple_reference.py:42-84   sorted listdir of query_dir / reference_dir
  NeighbourSelector + SamplerRandom    dataloading/dataset/nvs_dataset.py:14-84, utils/neighbour/sampler.py:15-38
  load_content / resize_all / crops / T.Normalize   nvs_dataset.py:218-279,429-470, task/predict.py:68-93
The pixel work (x/255, antialiased resize, crop, normalise) runs on the GPU through cs_op_preprocess_u8 straight from the decoded
uint8 image; only PNG/JPEG decoding stays on the host (PIL, as in utils/io/images.py:26-29).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .synth import IMAGENET_MEAN_STD

EMPTY = "empty_image"  # placeholder path the reference pads short reference lists with (sampler.py:22-27)


def resized_output_size(h: int, w: int, short: int) -> Tuple[int, int]:
    """torchvision T.Resize(int): the short side becomes `short`, the long side int(short * long / short_side)."""
    if h <= w:
        return short, int(short * w / h)
    return int(short * h / w), short


def list_paths(query_dir: str, reference_dir: str) -> Tuple[List[str], List[str]]:
    """simple_reference.py:53-58: every entry of the two directories, sorted by name."""
    query_dir, reference_dir = os.path.expanduser(query_dir), os.path.expanduser(reference_dir)
    q = [os.path.join(query_dir, p) for p in sorted(os.listdir(query_dir))]
    r = [os.path.join(reference_dir, p) for p in sorted(os.listdir(reference_dir))]
    return q, r


def sample_references(ref_list: Sequence[str], n_sample: int, deterministic: bool, rng=np.random) -> List[str]:
    """SamplerRandom.sample (utils/neighbour/sampler.py:19-38): first N when deterministic, else N without replacement from the
    global numpy RNG; a short list is padded with "empty_image" placeholders and permuted."""
    ref_list = list(ref_list)
    if n_sample > len(ref_list):
        result = ref_list + [EMPTY] * (n_sample - len(ref_list))
        return rng.permutation(result).tolist()
    if deterministic:
        return ref_list[:n_sample]
    return rng.choice(ref_list, n_sample, replace=False).tolist()


def read_image_u8(path: str) -> np.ndarray:
    """Decoded uint8 HWC RGB image (utils/io/images.py:26-29 keeps whatever channel count PIL returns; the model needs 3)."""
    from PIL import Image

    img = np.array(Image.open(path))
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    if img.shape[2] == 4:
        img = img[:, :, :3]
    if img.dtype != np.uint8 or img.shape[2] != 3:
        raise ValueError(f"{path}: expected an 8-bit RGB image, got {img.dtype} {img.shape}")
    return np.ascontiguousarray(img)


class InputStage:
    """uint8 HWC images -> the normalised fp32 batch tensors CrossScoreNet.forward takes, on `device`."""

    def __init__(self, device: torch.device, resize_short_side: int = 518, crop_size: Optional[int] = None, integer_patches: bool = False,
                 patch: int = 14, mean_std: Sequence[float] = IMAGENET_MEAN_STD):
        self.device = device
        self.resize_short_side = int(resize_short_side)
        self.crop_size = crop_size
        self.integer_patches = integer_patches
        self.patch = patch
        self.mean_std = tuple(float(v) for v in mean_std)
        self._mean = (C.c_float * 3)(*mean_std[:3])
        self._std = (C.c_float * 3)(*mean_std[3:])
        self._scratch: Optional[torch.Tensor] = None
        ms = torch.tensor(list(mean_std), dtype=torch.float32)
        self.zero_image_value = ((torch.zeros(3) - ms[:3]) / ms[3:]).to(device)  # a black pixel after T.Normalize

    def geometry(self, h: int, w: int):
        """(resized (h, w), crop (y, x, h, w)) for an input of h x w: resize_all, then the deterministic crop (crop.py:19-22: i = j = 0)
        or the integer-patch crop (nvs_dataset.py:227-241)."""
        rs = resized_output_size(h, w, self.resize_short_side) if self.resize_short_side > 0 else (h, w)
        oh, ow = rs
        if self.crop_size is not None:
            oh = ow = int(self.crop_size)
            if oh > rs[0] or ow > rs[1]:
                raise ValueError(f"crop {oh}x{ow} larger than the resized image {rs[0]}x{rs[1]}")
        elif self.integer_patches:
            oh, ow = rs[0] - rs[0] % self.patch, rs[1] - rs[1] % self.patch
        return rs, (0, 0, oh, ow)

    def __call__(self, img_u8: np.ndarray, out: torch.Tensor) -> None:
        """Writes the processed image into `out` ((3, oh, ow) fp32 slice on the device)."""
        lib = _lib.load()
        h, w, _ = img_u8.shape
        rs, crop = self.geometry(h, w)
        if tuple(out.shape) != (3, crop[2], crop[3]) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError(f"output slice must be contiguous fp32 (3,{crop[2]},{crop[3]}), got {tuple(out.shape)}")
        d_img = torch.from_numpy(img_u8).to(self.device, non_blocking=False)
        scratch = None
        if rs != (h, w):
            need = h * rs[1] * 3
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty((need,), dtype=torch.float32, device=self.device)
            scratch = C.c_void_p(self._scratch.data_ptr())
        _lib.check(lib.cs_op_preprocess_u8(C.c_void_p(d_img.data_ptr()), h, w, w * 3, rs[0], rs[1], crop[0], crop[1], crop[2], crop[3],
                                           self._mean, self._std, C.c_void_p(out.data_ptr()), scratch,
                                           C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        # d_img may be released once the stream has consumed it
        d_img.record_stream(torch.cuda.current_stream(self.device))


    # -- the one-pass form (SURVEY.md 8f-4 as worded): nothing is computed here, the patch-embedding launch does the pixel work --------------
    def describe(self, img_u8: np.ndarray):
        """The decoded image on the device with its geometry (model.U8Image) for CrossScoreNet.forward_u8 and its siblings."""
        from .model import U8Image
        h, w, _ = img_u8.shape
        rs, crop = self.geometry(h, w)
        return U8Image(torch.from_numpy(img_u8).to(self.device, non_blocking=False), h, w, rs, crop[0], crop[1])

    def placeholder(self, size):
        """The all-zero image (placeholders of short reference lists, zero_reference: nvs_dataset.py:459-470) of the processed size."""
        from .model import U8Image
        return U8Image(None, size[0], size[1], size)

    def batch(self, images, size):
        from .model import U8Batch
        return U8Batch(images, size, self.mean_std)


class SimpleReferenceItems:
    """Index -> file paths of one item, like NeighbourSelector.__getitem__ for the single-scene layout of SimpleReference."""

    def __init__(self, query_dir: str, reference_dir: str, neighbour_config) -> None:
        if neighbour_config["strategy"] != "random":
            raise NotImplementedError(f"neighbour strategy {neighbour_config['strategy']} (sampler.py:60-66 only knows 'random')")
        self.query_paths, self.reference_paths = list_paths(query_dir, reference_dir)
        self.n_cross = int(neighbour_config["cross"])
        self.deterministic = bool(neighbour_config["deterministic"])

    def __len__(self) -> int:
        return len(self.query_paths)

    def __getitem__(self, idx: int) -> Dict[str, object]:
        refs = sample_references(self.reference_paths, self.n_cross, self.deterministic) if self.n_cross > 0 else []
        return {"query/img": self.query_paths[idx], "query/score_map": EMPTY, "reference/cross/imgs": refs}


def decode_items(items: List[Dict[str, object]], zero_reference: bool = False, pool=None, skip=()) -> Dict[str, np.ndarray]:
    """path -> decoded uint8 image for every file the items name (host side; PIL releases the GIL while decoding, so a thread pool
    plays the role of the reference's DataLoader workers, task/predict.py:110-117).  Reference paths in `skip` (already in the
    token cache) are not decoded."""
    paths = []
    for it in items:
        paths.append(it["query/img"])
        if not zero_reference:
            paths += [p for p in it["reference/cross/imgs"] if p != EMPTY and p not in skip]
    uniq = list(dict.fromkeys(paths))
    imgs = list(pool.map(read_image_u8, uniq)) if pool is not None else [read_image_u8(p) for p in uniq]
    return dict(zip(uniq, imgs))


def load_batch(items: List[Dict[str, object]], stage: InputStage, zero_reference: bool = False,
               decoded: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, object]:
    """One batch dict with the keys `_core_step` reads (task/core.py:265-272) plus `item_paths` in the collated layout the
    writers expect (default_collate turns the per-item list of N reference paths into N lists of B paths).  `decoded` holds
    images already read by decode_items (a reference image shared by several items of the batch is decoded once)."""
    B = len(items)
    decoded = decoded if decoded is not None else {}
    get = lambda p: decoded[p] if p in decoded else read_image_u8(p)  # noqa: E731
    q_imgs = [get(it["query/img"]) for it in items]
    geo = {stage.geometry(*im.shape[:2])[1][2:] for im in q_imgs}
    if len(geo) != 1:
        raise ValueError(f"query images of one batch must share the processed size, got {sorted(geo)}")
    oh, ow = next(iter(geo))
    N = len(items[0]["reference/cross/imgs"])
    query = torch.empty((B, 3, oh, ow), dtype=torch.float32, device=stage.device)
    refs = torch.empty((B, N, 3, oh, ow), dtype=torch.float32, device=stage.device)
    for b, (it, qi) in enumerate(zip(items, q_imgs)):
        stage(qi, query[b])
        for n, p in enumerate(it["reference/cross/imgs"]):
            if p == EMPTY or zero_reference:
                # nvs_dataset.py:459-470: placeholders and zero_reference are all-zero images BEFORE T.Normalize -> (0 - mean) / std
                refs[b, n] = stage.zero_image_value[:, None, None]
                continue
            ri = get(p)
            if stage.geometry(*ri.shape[:2])[1][2:] != (oh, ow):
                raise ValueError(f"{p}: processed size differs from the query's {oh}x{ow}")
            stage(ri, refs[b, n])
    item_paths = {"query/img": [it["query/img"] for it in items], "query/score_map": [it["query/score_map"] for it in items],
                  "reference/cross/imgs": [[it["reference/cross/imgs"][n] for it in items] for n in range(N)]}
    return {"query/img": query, "reference/cross/imgs": refs, "item_paths": item_paths}


def load_batch_u8(items: List[Dict[str, object]], stage: InputStage, zero_reference: bool = False,
                  decoded: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, object]:
    """load_batch for the one-pass input stage: "query/img" and "reference/cross/imgs" are model.U8Batch objects (decoded images on the device +
    geometry) for CrossScoreNet.forward_u8; no processed fp32 tensor exists."""
    decoded = decoded if decoded is not None else {}
    get = lambda p: decoded[p] if p in decoded else read_image_u8(p)  # noqa: E731
    q_imgs = [get(it["query/img"]) for it in items]
    geo = {stage.geometry(*im.shape[:2])[1][2:] for im in q_imgs}
    if len(geo) != 1:
        raise ValueError(f"query images of one batch must share the processed size, got {sorted(geo)}")
    size = next(iter(geo))
    N = len(items[0]["reference/cross/imgs"])
    refs = []
    for it in items:
        for p in it["reference/cross/imgs"]:
            if p == EMPTY or zero_reference:
                refs.append(stage.placeholder(size))
                continue
            ri = get(p)
            if stage.geometry(*ri.shape[:2])[1][2:] != size:
                raise ValueError(f"{p}: processed size differs from the query's {size[0]}x{size[1]}")
            refs.append(stage.describe(ri))
    item_paths = {"query/img": [it["query/img"] for it in items], "query/score_map": [it["query/score_map"] for it in items],
                  "reference/cross/imgs": [[it["reference/cross/imgs"][n] for it in items] for n in range(N)]}
    return {"query/img": stage.batch([stage.describe(qi) for qi in q_imgs], size), "reference/cross/imgs": stage.batch(refs, size),
            "item_paths": item_paths}


class ReferenceTokenCache:
    """SURVEY.md 8f-3 inside the predict loop: references are sampled from one finite directory, so each reference image is
    pre-processed and encoded ONCE (CrossScoreNet.encode_references) and queries are scored with forward_cached -- the same bits
    as the full forward (tests/test_hip_forward.py), with 1 instead of 1 + N images through the encoder per query."""

    def __init__(self, net, stage: InputStage, keep_images: bool, max_images: int = 4096, from_u8: bool = False):
        self.net, self.stage, self.keep_images, self.max_images = net, stage, keep_images, max_images
        self.from_u8 = bool(from_u8) and not keep_images  # one-pass input stage: no processed fp32 image exists to keep
        self.tokens: Dict[object, torch.Tensor] = {}
        self.images: Dict[object, torch.Tensor] = {}

    def has(self, path: str) -> bool:
        return any(k[0] == path for k in self.tokens)

    def gather(self, ref_lists: List[List[str]], decoded: Dict[str, np.ndarray], size: Tuple[int, int], zero_reference: bool):
        """ref_lists: per item the N reference paths -> (tokens (B,N,Np,C) fp16, images (B,N,3,h,w) fp32 or None)."""
        oh, ow = size
        keys = [[(EMPTY if (p == EMPTY or zero_reference) else p, oh, ow) for p in refs] for refs in ref_lists]
        missing = list(dict.fromkeys(k for ks in keys for k in ks if k not in self.tokens))
        if missing:
            if len(self.tokens) + len(missing) > self.max_images:
                self.tokens.clear()
                self.images.clear()
                missing = list(dict.fromkeys(k for ks in keys for k in ks))
            if self.from_u8:
                descs = []
                for k in missing:
                    if k[0] == EMPTY:
                        descs.append(self.stage.placeholder((oh, ow)))
                        continue
                    img = decoded[k[0]] if k[0] in decoded else read_image_u8(k[0])
                    if self.stage.geometry(*img.shape[:2])[1][2:] != (oh, ow):
                        raise ValueError(f"{k[0]}: processed size differs from the query's {oh}x{ow}")
                    descs.append(self.stage.describe(img))
                tok = self.net.encode_references_u8(self.stage.batch(descs, (oh, ow)))
                for i, k in enumerate(missing):
                    self.tokens[k] = tok[i]
                missing = []
            buf = torch.empty((len(missing), 3, oh, ow), dtype=torch.float32, device=self.stage.device) if missing else None
            for i, k in enumerate(missing):
                if k[0] == EMPTY:
                    buf[i] = self.stage.zero_image_value[:, None, None]
                    continue
                img = decoded[k[0]] if k[0] in decoded else read_image_u8(k[0])
                if self.stage.geometry(*img.shape[:2])[1][2:] != (oh, ow):
                    raise ValueError(f"{k[0]}: processed size differs from the query's {oh}x{ow}")
                self.stage(img, buf[i])
            tok = self.net.encode_references(buf) if missing else None
            for i, k in enumerate(missing):
                self.tokens[k] = tok[i]
                if self.keep_images:
                    self.images[k] = buf[i].clone()
        tokens = torch.stack([torch.stack([self.tokens[k] for k in ks]) for ks in keys])
        images = torch.stack([torch.stack([self.images[k] for k in ks]) for ks in keys]) if self.keep_images else None
        return tokens, images


def load_query_batch(items: List[Dict[str, object]], stage: InputStage, decoded: Dict[str, np.ndarray]):
    """Query tensors + item paths of one batch (the references come from a ReferenceTokenCache)."""
    q_imgs = [decoded[it["query/img"]] if it["query/img"] in decoded else read_image_u8(it["query/img"]) for it in items]
    geo = {stage.geometry(*im.shape[:2])[1][2:] for im in q_imgs}
    if len(geo) != 1:
        raise ValueError(f"query images of one batch must share the processed size, got {sorted(geo)}")
    oh, ow = next(iter(geo))
    query = torch.empty((len(items), 3, oh, ow), dtype=torch.float32, device=stage.device)
    for b, qi in enumerate(q_imgs):
        stage(qi, query[b])
    N = len(items[0]["reference/cross/imgs"])
    item_paths = {"query/img": [it["query/img"] for it in items], "query/score_map": [it["query/score_map"] for it in items],
                  "reference/cross/imgs": [[it["reference/cross/imgs"][n] for it in items] for n in range(N)]}
    return {"query/img": query, "reference/cross/imgs": None, "item_paths": item_paths}, (oh, ow)


def load_query_batch_u8(items: List[Dict[str, object]], stage: InputStage, decoded: Dict[str, np.ndarray]):
    """load_query_batch for the one-pass input stage: "query/img" is a model.U8Batch."""
    q_imgs = [decoded[it["query/img"]] if it["query/img"] in decoded else read_image_u8(it["query/img"]) for it in items]
    geo = {stage.geometry(*im.shape[:2])[1][2:] for im in q_imgs}
    if len(geo) != 1:
        raise ValueError(f"query images of one batch must share the processed size, got {sorted(geo)}")
    size = next(iter(geo))
    N = len(items[0]["reference/cross/imgs"])
    item_paths = {"query/img": [it["query/img"] for it in items], "query/score_map": [it["query/score_map"] for it in items],
                  "reference/cross/imgs": [[it["reference/cross/imgs"][n] for it in items] for n in range(N)]}
    return {"query/img": stage.batch([stage.describe(qi) for qi in q_imgs], size), "reference/cross/imgs": None, "item_paths": item_paths}, size
