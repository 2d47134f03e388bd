"""Output side of the predict drop-in: what the reference writes after each batch and at the end of the run.

Mirrors
  BatchWriter                                   utils/io/batch_writer.py:26-135,155-270 (score maps, query / reference images, item-path
                                                json, attention-weight images of the centre query patch; ground-truth score maps do not
                                                exist in predict and are not written)
  attn2rgb                                      utils/misc/image.py:55-77
  get_vrange / metric_map_write / gray2rgb      batch_writer.py:9-21, utils/io/images.py:49-63, utils/misc/image.py:37-52
  SummaryWriterPredictedOnlineTestPrediction    utils/io/score_summariser.py:142-250 (per-image mean -> CSV, "%.4f")
The float -> integer image conversion runs on the GPU (cs_op_score_to_gray16 / cs_op_score_to_rgb: 2 or 3 bytes per pixel cross
PCIe instead of 4); PNG compression is PIL's, as in the reference.  The composite matplotlib "vis" figure (task/core.py:422-434) is
not reproduced.
"""
from __future__ import annotations

import csv
import ctypes as C
import json
import os
from pathlib import Path
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _lib


def get_vrange(metric_type: str, metric_min: float, metric_max: float):
    """batch_writer.py:9-21: gray PNGs use the metric's intrinsic range, RGB the model's prediction range."""
    if metric_type == "ssim":
        intrinsic = [-1, 1]
    elif metric_type in ("mse", "mae"):
        intrinsic = [0, 1]
    else:
        raise ValueError(f"metric_type {metric_type} not supported")
    return intrinsic, [metric_min, metric_max]


def colormap_table(name: str = "turbo") -> np.ndarray:
    """(256, 3) uint8: u8() of the colormap's 256 entries (utils/misc/image.py:45-52, utils/io/images.py:20-23)."""
    import matplotlib

    lut = np.asarray(matplotlib.colormaps[name](np.arange(256)))[:, :3]
    return (lut * 255.0).astype(np.uint8)


def attn2rgb(attn_map: np.ndarray, table: np.ndarray) -> np.ndarray:
    """utils/misc/image.py:55-77: softmax weights on a log scale (eps = 1e-8) through the colormap; (H, W) fp32 -> (H, W, 3) uint8."""
    eps = 1e-8
    a = np.asarray(attn_map, np.float32).clip(0, 1)
    a = (a + eps).clip(0, 1)
    a = np.log(a) - np.log(eps)
    x = (a - 0.0) / np.float32(-np.log(eps) - 0.0)   # plt.Normalize(vmin=0, vmax=-log(eps)) on a float32 array
    x = x * np.float32(256)
    idx = x.astype(np.int64)
    idx[x == 256] = 255
    idx[~(x >= 0)] = 0
    idx[x > 256] = 255
    return table[np.clip(idx, 0, 255)]


def name_stem(path: str) -> str:
    """batch_writer.py:112-115: last five path components joined by "_", ".png" removed."""
    return str(Path(*Path(path).parts[-5:])).replace("/", "_").replace(".png", "")


class ScoreMapEncoder:
    """Device score maps (B, H, W) fp32 -> host integer images, converted on the GPU."""

    def __init__(self, metric_type: str, metric_min: float, metric_max: float, colour_mode: str, device: torch.device):
        if colour_mode not in ("gray", "rgb"):
            raise ValueError(f"colour_mode {colour_mode} not supported")
        self.intrinsic, self.vis = get_vrange(metric_type, metric_min, metric_max)
        self.colour_mode = colour_mode
        self.device = device
        self._lut = torch.from_numpy(colormap_table("turbo").reshape(-1)).to(device) if colour_mode == "rgb" else None

    def __call__(self, score: torch.Tensor) -> np.ndarray:
        lib = _lib.load()
        score = score.contiguous()
        if score.dtype != torch.float32 or not score.is_cuda:
            raise ValueError("score maps must be fp32 CUDA tensors")
        n = score.numel()
        st = C.c_void_p(torch.cuda.current_stream(score.device).cuda_stream)
        if self.colour_mode == "gray":
            out = torch.empty(score.shape, dtype=torch.int16, device=score.device)  # 16-bit samples (viewed unsigned on the host)
            _lib.check(lib.cs_op_score_to_gray16(C.c_void_p(score.data_ptr()), n, 1 if self.intrinsic == [-1, 1] else 0, C.c_void_p(out.data_ptr()), st))
            return out.cpu().numpy().view(np.uint16)
        out = torch.empty(tuple(score.shape) + (3,), dtype=torch.uint8, device=score.device)
        _lib.check(lib.cs_op_score_to_rgb(C.c_void_p(score.data_ptr()), n, float(self.vis[0]), float(self.vis[1]), C.c_void_p(self._lut.data_ptr()),
                                         C.c_void_p(out.data_ptr()), st))
        return out.cpu().numpy()


def save_png(path, arr: np.ndarray) -> None:
    from PIL import Image

    # uint16 arrays become 16-bit grayscale PNGs ("I;16"): what imageio writes for the reference's int32 maps (utils/io/images.py:31-36)
    Image.fromarray(arr).save(path)


class BatchWriter:
    """PNG compression runs on a small thread pool (zlib releases the GIL): the arrays are materialised in the calling thread,
    the files are complete after finish()."""

    def __init__(self, cfg, phase: str, img_mean_std: torch.Tensor, device: torch.device, workers: int = 4):
        if phase not in ("test", "predict"):
            raise ValueError(f"Phase {phase} not supported. Has to be a Lightening phase test/predict.")
        self.cfg = cfg
        self.out_dir = Path(cfg.logger[phase].out_dir)
        self.write_config = cfg.logger[phase].write.config
        self.write_flag = cfg.logger[phase].write.flag
        m = cfg.model.predict.metric
        self.encoder = ScoreMapEncoder(m.type, m.min, m.max, self.write_config.score_map_colour_mode, device)
        self.img_mean_std = img_mean_std.detach().float().cpu()
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self._pending = []
        # batch_writer.py:42-45: attention images only when the model returns the weights
        self.write_attn = bool(self.write_flag["attn_weights"]) and bool(cfg.model.need_attn_weights)
        self.out_dir_dict = {"batch": Path(self.out_dir, "batch")}
        if self.write_flag["batch"]:
            for k in self.write_flag.keys():
                if k not in ("batch", "score_map_prediction") and self.write_flag[k] and k in ("item_path_json", "image_query", "image_reference", "attn_weights"):
                    self.out_dir_dict[k] = Path(self.out_dir_dict["batch"], k)
                    self.out_dir_dict[k].mkdir(parents=True, exist_ok=True)

    # batch_writer.py:63-104
    def write_out(self, batch_input, batch_output, local_rank: int, batch_idx: int) -> List[str]:
        written: List[str] = []
        if self.write_flag["score_map_prediction"]:
            written += self._write_score_map_prediction(batch_input, batch_output, local_rank, batch_idx)
        if self.write_flag["item_path_json"]:
            out_path = self.out_dir_dict["item_path_json"] / f"r{local_rank}_B{str(batch_idx).zfill(4)}.json"
            item_paths = dict(batch_input["item_paths"])
            if len(item_paths["reference/cross/imgs"]) > 0:  # transpose to (B, N_ref) like batch_writer.py:160-163
                item_paths["reference/cross/imgs"] = np.array(item_paths["reference/cross/imgs"]).T.tolist()
            with open(out_path, "w") as f:
                json.dump(item_paths, f, indent=2)
            written.append(str(out_path))
        if self.write_flag["image_query"]:
            stems = [name_stem(p) for p in batch_input["item_paths"]["query/img"]]
            for b, (stem, img) in enumerate(zip(stems, batch_input["query/img"])):
                path = self.out_dir_dict["image_query"] / f"r{local_rank}_B{batch_idx:04}_b{b:03}_{stem}.png"
                self._save(path, self._de_norm_u8(img))
                written.append(str(path))
        if self.write_flag["image_reference"] and len(batch_input["item_paths"]["reference/cross/imgs"]) > 0:
            stems = [name_stem(p) for p in batch_input["item_paths"]["query/img"]]
            ref_paths = np.array(batch_input["item_paths"]["reference/cross/imgs"]).T  # (B, N_ref)
            for b, stem in enumerate(stems):
                d = self.out_dir_dict["image_reference"] / f"r{local_rank}_B{batch_idx:04}_b{b:03}_{stem}" / "cross"
                d.mkdir(parents=True, exist_ok=True)
                for ref_idx, (rp, img) in enumerate(zip(ref_paths[b], batch_input["reference/cross/imgs"][b])):
                    path = d / f"ref{ref_idx:02}_{name_stem(rp)}.png"
                    self._save(path, self._de_norm_u8(img))
                    written.append(str(path))
        if self.write_attn and len(batch_input["item_paths"]["reference/cross/imgs"]) > 0:
            written += self._write_attn_weights(batch_input, batch_output, local_rank, batch_idx)
        return written

    def _write_attn_weights(self, batch_input, batch_output, local_rank, batch_idx) -> List[str]:
        """batch_writer.py:202-261 with check_patch_mode="centre": per query, the (N_ref, h, w) attention of its centre patch."""
        written = []
        table = colormap_table("turbo")
        stems = [name_stem(p) for p in batch_input["item_paths"]["query/img"]]
        ref_paths = np.array(batch_input["item_paths"]["reference/cross/imgs"]).T  # (B, N_ref)
        amap = batch_output["attn_weights_map_ref_cross"]                           # (B, h, w, N_ref, h, w)
        th, tw = amap.shape[1:3]
        for b, stem in enumerate(stems):
            d = self.out_dir_dict["attn_weights"] / f"r{local_rank}_B{batch_idx:04}_b{b:03}_{stem}" / "cross"
            d.mkdir(parents=True, exist_ok=True)
            maps = amap[b, th // 2, tw // 2].detach().float().cpu().numpy()        # (N_ref, h, w)
            for ref_idx, (rp, m) in enumerate(zip(ref_paths[b], maps)):
                path = d / f"ref{ref_idx:02}_{name_stem(rp)}.png"
                self._save(path, attn2rgb(m, table))
                written.append(str(path))
        return written

    def _save(self, path, arr: np.ndarray) -> None:
        self._pending.append(self._pool.submit(save_png, path, np.ascontiguousarray(arr)))

    def finish(self) -> None:
        """Waits for every queued file (re-raises the first failure)."""
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()

    def _de_norm_u8(self, img_chw: torch.Tensor) -> np.ndarray:
        """de_norm_img + u8 (utils/misc/image.py:25-34, utils/io/images.py:20-23): x*std + mean, *255, truncated to uint8."""
        x = img_chw.detach().float().cpu().permute(1, 2, 0)
        x = x * self.img_mean_std[3:][None, None] + self.img_mean_std[:3][None, None]
        return (x.numpy() * 255.0).astype(np.uint8)

    def _write_score_map_prediction(self, batch_input, batch_output, local_rank, batch_idx) -> List[str]:
        written = []
        stems = [name_stem(p) for p in batch_input["item_paths"]["query/img"]]
        for key in [k for k in batch_output.keys() if k.startswith("score_map")]:
            d = Path(self.out_dir_dict["batch"], key)
            d.mkdir(parents=True, exist_ok=True)
            if len(stems) != len(batch_output[key]):
                raise ValueError("num of query images and score maps are not equal")
            imgs = self.encoder(batch_output[key])
            for b, stem in enumerate(stems):
                path = d / f"r{local_rank}_B{batch_idx:04}_b{b:03}_{stem}.png"
                self._save(path, imgs[b])
                written.append(str(path))
        return written


class ScoreSummariser:
    """SummaryWriterPredictedOnlineTestPrediction: per-image mean scores -> score_summary/<dataset_type>/<rendering_method>.csv."""

    def __init__(self, metric_type: str, metric_min: float, dir_out):
        if metric_type == "ssim":
            metric_str = f"{metric_type}_-1_1" if metric_min == -1 else f"{metric_type}_0_1"
        else:
            metric_str = f"{metric_type}"
        self.columns = ["scene_name", "rendered_dir", "image_name", f"pred_{metric_str}"]
        self.csv_dir = Path(dir_out).expanduser() / "score_summary"
        self.csv_dir.mkdir(parents=True, exist_ok=True)
        self.rows: List[list] = []

    @staticmethod
    def _part(parts: Sequence[str], idx: int) -> str:
        return parts[idx] if -len(parts) <= idx < len(parts) else "unknown"  # the reference raises IndexError on such short paths

    def update(self, batch_input, batch_output, means: torch.Tensor = None) -> None:
        """score_summariser.py:166-195.  `means` = per-image means already reduced on the device (cs_forward's mean output); when
        absent they are taken from the single score-map entry like the reference does."""
        paths = batch_input["item_paths"]["query/img"]
        keys = [k for k in batch_output.keys() if k.startswith("score_map")]
        if len(keys) != 1:
            raise ValueError(f"Expect exactly one ref_type: self/cross, but got {keys}.")
        scores = means if means is not None else batch_output[keys[0]].mean(dim=[-1, -2])
        scores = scores.detach().float().cpu().tolist()
        for p, s in zip(paths, scores):
            parts = p.split("/")
            rendered = os.path.join(*parts[:-2]) if len(parts) > 2 else ""
            self.rows.append([self._part(parts, -5), rendered, parts[-1].replace("frame_", ""), s])

    def summarise(self) -> List[str]:
        """score_summariser.py:197-250: group by rendered_dir components, sort, write with float_format "%.4f"."""
        methods, datasets = [], []
        for r in self.rows:
            parts = r[1].split("/")
            m, d = self._part(parts, -6), self._part(parts, -5)
            if m not in methods:
                methods.append(m)
            if d not in datasets:
                datasets.append(d)
        written = []
        for d in datasets:
            for m in methods:
                rows = [r for r in self.rows if (m in r[1] or m == "unknown") and (d in r[1] or d == "unknown")]
                rows.sort(key=lambda r: (r[0], r[1], r[2]))
                out_dir = self.csv_dir / d
                out_dir.mkdir(parents=True, exist_ok=True)
                path = out_dir / f"{m}.csv"
                with open(path, "w", newline="") as f:
                    w = csv.writer(f, lineterminator="\n")
                    w.writerow(self.columns)
                    for r in rows:
                        w.writerow([r[0], r[1], r[2], "%.4f" % r[3]])
                written.append(str(path))
        return written

    def __len__(self) -> int:
        return len(self.rows)
