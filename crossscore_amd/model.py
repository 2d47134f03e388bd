"""Drop-in `CrossScoreNet` whose forward runs on the hand-written gfx950 kernels of libcrossscore_hip.so.

Mirrors the reference model boundary (task/core.py:26-161):
  * `CrossScoreNet(cfg)` reads the same config keys (config/model/model.yaml),
  * `state_dict()` has exactly the reference's keys and shapes, so a Lightning checkpoint's
    `ckpt["state_dict"]` (keys prefixed "model.") loads with `strict=True`,
  * `forward(query_img, ref_cross_imgs, need_attn_weights, need_attn_weights_head_id, norm_img)` returns
    `{"score_map_ref_cross": (B,14h,14w) fp32, "attn_weights_map_ref_cross": None | (B,h,w,N,h,w) fp32}`.
PyTorch is used here only for device buffers, the current stream and (de)serialisation: every arithmetic
stage of the forward is a HIP kernel launched through the C ABI.  There is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _lib, synth
from .synth import ArchSpec, BACKBONES


# ----- config helpers restating utils/check_config.py:1-28 and model/regression_layer.py:31-62 ---------------
def check_metric_prediction_config(metric_type, metric_min, metric_max) -> None:
    if metric_type not in ("ssim", "mse", "mae"):
        raise ValueError(f"Invalid metric type {metric_type}")
    valid_max = metric_max == 1
    valid_min = (metric_min in (-1, 0)) if metric_type == "ssim" else (metric_min == 0)
    if not (valid_min and valid_max):
        raise ValueError(f"Invalid metric range {metric_min} to {metric_max} for {metric_type}")


def regression_activation(metric_type, metric_min, metric_max, power_factor):
    """-> (act, p): act 0 = sigmoid, 1 = tanh; p = exponent applied after the activation."""
    check_metric_prediction_config(metric_type, metric_min, metric_max)
    if metric_min == -1:
        act = 1
    elif metric_min == 0:
        act = 0
    else:
        raise ValueError(f"metric_min={metric_min} not supported")
    if metric_min == 0:
        p = {"ssim": 1, "mae": 2, "mse": 4}[metric_type] if power_factor == "default" else power_factor
    else:
        p = 1
    return act, float(p)


def arch_from_cfg(cfg) -> ArchSpec:
    """Architecture constants from cfg.model (the reference gets hidden size / layers / heads from
    Dinov2Config.from_pretrained(cfg.model.backbone.from_pretrained), task/core.py:39; the hub is unreachable
    offline, so the two published backbones are tabulated; cfg.model.backbone may also carry explicit
    hidden_size / num_hidden_layers / num_attention_heads / image_size)."""
    m = cfg.model
    bb = m.backbone
    name = bb.from_pretrained
    if name in BACKBONES:
        a = BACKBONES[name]
    elif all(k in bb for k in ("hidden_size", "num_hidden_layers", "num_attention_heads")):
        a = ArchSpec(hidden=int(bb.hidden_size), enc_layers=int(bb.num_hidden_layers), enc_heads=int(bb.num_attention_heads),
                     pos_grid=int(bb.get("image_size", 518)) // int(m.patch_size), name=str(name), swiglu=bool(bb.get("use_swiglu_ffn", False)))
    else:
        raise ValueError(f"unknown backbone '{name}': known {sorted(BACKBONES)} (or give explicit sizes in cfg.model.backbone)")
    return dataclasses.replace(a, patch=int(m.patch_size), pe_h=int(m.pos_enc.multi_view.h), pe_w=int(m.pos_enc.multi_view.w),
                               do_self_attn=bool(m.decoder_do_self_attn))


class U8Image:
    """One decoded image on the device with its input-stage geometry (cs_u8_image): data (h, w, 3) uint8 CUDA tensor (rows may be padded:
    row_bytes), rs = size after T.Resize, (crop_y, crop_x) = top-left corner of the processed window.  data None = the all-zero placeholder
    image the reference pads short reference lists with (nvs_dataset.py:459-470)."""
    __slots__ = ("data", "h", "w", "row_bytes", "rs", "crop_y", "crop_x")

    def __init__(self, data: Optional[torch.Tensor], h: int, w: int, rs, crop_y: int = 0, crop_x: int = 0, row_bytes: Optional[int] = None):
        if data is not None and (not data.is_cuda or data.dtype != torch.uint8 or not data.is_contiguous()):
            raise ValueError("U8Image.data must be a contiguous uint8 CUDA tensor")
        self.data, self.h, self.w, self.rs, self.crop_y, self.crop_x = data, int(h), int(w), (int(rs[0]), int(rs[1])), int(crop_y), int(crop_x)
        self.row_bytes = int(row_bytes) if row_bytes is not None else 3 * self.w

    def c_struct(self) -> "_lib.CsU8Image":
        return _lib.CsU8Image(C.c_void_p(self.data.data_ptr()) if self.data is not None else None, self.h, self.w, self.row_bytes,
                              self.rs[0], self.rs[1], self.crop_y, self.crop_x)


class U8Batch:
    """The images of a forward as decoded uint8 (CrossScoreNet.forward_u8 and its siblings): `images` in batch order (references item-major:
    item 0's N views, item 1's, ...), all producing the same (H, W) window; mean_std = the six T.Normalize numbers (task/predict.py:68-74)."""

    def __init__(self, images, size, mean_std=synth.IMAGENET_MEAN_STD):
        self.images, self.size = list(images), (int(size[0]), int(size[1]))
        self.mean = (C.c_float * 3)(*mean_std[:3])
        self.std = (C.c_float * 3)(*mean_std[3:])

    def __len__(self) -> int:
        return len(self.images)

    def c_array(self, lo: int, hi: int):
        arr = (_lib.CsU8Image * (hi - lo))()
        for i in range(lo, hi):
            arr[i - lo] = self.images[i].c_struct()
        return arr

    def record_stream(self, s) -> None:
        for im in self.images:
            if im.data is not None:
                im.data.record_stream(s)

    @property
    def device(self):
        for im in self.images:
            if im.data is not None:
                return im.data.device
        return None


class _Node(torch.nn.Module):
    """Parameter container; the tree of _Nodes reproduces the reference module paths."""


class CrossScoreNet(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        m = cfg.model
        # model.do_reference_cross=False (task/core.py:55,89): the reference builds no CrossReferenceNet (its state dict has no `ref_cross.*`
        # entries), runs the encoder for nothing and returns an EMPTY result dict.  Same here, minus the wasted encoder pass: the module keeps
        # the backbone / PE parameters so that such a checkpoint loads strictly, and forward() returns {} after the reference's own input checks.
        self._do_reference_cross = bool(m.do_reference_cross)
        # model/positional_encoding.py:61-69 hands the mode to F.interpolate together with align_corners=True, which torch accepts for a 4-D tensor
        # only with the interpolating modes bilinear and bicubic: every other value raises there ("align_corners option can only be set with
        # the interpolating modes ...") -- at the first forward whose patch grid differs from the PE's (pe_h, pe_w), not before: a grid that
        # matches adds the parameter as it is (positional_encoding.py:51-56) whatever the mode says.  Same here: `_check_pe_mode`.
        self._pe_mode_name = str(m.pos_enc.multi_view.interpolate_mode)
        self._pe_mode = {"bilinear": 0, "bicubic": 1}.get(self._pe_mode_name, 0)  # (an unsupported name never reaches the resize kernel)
        self.arch = arch_from_cfg(cfg)
        metric = m.predict.metric
        self._act, self._pow = regression_activation(metric.type, metric.min, metric.max, metric.power_factor)
        # parameter tree with the checkpoint's key names (all frozen for inference)
        for name, shape, kind, _ in synth.state_dict_spec(self.arch):
            if not self._do_reference_cross and name.startswith("ref_cross."):
                continue
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Node())
                node = getattr(node, p)
            if name == "img_mean_std":
                node.register_buffer(parts[-1], torch.tensor(synth.IMAGENET_MEAN_STD, dtype=torch.float32))
            else:
                node.register_parameter(parts[-1], torch.nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False))
        self._handle = None
        self._handle_device = None
        self._dirty = True
        self.enc_chunk_images = 0  # 0 = library default
        self.lanes = 0             # 0 = library default (2 concurrent lanes); 1 = serial
        # encoder position-embedding resize: "size" (installed transformers, the goldens) or "scale_factor" (the reference's pinned 4.33.3);
        # an optional key of THIS build under model.backbone (the reference has no such key): pos_embed_interpolation
        self._pos_legacy = str(cfg.model.backbone.get("pos_embed_interpolation", "size")) == "scale_factor"
        self.enc_fused = 0         # 0 = token-panel kernel per encoder layer where supported (hidden 384), 1 = unfused kernels
        self.ln_fold = 0           # 1 = fold the encoder LayerNorms into the QKV / fc1 GEMM epilogues (opt-in; slower so far)
        # 16-bit operand type of the MFMA kernels: "fp16" (default: 11 significant bits, finite to 65504) or "bf16" (8 bits, fp32's range:
        # the mode for checkpoints whose activations leave the half range; nonfinite_count() tells).  An optional key of THIS build:
        # model.backbone.operand_dtype; the predict driver derives it from the reference's own trainer.precision key.
        self.operand_dtype = str(cfg.model.backbone.get("operand_dtype", "fp16"))
        self._capture = False      # debug taps (debug_capture / debug_read): stage-level parity tests only
        self.finite_check = True   # every forward counts the non-finite values of its score map on the device (~3 us)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # -- weights ----------------------------------------------------------------------------------------------
    def _mark_dirty(self):
        self._dirty = True

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .half() / .bfloat16() replace the parameter tensors: the handle's packed copies are stale afterwards
        self._mark_dirty()
        return super()._apply(fn, *args, **kwargs)

    def load_numpy_state_dict(self, sd: Dict[str, np.ndarray]) -> None:
        self.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)

    def _operand_code(self) -> int:
        if self.operand_dtype not in ("fp16", "bf16"):
            raise ValueError(f"operand_dtype must be 'fp16' or 'bf16', got {self.operand_dtype!r}")
        return 1 if self.operand_dtype == "bf16" else 0

    @property
    def token_dtype(self) -> torch.dtype:
        """dtype of the cached reference tokens (encode_references / forward_cached): the handle's 16-bit operand type"""
        return torch.bfloat16 if self._operand_code() else torch.float16

    def nonfinite_count(self) -> int:
        """Non-finite values the score maps of this module's forwards held since the last call (waits for the last forward; resets the
        count).  With fp16 operands an activation beyond 65504 turns into inf upstream and reaches the score map as NaN: a count > 0
        says "run this checkpoint with operand_dtype = 'bf16'"."""
        if self._handle is None:
            return 0
        n = C.c_longlong(0)
        _lib.check(_lib.load().cs_nonfinite_count(self._handle, C.byref(n)))
        return int(n.value)

    def forward_stats(self) -> dict:
        """Launch census of this module's last forward (cs_forward_stats): {"launches": kernel launches, "host_enqueue_ms": wall time the call
        took on the calling thread, "kernels": {name: count}}."""
        if self._handle is None:
            return {"launches": 0, "host_enqueue_ms": 0.0, "kernels": {}}
        n, ms, buf = C.c_int(0), C.c_double(0.0), C.create_string_buffer(1024)
        _lib.check(_lib.load().cs_forward_stats(self._handle, C.byref(n), C.byref(ms), buf, 1024))
        kern = {k: int(v) for k, v in (kv.split("=") for kv in buf.value.decode().split() if "=" in kv)}
        return {"launches": int(n.value), "host_enqueue_ms": float(ms.value), "kernels": kern}

    def _check_pe_mode(self, H: int, W: int) -> None:
        """The reference's F.interpolate call (positional_encoding.py:61-69) only runs when the patch grid differs from the PE table's."""
        P = self.arch.patch
        if self._pe_mode_name not in ("bilinear", "bicubic") and (H // P, W // P) != (self.arch.pe_h, self.arch.pe_w):
            raise ValueError(f"model.pos_enc.multi_view.interpolate_mode={self._pe_mode_name!r}: align_corners option can only be "
                             "set with the interpolating modes: linear | bilinear | bicubic | trilinear (4-D input: bilinear | bicubic)")

    def _release(self):
        if self._handle is not None:
            _lib.load().cs_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_handle(self, device: torch.device):
        if not self._do_reference_cross:
            raise ValueError("model.do_reference_cross=False: this module has no cross-reference predictor (task/core.py:55) -- nothing to score or cache")
        if self._handle is not None and not self._dirty and self._handle_device == device:
            return self._handle
        lib = _lib.load()
        self._release()
        a = self.arch
        m = self.cfg.model
        cc = _lib.CsConfig(hidden=a.hidden, enc_layers=a.enc_layers, enc_heads=a.enc_heads, mlp_ratio=a.mlp_ratio, patch=a.patch,
                           pos_grid=a.pos_grid, pe_h=a.pe_h, pe_w=a.pe_w, dec_layers=a.dec_layers, dec_heads=a.dec_heads,
                           do_self_attn=int(bool(m.decoder_do_self_attn)), do_short_cut=int(bool(m.decoder_do_short_cut)),
                           act=self._act, pow_p=self._pow, enc_chunk_images=int(self.enc_chunk_images), ln_fold=int(self.ln_fold),
                           lanes=int(self.lanes), pos_interp_legacy=int(self._pos_legacy), enc_fused=int(self.enc_fused),
                           operand_dtype=self._operand_code(), pe_interp_mode=int(self._pe_mode), skip_finite_check=int(not self.finite_check), swiglu=int(bool(a.swiglu)))
        with torch.cuda.device(device):
            h = lib.cs_create(C.byref(cc))
            if not h:
                msg = _lib.last_error()
                raise NotImplementedError(msg) if "not in" in msg else ValueError(msg)
            try:
                kinds = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}
                for name, t in self.state_dict().items():
                    t = t.detach()
                    t = (t if t.dtype in kinds else t.to(torch.float32)).contiguous()  # net.half() / net.bfloat16() modules load as they are
                    shape = (C.c_int64 * t.dim())(*t.shape)
                    _lib.check(lib.cs_set_weight_typed(h, name.encode(), C.c_void_p(t.data_ptr()), int(t.is_cuda), kinds[t.dtype], t.dim(), shape))
                _lib.check(lib.cs_finalize(h))
            except Exception:
                lib.cs_destroy(h)
                raise
        self._handle, self._handle_device, self._dirty = h, device, False
        if self._capture:
            _lib.check(lib.cs_debug_capture(h, 1))
        return h

    # -- forward ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, query_img, ref_cross_imgs, need_attn_weights=False, need_attn_weights_head_id=0, norm_img=False,
                return_mean=False):
        """
        :param query_img:       (B, 3, H, W) fp32, ImageNet-normalised
        :param ref_cross_imgs:  (B, N_ref_cross, 3, H, W)
        :param norm_img:        reference flag (task/core.py:76-81), bug-for-bug: the std used is the mean.
        """
        if not self._do_reference_cross:
            # task/core.py:84-117 with the flag off: features of the query (and of the references, when given) are computed and dropped, the
            # result dict stays empty.  The input checks of get_featmaps (:119-138: a 4-D query, references that concatenate with it) still hold.
            if query_img.dim() != 4 or query_img.shape[1] != 3:
                raise ValueError("expected query_img (B,3,H,W)")
            if ref_cross_imgs is not None and (ref_cross_imgs.dim() != 5 or ref_cross_imgs.shape[0] != query_img.shape[0] or
                                               tuple(ref_cross_imgs.shape[2:]) != tuple(query_img.shape[1:])):
                raise ValueError("expected ref_cross_imgs (B,N,3,H,W) matching query_img")
            return {}
        if ref_cross_imgs is None:
            raise ValueError("ref_cross_imgs is required when model.do_reference_cross=True (task/core.py:90)")
        if query_img.dim() != 4 or query_img.shape[1] != 3 or ref_cross_imgs.dim() != 5 or ref_cross_imgs.shape[2] != 3:
            raise ValueError("expected query_img (B,3,H,W) and ref_cross_imgs (B,N,3,H,W)")
        if ref_cross_imgs.shape[0] != query_img.shape[0] or ref_cross_imgs.shape[-2:] != query_img.shape[-2:]:
            raise ValueError("query and reference batch / image sizes differ")
        if not query_img.is_cuda:
            raise _lib.CrossScoreHipError("CrossScoreNet.forward needs CUDA(HIP) tensors: the hot path has no CPU fallback")
        dev = query_img.device
        if norm_img:
            mean = self.img_mean_std.to(dev)[None, :3, None, None]
            query_img = (query_img - mean) / mean  # sic: task/core.py:77-79 divides by the mean
            ref_cross_imgs = (ref_cross_imgs - mean[:, None]) / mean[:, None]
        q = query_img.to(torch.float32).contiguous()
        r = ref_cross_imgs.to(device=dev, dtype=torch.float32).contiguous()
        B, _, H, W = q.shape
        N = r.shape[1]
        P = self.arch.patch
        h, w = H // P, W // P
        self._check_pe_mode(H, W)
        lib = _lib.load()
        handle = self._ensure_handle(dev)
        with torch.cuda.device(dev):
            score = torch.empty((B, h * P, w * P), dtype=torch.float32, device=dev)
            attn = torch.empty((B, h, w, N, h, w), dtype=torch.float32, device=dev) if need_attn_weights else None
            mean_out = torch.empty((B,), dtype=torch.float32, device=dev) if return_mean else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            for b0, b1 in self._sub_batches(B, N, h * w):
                rc = lib.cs_forward(handle, C.c_void_p(q[b0:b1].data_ptr()), C.c_void_p(r[b0:b1].data_ptr()), b1 - b0, N, H, W,
                                    C.c_void_p(score[b0:b1].data_ptr()),
                                    C.c_void_p(attn[b0:b1].data_ptr()) if attn is not None else None,
                                    int(need_attn_weights_head_id),
                                    C.c_void_p(mean_out[b0:b1].data_ptr()) if mean_out is not None else None,
                                    C.c_void_p(stream))
                _lib.check(rc)
        results = {"score_map_ref_cross": score, "attn_weights_map_ref_cross": attn}
        if return_mean:
            results["score_mean_ref_cross"] = mean_out  # key does not start with "score_map": writers ignore it
        return results

    def _sub_batches(self, B: int, N: int, Np: int):
        """Items are independent, so a batch whose decoder buffers would overflow the kernels' 32-bit element offsets
        (B*N*Np*4C >= 2^31, e.g. ViT-B with 128 items) is scored in sub-batches; results are identical either way."""
        kv_per_item = N * Np * 2 * self.arch.hidden * self.arch.dec_layers
        step = max(1, min(B, (2 ** 31 - 1) // max(kv_per_item, 1)))
        return [(b0, min(B, b0 + step)) for b0 in range(0, B, step)]

    # -- reference-feature cache: a separate mode, bit-identical results ---------------------------
    @torch.no_grad()
    def encode_references(self, ref_imgs):
        """(R,3,H,W) normalised reference images -> (R, h*w, C) fp16 decoder-ready tokens (final LN + multi-view PE).
        A reference's tokens depend neither on the query nor on its view slot, so they can be computed once per image of
        reference_dir and gathered per query."""
        if ref_imgs.dim() != 4 or ref_imgs.shape[1] != 3 or not ref_imgs.is_cuda:
            raise ValueError("expected CUDA ref_imgs (R,3,H,W)")
        dev = ref_imgs.device
        x = ref_imgs.to(torch.float32).contiguous()
        R, _, H, W = x.shape
        self._check_pe_mode(H, W)
        P = self.arch.patch
        handle = self._ensure_handle(dev)
        with torch.cuda.device(dev):
            tok = torch.empty((R, (H // P) * (W // P), self.arch.hidden), dtype=self.token_dtype, device=dev)
            _lib.check(_lib.load().cs_encode_references(handle, C.c_void_p(x.data_ptr()), R, H, W, C.c_void_p(tok.data_ptr()),
                                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return tok

    @torch.no_grad()
    def forward_cached(self, query_img, ref_tokens, need_attn_weights=False, need_attn_weights_head_id=0, return_mean=False):
        """forward() with the reference views given as cached tokens (B, N, h*w, C) fp16 from encode_references."""
        if query_img.dim() != 4 or ref_tokens.dim() != 4 or ref_tokens.shape[0] != query_img.shape[0]:
            raise ValueError("expected query_img (B,3,H,W) and ref_tokens (B,N,h*w,C)")
        if not query_img.is_cuda or not ref_tokens.is_cuda or ref_tokens.device != query_img.device:
            raise _lib.CrossScoreHipError("forward_cached needs CUDA(HIP) tensors on one device: the hot path has no CPU fallback")
        dev = query_img.device
        q = query_img.to(torch.float32).contiguous()
        if ref_tokens.dtype != self.token_dtype:
            raise ValueError(f"ref_tokens are {ref_tokens.dtype}, this module's operand type is {self.token_dtype}: encode them with the same module")
        t = ref_tokens.contiguous()
        B, _, H, W = q.shape
        N = t.shape[1]
        P = self.arch.patch
        h, w = H // P, W // P
        self._check_pe_mode(H, W)
        if t.shape[2] != h * w or t.shape[3] != self.arch.hidden:
            raise ValueError("ref_tokens do not match the query's patch grid / hidden size")
        handle = self._ensure_handle(dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            score = torch.empty((B, h * P, w * P), dtype=torch.float32, device=dev)
            attn = torch.empty((B, h, w, N, h, w), dtype=torch.float32, device=dev) if need_attn_weights else None
            mean_out = torch.empty((B,), dtype=torch.float32, device=dev) if return_mean else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            for b0, b1 in self._sub_batches(B, N, h * w):
                _lib.check(lib.cs_forward_cached(
                    handle, C.c_void_p(q[b0:b1].data_ptr()), C.c_void_p(t[b0:b1].data_ptr()), b1 - b0, N, H, W,
                    C.c_void_p(score[b0:b1].data_ptr()),
                    C.c_void_p(attn[b0:b1].data_ptr()) if attn is not None else None, int(need_attn_weights_head_id),
                    C.c_void_p(mean_out[b0:b1].data_ptr()) if mean_out is not None else None, C.c_void_p(stream)))
        results = {"score_map_ref_cross": score, "attn_weights_map_ref_cross": attn}
        if return_mean:
            results["score_mean_ref_cross"] = mean_out
        return results

    # -- the same three calls fed from decoded uint8 images (SURVEY.md 8f-4 as worded: uint8 in, tokens out) --------------------------------
    def u8_input_supported(self, img: U8Image, size, device=None) -> bool:
        """Whether forward_u8 / encode_references_u8 / forward_cached_u8 take this image geometry (else: InputStage + the fp32 calls)."""
        dev = device or (img.data.device if img.data is not None else None)
        if dev is None or not self._do_reference_cross:
            return False
        st = img.c_struct()
        return bool(_lib.load().cs_u8_input_supported(self._ensure_handle(dev), C.byref(st), int(size[0]), int(size[1])))

    def _u8_out(self, B, N, H, W, dev, need_attn_weights, return_mean):
        P = self.arch.patch
        h, w = H // P, W // P
        score = torch.empty((B, h * P, w * P), dtype=torch.float32, device=dev)
        attn = torch.empty((B, h, w, N, h, w), dtype=torch.float32, device=dev) if need_attn_weights else None
        mean_out = torch.empty((B,), dtype=torch.float32, device=dev) if return_mean else None
        return score, attn, mean_out

    @torch.no_grad()
    def forward_u8(self, query: U8Batch, refs: U8Batch, need_attn_weights=False, need_attn_weights_head_id=0, return_mean=False):
        """forward() from decoded uint8 images: query = B images, refs = B * N images (item-major).  The resize / crop / normalise of the input
        stage happens inside the patch-embedding launch; results are bit-identical to InputStage + forward()."""
        B = len(query)
        if B == 0 or len(refs) % B or query.size != refs.size:
            raise ValueError("expected B query images and B * N reference images of one processed size")
        N = len(refs) // B
        H, W = query.size
        dev = query.device
        if dev is None:
            raise _lib.CrossScoreHipError("forward_u8 needs device images: the hot path has no CPU fallback")
        self._check_pe_mode(H, W)
        lib = _lib.load()
        handle = self._ensure_handle(dev)
        P = self.arch.patch
        with torch.cuda.device(dev):
            score, attn, mean_out = self._u8_out(B, N, H, W, dev, need_attn_weights, return_mean)
            stream = torch.cuda.current_stream(dev).cuda_stream
            for b0, b1 in self._sub_batches(B, N, (H // P) * (W // P)):
                _lib.check(lib.cs_forward_u8(handle, query.c_array(b0, b1), refs.c_array(b0 * N, b1 * N), b1 - b0, N, H, W, query.mean, query.std,
                                             C.c_void_p(score[b0:b1].data_ptr()), C.c_void_p(attn[b0:b1].data_ptr()) if attn is not None else None,
                                             int(need_attn_weights_head_id),
                                             C.c_void_p(mean_out[b0:b1].data_ptr()) if mean_out is not None else None, C.c_void_p(stream)))
        results = {"score_map_ref_cross": score, "attn_weights_map_ref_cross": attn}
        if return_mean:
            results["score_mean_ref_cross"] = mean_out
        return results

    @torch.no_grad()
    def encode_references_u8(self, imgs: U8Batch):
        """encode_references() from decoded uint8 images."""
        R = len(imgs)
        H, W = imgs.size
        dev = imgs.device
        if R == 0 or dev is None:
            raise ValueError("expected at least one device image")
        self._check_pe_mode(H, W)
        P = self.arch.patch
        handle = self._ensure_handle(dev)
        with torch.cuda.device(dev):
            tok = torch.empty((R, (H // P) * (W // P), self.arch.hidden), dtype=self.token_dtype, device=dev)
            _lib.check(_lib.load().cs_encode_references_u8(handle, imgs.c_array(0, R), R, H, W, imgs.mean, imgs.std, C.c_void_p(tok.data_ptr()),
                                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return tok

    @torch.no_grad()
    def forward_cached_u8(self, query: U8Batch, ref_tokens, need_attn_weights=False, need_attn_weights_head_id=0, return_mean=False):
        """forward_cached() with the query images as decoded uint8."""
        B = len(query)
        H, W = query.size
        dev = ref_tokens.device
        if ref_tokens.dim() != 4 or ref_tokens.shape[0] != B or not ref_tokens.is_cuda or ref_tokens.dtype != self.token_dtype:
            raise ValueError("expected CUDA ref_tokens (B,N,h*w,C) of this module's operand type")
        t = ref_tokens.contiguous()
        N = t.shape[1]
        P = self.arch.patch
        if t.shape[2] != (H // P) * (W // P) or t.shape[3] != self.arch.hidden:
            raise ValueError("ref_tokens do not match the query's patch grid / hidden size")
        self._check_pe_mode(H, W)
        lib = _lib.load()
        handle = self._ensure_handle(dev)
        with torch.cuda.device(dev):
            score, attn, mean_out = self._u8_out(B, N, H, W, dev, need_attn_weights, return_mean)
            stream = torch.cuda.current_stream(dev).cuda_stream
            for b0, b1 in self._sub_batches(B, N, (H // P) * (W // P)):
                _lib.check(lib.cs_forward_cached_u8(handle, query.c_array(b0, b1), C.c_void_p(t[b0:b1].data_ptr()), b1 - b0, N, H, W, query.mean, query.std,
                                                    C.c_void_p(score[b0:b1].data_ptr()), C.c_void_p(attn[b0:b1].data_ptr()) if attn is not None else None,
                                                    int(need_attn_weights_head_id),
                                                    C.c_void_p(mean_out[b0:b1].data_ptr()) if mean_out is not None else None, C.c_void_p(stream)))
        results = {"score_map_ref_cross": score, "attn_weights_map_ref_cross": attn}
        if return_mean:
            results["score_mean_ref_cross"] = mean_out
        return results

    def calibrate_lanes(self, query_img, ref_cross_imgs, tries: int = 3, steps: int = 4, min_gain: float = 0.08) -> Dict[str, Any]:
        """Checks that this module's multi-lane forward really overlaps its lanes, and repairs it if not.  The lanes' streams are probed
        when they are drawn (cs_forward), but streams that passed the probe can still end up on one hardware queue when other queues were
        created in between: the forward then silently runs 8.3-8.7 instead of 7.2 ms per cfg-2 batch.  This times `steps`
        forwards limited to one lane, then as configured; if the configured form does not win by `min_gain`, the lane streams are drawn
        again, up to `tries` times or until a set wins.  Call once with inputs of the working shape, outside any timed region.
        Returns what it measured.  Results of forwards are bit-identical whatever the lanes."""
        import time

        if self.lanes == 1:
            return {"one_lane_s": None, "lanes_s": []}
        lib = _lib.load()
        dev = query_img.device

        def run(n):
            for _ in range(n):
                self(query_img, ref_cross_imgs, False, 0, False)
            torch.cuda.synchronize(dev)

        def timed():
            run(1)
            t0 = time.perf_counter()
            run(steps)
            return (time.perf_counter() - t0) / steps

        run(2)  # handle, workspace, tables, lane streams exist from here on
        _lib.check(lib.cs_set_lanes(self._handle, 1))
        try:
            one = timed()
        finally:
            _lib.check(lib.cs_set_lanes(self._handle, 0))
        seen = []
        for attempt in range(max(1, tries)):
            if attempt:
                _lib.check(lib.cs_redraw_lane_streams(self._handle))
            seen.append(timed())
            if seen[-1] < (1.0 - min_gain) * one:
                break
        return {"one_lane_s": one, "lanes_s": seen}

    # -- debug taps used by the stage-level parity tests (tests/test_hip_stages.py) ------------------------------------
    def debug_capture(self, on: bool = True) -> None:
        """Following forwards also keep their intermediate tensors (cs_debug_capture): not for timed runs."""
        self._capture = bool(on)
        if self._handle is not None:
            _lib.check(_lib.load().cs_debug_capture(self._handle, int(self._capture)))

    def debug_read(self, name: str) -> torch.Tensor:
        """Tap `name` of the last captured forward (include/crossscore_hip.h lists the names), in its own dtype and shape."""
        if self._handle is None:
            raise _lib.CrossScoreHipError("debug_read before any forward")
        lib = _lib.load()
        dt, nd, shp = C.c_int(), C.c_int(), (C.c_int64 * 4)()
        stream = C.c_void_p(torch.cuda.current_stream(self._handle_device).cuda_stream)
        _lib.check(lib.cs_debug_read(self._handle, name.encode(), None, 0, C.byref(dt), C.byref(nd), shp, stream))
        dtype = {_lib.DTYPE_F32: torch.float32, _lib.DTYPE_F16: torch.float16, _lib.DTYPE_BF16: torch.bfloat16}[dt.value]
        out = torch.empty(tuple(int(shp[k]) for k in range(nd.value)), dtype=dtype, device=self._handle_device)
        with torch.cuda.device(self._handle_device):
            _lib.check(lib.cs_debug_read(self._handle, name.encode(), C.c_void_p(out.data_ptr()), out.numel() * out.element_size(),
                                         C.byref(dt), C.byref(nd), shp, stream))
        return out

    # -- profiling hooks used by bench.py --------------------------------------------------------------------
    def profile_enable(self, on: bool) -> None:
        _lib.check(_lib.load().cs_profile_enable(self._handle, int(on)))

    def profile_read(self, family: int):
        ms, n, fl = C.c_double(), C.c_int(), C.c_double()
        _lib.check(_lib.load().cs_profile_read(self._handle, family, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    def profile_read_bytes(self, family: int) -> float:
        b = C.c_double()
        _lib.check(_lib.load().cs_profile_read_bytes(self._handle, family, C.byref(b)))
        return b.value


def load_lightning_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """`state_dict` of a CrossScore Lightning checkpoint with the "model." prefix removed
    (CrossScoreLightningModule.model = CrossScoreNet, task/core.py:173).  Optimizer / scheduler / callback
    state is ignored."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    return {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")} or dict(sd)
