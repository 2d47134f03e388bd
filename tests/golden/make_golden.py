#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE code itself (build container only).

Imports /root/reference's CrossScoreNet (task/core.py) with stub modules for packages that are absent
offline (lightning, wandb, omegaconf, imageio, torchvision) and with Dinov2{Config,Model}.from_pretrained
patched to build a random-init model from explicit kwargs (the HF hub is unreachable), loads the portable
synthetic state dict of crossscore_amd.synth with load_state_dict(strict=True), runs forward() in fp32 and
stores compact outputs as .npz next to this script.  Neither reference source nor bytecode is written to
the repo; only inputs' seeds and output numbers are.

    python tests/golden/make_golden.py [--only g0,g1,...]
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from crossscore_amd import synth  # noqa: E402


def _import_reference():
    from transformers import Dinov2Config, Dinov2Model  # must precede the stubs (find_spec probes)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("imageio")
    stub("wandb", Image=object, Histogram=object)
    stub("lightning", LightningModule=torch.nn.Module, seed_everything=lambda *a, **k: None)
    stub("lightning.pytorch")
    stub("lightning.pytorch.utilities", rank_zero_only=lambda f: f)
    stub("omegaconf", DictConfig=dict, OmegaConf=object, ListConfig=list)
    stub("torchvision")
    stub("torchvision.utils", make_grid=None)
    sys.path[:0] = [REF, os.path.join(REF, "task")]
    import core  # noqa

    return core, Dinov2Config, Dinov2Model


def _patch_backbone(Dinov2Config, Dinov2Model, arch: synth.ArchSpec):
    kw = dict(hidden_size=arch.hidden, num_hidden_layers=arch.enc_layers, num_attention_heads=arch.enc_heads,
              image_size=arch.pos_grid * arch.patch, patch_size=arch.patch, mlp_ratio=arch.mlp_ratio, use_swiglu_ffn=bool(arch.swiglu))
    Dinov2Config.from_pretrained = classmethod(lambda cls, *_a, **_k: cls(**kw))
    Dinov2Model.from_pretrained = classmethod(lambda cls, *_a, **_k: cls(Dinov2Config(**kw)))


def make_cfg(arch: synth.ArchSpec, **over):
    metric = NS(type=over.get("metric_type", "ssim"), min=over.get("metric_min", 0), max=1,
                power_factor=over.get("power_factor", "default"))
    model = NS(
        patch_size=arch.patch, do_reference_cross=True,
        decoder_do_self_attn=over.get("do_self_attn", True), decoder_do_short_cut=over.get("do_short_cut", True),
        need_attn_weights=False, need_attn_weights_head_id=0,
        backbone=NS(from_pretrained=arch.name),
        pos_enc=NS(multi_view=NS(interpolate_mode=over.get("pe_interpolate_mode", "bilinear"), req_grad=False, h=arch.pe_h, w=arch.pe_w)),
        predict=NS(metric=metric),
    )
    return NS(model=model)


def run_reference(core, Dinov2Config, Dinov2Model, arch, seed, B, N, H, W, need_w=False, head_id=0, **over):
    import dataclasses
    arch = dataclasses.replace(arch, do_self_attn=over.get("do_self_attn", True))
    _patch_backbone(Dinov2Config, Dinov2Model, arch)
    net = core.CrossScoreNet(make_cfg(arch, **over)).eval()
    sd = synth.make_state_dict(arch, seed)
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    q, r = synth.make_inputs(B, N, H, W, seed)
    with torch.no_grad():
        t0 = time.time()
        out = net(torch.from_numpy(q), torch.from_numpy(r), need_w, head_id, False)
        dt = time.time() - t0
    return net, out, dt


def compact(score: np.ndarray, P: int = 14, rows=(0, 7, 100)):
    """Compact summary of a (B,Hs,Ws) map: per-patch mean grid, a few full rows, stats, per-image mean."""
    B, Hs, Ws = score.shape
    h, w = Hs // P, Ws // P
    grid = score.reshape(B, h, P, w, P).mean(axis=(2, 4), dtype=np.float64).astype(np.float32)
    rr = [r for r in rows if r < Hs] + [Hs - 1]
    return dict(patch_mean=grid, rows_idx=np.asarray(rr), rows=score[:, rr, :].copy(),
                mean=score.mean(axis=(1, 2), dtype=np.float64), min=score.min(axis=(1, 2)),
                max=score.max(axis=(1, 2)), shape=np.asarray(score.shape))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    core, DC, DM = _import_reference()
    torch.set_num_threads(os.cpu_count() or 8)
    tiny = synth.BACKBONES["synthetic/dinov2-tiny"]
    small = synth.BACKBONES["facebook/dinov2-small"]
    base = synth.BACKBONES["facebook/dinov2-base"]

    def want(n):
        return not only or n in only

    # G0: tiny, all intermediates, non-square (70x84 -> 5x6 patches; encoder bicubic 5x5->5x6, PE 40x40->5x6),
    # attention weights of head 3.
    if want("g0"):
        import dataclasses
        _patch_backbone(DC, DM, tiny)
        net = core.CrossScoreNet(make_cfg(tiny)).eval()
        sd = synth.make_state_dict(tiny, 3)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        q, r = synth.make_inputs(2, 2, 75, 90, 3)  # 75%14=5, 90%14=6 -> floor-drop exercised
        tq, tr = torch.from_numpy(q), torch.from_numpy(r)
        taps = {}
        hooks = []
        bb = net.backbone
        hooks.append(bb.embeddings.patch_embeddings.register_forward_hook(lambda m, i, o: taps.__setitem__("patch_embed", o.detach().numpy().copy())))
        hooks.append(bb.embeddings.register_forward_hook(lambda m, i, o: taps.__setitem__("embeddings", o.detach().numpy().copy())))
        for li, layer in enumerate(bb.encoder.layer):
            hooks.append(layer.register_forward_hook(
                lambda m, i, o, li=li: taps.__setitem__(f"enc_layer_{li}", (o[0] if isinstance(o, tuple) else o).detach().numpy().copy())))
        for li, layer in enumerate(net.ref_cross.attn.layers):
            hooks.append(layer.register_forward_hook(lambda m, i, o, li=li: taps.__setitem__(f"dec{li}_out", o[0].detach().numpy().copy())))
        hooks.append(net.ref_cross.head[2].register_forward_hook(lambda m, i, o: taps.__setitem__("head_pre_activation", o.detach().numpy().copy())))
        with torch.no_grad():
            fm = net.get_featmaps(tq, tr)
            taps["featmap_query"] = net.pos_enc_fn(fm["query"], N_view=1, img_h=75, img_w=90).numpy().copy()
            taps["featmap_ref"] = net.pos_enc_fn(fm["ref_cross"], N_view=2, img_h=75, img_w=90).numpy().copy()
            taps["last_hidden_state"] = bb(torch.cat([tq.view(2, 1, 3, 75, 90), tr], 1).view(6, 3, 75, 90)).last_hidden_state.numpy().copy()
            out = net(tq, tr, True, 3, False)
            out_nw = net(tq, tr, False, 0, False)
        for hk in hooks:
            hk.remove()
        np.savez_compressed(os.path.join(HERE, "g0_tiny_all.npz"), seed=3, B=2, N=2, H=75, W=90,
                            score=out["score_map_ref_cross"].numpy(), attn_head3=out["attn_weights_map_ref_cross"].numpy(),
                            score_no_weights=out_nw["score_map_ref_cross"].numpy(), **taps)
        print("g0 done", out["score_map_ref_cross"].shape, out["attn_weights_map_ref_cross"].shape)

    # G5: flag variants on the tiny net (square 70x70 = native 5x5 grid: no encoder interpolation).
    if want("g5"):
        res = {}
        variants = {
            "no_self_attn": dict(do_self_attn=False),
            "no_short_cut": dict(do_short_cut=False),
            "tanh": dict(metric_type="ssim", metric_min=-1),
            "mae_pow2": dict(metric_type="mae"),
            "mse_pow4": dict(metric_type="mse"),
            "scalar_p": dict(power_factor=0.5),
        }
        for name, over in variants.items():
            _, out, _ = run_reference(core, DC, DM, tiny, 5, 1, 3, 70, 70, **over)
            res[name] = out["score_map_ref_cross"].numpy()
        np.savez_compressed(os.path.join(HERE, "g5_tiny_flags.npz"), seed=5, B=1, N=3, H=70, W=70, **res)
        print("g5 done")

    # G6: the encoder position-embedding resize of the reference's PINNED transformers 4.33.3 (environment.yaml:340), which the installed
    # 5.x no longer runs: F.interpolate(scale_factor=((h + 0.1) / G, (w + 0.1) / G), mode="bicubic", align_corners=False).  The call
    # is restated here from the published 4.33.3 source and EXECUTED by the installed torch, (a) on the synthetic position tables of
    # the three backbones at the grids the path sees, (b) end to end: the imported reference model with its embeddings'
    # interpolate_pos_encoding replaced by that call (tiny net, 75 x 90 -> 5 x 6 patches, so the branch is live).
    if want("g6"):
        import math

        def legacy_table(pos, h, w):  # pos (1, 1 + G*G, C) -> (1 + h*w, C)
            G2 = pos.shape[1] - 1
            G = int(math.sqrt(G2))
            dim = pos.shape[-1]
            cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
            patch_pos = patch_pos.reshape(1, G, G, dim).permute(0, 3, 1, 2)
            patch_pos = torch.nn.functional.interpolate(patch_pos, scale_factor=((h + 0.1) / math.sqrt(G2), (w + 0.1) / math.sqrt(G2)),
                                                        mode="bicubic", align_corners=False)
            assert patch_pos.shape[-2:] == (h, w), patch_pos.shape
            return torch.cat([cls_pos.unsqueeze(0), patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)[0]

        res = {}
        for tag, arch, seed, grids in (("tiny", tiny, 3, [(5, 6), (7, 4)]), ("small", small, 4, [(37, 49), (74, 74), (20, 31)]), ("base", base, 2, [(37, 49)])):
            pos = torch.from_numpy(synth.make_state_dict(arch, seed)["backbone.embeddings.position_embeddings"])
            for (h, w) in grids:
                tab = legacy_table(pos, h, w).numpy()
                # compact form for the big tables: every 41st position row in full + the mean over channels of every position
                rows = np.arange(0, tab.shape[0], 1 if tab.shape[0] <= 64 else 41)
                res[f"table_{tag}_{h}x{w}_rows_idx"] = rows
                res[f"table_{tag}_{h}x{w}_rows"] = tab[rows]
                res[f"table_{tag}_{h}x{w}_chmean"] = tab.mean(axis=1, dtype=np.float64).astype(np.float32)
                res[f"table_{tag}_{h}x{w}_seed"] = np.asarray(seed)
        _patch_backbone(DC, DM, tiny)
        net = core.CrossScoreNet(make_cfg(tiny)).eval()
        sd = synth.make_state_dict(tiny, 31)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        emb = net.backbone.embeddings
        P = tiny.patch
        emb.interpolate_pos_encoding = lambda embeddings, height, width: legacy_table(emb.position_embeddings, height // P, width // P)[None]
        q, r = synth.make_inputs(1, 2, 75, 90, 6)
        with torch.no_grad():
            out = net(torch.from_numpy(q), torch.from_numpy(r), False, 0, False)
            lhs = net.backbone(torch.from_numpy(np.concatenate([q[:, None], r], 1).reshape(3, 3, 75, 90))).last_hidden_state
        np.savez_compressed(os.path.join(HERE, "g6_pos_legacy.npz"), seed=31, input_seed=6, B=1, N=2, H=75, W=90,
                            score=out["score_map_ref_cross"].numpy(), last_hidden_state=lhs.numpy(), **res)
        print("g6 done", out["score_map_ref_cross"].shape, sorted(k for k in res if k.endswith("_rows")))

    # G7: model.pos_enc.multi_view.interpolate_mode = bicubic (positional_encoding.py:61-69 passes the key to F.interpolate, align_corners=True):
    # tiny net at 75x90 (5x6 patch grid != 40x40: the resize is live), the reference's score map and its featmaps behind the PE.
    if want("g7"):
        _patch_backbone(DC, DM, tiny)
        net = core.CrossScoreNet(make_cfg(tiny, pe_interpolate_mode="bicubic")).eval()
        sd = synth.make_state_dict(tiny, 13)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        q, r = synth.make_inputs(2, 2, 75, 90, 13)
        taps = {}
        hk = net.ref_cross.attn.layers[0].register_forward_pre_hook(
            lambda m, a, k: (taps.__setitem__("featmap_query", (a[0] if a else k["tgt"]).detach().numpy().copy()),
                             taps.__setitem__("featmap_ref", (a[1] if len(a) > 1 else k["memory"]).detach().numpy().copy()))[0] and None,
            with_kwargs=True)
        with torch.no_grad():
            out = net(torch.from_numpy(q), torch.from_numpy(r), False, 0, False)
        hk.remove()
        np.savez_compressed(os.path.join(HERE, "g7_tiny_pe_bicubic.npz"), seed=13, B=2, N=2, H=75, W=90,
                            score=out["score_map_ref_cross"].numpy(), **taps)
        print("g7 done", out["score_map_ref_cross"].shape, {k: v.shape for k, v in taps.items()})

    # G8: the ViT-S WIDTH (C = 384, 6 encoder heads of 64, decoder heads of 48) with two encoder layers, every module output of the reference:
    # at this width the HIP path runs its production kernels (token-panel kernel, 256-tile GEMM at >= 256 rows, row-complete linear +
    # LayerNorm, one-launch patch embedding, attention at dh 64 / 48), which g0's C = 128 net never reaches.  B = 2, N = 3, 98 x 112 ->
    # 7 x 8 patches: 8 images x 57 tokens = 456 encoder rows, 336 memory rows; encoder table 5 x 5 -> 7 x 8 bicubic, PE 40 x 40 -> 7 x 8.
    if want("g8"):
        s2 = synth.BACKBONES["synthetic/dinov2-small-2l"]
        _patch_backbone(DC, DM, s2)
        net = core.CrossScoreNet(make_cfg(s2)).eval()
        sd = synth.make_state_dict(s2, 8)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        Bq, Nq, Hq, Wq = 2, 3, 98, 112
        q, r = synth.make_inputs(Bq, Nq, Hq, Wq, 8)
        tq, tr = torch.from_numpy(q), torch.from_numpy(r)
        taps = {}
        hooks = []
        bb = net.backbone
        hooks.append(bb.embeddings.register_forward_hook(lambda m, i, o: taps.__setitem__("embeddings", o.detach().numpy().copy())))
        for li, layer in enumerate(bb.encoder.layer):
            hooks.append(layer.register_forward_hook(
                lambda m, i, o, li=li: taps.__setitem__(f"enc_layer_{li}", (o[0] if isinstance(o, tuple) else o).detach().numpy().copy())))
        for li, layer in enumerate(net.ref_cross.attn.layers):
            hooks.append(layer.register_forward_hook(lambda m, i, o, li=li: taps.__setitem__(f"dec{li}_out", o[0].detach().numpy().copy())))
        hooks.append(net.ref_cross.head[2].register_forward_hook(lambda m, i, o: taps.__setitem__("head_pre_activation", o.detach().numpy().copy())))
        hooks.append(net.ref_cross.attn.layers[0].register_forward_pre_hook(
            lambda m, a, k: (taps.__setitem__("featmap_query", (a[0] if a else k["tgt"]).detach().numpy().copy()),
                             taps.__setitem__("featmap_ref", (a[1] if len(a) > 1 else k["memory"]).detach().numpy().copy()))[0] and None,
            with_kwargs=True))
        with torch.no_grad():
            out = net(tq, tr, True, 3, False)
        for hk in hooks:
            hk.remove()
        np.savez_compressed(os.path.join(HERE, "g8_vits_width_all.npz"), seed=8, B=Bq, N=Nq, H=Hq, W=Wq,
                            score=out["score_map_ref_cross"].numpy(), attn_head3=out["attn_weights_map_ref_cross"].numpy(), **taps)
        print("g8 done", out["score_map_ref_cross"].shape, {k: v.shape for k, v in taps.items()})

    # G9 (round 6): the SwiGLU MLP of facebook/dinov2-giant (HF Dinov2SwiGLUFFN, use_swiglu_ffn) -- the giant's width (1536, decoder heads of 192
    # channels) and the base width, two encoder layers each; B = 2, N = 2, 98 x 112; the last encoder layer's output of image 0 beside the map
    if want("g9"):
        for tag, key, seed in (("g9_swiglu_base_width", "synthetic/dinov2-swiglu-2l", 9), ("g9_swiglu_giant_width", "synthetic/dinov2-giant-2l", 19)):
            a9 = synth.BACKBONES[key]
            _patch_backbone(DC, DM, a9)
            net = core.CrossScoreNet(make_cfg(a9)).eval()
            assert type(net.backbone.encoder.layer[0].mlp).__name__ == "Dinov2SwiGLUFFN"
            net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(a9, seed).items()}, strict=True)
            q, r = synth.make_inputs(2, 2, 98, 112, seed)
            taps = {}
            hk = net.backbone.encoder.layer[-1].register_forward_hook(
                lambda m, i, o: taps.__setitem__("enc_last_img0", (o[0] if isinstance(o, tuple) else o)[0].detach().numpy().copy()))
            with torch.no_grad():
                out = net(torch.from_numpy(q), torch.from_numpy(r), False, 0, False)
            hk.remove()
            np.savez_compressed(os.path.join(HERE, tag + ".npz"), seed=seed, B=2, N=2, H=98, W=112, score=out["score_map_ref_cross"].numpy(), **taps)
            print(tag, "done", out["score_map_ref_cross"].shape, taps["enc_last_img0"].shape)

    def big(name, arch, seed, B, N, H, W):
        _, out, dt = run_reference(core, DC, DM, arch, seed, B, N, H, W)
        s = out["score_map_ref_cross"].numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, B=B, N=N, H=H, W=W,
                            ref_seconds=dt, threads=torch.get_num_threads(), **compact(s))
        print(name, "done", s.shape, f"{dt:.2f}s", float(s.mean()), float(s.min()), float(s.max()))

    if want("g1"):
        big("g1_vits_518_n5", small, 1, 1, 5, 518, 518)
    if want("g4"):
        big("g4_vits_518x690_n2", small, 4, 1, 2, 518, 690)
    if want("g2"):
        big("g2_vitb_518_n10", base, 2, 1, 10, 518, 518)
    if want("g3"):
        big("g3_vits_1036_n5", small, 6, 1, 5, 1036, 1036)


if __name__ == "__main__":
    main()
