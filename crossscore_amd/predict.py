"""`task/predict.py`-compatible driver for the MI355X path.

    python -m crossscore_amd.predict trainer.ckpt_path_to_load=ckpt/CrossScore-v1.0.0.ckpt \\
        data.dataset.query_dir=<dir> data.dataset.reference_dir=<dir> [any a.b=c override of config/default_predict.yaml]

Same key tree and defaults as the reference's Hydra config (crossscore_amd/config/), same output tree:
  out_dir naming                        task/predict.py:47-65
  dataset / sampling / transforms       task/predict.py:67-108 -> crossscore_amd/data.py (GPU input stage)
  model + checkpoint                    task/predict.py:120-141, task/core.py:173 -> crossscore_amd/model.py (state_dict keys "model.*")
  per-batch writers, end-of-run CSV     task/core.py:214-225,419-441,483-484 -> crossscore_amd/writers.py
What is not reproduced: Lightning's Trainer (one process per GPU is launched with torchrun instead of DDPStrategy; ranks take
contiguous shards of the query list), DataLoader worker processes (so with `neighbour_config.deterministic=False` the random
reference choice follows numpy's global RNG seeded with `lightning.seed` in this process, not Lightning's per-worker seeds), and the
composite matplotlib "vis" figure.
"""
from __future__ import annotations

import random
import sys
import time
from datetime import datetime
from pathlib import Path
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from . import parallel, synth
from .config import load_config
from .data import (InputStage, ReferenceTokenCache, SimpleReferenceItems, decode_items, load_batch, load_batch_u8, load_query_batch,
                   load_query_batch_u8, read_image_u8)
from .model import CrossScoreNet, load_lightning_checkpoint
from .pipeline import ForwardPipeline
from .writers import BatchWriter, ScoreSummariser


def resolve_out_dir(cfg, now: Optional[str] = None) -> str:
    """task/predict.py:47-65."""
    now = now or datetime.now().strftime("%Y%m%d_%H%M%S.%f")
    if cfg.trainer.ckpt_path_to_load is None:
        log_dir, test_dir_name = Path("log") / now, "predict_empty_ckpt"
    else:
        log_dir, test_dir_name = Path(cfg.trainer.ckpt_path_to_load).parents[1], "predict"
    log_dir = log_dir / test_dir_name
    log_dir.mkdir(parents=True, exist_ok=True)
    out_dir = cfg.logger.predict.out_dir
    if out_dir is None:
        out_dir = f"{log_dir}/{now}"
    if cfg.alias != "":
        out_dir += f"_{cfg.alias}"
    return out_dir


def seed_everything(seed: int) -> None:
    """lightning.seed_everything (task/predict.py:23): python, numpy and torch global generators."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def predict(cfg, state_dict: Optional[Dict[str, torch.Tensor]] = None, now: Optional[str] = None) -> Dict[str, object]:
    """Runs the predict loop; returns {"out_dir", "files", "rows", "query_images_per_sec"}."""
    if not torch.cuda.is_available():
        raise RuntimeError("crossscore_amd.predict needs a GPU: the scoring path has no CPU fallback")
    seed_everything(int(cfg.lightning.seed))
    rank, local_rank, world = parallel.init_from_env()
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)
    cfg.logger.predict.out_dir = resolve_out_dir(cfg, now)
    Path(cfg.logger.predict.out_dir).mkdir(parents=True, exist_ok=True)

    if cfg.this_main.crop_mode not in (None, "dataset_default"):
        raise ValueError(f"crop_mode {cfg.this_main.crop_mode} not supported (task/predict.py:76-86 knows null and dataset_default)")
    stage = InputStage(device, resize_short_side=int(cfg.this_main.resize_short_side),
                       crop_size=int(cfg.data.transforms.crop_size) if cfg.this_main.crop_mode == "dataset_default" else None)
    items = SimpleReferenceItems(cfg.data.dataset.query_dir, cfg.data.dataset.reference_dir, cfg.data.neighbour_config)

    net = CrossScoreNet(cfg)
    # trainer.precision (config/default_predict.yaml:25, the reference's own key; Lightning names): "16-mixed" (default) / "16" -> IEEE
    # half MFMA operands, "bf16-mixed" / "bf16" -> bfloat16 operands (fp32's range: for checkpoints whose activations leave the half
    # range).  "32" / "32-true" have no counterpart here (the MFMA operands are 16 bits wide; everything else is fp32 already) and
    # select the more accurate of the two, half.  An explicit model.backbone.operand_dtype wins.
    if "operand_dtype" not in cfg.model.backbone:
        net.operand_dtype = "bf16" if str(cfg.trainer.precision).startswith("bf16") else "fp16"
    if state_dict is None:
        if cfg.trainer.ckpt_path_to_load is not None:
            state_dict = load_lightning_checkpoint(cfg.trainer.ckpt_path_to_load)
        else:  # the reference would start from the hub's DINOv2 weights + a random decoder; there is no hub here
            print("[crossscore_amd.predict] no checkpoint: seeded synthetic weights (scores are meaningless)", file=sys.stderr)
            state_dict = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(net.arch, int(cfg.lightning.seed)).items()}
    net.load_state_dict(state_dict, strict=True)
    net = net.to(device)

    writer = (BatchWriter(cfg, "predict", net.img_mean_std, device, workers=max(1, int(cfg.data.loader.validation.num_workers) // 2))
              if cfg.logger.predict.write.flag.batch else None)
    summariser = ScoreSummariser(cfg.model.predict.metric.type, cfg.model.predict.metric.min, cfg.logger.predict.out_dir)

    lo, hi = parallel.shard_bounds(len(items), world, rank)
    bs = int(cfg.data.loader.validation.batch_size)
    zero_ref = bool(cfg.data.dataset.zero_reference)
    # Host side of the loader: the file lists of all batches are fixed up front (sampling order = item order, like a shuffle=False
    # DataLoader), images are decoded by `num_workers` threads, and the next batch is decoded while the GPU scores this one.
    batches = [[items[i] for i in range(start, min(start + bs, hi))] for start in range(lo, hi, bs)]
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, int(cfg.data.loader.validation.num_workers)))
    prefetch = ThreadPoolExecutor(max_workers=1)
    # this_main.cache_reference_tokens (this build's key, default on): every reference image goes through the encoder once per run
    # instead of once per query that samples it; the score maps are bit-identical
    use_cache = bool(cfg.this_main.get("cache_reference_tokens", True)) and int(cfg.data.neighbour_config.cross) > 0
    # this_main.batches_in_flight (this build's key, default 3: with cached reference tokens a batch is 1/3 decoder, measured 2837 /
    # 4152 / 4473 query-images/s with 1 / 2 / 3 in flight): the batch loop keeps that many forwards queued on replicas of the
    # module (pipeline.py), so one batch's decoder runs beside the next batch's encoder; outputs are consumed depth-1 submits later,
    # in batch order, bit-identical to the one-at-a-time loop
    pipe = ForwardPipeline(net, depth=max(1, int(cfg.this_main.get("batches_in_flight", 3))))
    # this_main.fused_input_stage (this build's key, default "auto"): uint8 in, tokens out -- the resize / crop / normalise of the
    # input stage happens inside the patch-embedding launch and no processed fp32 image is written.  The writers' image_query / image_reference
    # outputs (on by default, as in the reference's config) ARE that processed image, so "auto" takes the one-pass form only when neither is
    # written, and only for geometries it holds (cs_u8_input_supported); True insists, False never.  Results are bit-identical either way.
    want_imgs = writer is not None and bool(cfg.logger.predict.write.flag.image_query or cfg.logger.predict.write.flag.image_reference)
    fused_cfg = cfg.this_main.get("fused_input_stage", "auto")
    fused_in = False
    if fused_cfg not in (False, "false", "False", 0) and len(items) > lo and not want_imgs:
        probe = read_image_u8(items[lo]["query/img"])
        rs0, crop0 = stage.geometry(*probe.shape[:2])
        from .model import U8Image
        fused_in = net.u8_input_supported(U8Image(None, probe.shape[0], probe.shape[1], rs0, crop0[0], crop0[1]), crop0[2:], device)
    if fused_cfg in (True, "true", "True", 1) and not fused_in:
        raise ValueError("this_main.fused_input_stage=True, but " + ("the writers need the processed images (logger.predict.write.flag.image_query / "
                         "image_reference)" if want_imgs else "this backbone / image geometry is not taken by the one-pass input stage"))
    cache = ReferenceTokenCache(pipe, stage, keep_images=bool(writer is not None and cfg.logger.predict.write.flag.image_reference),
                                max_images=int(cfg.this_main.get("reference_cache_max_images", 4096)), from_u8=fused_in) if use_cache else None
    cached_paths = lambda: {k[0] for k in cache.tokens} if cache is not None else ()  # noqa: E731
    pending = prefetch.submit(decode_items, batches[0], zero_ref, pool, cached_paths()) if batches else None
    need_w, head_id = bool(cfg.model.need_attn_weights), int(cfg.model.need_attn_weights_head_id)
    files, n_done = [], 0

    def consume(entry):
        ticket, batch, idx = entry
        out = pipe.result(ticket)
        summariser.update(batch, out, means=out.get("score_mean_ref_cross"))  # the per-image means the head launch left (score_summariser.py:180-192)
        if writer is not None:
            files.extend(writer.write_out(batch, out, local_rank, idx))

    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    queued = []  # batches submitted and not yet consumed, oldest first: depth of them stay in flight
    for batch_idx, its in enumerate(batches):
        decoded = pending.result()
        if cache is None:
            pending = prefetch.submit(decode_items, batches[batch_idx + 1], zero_ref, pool) if batch_idx + 1 < len(batches) else None
            batch = (load_batch_u8 if fused_in else load_batch)(its, stage, zero_ref, decoded)
            if batch_idx == 0 and len(batches) >= 16:  # a long run: make sure the batches in flight really overlap (pipeline.py)
                pipe.calibrate(batch["query/img"], batch["reference/cross/imgs"], u8=fused_in)
                if pipe.depth == 1 and not fused_in:  # one batch at a time: then it is the forward's two lanes that have to overlap (model.py)
                    net.calibrate_lanes(batch["query/img"], batch["reference/cross/imgs"])
            if fused_in:
                ticket = pipe.submit_u8(batch["query/img"], batch["reference/cross/imgs"], need_w, head_id, True)
            else:
                ticket = pipe.submit(batch["query/img"], batch["reference/cross/imgs"], need_w, head_id, False, return_mean=True)
        else:
            batch, size = (load_query_batch_u8 if fused_in else load_query_batch)(its, stage, decoded)
            tokens, ref_imgs = cache.gather([it["reference/cross/imgs"] for it in its], decoded, size, zero_ref)
            batch["reference/cross/imgs"] = ref_imgs
            # (submitted after gather so that the set of cached paths is current; decoding overlaps the forward below)
            pending = (prefetch.submit(decode_items, batches[batch_idx + 1], zero_ref, pool, cached_paths())
                       if batch_idx + 1 < len(batches) else None)
            if batch_idx == 0 and len(batches) >= 16:  # the same check for the (default) cached mode
                pipe.calibrate(batch["query/img"], tokens, cached=True, u8=fused_in)
            ticket = (pipe.submit_cached_u8 if fused_in else pipe.submit_cached)(batch["query/img"], tokens, need_w, head_id, True)
        n_done += len(batch["query/img"])
        queued.append((ticket, batch, batch_idx))
        while len(queued) >= pipe.depth:
            consume(queued.pop(0))
    while queued:
        consume(queued.pop(0))
    torch.cuda.synchronize(device)
    t_loop = time.perf_counter() - t0
    # finish the outputs and the pools first, then decide about the overflow report together with the other ranks: a rank that raised on its
    # own before the barrier would leave the others waiting in it, and the files written so far are worth keeping either way
    # A rank whose output stage raises must still take part in the collective below, or the other ranks wait in it forever: the
    # exception is kept, the failure travels with the overflow count (one max-reduction: value = count + 1e12 for a failed rank) and is
    # re-raised behind the barrier -- on the failing rank as it was, on the others as a RuntimeError naming the situation.
    failure: Optional[BaseException] = None
    try:
        if writer is not None:
            writer.finish()
        files += summariser.summarise()
    except BaseException as exc:  # noqa: BLE001 -- re-raised below, behind the collective
        failure = exc
    finally:
        prefetch.shutdown()
        pool.shutdown()
    bad = pipe.nonfinite_count()
    FAILED = 1.0e12
    worst = parallel.max_over_ranks(float(bad) + (FAILED if failure is not None else 0.0), device)  # every rank learns about ANY rank
    parallel.barrier()
    if failure is not None:
        raise failure
    if worst >= FAILED:
        raise RuntimeError("another rank failed while finishing its outputs (see its traceback); this rank's outputs are under "
                           f"{cfg.logger.predict.out_dir}")
    bad_any = int(worst)
    if bad_any:  # an activation left the range of the 16-bit operand type somewhere upstream (fp16: |x| > 65504)
        raise FloatingPointError(f"{bad} non-finite score-map values on this rank (up to {bad_any} on one rank) with {net.operand_dtype} MFMA operands: "
                                 "run with trainer.precision=bf16-mixed (model.backbone.operand_dtype=bf16); the outputs written are under "
                                 f"{cfg.logger.predict.out_dir}")
    return {"out_dir": cfg.logger.predict.out_dir, "files": files, "rows": summariser.rows,
            "input_stage": "one-pass (uint8 in, tokens out)" if fused_in else "two-launch (uint8 -> fp32 image -> tokens)",
            "query_images_per_sec": n_done / t_loop if t_loop > 0 else 0.0}  # the whole scoring loop: input stage, forwards, output stage


def main(argv: Optional[Iterable[str]] = None) -> int:
    overrides = list(sys.argv[1:] if argv is None else argv)
    from . import configure_runtime
    configure_runtime()  # hardware queues for the batches in flight; before the first HIP call (a no-op afterwards)
    cfg = load_config("default_predict", overrides)
    if cfg.data.dataset.query_dir is None or cfg.data.dataset.reference_dir is None:
        print("usage: python -m crossscore_amd.predict data.dataset.query_dir=<dir> data.dataset.reference_dir=<dir> "
              "[trainer.ckpt_path_to_load=<ckpt>] [a.b=c ...]", file=sys.stderr)
        return 2
    with torch.no_grad():
        res = predict(cfg)
    print(f"[crossscore_amd.predict] {len(res['rows'])} query images, {res['query_images_per_sec']:.1f} query-images/s through the scoring loop, "
          f"outputs under {res['out_dir']}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
