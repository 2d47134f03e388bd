#!/usr/bin/env python3
"""Benchmark of the CrossScore hot path on MI355X: query-images/sec of CrossScoreNet.forward on synthetic input.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run, one rank per GPU).
A step = one forward over one per-GPU batch of (query, 5 refs) items already resident in HBM.  Workload =
BASELINE.json configs[1]: ViT-S/14 encoder, 518x518, 5 refs, batch 8 per GPU (weak scaling: every rank runs its own
independent batch; no data-path collective -- items never interact).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from crossscore_amd import synth  # noqa: E402
from crossscore_amd.config import model_config  # noqa: E402
from crossscore_amd.model import CrossScoreNet  # noqa: E402
from crossscore_amd import parallel  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0      # HBM3E peak, same table
GEMM_EPI = ["BIAS_BF16", "BIAS_GELU_BF16", "BIAS_RELU_BF16", "BIAS_LEAKY_BF16", "RESID_F32", "PATCH_F32", "HEAD_SCORE",
            "LN_BF16", "LN_GELU_BF16", "RESID_F32_LN"]
WORKLOADS = {
    # name: (backbone, H, W, N refs, per-GPU batch)
    "cfg2": ("facebook/dinov2-small", 518, 518, 5, 8),
    "cfg3": ("facebook/dinov2-base", 518, 518, 10, 8),
    "cfg4": ("facebook/dinov2-base", 518, 518, 5, 16),
    "cfg5": ("facebook/dinov2-small", 1036, 1036, 5, 2),
}


def algorithmic_flops_per_query(C, L, H, W, N, P=14, dec_layers=2):
    """2 FLOPs per MAC."""
    Np = (H // P) * (W // P)
    T, Lq, Lk = Np + 1, Np, N * Np
    enc = 2 * Np * 3 * P * P * C + L * (24 * T * C * C + 4 * T * T * C)
    dec = dec_layers * (16 * Lq * C * C + 4 * Lk * C * C + 4 * Lq * Lq * C + 4 * Lq * Lk * C)
    head = 2 * Lq * C * C + 2 * Lq * C * P * P
    return (1 + N) * enc + dec + head


def kernel_table(net):
    rows = []
    for fam in list(range(10)) + [16 + d // 16 for d in (16, 48, 64, 96)] + [40, 32]:
        ms, n, fl = net.profile_read(fam)
        if n == 0:
            continue
        name = (f"cs_gemm_kernel<{GEMM_EPI[fam]}>" if fam < 10 else f"cs_attn_kernel<{(fam - 16) * 16}>" if fam < 32
                else "cs_panel_kernel" if fam == 40 else "layernorm/im2col/other")
        by = net.profile_read_bytes(fam)
        rows.append(dict(kernel=name, launches=n, total_ms=ms, avg_us=1e3 * ms / n, tflops=(fl / ms / 1e9) if fl else None, flops=fl,
                         gbytes_per_s=(by / ms / 1e6) if by else None, flop_per_byte=(fl / by) if by else None, bytes=by))
    return rows


def roofline_of(dom):
    """Roofline entry of the dominant kernel family.  Its bound follows from its arithmetic intensity (algorithmic FLOPs over
    algorithmic HBM bytes per launch, both recorded by the library next to the HIP-event timings) against the ridge point
    peak_flops / peak_bandwidth = 312.5 FLOP/B: below it the kernel is HBM-bound and `achieved` is algorithmic bytes over the
    launch duration, above it MFMA-bound and `achieved` is algorithmic FLOPs over the duration.  The other view is kept too."""
    ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    ai = dom["flop_per_byte"]
    hbm = ai is not None and ai < ridge
    r = {"bound": "hbm" if hbm else "mfma", "kernel": dom["kernel"],
         "achieved": dom["gbytes_per_s"] if hbm else dom["tflops"], "peak": PEAK_HBM_GBS if hbm else PEAK_BF16_TFLOPS,
         "unit": "GB/s" if hbm else "TFLOP/s"}
    r["frac"] = r["achieved"] / r["peak"]
    r.update({"traffic": pmc_traffic(dom["kernel"]),
              "measured": "HIP events around each launch on its stream, kernel alone on the GPU (1 lane, whole batch per chunk)",
              "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_us"],
              "flop_per_launch": dom["flops"] / dom["launches"], "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
              "flop_per_byte": ai, "ridge_flop_per_byte": ridge,
              "mfma_view": {"achieved_tflops": dom["tflops"], "frac_of_peak": dom["tflops"] / PEAK_BF16_TFLOPS,
                            "frac_of_attainable": dom["tflops"] / min(PEAK_BF16_TFLOPS, ai * PEAK_HBM_GBS / 1e3) if ai else None}})
    return r


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary (FETCH_SIZE x2-corrected + WRITE_SIZE,
    separate --pmc passes, tools/summarise_prof.py); None when no summary is committed."""
    path = os.path.join(REPO, "profiles", "r01_hbm_traffic.json")
    if not os.path.exists(path):
        return None
    tab = json.load(open(path))
    if kernel_name.startswith("cs_gemm_kernel<"):
        epi = GEMM_EPI.index(kernel_name[len("cs_gemm_kernel<"):-1])
        keys = [k for k in tab if k.startswith(f"cs_gemm_kernel<{epi},")]
    else:
        keys = [k for k in tab if k == kernel_name]
    n = sum(tab[k]["launches"] for k in keys)
    return (sum(tab[k]["hbm_bytes_per_launch"] * tab[k]["launches"] for k in keys) / n) if n else None


def cpu_baseline(arch, sd, H, W, N, hip_score_item0, seed, max_seconds=40.0):
    """Oracle (CPU fp32 restatement = 'port') timed on the host cores on a bounded sample of the same workload:
    B=1 items of the benchmark batch, 1 untimed warm-up + timed repeats while under ~max_seconds."""
    from oracle import crossscore_oracle as orc

    torch.set_num_threads(min(os.cpu_count() or 1, 32))  # pure-torch oracle: more threads than ~32 only adds contention
    Wt = orc.to_torch(sd)
    q, r = synth.make_inputs_shard(0, 1, N, H, W, seed)
    tq, tr = torch.from_numpy(q), torch.from_numpy(r)
    cfg = dict(enc_heads=arch.enc_heads)
    t0 = time.time()
    ref = orc.forward(Wt, cfg, tq, tr)["score_map_ref_cross"]  # warm-up (also the parity reference)
    warm = time.time() - t0
    n, t_sum = 0, 0.0
    while n < 1 or (t_sum + warm * (n + 1) / max(n, 1) < max_seconds and n < 3):
        t0 = time.time()
        orc.forward(Wt, cfg, tq, tr)
        t_sum += time.time() - t0
        n += 1
    mae = float((hip_score_item0.cpu() - ref[0]).abs().mean())
    return dict(value=n / t_sum, unit="query-images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} timed + 1 warm-up forward(s) of batch item 0 (B=1, {N} refs, {H}x{W}), fp32 oracle"), mae


def eager_baseline(arch, sd, tq, tr, dev, steps=3):
    """The 'PyTorch-ROCm eager reference' of the north-star target: the same fp32 restatement (plain torch ops, i.e. what the
    reference's nn.Linear / SDPA-free attention / LayerNorm sequence executes) run on the GPU, in fp32 and under bf16 autocast
    (mirrors trainer.precision=16-mixed).  Reported next to `value`; never part of it."""
    from oracle import crossscore_oracle as orc

    Wt = {k: v.to(dev) for k, v in orc.to_torch(sd).items()}
    cfg = dict(enc_heads=arch.enc_heads)
    out = {}

    def sdpa_attention(q, k, v, heads, rnd, need_weights=False):
        # what the reference actually executes on a GPU: F.scaled_dot_product_attention (HF sdpa attention and the
        # need_weights=False branch of nn.MultiheadAttention, torch functional.py:6613-6642)
        B, Lq, C = q.shape
        dh = C // heads
        qh, kh, vh = (t.view(B, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
        o = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)
        return o.transpose(1, 2).reshape(B, Lq, C), None

    explicit_attention = orc.attention
    variants = (("fp32", torch.autocast("cuda", enabled=False), explicit_attention),
                ("bf16_autocast", torch.autocast("cuda", dtype=torch.bfloat16), explicit_attention),
                ("fp32_sdpa", torch.autocast("cuda", enabled=False), sdpa_attention),
                ("bf16_autocast_sdpa", torch.autocast("cuda", dtype=torch.bfloat16), sdpa_attention))
    for name, ctx, attn_fn in variants:
        orc.attention = attn_fn
        try:
            with torch.no_grad(), ctx:
                orc.forward(Wt, cfg, tq, tr)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    orc.forward(Wt, cfg, tq, tr)
                torch.cuda.synchronize(dev)
            out[name] = dict(value=tq.shape[0] * steps / (time.perf_counter() - t0), unit="query-images/sec", batch=int(tq.shape[0]))
        except Exception as e:  # e.g. out of memory for the materialised attention at large batch
            out[name] = dict(error=str(e)[:200])
    orc.attention = explicit_attention
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernels", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--cached-refs", action="store_true", help="also time the reference-token cache mode (separate metric)")
    ap.add_argument("--eager", action="store_true", help="also time the plain-PyTorch (eager, GPU) restatement: fp32 and bf16 autocast")
    ap.add_argument("--lanes", type=int, default=0, help="internal streams of the forward (0 = library default, 2); 1 makes every "
                    "kernel run alone, which is how the per-kernel table below is measured")
    ap.add_argument("--chunk", type=int, default=0, help="encoder chunk in images (0 = library default); the per-kernel table uses "
                    "one chunk for the whole batch")
    args = ap.parse_args()

    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    backbone, H, W, N, B = WORKLOADS[args.workload]
    seed = 1
    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": backbone}))
    arch = net.arch
    sd = synth.make_state_dict(arch, seed)  # every rank builds the same replica from the seed (no broadcast needed)
    net.load_numpy_state_dict(sd)
    net = net.to(dev)
    if args.lanes > 0:
        net.lanes = args.lanes
    if args.chunk > 0:
        net.enc_chunk_images = args.chunk
    lo = rank * B  # weak scaling: rank r scores global items [r*B, (r+1)*B)
    q, r = synth.make_inputs_shard(lo, lo + B, N, H, W, seed)
    tq, tr = torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev)  # inputs resident in HBM before timing

    for _ in range(max(args.warmup, 1)):
        out = net(tq, tr, False, 0, False)
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = net(tq, tr, False, 0, False)
    torch.cuda.synchronize(dev)
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, dev)
    score = out["score_map_ref_cross"]

    # ---- per-kernel HIP-event timing (separate pass so the events do not perturb the timed region) ----
    result = None
    if rank == 0:
        # every kernel alone on the GPU: one lane, the whole batch as one encoder chunk (the timed region above overlaps two lanes,
        # where a kernel's wall duration also contains its neighbour's work); `bench.py --lanes 1 --chunk <images>` under rocprofv3
        # reproduces exactly these launches (profiles/)
        saved = (net.lanes, net.enc_chunk_images)
        net.lanes, net.enc_chunk_images = 1, B * (1 + N)
        net._mark_dirty()
        net(tq, tr, False, 0, False)  # rebuilds the handle with the new lane / chunk settings
        net.profile_enable(True)
        prof_steps = 3
        for _ in range(prof_steps):
            net(tq, tr, False, 0, False)
        rows = kernel_table(net)
        net.profile_enable(False)
        net.lanes, net.enc_chunk_images = saved
        net._mark_dirty()
        for rrow in rows:
            rrow["launches_per_step"] = rrow["launches"] // prof_steps
        dom = max((x for x in rows if x["tflops"]), key=lambda x: x["total_ms"])
        flops_q = algorithmic_flops_per_query(arch.hidden, arch.enc_layers, H, W, N)
        value = world * B * args.steps / elapsed
        result = {
            "metric": "query-images/sec at 518x518, 5 ref views, bs=8; score-map MAE vs ref",
            "value": value, "unit": "query-images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {backbone} encoder, {H}x{W}, {N} refs, batch {B} per GPU "
                                   f"(global batch {B * world}), seeded synthetic weights + N(0,1) inputs",
                       "gflop_per_query": flops_q / 1e9, "parallelism": f"batch-shard x{world} (replicas, no data-path collective)"},
            "whole_path": {"achieved_tflops": value * flops_q / 1e12 / world, "peak_tflops": PEAK_BF16_TFLOPS,
                           "frac": value * flops_q / 1e12 / world / PEAK_BF16_TFLOPS},
            "roofline": roofline_of(dom),
            "kernels": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in x.items() if k not in ("flops", "bytes")} for x in rows],
        }
        if args.kernels:
            for x in rows:
                print(x, file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            cb, mae = cpu_baseline(arch, sd, H, W, N, score[0], seed)
            result["cpu_baseline"] = cb
            result["score_map_mae"] = mae
            result["speedup_vs_cpu_baseline"] = value / cb["value"]
        if world == 1 and args.cached_refs:
            # separate mode: references pre-encoded once, queries scored against gathered tokens.  Not the
            # headline metric: the encoder FLOPs per query drop from 1+N images to 1.
            tok = net.encode_references(tr.reshape(-1, 3, H, W)).reshape(B, N, -1, arch.hidden)
            for _ in range(3):
                net.forward_cached(tq, tok)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                oc = net.forward_cached(tq, tok)
            torch.cuda.synchronize(dev)
            tc = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(5):
                net.encode_references(tr.reshape(-1, 3, H, W))
            torch.cuda.synchronize(dev)
            te = time.perf_counter() - t0
            result["cached_refs_mode"] = {"value": B * args.steps / tc, "unit": "query-images/sec (references pre-encoded)",
                                          "encode_images_per_sec": 5 * B * N / te,
                                          "bit_identical_to_full_forward": bool(torch.equal(oc["score_map_ref_cross"], score)),
                                          "encoder_images_per_query": 1}
        if world == 1 and args.eager:
            eb = eager_baseline(arch, sd, tq, tr, dev)
            result["eager_baseline"] = eb
            for k, v in eb.items():
                if "value" in v:
                    result[f"speedup_vs_eager_{k}"] = value / v["value"]
    parallel.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
