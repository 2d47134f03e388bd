"""Batches in flight: the predict loop (task/predict.py:119-135 -> trainer.predict over a DataLoader) scores independent batches one
after another.  Inside one forward the decoder phase (small GEMMs, one attention launch at a time) cannot fill 256 CUs and the next
batch's encoder cannot start before it ends, because one stream orders them.  ForwardPipeline keeps `depth` replicas of a
CrossScoreNet -- same parameters (aliased, not copied), own C-ABI handle, own workspace, own stream -- and feeds them round-robin, so
batch i+1's encoder runs beside batch i's decoder.  Every batch is still computed by exactly the same launches: score maps are
bit-identical to the one-at-a-time forward (tests/test_hip_forward.py).

    pipe = ForwardPipeline(net, depth=2)
    prev = None
    for batch in batches:
        t = pipe.submit(batch.query, batch.refs)      # returns at once; work is queued on the replica's stream
        if prev is not None:
            consume(pipe.result(prev))                # makes the current stream wait for that batch only
        prev = t
    consume(pipe.result(prev))
"""
from typing import Any, Dict, List, NamedTuple, Optional

import torch

from .model import CrossScoreNet


class Ticket(NamedTuple):
    out: Dict[str, Optional[torch.Tensor]]
    done: Any  # torch.cuda.Event
    stream: Any


def _replica(net: CrossScoreNet, lanes: int) -> CrossScoreNet:
    """A second module over the SAME parameter tensors (no copy); its handle packs its own fp16 images at first use."""
    rep = CrossScoreNet(net.cfg)
    src = dict(net.named_parameters())
    src.update(dict(net.named_buffers()))
    for name, p in list(rep.named_parameters()) + list(rep.named_buffers()):
        p.data = src[name].data
    for attr in ("enc_chunk_images", "enc_fused", "ln_fold", "operand_dtype", "finite_check"):
        setattr(rep, attr, getattr(net, attr))
    rep.lanes = lanes
    rep._mark_dirty()
    return rep


class ForwardPipeline:
    def __init__(self, net: CrossScoreNet, depth: int = 2, lanes: Optional[int] = None):
        """depth 1 is the plain forward on the caller's stream.  With depth > 1 each replica runs ONE encoder lane by default: the
        batches in flight supply the kernel-level concurrency the two lanes of a single forward otherwise provide (measured: 2 in
        flight x 1 lane beats 1 x 2 and 2 x 2)."""
        if depth < 1:
            raise ValueError("ForwardPipeline: depth must be >= 1")
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("ForwardPipeline needs the module on a GPU (net.to('cuda')): the scoring path has no CPU fallback")
        self.depth = depth
        self.device = dev
        if depth == 1:
            self.nets: List[CrossScoreNet] = [net]
            self.streams: List[Any] = [None]
        else:
            lanes = 1 if lanes is None else lanes
            # the caller's module is never re-configured: when it runs another lane count than the pipeline's replicas, replica 0 is a
            # module of its own over the same parameters (a plain net(...) call afterwards still runs the caller's two lanes)
            first = net if net.lanes == lanes else _replica(net, lanes)
            self.nets = [first] + [_replica(net, lanes) for _ in range(depth - 1)]
            self.streams = self._overlapping_streams(depth)
            # new weights loaded into `net` land in the shared tensors: the replicas must re-pack their fp16 images too
            # (build the pipeline after net.to(device); moving the module afterwards would break the sharing)
            reps = [r for r in self.nets if r is not net]

            def _replicas_dirty(module, incompatible):
                for r in reps:
                    r._mark_dirty()

            net.register_load_state_dict_post_hook(_replicas_dirty)
        self._n = 0
        self._marks = None  # record_timeline(): [(replica, start event, end event)] of every submit

    def last_replica(self) -> CrossScoreNet:
        """The module that ran (or is running) the most recently submitted batch -- e.g. to read its forward_stats()."""
        return self.nets[(self._n - 1) % self.depth] if self._n else self.nets[0]

    def _overlapping_streams(self, n: int) -> List[Any]:
        """n streams on which kernels really run side by side.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
        queues and two streams that land on one queue serialise (the replicas would then run one after another: measured 9.0
        instead of 6.7 ms per cfg-2 batch); which streams collide depends on what else the process created before.  Each new
        stream is probed against the ones already chosen (cs_op_streams_overlap: a large idle grid beside one idle wave) and replaced until it
        overlaps with all of them (8 candidates per stream, 24 probes in all; a failing probe accepts the candidate)."""
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        with torch.cuda.device(self.device):
            chosen = [torch.cuda.Stream(device=self.device)]
            budget = 24  # torch hands streams out of a round-robin pool of 32 per device: stay below one lap
            while len(chosen) < n:
                cand = torch.cuda.Stream(device=self.device)
                for _ in range(8):
                    budget -= 1
                    ok = True
                    for c in chosen:
                        if cand.cuda_stream == c.cuda_stream:  # the pool wrapped around: the same stream cannot run beside itself
                            ok = False
                            break
                        flag = C.c_int(0)
                        try:
                            _lib.check(lib.cs_op_streams_overlap(C.c_void_p(c.cuda_stream), C.c_void_p(cand.cuda_stream), C.byref(flag)))
                        except Exception:  # a set-up heuristic must never take the caller down: accept the candidate
                            flag.value = 1
                        ok = ok and bool(flag.value)
                    if ok or budget <= 0:
                        break
                    cand = torch.cuda.Stream(device=self.device)
                if any(cand.cuda_stream == c.cuda_stream for c in chosen):  # no distinct stream left: a fresh pool entry, unprobed
                    cand = torch.cuda.Stream(device=self.device)
                chosen.append(cand)
        return chosen

    def calibrate(self, query_img, ref_cross_imgs, tries: int = 4, steps: int = 6, cached: bool = False, u8: bool = False) -> Dict[str, Any]:
        """The HIP runtime multiplexes streams onto a few hardware queues, and two streams that land on one queue serialise: the
        replicas then run one after another (measured 9.0 instead of 6.7 ms per cfg-2 batch); which streams collide depends on
        what else the process created before.  This times `steps` batches one at a time on replica 0, then in flight on up to
        `tries` fresh sets of streams, and keeps the first set that beats the serial time by 15 % (else the fastest seen).  Call
        it once, outside any timed region, with inputs of the working shape; returns what it measured.  cached=True: the second argument
        holds reference TOKENS (submit_cached: the predict driver's default mode); u8=True: the images are model.U8Batch objects (the *_u8 calls)."""
        if self.depth == 1:
            return {"serial_s": None, "in_flight_s": []}
        import time

        def in_flight(n):
            last = None
            for _ in range(n):
                if u8:
                    last = self.submit_cached_u8(query_img, ref_cross_imgs) if cached else self.submit_u8(query_img, ref_cross_imgs)
                else:
                    last = self.submit_cached(query_img, ref_cross_imgs) if cached else self.submit(query_img, ref_cross_imgs, False, 0, False)
            self.result(last)
            torch.cuda.synchronize(self.device)

        in_flight(2 * self.depth)  # handles, workspaces and tables exist from here on
        t0 = time.perf_counter()
        with torch.cuda.stream(self.streams[0]):
            for _ in range(steps):
                if u8:
                    (self.nets[0].forward_cached_u8 if cached else self.nets[0].forward_u8)(query_img, ref_cross_imgs)
                elif cached:
                    self.nets[0].forward_cached(query_img, ref_cross_imgs)
                else:
                    self.nets[0](query_img, ref_cross_imgs, False, 0, False)
        torch.cuda.synchronize(self.device)
        serial = (time.perf_counter() - t0) / steps
        seen, best = [], None
        for attempt in range(max(1, tries)):
            if attempt:
                self.streams = self._overlapping_streams(self.depth)
                in_flight(self.depth)
            t0 = time.perf_counter()
            in_flight(steps)
            dt = (time.perf_counter() - t0) / steps
            seen.append(dt)
            if best is None or dt < best[0]:
                best = (dt, self.streams)
            if dt < 0.85 * serial:
                break
        self.streams = best[1]
        return {"serial_s": serial, "in_flight_s": seen}

    def record_timeline(self, on: bool = True) -> None:
        """From now on every submit is bracketed by two timing events on its replica's stream (in-run evidence of how many batches are in
        flight: in_flight_fractions()).  Two event records per forward; off by default."""
        self._marks = [] if on else None

    def in_flight_fractions(self) -> Dict[str, Any]:
        """Fractions of the recorded window (first start .. last end) during which 0 / 1 / >= 2 forwards were executing, from the HIP events of
        record_timeline() (device time, on the streams the forwards ran on).  Waits for the recorded work."""
        if not self._marks:
            return {}
        torch.cuda.synchronize(self.device)
        t0 = self._marks[0][1]
        iv = sorted((t0.elapsed_time(a), t0.elapsed_time(b)) for _, a, b in self._marks)  # ms since the first start
        pts = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        lo, hi = min(a for a, _ in iv), max(b for _, b in iv)
        acc, depth, last = [0.0, 0.0, 0.0], 0, lo
        for t, d in pts:
            acc[min(depth, 2)] += t - last
            last, depth = t, depth + d
        span = max(hi - lo, 1e-9)
        n = len(self._marks)
        self._marks = []
        return {"forwards": n, "window_ms": span, "fraction_two_in_flight": acc[2] / span, "fraction_one_in_flight": acc[1] / span,
                "fraction_idle": acc[0] / span,
                "measured": "HIP events at the start and the end of every forward of the window, recorded on the stream the forward ran on"}

    def _run(self, method: str, args, kwargs) -> Ticket:
        i = self._n % self.depth
        self._n += 1
        net = self.nets[i]
        if self.depth == 1:
            if self._marks is not None:
                a = torch.cuda.Event(enable_timing=True)
                a.record(torch.cuda.current_stream(self.device))
            out = getattr(net, method)(*args, **kwargs)
            ev = torch.cuda.Event(enable_timing=self._marks is not None)
            ev.record(torch.cuda.current_stream(self.device))
            if self._marks is not None:
                self._marks.append((0, a, ev))
            return Ticket(out, ev, None)
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))  # inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            if self._marks is not None:
                a = torch.cuda.Event(enable_timing=True)
                a.record(s)
            out = getattr(net, method)(*args, **kwargs)
            ev = torch.cuda.Event(enable_timing=self._marks is not None)
            ev.record(s)
            if self._marks is not None:
                self._marks.append((i, a, ev))
        for a in list(args) + list(kwargs.values()):
            if (isinstance(a, torch.Tensor) and a.is_cuda) or hasattr(a, "images"):  # (model.U8Batch: the decoded images of the one-pass input stage)
                a.record_stream(s)  # the caching allocator must not recycle an input before the replica has read it
        return Ticket(out, ev, s)

    def submit(self, query_img, ref_cross_imgs, *args, **kwargs) -> Ticket:
        """CrossScoreNet.forward on the next replica; arguments as there."""
        return self._run("forward", (query_img, ref_cross_imgs) + args, kwargs)

    def submit_cached(self, query_img, ref_tokens, *args, **kwargs) -> Ticket:
        """CrossScoreNet.forward_cached on the next replica; the tokens may come from any replica's encode_references."""
        return self._run("forward_cached", (query_img, ref_tokens) + args, kwargs)

    def submit_u8(self, query, refs, *args, **kwargs) -> Ticket:
        """CrossScoreNet.forward_u8 on the next replica (query / refs: model.U8Batch)."""
        return self._run("forward_u8", (query, refs) + args, kwargs)

    def submit_cached_u8(self, query, ref_tokens, *args, **kwargs) -> Ticket:
        """CrossScoreNet.forward_cached_u8 on the next replica."""
        return self._run("forward_cached_u8", (query, ref_tokens) + args, kwargs)

    def encode_references_u8(self, imgs) -> torch.Tensor:
        """CrossScoreNet.encode_references_u8 through replica 0 (see encode_references)."""
        net = self.nets[0]
        if self.depth == 1:
            return net.encode_references_u8(imgs)
        cur, s = torch.cuda.current_stream(self.device), self.streams[0]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            tok = net.encode_references_u8(imgs)
        cur.wait_stream(s)
        imgs.record_stream(s)
        tok.record_stream(cur)
        return tok

    def encode_references(self, ref_imgs: torch.Tensor) -> torch.Tensor:
        """CrossScoreNet.encode_references through replica 0, ordered with that replica's own forwards (a handle's workspace serves
        one stream at a time); the tokens are ready on the current stream when this returns."""
        net = self.nets[0]
        if self.depth == 1:
            return net.encode_references(ref_imgs)
        cur, s = torch.cuda.current_stream(self.device), self.streams[0]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            tok = net.encode_references(ref_imgs)
        cur.wait_stream(s)
        ref_imgs.record_stream(s)
        tok.record_stream(cur)
        return tok

    def nonfinite_count(self) -> int:
        """Sum of CrossScoreNet.nonfinite_count() over the replicas (waits for their last forwards)."""
        return sum(n.nonfinite_count() for n in self.nets)

    def result(self, t: Ticket) -> Dict[str, Optional[torch.Tensor]]:
        """The batch's outputs, ordered after its kernels on the CURRENT stream (no host synchronisation)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(t.done)
        if t.stream is not None:
            for v in t.out.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        return t.out
