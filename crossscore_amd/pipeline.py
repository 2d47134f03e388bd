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
    for attr in ("enc_chunk_images", "enc_fused", "ln_fold"):
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
            if net.lanes != lanes:
                net.lanes = lanes
                net._mark_dirty()
            self.nets = [net] + [_replica(net, lanes) for _ in range(depth - 1)]
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self._n = 0

    def _run(self, method: str, args, kwargs) -> Ticket:
        i = self._n % self.depth
        self._n += 1
        net = self.nets[i]
        if self.depth == 1:
            out = getattr(net, method)(*args, **kwargs)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return Ticket(out, ev, None)
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))  # inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            out = getattr(net, method)(*args, **kwargs)
            ev = torch.cuda.Event()
            ev.record(s)
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                a.record_stream(s)  # the caching allocator must not recycle an input before the replica has read it
        return Ticket(out, ev, s)

    def submit(self, query_img, ref_cross_imgs, *args, **kwargs) -> Ticket:
        """CrossScoreNet.forward on the next replica; arguments as there."""
        return self._run("forward", (query_img, ref_cross_imgs) + args, kwargs)

    def submit_cached(self, query_img, ref_tokens, *args, **kwargs) -> Ticket:
        """CrossScoreNet.forward_cached on the next replica; the tokens may come from any replica's encode_references."""
        return self._run("forward_cached", (query_img, ref_tokens) + args, kwargs)

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

    def result(self, t: Ticket) -> Dict[str, Optional[torch.Tensor]]:
        """The batch's outputs, ordered after its kernels on the CURRENT stream (no host synchronisation)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(t.done)
        if t.stream is not None:
            for v in t.out.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        return t.out
