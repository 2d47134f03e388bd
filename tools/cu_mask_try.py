"""do CU-masked streams (hipExtStreamCreateWithCUMask) make the overlap of two batches in flight deterministic -- each
replica on its own set of CUs instead of relying on the hardware queues' dispatch order?  cfg-2, two batches in flight x one encoder lane,
replica streams replaced by masked ones (wrapped as torch ExternalStreams): no mask (the product), two halves of the mask words, interleaved
mask words, 192 + 64 CUs.  Prints ms per batch and the measured fractions of time with two batches in flight."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import crossscore_amd
crossscore_amd.configure_runtime(hw_queues=8)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.pipeline import ForwardPipeline

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "facebook/dinov2-small"}))
net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1))
net = net.to(dev)
q, r = synth.make_inputs_shard(0, 8, 5, 518, 518, 1)
tq, tr = torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev)


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(name, masks):
    pipe = ForwardPipeline(net, depth=2)
    if masks is not None:
        pipe.streams = [masked_stream(m) for m in masks]
    last = None
    for _ in range(6):
        last = pipe.submit(tq, tr, False, 0, False)
    pipe.result(last); torch.cuda.synchronize()
    pipe.record_timeline(True)
    t0 = time.perf_counter()
    for _ in range(30):
        last = pipe.submit(tq, tr, False, 0, False)
    pipe.result(last); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 30
    f = pipe.in_flight_fractions()
    print(f"{name:34s} {ms:6.2f} ms / batch   two in flight {f['fraction_two_in_flight']:.2f}  one {f['fraction_one_in_flight']:.2f}  idle {f['fraction_idle']:.2f}", flush=True)


F = 0xFFFFFFFF
run("no mask (hardware queues)", None)
run("no mask (hardware queues), again", None)
run("words 0-3 | words 4-7", [[F, F, F, F, 0, 0, 0, 0], [0, 0, 0, 0, F, F, F, F]])
run("even words | odd words", [[F, 0, F, 0, F, 0, F, 0], [0, F, 0, F, 0, F, 0, F]])
run("even bits | odd bits", [[0x55555555] * 8, [0xAAAAAAAA] * 8])
run("all CUs on both (masked streams)", [[F] * 8, [F] * 8])
run("192 CUs | 64 CUs", [[F, F, F, F, F, F, 0, 0], [0, 0, 0, 0, 0, 0, F, F]])
