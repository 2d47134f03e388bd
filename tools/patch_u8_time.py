"""The patch embedding of 48 images at the BASELINE geometry (540 x 720 decoded -> 518 x 690 -> 518 x 686 window, ViT-S width), two ways:
cs_op_preprocess_u8 per image (width pass, height pass: an fp32 CHW tensor written and re-read) + the one-launch patch embedding, against the
one-pass form (uint8 in, tokens out: cs_patch_fused_kernel<.., U8>).  HIP events over the launches only (the op entry points pack weights and
synchronise per call, so both forms are timed through the forward of a 2-layer ViT-S-width model with the encoder work common to both)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.data import InputStage
from crossscore_amd.model import CrossScoreNet
dev = torch.device("cuda:0")
net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-small-2l"}))
net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
rng = np.random.Generator(np.random.PCG64(0))
for (h, w) in ((540, 720), (518, 518), (1080, 1440)):
    stage = InputStage(dev, resize_short_side=518, integer_patches=True)
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(8)]
    size = stage.geometry(h, w)[1][2:]
    R_ = 48
    descs = [stage.describe(imgs[i % 8]) for i in range(R_)]
    ub = stage.batch(descs, size)
    buf = torch.empty((R_, 3) + size, device=dev)
    def two():
        for i in range(R_): stage_from_dev(i)
        return net.encode_references(buf)
    d_imgs = [d.data for d in descs]
    import ctypes as C
    from crossscore_amd import _lib
    lib = _lib.load()
    rs, crop = stage.geometry(h, w)
    scratch = torch.empty((h * rs[1] * 3,), dtype=torch.float32, device=dev)
    def stage_from_dev(i):  # the two-launch input stage without the host-to-device copy (both forms start from device uint8)
        _lib.check(lib.cs_op_preprocess_u8(C.c_void_p(d_imgs[i].data_ptr()), h, w, w * 3, rs[0], rs[1], crop[0], crop[1], crop[2], crop[3], stage._mean, stage._std,
                                           C.c_void_p(buf[i].data_ptr()), C.c_void_p(scratch.data_ptr()) if rs != (h, w) else None,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    def one():
        return net.encode_references_u8(ub)
    assert torch.equal(two(), one())
    for name, f in (("two-launch stage + patch embedding", two), ("one-pass (uint8 in, tokens out)", one)) * 2:
        for _ in range(3): f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        print(f"{h}x{w} -> {size[0]}x{size[1]}, 48 images, 2-layer encoder included: {name:38s} {1e3 * a.elapsed_time(b) / 10:8.1f} us", flush=True)
