import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import hip_helpers as hh
x = torch.randn(48, 3, 518, 518, device="cuda")
for _ in range(3): o = hh.im2col(x, 14, 640)
torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
for _ in range(20): o = hh.im2col(x, 14, 640)
e1.record(); torch.cuda.synchronize(); print("im2col 48x518x518: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
