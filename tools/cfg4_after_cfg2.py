"""Why the cfg-4 leg of the default bench line ran slower than a dedicated cfg-4 run: times cfg-4 alone, after a cfg-2 workload that stays
alive, and after one that was released."""
import gc, os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
dev = torch.device("cuda:0")
sync = lambda: torch.cuda.synchronize(dev)
def run(name, steps=8, warm=3, inflight=2):
    w = bench.Workload(name, 0, dev, inflight=inflight).start_pipeline()
    e, _ = bench.timed_steps(w.step, sync, steps, warm, dev)
    return w, 1e3 * e / steps
mode = sys.argv[1]
if mode == "alone":
    _, ms = run("cfg4"); print("cfg4 alone", round(ms, 2))
elif mode == "after_alive":
    w2, ms2 = run("cfg2", 20); print("cfg2", round(ms2, 2))
    _, ms = run("cfg4"); print("cfg4 after cfg2 (alive)", round(ms, 2))
elif mode == "after_released":
    w2, ms2 = run("cfg2", 20); print("cfg2", round(ms2, 2))
    del w2; gc.collect(); torch.cuda.empty_cache()
    _, ms = run("cfg4"); print("cfg4 after cfg2 (released)", round(ms, 2))
