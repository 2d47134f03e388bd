"""What the chip reports while the cfg-2 pipeline runs: rocm-smi power / clocks / temperature sampled twice a second beside `bench.py` (a child
process), and the same at idle before it -- is the clock the dominant kernels run at (1.4 - 1.8 GHz instead of 2.4) the power cap's?"""
import json, os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d.get("card0", {})
        keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "temperature (sensor junction", "fclk"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}
print("idle:", json.dumps(sample()), flush=True)
p = subprocess.Popen([sys.executable, os.path.join(R, "bench.py"), "--no-eager", "--no-cpu-baseline", "--no-cfg4", "--no-more-configs", "--steps", "400"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
t0 = time.time()
while p.poll() is None and time.time() - t0 < 120:
    time.sleep(0.5)
    print(f"t={time.time() - t0:5.1f}s", json.dumps(sample()), flush=True)
out = p.communicate()[0]
try:
    b = json.loads(out.strip().splitlines()[-1])
    print("bench:", b["value"], "query-images/s")
except Exception:  # noqa: BLE001
    print("bench output:", out[-300:])
