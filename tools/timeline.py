"""Concurrency timeline of one bench run from a rocprofv3 --kernel-trace CSV: how much of the steady-state time has 0 / 1 / >=2
kernels in flight, and which kernels run while nothing else does.  usage: timeline.py <dir with *kernel_trace.csv> [summary.json]"""
import collections, csv, glob, json, os, sys
fn = glob.glob(os.path.join(sys.argv[1], "**/*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state: the longest run of back-to-back forwards (one regression-head GEMM each, less than 1.6x the median spacing apart) -- the
# timed region of the bench; calibration passes, the one-batch-at-a-time leg and the per-kernel pass are other, shorter runs
heads = sorted(e for s, e, n in ev if "cs_gemm_kernel<6," in n or "cs_gemm_kernel<(CsEpilogue)6" in n)
if len(heads) < 8: raise SystemExit(f"only {len(heads)} forwards in the trace")
gaps = [b - a for a, b in zip(heads, heads[1:])]
med = sorted(gaps)[len(gaps) // 2]
best, cur = (0, 0), 0
for i, g in enumerate(gaps + [10 ** 18]):
    if g > 1.6 * med:
        if i - cur > best[1] - best[0]: best = (cur, i)
        cur = i + 1
lo, hi = heads[best[0] + 1], heads[best[1] - 1]
steps = best[1] - best[0] - 2
pts = []
for s, e, n in ev:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2: pts += [(s2, 1, n), (e2, -1, n)]
pts.sort()
depth, last, hist, solo = 0, lo, collections.Counter(), collections.Counter()
active = collections.Counter()
for t, d, n in pts:
    hist[min(depth, 3)] += t - last
    if depth == 1:
        k = next(iter(k for k, v in active.items() if v > 0))
        solo[k.replace("void (anonymous namespace)::", "").split("(")[0][:40]] += t - last
    last = t
    depth += d; active[n] += d
hist[min(depth, 3)] += hi - last
tot = sum(hist.values())
print("%d steps, %.2f ms per step" % (steps, (hi - lo) / 1e6 / steps))
print("steady-state window %.1f ms: idle %.1f %%, one kernel %.1f %%, two %.1f %%, three+ %.1f %%" % (tot / 1e6, *(100 * hist[i] / tot for i in range(4))))
print("time with exactly one kernel in flight, by kernel:")
for k, v in solo.most_common(8): print("   %-42s %.1f %%" % (k, 100 * v / tot))

if len(sys.argv) > 2:
    # per-kernel solo durations of the same window (average wall time of a launch; overlapped launches include their neighbours' work)
    per = collections.defaultdict(list)
    for s_, e_, n in ev:
        if s_ >= lo and e_ <= hi:
            per[n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]].append((e_ - s_) / 1e3)
    json.dump({"source": "rocprofv3 --kernel-trace of the default bench.py command; tools/timeline.py", "steps": steps,
               "ms_per_step": (hi - lo) / 1e6 / steps, "window_ms": tot / 1e6,
               "fraction_of_time_with_kernels_in_flight": {"0": hist[0] / tot, "1": hist[1] / tot, "2": hist[2] / tot, "3+": hist[3] / tot},
               "alone_in_flight_by_kernel": {k: v / tot for k, v in solo.most_common(12)},
               "launch_wall_us_in_this_window": {k: {"launches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)}
                                                 for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]}},
              open(sys.argv[2], "w"), indent=1)
