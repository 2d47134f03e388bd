"""Wall time of the four-wave token-panel kernel in one or more variant builds (tools/build_variant.sh NAME panel4.hip -D...), one child process per
variant, same box: CS_VARIANTS=a,b,c python tools/panel4_time.py  ("-" = the in-tree build).  Ablation variants compute wrong results by design."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_P4_CHILD"):
    v = os.environ["CS_P4_CHILD"]
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    if v != "-": sys.path.insert(0, os.path.join(R, "tools", "_var", v))
    import torch
    import hip_helpers as hh
    from crossscore_amd import _lib
    from test_hip_panel import _make, _reference
    lib = _lib.load(); lib.cs_debug_panel_impl(int(os.environ.get("CS_PANEL_IMPL", "1")))
    dev = torch.device("cuda:0")
    out = []
    for M in (48 * 1370, 256 * 128):
        x, o, w = _make(M, 21, dev)
        img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
        xk = x.clone(); u = hh.encoder_panel(xk, o, img, w["bo"], w["b1"], w["b2"])
        ref_x, _ = _reference(x[:1024], o[:1024], w, True, emulate=True)
        err = float((xk[:1024] - ref_x).abs().max())
        xs = x.clone()
        for _ in range(5): hh.encoder_panel(xs, o, img, w["bo"], w["b1"], w["b2"])
        ts = []
        for r in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): hh.encoder_panel(xs, o, img, w["bo"], w["b1"], w["b2"])
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 50)
        out.append(f"M={M}: {sorted(ts)[2]:.1f} us (err {err:.1e})")
    print(f"{v:12s} " + "   ".join(out), flush=True)
    sys.exit(0)
for v in os.environ.get("CS_VARIANTS", "-").split(","):
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_P4_CHILD=v))
