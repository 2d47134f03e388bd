"""The panel kernel's packed-half GELU (csrc/panel.hip, round 5), checked on the CPU: the half constants compiled into the kernel are the ones
tools/gelu_pk16_fit.py derives, and the kernel's instruction sequence -- emulated with one half rounding per packed fma, exactly as the tool
does -- stays inside the error figures DESIGN.md / panel.hip quote against the exact erf-GELU of HF ACT2FN["gelu"] (modeling_dinov2.py:293-297).
The GPU side of the same claim is tests/test_hip_panel.py (the kernel against an exact GELU through fc2) and tests/test_hip_stages.py."""
import importlib.util
import os
import re

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("gelu_pk16_fit", os.path.join(REPO, "tools", "gelu_pk16_fit.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _kernel_constants():
    """{name: half value} of pk_gelu_consts() in panel_shared.h (both halves of every packed constant must be equal)."""
    src = open(os.path.join(REPO, "crossscore_amd", "csrc", "panel_shared.h")).read()  # (shared by panel.hip and panel4.hip since round 6)
    out = {}
    for name, hexv in re.findall(r'asm volatile\("[sv]_mov_b32 %0, 0x([0-9a-f]{8})" : "=[sv]"\(k\.(\w+)\)\)', src):
        name, hexv = hexv, name
        lo, hi = int(hexv[4:], 16), int(hexv[:4], 16)
        assert lo == hi, (name, hexv)
        out[name] = float(np.array([lo], dtype=np.uint16).view(np.float16)[0])
    return out


def test_kernel_constants_are_the_fitted_ones_and_the_error_figures_hold():
    t = _tool()
    k = _kernel_constants()
    assert set(k) == {"nk", "c5", "c4", "c3", "c2", "c1", "c0", "vc6"} and k["nk"] == -0.25  # -1 / R, R = 4
    c, fit_err = t.fit()
    assert fit_err < 1.0e-4  # 8.2e-5: minimax fit of -|x| Phi(-|x|) in z on |x| <= 4
    fitted = [float(np.float16(v)) for v in c]
    compiled = [k["c0"], k["c1"], k["c2"], k["c3"], k["c4"], k["c5"], k["vc6"]]
    assert compiled == fitted, (compiled, fitted)
    # the kernel's arithmetic on its own constants
    rng = np.random.default_rng(0)
    for sigma, rms_bound in ((0.5, 2.6e-4), (1.0, 3.2e-4), (2.0, 5.0e-4)):   # measured 2.1e-4 / 2.6e-4 / 4.0e-4
        x = (rng.standard_normal(200000) * sigma).astype(np.float32).astype(np.float64)
        e = t.kernel_gelu(x, np.asarray(compiled)) - t.gelu(x)
        assert float(np.sqrt((e ** 2).mean())) < rms_bound, sigma
        # never much worse than the reference's own 16-mixed arithmetic (GELU of the half-rounded input, rounded to half)
        ref16 = t.r16(t.gelu(t.r16(x))) - t.gelu(x)
        assert float(np.sqrt((e ** 2).mean())) < 2.4 * float(np.sqrt((ref16 ** 2).mean()))
    xl = np.linspace(-8.0, 8.0, 400001)
    e = np.abs(t.kernel_gelu(xl, np.asarray(compiled)) - t.gelu(xl))
    assert e.max() < 2.5e-3                      # 2.1e-3 at x = 2.30 (one half ulp is 9.8e-4 there)
    # beyond the fitted range the result is relu(x) + P6(-1/2) = relu(x) - 1.3e-4: exact to 2e-4 below -4, the output's half rounding above 4
    assert e[xl <= -4.0].max() < 2e-4 and (e[xl >= 4.0] / xl[xl >= 4.0]).max() < 2.0 ** -11 + 5e-5
    # values outside the half range behave as relu (what bf16 operand mode relies on for its correction term)
    big = np.array([7.0e4, -7.0e4, 1.0e9, -1.0e9, np.inf, -np.inf])
    with np.errstate(over="ignore"):  # (the half conversion of 7e4 / 1e9 overflows to inf by design: v_cvt_pk_f16_f32 does the same)
        y = t.kernel_gelu(big, np.asarray(compiled))
    assert np.all(y[[1, 3, 5]] <= 0) and np.all(np.abs(y[[1, 3, 5]]) < 2e-4) and np.all(np.isinf(y[[0, 2, 4]]))


def test_the_horner_form_of_phi_does_not_survive_half_precision():
    """Why the kernel does not evaluate Phi(x) = 0.5 + x Q(x^2) in halves (the form of the fp32 epilogues, cs_common.h gelu_erf4)."""
    t = _tool()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(100000) * 2.0).astype(np.float32).astype(np.float64)
    c, _ = t.fit()
    good = float(np.sqrt(((t.kernel_gelu(x, c) - t.gelu(x)) ** 2).mean()))
    naive = float(np.sqrt(((t.naive_gelu(x) - t.gelu(x)) ** 2).mean()))
    assert naive > 4 * good and naive > 1.5e-3, (naive, good)
