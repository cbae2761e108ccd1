#!/usr/bin/env python3
"""Generates tests/golden/coeff_golden.npz: outputs of the REFERENCE's own coefficient builders
(mind_the_gaps/models/celerite_models.py: Lorentzian.get_real_coefficients :9-15 and .get_complex_coefficients
:17-31, Cosinus :39-52, DampedRandomWalk :58-66, BendingPowerlaw :77-83).

The classes derive from celerite's ``Term`` (not installed), but these methods are plain functions of ``params`` --
none of them touches ``self`` -- so each FunctionDef is compiled here from the reference file where it lies and called
with ``self = None``: read at generation time, never copied; inputs and outputs only are committed.  What celerite's
``Term.coefficients`` does with the returned tuples is SURVEY.md Appendix A.2 (``np.atleast_1d`` of every entry; a complex
3-tuple means b = 0; missing families are empty): applied here so that a fixture row is (a_real, c_real, a_comp, b_comp,
c_comp, d_comp).

Run from the repo root (needs /root/reference):  python tests/golden/make_coeff_golden.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/mind_the_gaps/models/celerite_models.py"

methods = {}
for cls in ast.parse(open(SRC).read()).body:
    if isinstance(cls, ast.ClassDef):
        for node in cls.body:
            if isinstance(node, ast.FunctionDef) and node.name in ("get_real_coefficients", "get_complex_coefficients"):
                ns = {"np": np}
                exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
                methods[(cls.name, node.name)] = ns[node.name]

NPAR = {"Lorentzian": 3, "Cosinus": 2, "DampedRandomWalk": 2, "BendingPowerlaw": 3}
rng = np.random.default_rng(20250704)
out = {"classes": np.array(sorted(NPAR))}
for name in sorted(NPAR):
    # the tutorials' box: amplitudes (-10, 50), the rest (-10, 10); plus the tutorial's own values
    params = np.column_stack([rng.uniform(-10, 50, 40)] + [rng.uniform(-10, 10, 40) for _ in range(NPAR[name] - 1)])
    params[0, :] = [np.log(100.0), np.log(80.0), np.log(2 * np.pi / 10.0)][:NPAR[name]] if NPAR[name] == 3 else [np.log(100.0), np.log(2 * np.pi / 20.0)]
    rows = []
    for p in params:
        real = methods[(name, "get_real_coefficients")](None, p) if (name, "get_real_coefficients") in methods else ()
        comp = methods[(name, "get_complex_coefficients")](None, p) if (name, "get_complex_coefficients") in methods else ()
        real = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in real] or [np.empty(0), np.empty(0)]
        comp = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in comp]
        if len(comp) == 3:
            comp = [comp[0], np.zeros_like(comp[0]), comp[1], comp[2]]
        comp = comp or [np.empty(0)] * 4
        rows.append([v for v in real + comp])
    out[name + "/params"] = params
    for j, key in enumerate(("a_real", "c_real", "a_comp", "b_comp", "c_comp", "d_comp")):
        out["%s/%s" % (name, key)] = np.array([r[j] for r in rows])
np.savez_compressed(os.path.join(HERE, "coeff_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
