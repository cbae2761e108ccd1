"""The C-ABI from plain C (examples/c_api_demo.c): the header is valid C99, the library links
without Python or torch, fails loudly without a GPU, and on a GPU prints the oracle's values."""
import os
import subprocess

import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "mind_the_gaps_amd")


def build(tmp_path):
    exe = str(tmp_path / "c_api_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_api_demo.c"), "-o", exe, "-L", LIBDIR, "-lmtg_hip",
                           "-Wl,-rpath," + LIBDIR, "-lm"])
    return exe


def test_header_is_c99_and_cxx11():
    hdr = os.path.join(ROOT, "include", "mtg.h")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])


def test_c_program_links_and_fails_loudly_without_a_gpu(tmp_path):
    from mind_the_gaps_amd import engine
    exe = build(tmp_path)
    if engine.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_c_program_matches_oracle(tmp_path):
    from oracle import celerite as oracle_c
    r = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    out = np.array([float(v[0]) for v in rows]); st = np.array([int(v[1]) for v in rows])
    # the same inputs as examples/c_api_demo.c
    n = np.arange(400)
    t = np.cumsum(0.3 + 0.7 * np.abs(np.sin(1.7 * n)) + np.where(n % 150 == 149, 40.0, 0.0))
    y = 100.0 + 8.0 * np.sin(0.9 * t) + 3.0 * np.cos(0.13 * n * n)
    dy = 1.0 + 0.5 * np.abs(np.cos(2.3 * n))
    full = np.array([np.log(100.0), np.log(2 * np.pi / 20.0), np.log(50.0), np.log(3.0), np.log(2 * np.pi / 7.0), 100.0])
    theta = np.array([[full[p] * (1.0 + 0.04 * (b - 2) * (1 if p % 2 else -1)) for p in range(5)] for b in range(6)])
    theta[5, 1] = 11.0
    bounds = np.vstack([synth.bounds_for(synth.NULL_MODEL), [(-np.inf, np.inf)]])
    ref, rst = oracle_c.logprob_batch(t, y, dy, synth.NULL_MODEL, np.hstack([theta, np.full((6, 1), 100.0)]),
                                      bounds=bounds, add_prior=True)
    assert np.array_equal(st, rst) and list(st) == [0, 0, 0, 0, 0, 1] and np.isneginf(out[5])
    assert np.max(np.abs(out[:5] - ref[:5]) / np.abs(ref[:5])) <= 1e-8
