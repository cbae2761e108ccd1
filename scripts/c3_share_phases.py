"""Phases of the two refits of the configs[3] share of one GPU at 8 GPUs (250 light curves), sequential and side by side."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("config3_probe", os.path.join(os.path.dirname(os.path.abspath(__file__)), "config3_probe.py"))
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
import numpy as np
from mind_the_gaps_amd import ppp
orig = ppp.derive_posteriors_batch
def traced(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    print("   refit P=%d own_engine=%r: %.3f s  phases %s  [%.3f .. %.3f]" % (r.max_parameters.shape[1], k.get("own_engine"), time.perf_counter() - t0,
          {x: round(v, 3) for x, v in r.seconds.items()}, t0 - T0, time.perf_counter() - T0), flush=True)
    return r
ppp.derive_posteriors_batch = traced
for mode, pair in ((False, "1"), ("unpaired", "0"), ("auto", "1"), ("unpaired", "0"), ("auto", "1")):
    # side by side: the two contexts' pipelined half-steps in one launch ("auto": paired) or not ("unpaired")
    T0 = time.perf_counter()
    d = probe.run(250, concurrent_refits=mode)
    print("mode %r pair %s: whole %.3f s, refits %.3f s, p = %.10f, T_sim checksum %.12f" % (
        mode, pair, d["whole_test_s"], d["seconds"]["refit_null"] + d["seconds"]["refit_alt"], d["p_value"], d["T_sim_checksum"]), flush=True)
