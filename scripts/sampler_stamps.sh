#!/bin/bash
# phase times inside the speculative sampler kernel at tutorial size (N = 1000, 30 walkers) and configs[1] (N = 1e4, 128 walkers):
# a library built with -DMTG_SAMPLER_STAMPS in a scratch copy.   gpurun -- bash scripts/sampler_stamps.sh
set -e
REPO=$PWD; W=/tmp/stamps; rm -rf $W; mkdir -p $W; cp -a $REPO/mind_the_gaps_amd $REPO/oracle $REPO/tests $REPO/scripts $REPO/include $W/ 2>/dev/null
cd $W/mind_the_gaps_amd/csrc && rm -f mtg_sampler.o && make HIPFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DMTG_SAMPLER_STAMPS" > $W/make.log 2>&1 || { tail -5 $W/make.log; exit 1; }
cd $W && python3 - <<'PY'
import sys, warnings; sys.path.insert(0, "/tmp/stamps")
import numpy as np
from mind_the_gaps_amd import terms, synthetic as synth
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
for N, W, kernel, name in ((1000, 30, lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=dict(log_a=(-10, 50), log_c=(-10, 10))), "tutorial null (J = 1)"),
                           (1000, 30, lambda: terms.ComplexTerm(log_a=np.log(100.0), log_c=-5.0, log_d=-0.46, bounds=dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5)))
                            + terms.RealTerm(np.log(100.0), np.log(0.3), bounds=dict(log_a=(-10, 50), log_c=(-10, 10))), "tutorial alternative (J = 3)"),
                           (10000, 128, lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=dict(log_a=(-10, 50), log_c=(-10, 10)))
                            + terms.SHOTerm(np.log(50.0), np.log(3.0), np.log(0.9), bounds=[(-10, 50), (-10, 10), (-10, 10)]), "configs[1]")):
    t, y, dy = synth.make_lightcurves(N, 1, seed=3)
    lc = GappyLightcurve(t, y[0], dy[0])
    print("==", name, "N", N, "walkers", W, flush=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = GPModelling(lc, kernel())
        import time; t0 = time.perf_counter()
        m.derive_posteriors(fit=True, converge=False, max_steps=400, walkers=W, progress=False)
        print("   %.1f us per iteration (wall, incl. fit and set-up)" % (1e6 * (time.perf_counter() - t0) / 400), flush=True)
PY
