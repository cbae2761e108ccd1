"""CPU tests of the post-processing helpers."""
import numpy as np

from mind_the_gaps_amd import stats
from mind_the_gaps_amd.lightcurves import GappyLightcurve


def test_information_criteria_match_reference_formulas():
    # /root/reference/mind_the_gaps/stats.py:155-195
    lnl, n, k = -1234.5, 400, 5
    assert stats.bic(lnl, n, k) == -2.0 * lnl + k * np.log(n)
    assert stats.aic(lnl, k) == 2 * k - 2 * lnl
    assert stats.aicc(lnl, n, k) == stats.aic(lnl, k) + 2 * k * (k + 1) / (n - k - 1)


def test_information_criteria_match_the_reference_module():
    """tests/golden/stats_golden.json holds outputs of the reference's own stats.py (the one
    module of the reference that imports here); see tests/golden/make_stats_golden.py."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stats_golden.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 60
    for c in cases:
        assert stats.bic(c["lnL"], c["n"], c["k"]) == c["bic"]
        assert stats.aic(c["lnL"], c["k"]) == c["aic"]
        assert stats.aicc(c["lnL"], c["n"], c["k"]) == c["aicc"]


def test_lrt_statistic_and_pvalue():
    null, alt = np.array([-100.0, -90.0, -80.0]), np.array([-95.0, -90.0, -70.0])
    assert np.array_equal(stats.lrt_statistic(null, alt), [10.0, 0.0, 20.0])     # tutorial_ppp.ipynb:406-411
    sims = np.arange(99.0)
    assert stats.lrt_pvalue(98.0, sims) == 2 / 100 and stats.lrt_pvalue(1e9, sims) == 1 / 100
    assert stats.lrt_pvalue(-1.0, sims) == 1.0


def test_lightcurve_csv_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    t = np.cumsum(rng.uniform(1, 2, 20))
    lc = GappyLightcurve(t, rng.normal(10, 1, 20), rng.uniform(0.1, 0.2, 20), exposures=0.5)
    f = tmp_path / "lc.csv"
    lc.to_csv(str(f))
    assert open(f).readline().startswith("# t\trate\terror\texposure")
    back = GappyLightcurve.from_csv(str(f))
    assert np.allclose(back.times, lc.times, rtol=1e-8) and np.allclose(back.y, lc.y, atol=1e-5)
    assert np.allclose(back.dy, lc.dy, atol=1e-5) and np.allclose(back.exposures, 0.5)


def test_distributions_match_the_reference_module():
    """kraft_pdf, lognormal, create_log_normal, create_uniform_distribution (reference stats.py:10-29, 116-146) against the
    reference module's own outputs (tests/golden/make_stats_golden.py)."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stats_golden.json")) as f:
        dists = json.load(f)["distributions"]
    assert len(dists) == 20
    for d in dists:
        x = np.array(d["x"])
        ln, un = stats.create_log_normal(d["mean"], d["std"]), stats.create_uniform_distribution(d["mean"], d["std"])
        np.testing.assert_allclose(ln.pdf(x), d["lognormal_pdf"], rtol=1e-13)
        np.testing.assert_allclose(ln.ppf([0.1, 0.5, 0.9]), d["lognormal_ppf"], rtol=1e-13)
        np.testing.assert_allclose([ln.mean(), ln.std()], d["lognormal_moments"], rtol=1e-13)
        np.testing.assert_allclose([ln.mean(), ln.std()], [d["mean"], d["std"]], rtol=1e-10)       # what the function is for
        np.testing.assert_allclose(un.pdf(x), d["uniform_pdf"], rtol=1e-13)
        np.testing.assert_allclose([un.ppf(0.0), un.ppf(1.0)], d["uniform_support"], rtol=1e-13)
        np.testing.assert_allclose([un.mean(), un.std()], [d["mean"], d["std"]], rtol=1e-10)
        got = [float(stats.kraft_pdf(a=0)._pdf(v, d["N"], d["B"])) for v in d["s"]]
        np.testing.assert_allclose(got, d["kraft_pdf"], rtol=1e-12)
        np.testing.assert_allclose(stats.lognormal(a=0)._pdf(x, d["center"], d["sigma"]), d["lognormal_class_pdf"], rtol=1e-13)
    # the Kraft posterior integrates to one and its median is the simulator's (simulator.py kraft helpers)
    k = stats.kraft_pdf(a=0)
    from scipy.integrate import quad
    assert abs(quad(lambda s: k._pdf(s, 7, 2.5), 0, 80)[0] - 1.0) < 1e-9


def test_tutorial_pvalue_expression_and_neg_log_like():
    from scipy.stats import percentileofscore
    sims = np.random.default_rng(1).chisquare(3, 500)
    for t in (0.5, 3.0, 11.0, 1e3, -1.0):
        assert stats.lrt_pvalue_percentile(t, sims) == 1 - percentileofscore(sims, t) / 100
        assert abs(stats.lrt_pvalue_percentile(t, sims) - stats.lrt_pvalue(t, sims)) <= 1.0 / 500 + 1e-12

    class FakeGP:
        def set_parameter_vector(self, p): self.p = np.asarray(p)
        def log_likelihood(self, y): return -float(np.sum((y - self.p[0]) ** 2))
    g = FakeGP()
    assert stats.neg_log_like([2.0], np.array([1.0, 4.0]), g) == 5.0 and g.p[0] == 2.0


def test_simple_lightcurve_reads_what_to_csv_writes_and_plain_tables(tmp_path):
    """lightcurves/simplelightcurve.py:16-59: columns by position, a header line of names, days -> seconds by the
    time column's name, missing exposure / background columns -> zeros (with the reference's warning)."""
    import pytest
    from mind_the_gaps_amd.lightcurves import SimpleLightcurve
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.uniform(100, 200, 15))
    lc = GappyLightcurve(t, rng.normal(10, 1, 15), rng.uniform(0.1, 0.2, 15), exposures=50.0, bkg_rate=rng.uniform(0, 1, 15),
                         bkg_rate_err=rng.uniform(0, 0.1, 15))
    f = tmp_path / "lc.csv"
    lc.to_csv(str(f))
    back = SimpleLightcurve(str(f))
    assert back.n == 15 and np.allclose(back.times, t, rtol=1e-8) and np.allclose(back.y, lc.y, atol=1e-5)
    assert np.allclose(back.exposures, 50.0) and np.allclose(back.bkg_rate, lc.bkg_rate, atol=1e-5) and np.allclose(back.bkg_rate_err, lc.bkg_rate_err, atol=1e-5)
    g = tmp_path / "three.txt"
    g.write_text("mjd,flux,err\n1.0,5.0,0.5\n2.5,6.0,0.4\n4.0,5.5,0.6\n")
    with pytest.warns(UserWarning, match="no exposures"):
        three = SimpleLightcurve(str(g), delimiter=",")
    assert np.array_equal(three.times, np.array([1.0, 2.5, 4.0]) * 86400.0) and np.array_equal(three.y, [5.0, 6.0, 5.5])
    assert np.array_equal(three.exposures, np.zeros(3)) and np.array_equal(three.bkg_rate, np.zeros(3)) and three.duration == 3.0 * 86400
    h = tmp_path / "four.txt"
    h.write_text("junk line\ntime rate error exposure\n0 1 0.1 2\n10 2 0.1 2\n20 3 0.1 2\n")
    four = SimpleLightcurve(str(h), skip_header=1)
    assert np.array_equal(four.times, [0, 10, 20]) and np.array_equal(four.exposures, [2, 2, 2]) and np.array_equal(four.bkg_rate_err, np.zeros(3))


def test_truncate_split_rand_remove_get_simulator():
    """gappylightcurve.py:174-293"""
    import pytest
    t = np.concatenate([np.arange(0.0, 50.0, 5.0), np.arange(200.0, 240.0, 5.0), np.arange(1000.0, 1030.0, 5.0)])
    n = len(t)
    lc = GappyLightcurve(t, np.arange(n) + 100.0, np.full(n, 0.5), exposures=2.0, bkg_rate=np.full(n, 0.1), bkg_rate_err=np.full(n, 0.01))
    cut = lc.truncate(20.0, 205.0)
    assert np.array_equal(cut.times, [20, 25, 30, 35, 40, 45, 200, 205]) and np.array_equal(cut.y, lc.y[4:12]) and np.array_equal(cut.exposures, np.full(8, 2.0))
    assert lc.truncate().n == n and lc.truncate(tmin=1000.0).n == 6
    with pytest.raises(ValueError, match="greater than or equal"):
        lc.truncate(10.0, 10.0)
    with pytest.raises(ValueError, match="lower than initial"):
        lc.truncate(-20.0, -10.0)
    pieces = lc.split(100.0)
    assert [p.n for p in pieces] == [10, 8, 6] and np.array_equal(np.concatenate([p.times for p in pieces]), t)
    assert [p.n for p in lc.split(1e6)] == [n]
    import random
    random.seed(4)
    fewer = lc.rand_remove(7)
    assert fewer.n == n - 7 and np.all(np.isin(fewer.times, t)) and np.all(np.diff(fewer.times) > 0)
    assert np.array_equal(fewer.y, lc.y[np.isin(t, fewer.times)])
    with pytest.raises(ValueError, match="greater than number"):
        lc.rand_remove(n + 1)
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw
    sim = lc.get_simulator(BendingPowerlaw(S0=1.0, omega0=0.1), "Lognormal", sigma_noise=2.0, extension_factor=3)
    assert sim.pdf == "Lognormal" and np.array_equal(sim._times, t) and sim.mean == lc.mean and sim.noise_name == "Gaussian"
