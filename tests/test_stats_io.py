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
