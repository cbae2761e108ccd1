"""``GPModelling.spread_walkers`` against the REFERENCE's own outputs.

tests/golden/spread_golden.npz holds what /root/reference/mind_the_gaps/gpmodelling.py:289-350
returned for 18 cases x 3 seeds (generator: tests/golden/make_spread_golden.py, which compiles
that one function from the reference file).  Identity is asked for, not closeness: the same
array, the same number of warnings raised or none, and numpy's global generator left at the same
place -- so whatever is drawn next (emcee's private RandomState is seeded from it, SURVEY
Appendix B) is the reference's too.
"""
import os
import warnings

import numpy as np
import pytest

from mind_the_gaps_amd import walkers
from mind_the_gaps_amd.gpmodelling import GPModelling

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "spread_golden.npz"))
NAMES = [str(n) for n in GOLD["names"]]


def box_of(name):
    return [(None if np.isnan(lo) else lo, None if np.isnan(hi) else hi) for lo, hi in GOLD[name + "/box"]]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_identical_to_the_reference(name, seed):
    key = "%s/%d" % (name, seed)
    np.random.seed(seed)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        # the method reads no state of the object (nor does the reference's): unbound call
        got = GPModelling.spread_walkers(None, int(GOLD[name + "/walkers"]), GOLD[name + "/centre"], box_of(name),
                                         percent=float(GOLD[name + "/percent"]),
                                         max_attempts=int(GOLD[name + "/max_attempts"]))
    assert np.array_equal(got, GOLD[key + "/p0"])
    assert len(caught) == int(GOLD[key + "/warned"])
    assert np.random.random_sample() == float(GOLD[key + "/next_uniform"])


def test_fixtures_exercise_redraws_and_clamps():
    """The cases are not all the easy kind: some redraw, some use up their attempts."""
    redraw = clamp = 0
    for name in NAMES:
        centre, box = GOLD[name + "/centre"], GOLD[name + "/box"]
        lo, hi = np.where(np.isnan(box[:, 0]), -np.inf, box[:, 0]), np.where(np.isnan(box[:, 1]), np.inf, box[:, 1])
        for seed in (0, 1, 2):
            np.random.seed(seed)
            first = np.random.normal(centre, np.abs(centre) * float(GOLD[name + "/percent"]),
                                     size=(int(GOLD[name + "/walkers"]), len(centre)))
            redraw += int(np.any((first < lo) | (first > hi)))
            clamp += int(GOLD["%s/%d/warned" % (name, seed)] > 0)
    assert redraw >= 30 and clamp >= 15


def test_batched_form_agrees_when_nothing_is_redrawn():
    """walkers.spread (attempt-major, for the Protassov batches) and the reference order are the
    same array whenever no walker leaves the box -- they differ only in the order of redraws."""
    name = "tutorial_alt"
    box = GOLD[name + "/box"]
    np.random.seed(1)
    got = walkers.spread(np.random.normal, GOLD[name + "/centre"][None, :], box[:, 0], box[:, 1],
                         int(GOLD[name + "/walkers"]))[0]
    assert np.array_equal(got, GOLD[name + "/1/p0"])
