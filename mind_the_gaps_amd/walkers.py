"""Initial walker positions: the law of ``GPModelling.spread_walkers``
(/root/reference/mind_the_gaps/gpmodelling.py:289-350) for any number of ensembles at once.

Every walker is drawn from N(centre, percent * |centre|); a walker with a coordinate outside
the box is redrawn as a whole, at most ``max_attempts`` times; what is still outside then is
put next to the violated bound: ``bound * 1.05`` or ``bound * 0.95``, whichever lies inside
for that bound's sign (gpmodelling.py:327-328, 346-349).  All ensembles advance together, so
the Protassov sweep's 2000 x 256 walkers cost twenty vectorised passes, not 512 000 loops.
"""
import warnings

import numpy as np

__all__ = ["spread"]


def spread(normal, centers, lower, upper, walkers, percent=0.1, max_attempts=20):
    """``normal(loc, scale)`` draws like numpy's; centers [E, P]; lower / upper [P] (+-inf = open).
    Returns [E, walkers, P]."""
    if percent < 0 or percent > 1:
        raise ValueError("The 'percent' parameter must be between 0 and 1 (inclusive).")
    centers = np.atleast_2d(np.asarray(centers, dtype=np.float64))
    lower, upper = np.asarray(lower, dtype=np.float64), np.asarray(upper, dtype=np.float64)
    E, P = centers.shape
    loc = np.broadcast_to(centers[:, None, :], (E, walkers, P))
    scale = np.broadcast_to(np.abs(centers)[:, None, :] * percent, (E, walkers, P))
    p0 = np.asarray(normal(loc, scale), dtype=np.float64)
    stuck = np.zeros((E, walkers), dtype=bool)
    for attempt in range(max_attempts):
        stuck = np.any((p0 < lower) | (p0 > upper), axis=2)
        if not stuck.any():
            break
        e, w = np.nonzero(stuck)
        p0[e, w] = normal(loc[e, w], scale[e, w])
    if stuck.any():   # these walkers used up their attempts
        warnings.warn("Some walkers are out of bounds! Setting them to values close to the bounds")
        near_lower = np.where(lower > 0, 1.05, 0.95) * lower
        near_upper = np.where(upper > 0, 0.95, 1.05) * upper
        p0 = np.where(p0 < lower, near_lower, np.where(p0 > upper, near_upper, p0))
    return p0
