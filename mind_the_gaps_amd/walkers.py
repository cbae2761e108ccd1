"""Initial walker positions: the law of ``GPModelling.spread_walkers``
(/root/reference/mind_the_gaps/gpmodelling.py:289-350) for any number of ensembles at once.

Every walker is drawn from N(centre, percent * |centre|); a walker with a coordinate outside
the box is redrawn as a whole, at most ``max_attempts`` times; what is still outside then is
put next to the violated bound: ``bound * 1.05`` or ``bound * 0.95``, whichever lies inside
for that bound's sign (gpmodelling.py:327-328, 346-349).

Two forms of the same law, differing only in the order the generator is consumed in:

``spread_reference_order``  one ensemble, walker by walker exactly as the reference loops
    (gpmodelling.py:330-349): for a seed it returns the reference's own array, and leaves the
    generator where the reference leaves it (tests/golden/spread_golden.npz, produced by the
    reference's function).  This is what ``GPModelling.spread_walkers`` calls.
``spread``  any number of ensembles, attempt by attempt over all stuck walkers at once -- the
    Protassov sweep's 2000 x 256 walkers cost twenty vectorised passes, not 512 000 loops.  No
    reference stream exists for that batch (the reference runs one process per light curve), so
    only the law is shared.
"""
import warnings

import numpy as np

__all__ = ["spread", "spread_reference_order"]


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


def spread_reference_order(normal, center, lower, upper, walkers, percent=0.1, max_attempts=20):
    """One ensemble in the reference's draw order.  center / lower / upper [P]; returns [walkers, P].

    Walker i is tested and redrawn up to ``max_attempts`` times before walker i + 1 is looked at;
    each redraw is one ``normal(center, std)`` call of P values.  The test comes before the
    redraw, so the last redraw is never tested, and a walker whose loop ended on its last attempt
    -- redrawn or found inside just then -- is warned about and clamped (a no-op for a walker that
    is inside): gpmodelling.py:332-349."""
    if percent < 0 or percent > 1:
        raise ValueError("The 'percent' parameter must be between 0 and 1 (inclusive).")
    center = np.asarray(center, dtype=np.float64)
    lower, upper = np.asarray(lower, dtype=np.float64), np.asarray(upper, dtype=np.float64)
    std = np.abs(center) * percent
    p0 = np.asarray(normal(center, std, size=(walkers, len(center))), dtype=np.float64)
    near_lower = np.where(lower > 0, 1.05, 0.95) * lower
    near_upper = np.where(upper > 0, 0.95, 1.05) * upper
    if max_attempts < 1:      # the reference's loop variable would be unbound here: nothing to retry
        return p0
    # walkers that are inside at the first look (nearly all of them) never touch the generator
    outside = np.nonzero(~np.all((lower <= p0) & (p0 <= upper), axis=1))[0]
    last = np.zeros(walkers, dtype=bool)
    last[:] = max_attempts == 1
    for i in outside:
        for attempt in range(max_attempts):
            if np.all((lower <= p0[i]) & (p0[i] <= upper)):
                break
            p0[i] = normal(center, std)
        last[i] = attempt == max_attempts - 1
    for i in np.nonzero(last)[0]:
        warnings.warn("Some walkers are out of bounds! Setting them to values close to the bounds")
        p0[i] = np.where(p0[i] > upper, near_upper, np.where(p0[i] < lower, near_lower, p0[i]))
    return p0
