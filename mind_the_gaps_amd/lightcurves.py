"""Light-curve container of the hot path: the six per-epoch columns ``GPModelling`` and the
simulator read, and the tab-separated file they travel in.

Contract (reference lightcurves/gappylightcurve.py): a light curve is
``times, y, dy, exposures, bkg_rate, bkg_rate_err`` (one value per epoch, timestamps in seconds),
with ``n``, ``duration`` and ``mean`` derived from them; epochs whose exposures would overlap are an
error; the CSV has exactly these six columns.  Mission file readers, truncation / splitting and
``get_simulator`` are not on the path.
"""
import numpy as np

#           attribute        CSV header      CSV format
COLUMNS = (("times", "t", "%.8e"),
           ("y", "rate", "%.5f"),
           ("dy", "error", "%.5f"),
           ("exposures", "exposure", "%.3f"),
           ("bkg_rate", "bkg_rate", "%.5f"),
           ("bkg_rate_err", "bkg_rate_err", "%.5f"))
# two epochs may be this much closer than the exposures allow before it counts as an overlap
# (equal spacings that differ in the last digits)
SPACING_TOLERANCE = 0.01


class ExposureTimeError(Exception):
    """Consecutive epochs closer together than their exposures allow."""


def _per_epoch(value, n):
    """None -> zeros; a scalar -> the same value at every epoch; an array -> itself."""
    if value is None:
        return np.zeros(n)
    if np.ndim(value) == 0:
        return np.full(n, float(value))
    return value


class GappyLightcurve:
    """An irregularly sampled light curve."""

    def __init__(self, times, y, dy=None, exposures=None, bkg_rate=None, bkg_rate_err=None):
        n = len(times)
        self.times, self.y, self.dy = times, y, dy
        self.exposures = _per_epoch(exposures, n)
        self.bkg_rate = _per_epoch(bkg_rate, n)
        self.bkg_rate_err = _per_epoch(bkg_rate_err, n)
        if exposures is not None and n > 1:
            # an exposure is centred on its timestamp: the next epoch must start at least half of it later
            gap = np.diff(np.asarray(self.times, dtype=float))
            needed = 0.5 * (1.0 + SPACING_TOLERANCE) * np.asarray(self.exposures, dtype=float)[:-1]
            overlapping = int(np.sum(gap < needed))
            if overlapping:
                raise ExposureTimeError("%d epochs follow their predecessor sooner than its exposure time allows"
                                        % overlapping)

    n = property(lambda self: len(self.times), doc="number of epochs")
    duration = property(lambda self: self.times[-1] - self.times[0], doc="last minus first timestamp")
    mean = property(lambda self: np.mean(self.y), doc="mean count rate")

    def to_csv(self, outname):
        """One row per epoch, the COLUMNS in order (the layout the reference's files have)."""
        table = np.column_stack([getattr(self, name) for name, _, _ in COLUMNS])
        np.savetxt(outname, table, fmt="\t".join(fmt for _, _, fmt in COLUMNS),
                   header="\t".join(head for _, head, _ in COLUMNS))

    @classmethod
    def from_csv(cls, filename):
        """Read back a file written by ``to_csv``."""
        cols = dict(zip((name for name, _, _ in COLUMNS), np.loadtxt(filename, unpack=True, ndmin=2)))
        exposures = cols.pop("exposures")
        return cls(exposures=exposures if np.any(exposures) else None, **cols)
