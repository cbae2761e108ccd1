"""Light-curve container of the hot path: the six per-epoch columns ``GPModelling`` and the
simulator read, and the tab-separated file they travel in.

Contract (reference lightcurves/gappylightcurve.py): a light curve is
``times, y, dy, exposures, bkg_rate, bkg_rate_err`` (one value per epoch, timestamps in seconds),
with ``n``, ``duration`` and ``mean`` derived from them; epochs whose exposures would overlap are an
error; the CSV has exactly these six columns.  ``SimpleLightcurve`` reads such a file (or any table with time, rate,
error[, exposure[, bkg_rate, bkg_rate_err]] columns) the way reference lightcurves/simplelightcurve.py:16-59 does;
``truncate`` / ``split`` / ``rand_remove`` / ``get_simulator`` are gappylightcurve.py:174-293.  Mission file readers
(Swift, Fermi) are not on the path.
"""
import random
import warnings

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

    def _subset(self, keep):
        """the epochs of a boolean mask as a new light curve"""
        cols = [None if getattr(self, name) is None else np.asarray(getattr(self, name))[keep] for name, _, _ in COLUMNS]
        return GappyLightcurve(*cols)

    def truncate(self, tmin=-np.inf, tmax=np.inf):
        """The epochs with tmin <= t <= tmax as a new light curve (gappylightcurve.py:174-207)."""
        if tmin >= tmax:
            raise ValueError("Minimum truncation time (%.2es) is greater than or equal to maximum truncation time (%.3es)!" % (tmin, tmax))
        if tmax < self.times[0]:
            raise ValueError("Maximum truncation time (%.2f) is lower than initial lightcurve time (%.2f)" % (tmax, self.times[0]))
        t = np.asarray(self.times)
        return self._subset((t >= tmin) & (t <= tmax))

    def split(self, interval):
        """Cut wherever two consecutive epochs are more than ``interval`` apart (gappylightcurve.py:209-235)."""
        t = np.asarray(self.times)
        last_of_piece = np.append(np.flatnonzero(np.diff(t) > interval), len(t) - 1)
        pieces, first = [], 0
        for last in last_of_piece:
            pieces.append(self.truncate(t[first], t[last]))
            first = last + 1
        return pieces

    def rand_remove(self, points_remove):
        """A copy with ``points_remove`` epochs taken out at random (``random.sample``, gappylightcurve.py:237-254; the
        reference RETURNS the ValueError for too large a request instead of raising it -- raised here)."""
        if points_remove > self.n:
            raise ValueError("Number of points to remove (%d) is greater than number of lightcurve datapoints (%d)" % (points_remove, self.n))
        keep = np.ones(self.n, dtype=bool)
        keep[random.sample(range(self.n), points_remove)] = False
        return self._subset(keep)

    def get_simulator(self, psd_model, pdf="gaussian", **kwargs):
        """A ``Simulator`` with this light curve's sampling, exposures, mean and background (gappylightcurve.py:265-293)."""
        from .simulator import Simulator
        return Simulator(psd_model, self.times, self.exposures, self.mean, pdf, self.bkg_rate, self.bkg_rate_err, **kwargs)

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


class SimpleLightcurve(GappyLightcurve):
    """A light curve read from a text table with a header line (reference lightcurves/simplelightcurve.py:12-59): columns
    time, rate, error and optionally exposure, background rate and its error, in that order whatever their names.  A time
    column named "mjd", "jd" or "day" is in days and converted to seconds."""

    DAY = 86400.0

    def __init__(self, input_file, skip_header=0, delimiter=None):
        time, y, yerr, exposures, bkg_rate, bkg_err = self.readdata(input_file, skip_header, delimiter)
        super().__init__(time, y, yerr, exposures, bkg_rate, bkg_err)

    def readdata(self, input_file, skip_header, delimiter):
        data = np.atleast_1d(np.genfromtxt("%s" % input_file, names=True, skip_header=skip_header, delimiter=delimiter))
        names = data.dtype.names
        time = data[names[0]] * (self.DAY if names[0] in ("mjd", "jd", "day") else 1.0)
        n = len(time)
        if len(names) > 3:
            exposures = data[names[3]]
            bkg_rate, bkg_err = (data[names[4]], data[names[5]]) if len(names) >= 6 else (np.zeros(n), np.zeros(n))
        else:
            warnings.warn("Lightcurve has no exposures!")
            exposures, bkg_rate, bkg_err = np.zeros(n), np.zeros(n), np.zeros(n)
        return time, data[names[1]], data[names[2]], exposures, bkg_rate, bkg_err
