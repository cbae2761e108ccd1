"""Minimal light-curve container for the hot path.

Mirror of the part of /root/reference/mind_the_gaps/lightcurves/gappylightcurve.py
that ``GPModelling`` reads (constructor :24-70 and the properties :72-171):
``times, y, dy, exposures, bkg_rate, bkg_rate_err, n, duration, mean``, plus the CSV
round trip of :256-262.  Mission file readers,
truncation/splitting and ``get_simulator`` are outside the hot path.
"""
import numpy as np


class ExposureTimeError(Exception):
    def __init__(self, message):
        super().__init__(message)


class GappyLightcurve:
    """An irregularly sampled light curve (timestamps always in seconds)."""

    def __init__(self, times, y, dy=None, exposures=None, bkg_rate=None, bkg_rate_err=None):
        self._times = times
        self._y = y
        self._dy = dy
        if exposures is not None:
            if np.isscalar(exposures):
                self._exposures = np.full(len(times), exposures)
            else:
                self._exposures = exposures
            epsilon = 1.01  # numerically distinct but equal spacings
            wrong = np.count_nonzero(np.diff(self._times) < self._exposures[:-1] * epsilon / 2)
            if wrong > 0:
                raise ExposureTimeError(
                    "Some timestamps (%d) have a spacing below the exposure sampling time!" % wrong)
        else:
            self._exposures = np.zeros(len(times))
        self._bkg_rate = bkg_rate if bkg_rate is not None else np.zeros(len(times))
        self._bkg_rate_err = bkg_rate_err if bkg_rate_err is not None else np.zeros(len(times))

    @property
    def times(self):
        return self._times

    @property
    def n(self):
        return len(self._times)

    @property
    def y(self):
        return self._y

    @property
    def dy(self):
        return self._dy

    @property
    def exposures(self):
        return self._exposures

    @property
    def bkg_rate(self):
        return self._bkg_rate

    @property
    def bkg_rate_err(self):
        return self._bkg_rate_err

    @property
    def duration(self):
        return self._times[-1] - self._times[0]

    @property
    def mean(self):
        return np.mean(self._y)

    def to_csv(self, outname):
        """Save the light curve (gappylightcurve.py:256-262: same columns and formats)."""
        outputs = np.array([self._times, self._y, self._dy, self._exposures, self._bkg_rate, self._bkg_rate_err])
        np.savetxt(outname, outputs.T, fmt="%.8e\t%.5f\t%.5f\t%.3f\t%.5f\t%.5f",
                   header="t\trate\terror\texposure\tbkg_rate\tbkg_rate_err")

    @classmethod
    def from_csv(cls, filename):
        """Read back what ``to_csv`` wrote."""
        t, y, dy, exp, bkg, bkg_err = np.loadtxt(filename, unpack=True, ndmin=2)
        return cls(t, y, dy, exposures=exp if np.any(exp) else None, bkg_rate=bkg, bkg_rate_err=bkg_err)
