"""Small value types shared by the message consumers (reference: python/modes_types.py:24-107)."""
from collections import namedtuple


class stamp:
    """Timestamp as whole seconds + fraction, so that UTC-sized values keep sub-microsecond
    precision (modes_types.py:27-103).  The fraction is normalised into [0, 1) on construction."""

    __hash__ = None     # mutable pair; the reference deliberately leaves it unhashable

    def __init__(self, secs, frac_secs):
        whole = int(frac_secs)
        self.secs = secs + whole
        self.frac_secs = frac_secs - whole

    def _key(self):
        return (self.secs, self.frac_secs)

    def __float__(self):
        return self.secs + self.frac_secs

    def __str__(self):
        return "%f" % float(self)

    def __eq__(self, other):
        if isinstance(other, stamp):
            return self._key() == other._key()
        if isinstance(other, float):
            return float(self) == other
        raise TypeError

    def __ne__(self, other):
        return not self == other

    def __lt__(self, other):
        if isinstance(other, stamp):
            return self._key() < other._key()
        if isinstance(other, float):
            return float(self) > other          # sic: modes_types.py:41
        raise TypeError

    def __gt__(self, other):
        if type(other) is type(self):
            return self._key() > other._key()
        raise TypeError                          # (the reference's float branch can never match)

    def __le__(self, other):
        return self == other or self < other

    def __ge__(self, other):
        return self == other or self > other

    def _coerce(self, other):
        if isinstance(other, stamp):
            return other
        if isinstance(other, float):
            return stamp(0, other)
        if isinstance(other, int):
            return stamp(other, 0)
        raise TypeError

    def __add__(self, other):
        o = self._coerce(other)
        return stamp(self.secs + o.secs, self.frac_secs + o.frac_secs)

    def __sub__(self, other):
        o = self._coerce(other)
        return stamp(self.secs - o.secs, self.frac_secs - o.frac_secs)


# one received Mode S reply: parsed fields, CRC syndrome, signal level in dB, time of arrival
modes_report = namedtuple("modes_report", ["data", "ecc", "rssi", "timestamp"])
llh = namedtuple("llh", ["lat", "lon", "alt"])
mlat_report = namedtuple("mlat_report", ["data", "nreps", "timestamp", "llh", "hdop", "vdop"])
