"""Compact Position Reporting (reference: python/cpr.py:31-236; RTCA DO-260B A.1.7).

The arithmetic follows the reference expression by expression (same operand order), so that the
decoded degrees are the same doubles and print identically.
"""
import math
import time

from .exceptions import CPRBoundaryStraddleError, CPRNoPositionError

LATZ = 15                       # latitude zones per quadrant (NZ)
_FULL = float(2 ** 17)          # 17-bit encodings


def nz(ctype):
    return 4 * LATZ - ctype


def dlat(ctype, surface):
    span = 90.0 if surface == 1 else 360.0
    zones = nz(ctype)
    return span if zones == 0 else span / zones


def nl(declat_in):
    """Number of longitude zones at a latitude (cpr.py:47-50)."""
    if abs(declat_in) >= 87.0:
        return 1.0
    a = 1.0 - math.cos(math.pi / (2.0 * LATZ))
    b = math.cos((math.pi / 180.0) * abs(declat_in)) ** 2
    return math.floor((2.0 * math.pi) * math.acos(1.0 - a / b) ** -1)


def dlon(declat_in, ctype, surface):
    span = 90.0 if surface else 360.0
    return span / max(nl(declat_in) - ctype, 1)


def _nearest_zone(size, here, frac):
    # zone index whose encoded fraction `frac` lies closest to the reference coordinate `here`
    return math.floor(here / size) + math.floor(0.5 + ((here % size) / size) - frac)


def decode_lat(enclat, ctype, my_lat, surface):
    size = dlat(ctype, surface)
    frac = float(enclat) / (2 ** 17)
    return size * (_nearest_zone(size, my_lat, frac) + frac)


def decode_lon(declat, enclon, ctype, my_lon, surface):
    size = dlon(declat, ctype, surface)
    frac = float(enclon) / (2 ** 17)
    return size * (_nearest_zone(size, my_lon, frac) + frac)


def cpr_resolve_local(my_location, encoded_location, ctype, surface):
    """Position from ONE report and a reference position within half a zone (cpr.py:70-77)."""
    lat = decode_lat(encoded_location[0], ctype, my_location[0], surface)
    lon = decode_lon(lat, encoded_location[1], ctype, my_location[1], surface)
    return [lat, lon]


def cpr_resolve_global(evenpos, oddpos, mypos, mostrecent, surface):
    """Position from an even/odd pair (cpr.py:79-150).  `mostrecent` selects the report (0 even,
    1 odd) whose position is returned."""
    if surface and mypos is None:
        raise CPRNoPositionError        # a surface pair is ambiguous without the receiver position
    d_even, d_odd = dlat(0, surface), dlat(1, surface)
    ev = [float(evenpos[0]), float(evenpos[1])]
    od = [float(oddpos[0]), float(oddpos[1])]

    j = math.floor(((nz(1) * ev[0] - nz(0) * od[0]) / 2 ** 17) + 0.5)
    rlat_even = d_even * ((j % nz(0)) + ev[0] / 2 ** 17)
    rlat_odd = d_odd * ((j % nz(1)) + od[0] / 2 ** 17)
    if rlat_even > 270.0:
        rlat_even -= 360.0
    if rlat_odd > 270.0:
        rlat_odd -= 360.0
    if nl(rlat_even) != nl(rlat_odd):
        raise CPRBoundaryStraddleError
    rlat = rlat_even if mostrecent == 0 else rlat_odd
    if surface and mypos[0] < 0:
        rlat -= 90

    dl = dlon(rlat, mostrecent, surface)
    zones = nl(rlat)
    m = math.floor(((ev[1] * (zones - 1) - od[1] * zones) / 2 ** 17) + 0.5)
    enclon = ev[1] if mostrecent == 0 else od[1]
    rlon = dl * ((m % max(zones - mostrecent, 1)) + enclon / 2. ** 17)
    if surface:
        # nearest 90-degree segment to the receiver; cpr.py:136 as written under Python 3 (true
        # division): the "zone" of x is 90 * (int(x) / 90)
        wat = mypos[1]
        if wat < 0:
            wat += 360
        rlon += (90 * (int(wat) / 90) - 90 * (int(rlon) / 90))
    if rlon > 180:
        rlon -= 360.0
    return [rlat, rlon]


def range_bearing(loc_a, loc_b):
    """Distance (statute miles) and bearing (degrees) from a to b on the WGS-84 ellipsoid, flat
    approximation around the mean latitude (cpr.py:155-179)."""
    flattening = 1 / 298.257223563
    esquared = flattening * (2 - flattening)
    earth_radius_mi = 3963.19059 * (math.pi / 180)
    delta_lat = loc_b[0] - loc_a[0]
    delta_lon = loc_b[1] - loc_a[1]
    avg_lat = ((loc_a[0] + loc_b[0]) / 2.0) * math.pi / 180
    r_meridian = earth_radius_mi * (1.0 - esquared) / pow((1.0 - esquared * pow(math.sin(avg_lat), 2)), 1.5)
    r_normal = earth_radius_mi / math.sqrt(1.0 - esquared * pow(math.sin(avg_lat), 2))
    north = r_meridian * delta_lat
    east = r_normal * math.cos(avg_lat) * delta_lon
    bearing = math.atan2(east, north) * (180.0 / math.pi)
    if bearing < 0.0:
        bearing += 360.0
    return [math.hypot(east, north), bearing]


class cpr_decoder:
    """Keeps the latest even and odd report of every aircraft (airborne: 10 s, surface: 25 s)
    and resolves positions globally from pairs (cpr.py:181-236)."""

    _MAX_AGE = {0: 10, 1: 25}

    def __init__(self, my_location):
        self.my_location = my_location
        # [surface][format] -> {icao24: [enclat, enclon, time]}
        self._seen = {0: {0: {}, 1: {}}, 1: {0: {}, 1: {}}}
        self.evenlist, self.oddlist = self._seen[0][0], self._seen[0][1]
        self.evenlist_sfc, self.oddlist_sfc = self._seen[1][0], self._seen[1][1]

    def set_location(self, new_location):
        self.my_location = new_location

    def weed_poslists(self):
        for sfc, age in self._MAX_AGE.items():
            for table in self._seen[sfc].values():
                for icao in [k for k, rec in table.items() if time.time() - rec[2] > age]:
                    del table[icao]

    def decode(self, icao24, encoded_lat, encoded_lon, cpr_format, surface):
        tables = self._seen[1 if surface else 0]
        tables[1 if cpr_format == 1 else 0][icao24] = [encoded_lat, encoded_lon, time.time()]
        self.weed_poslists()
        even, odd = tables[0].get(icao24), tables[1].get(icao24)
        if even is None or odd is None:
            raise CPRNoPositionError
        newer = (odd[2] - even[2]) > 0
        lat, lon = cpr_resolve_global(even[0:2], odd[0:2], self.my_location, newer, surface)
        if self.my_location is not None:
            rnge, bearing = range_bearing(self.my_location, [lat, lon])
        else:
            rnge = bearing = None
        return [lat, lon, rnge, bearing]


def cpr_encode(lat, lon, ctype, surface):
    """17-bit CPR encoding of a position (cpr.py:239-258)."""
    scalar = 2. ** 19 if surface is True else 2. ** 17
    size_lat = dlat(ctype, False)
    yz = math.floor(scalar * ((lat % size_lat) / size_lat) + 0.5)
    size_lon = dlon(lat, ctype, False)
    xz = math.floor(scalar * ((lon % size_lon) / size_lon) + 0.5)
    return (int(yz) & (2 ** 17 - 1), int(xz) & (2 ** 17 - 1))
