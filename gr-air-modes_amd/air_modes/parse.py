"""Mode S reply parser (reference: python/parse.py:29-436).

A reply is an integer; a record class describes, per (sub)type, where its named fields sit.
Field positions are written as in the ICAO Annex 10 / DO-260B tables: 1-based, counted from the
most significant bit of the enclosing field ("name:first_bit:width[:record]").  Sub-records
(ME, MB, BDS0,9 and the TCAS threat identity) number their bits in the coordinates of the
documents that define them, hence the per-class `offset`.

The behaviour mirrors the reference, including what it drops: unknown formats raise
NoHandlerError from the constructor, make_parser() swallows every ADSBError.
"""
import math

from .altitude import decode_alt
from .exceptions import ADSBError, FieldNotInPacket, NoHandlerError
from .modes_types import modes_report, stamp


def _layout(spec, records=None):
    """'a:1:5 b:6:3:sub' -> {'a': (1, 5), 'b': (6, 3, <class sub>)} (insertion order kept)."""
    out = {}
    for item in spec.split():
        parts = item.split(":")
        entry = (int(parts[1]), int(parts[2]))
        if len(parts) == 4:
            entry += (records[parts[3]],)
        out[parts[0]] = entry
    return out


class data_field:
    """A bit string that can name its own fields (parse.py:29-86)."""

    dtypes = {}      # type -> {field: (first_bit, width[, record class])}
    offset = 1       # number of the first bit of this record in its defining document

    def __init__(self, data):
        self.data = data
        self.fields = self.parse()

    def get_type(self):
        raise NotImplementedError

    def get_numbits(self):
        raise NotImplementedError

    def get_bits(self, startbit, num):
        shift = self.get_numbits() - startbit - num + self.offset
        if shift < 0:
            # a short frame that announces a long format: the reference swallows the negative
            # shift (ValueError) and reads the field as zero (parse.py:73-86)
            return 0
        return (self.data >> shift) & ((1 << num) - 1)

    def parse(self):
        kind = self.get_type()
        if kind not in self.dtypes:
            raise NoHandlerError(kind)
        found = {}
        for name, where in self.dtypes[kind].items():
            value = self.get_bits(where[0], where[1])
            if len(where) == 3:
                value = where[2](value)
                found.update(value.parse())      # a sub-record's fields are also visible here
            found[name] = value
        return found

    def __getitem__(self, fieldname):
        kind = self.get_type()
        if kind not in self.dtypes:
            raise NoHandlerError(kind)
        if fieldname not in self.fields:
            raise FieldNotInPacket(fieldname)
        return self.fields[fieldname]


class bds09_reply(data_field):
    """Airborne velocity, the 51 bits after the format type code of a BDS0,9 squitter."""
    offset = 6
    _gs = "sub:6:3 icf:9:1 ifr:10:1 nuc:11:3 dew:14:1 vew:15:10 dns:25:1 vns:26:10 vrsrc:36:1 dvr:37:1 " \
          "vr:38:9 dhd:49:1 hd:50:6"
    _as = "sub:6:3 icf:9:1 ifr:10:1 nuc:11:3 mhs:14:1 hdg:15:10 ast:25:1 spd:26:10 vrsrc:36:1 dvr:37:1 " \
          "vr:38:9 dhd:49:1 hd:50:6"
    dtypes = {0: _layout("sub:6:3 dew:10:1 vew:11:11 dns:22:1 vns:23:11 str:34:1 tr:35:6 dvr:41:1 vr:42:9"),
              1: _layout(_gs),         # subtypes 1-2: velocity over ground (2 = supersonic scale)
              3: _layout(_as)}         # subtypes 3-4: airspeed and heading

    def get_type(self):
        sub = self.get_bits(6, 3)
        return {0: 0, 1: 1, 2: 1, 3: 3, 4: 3}.get(sub)       # 5-7: None -> no handler

    def get_numbits(self):
        return 51


class me_reply(data_field):
    """The 56-bit ME field of an extended squitter, keyed by the BDS register it reports."""
    _pos = "time:21:1 cpr:22:1 lat:23:17 lon:40:17"
    dtypes = {0x05: _layout("ftc:1:5 ss:6:2 saf:8:1 alt:9:12 " + _pos),              # airborne position
              0x06: _layout("ftc:1:5 mvt:6:7 gts:13:1 gtk:14:7 " + _pos),            # surface position
              0x07: _layout("ftc:1:5"),                                              # status (unused)
              0x08: _layout("ftc:1:5 cat:6:3 ident:9:48"),                           # identification
              0x09: _layout("ftc:1:5 bds09:6:51:bds09", {"bds09": bds09_reply}),      # velocity
              0x61: _layout("ftc:1:5 eps:9:3")}                                      # emergency status

    def get_type(self):
        ftc = self.get_bits(1, 5)
        if 1 <= ftc <= 4:
            return 0x08
        if 5 <= ftc <= 8:
            return 0x06
        if 9 <= ftc <= 18 and ftc != 15:
            return 0x05
        if ftc == 19:
            return 0x09
        if ftc == 28:
            return 0x61
        return None                      # no table: parse() raises NoHandlerError

    def get_numbits(self):
        return 56


class tcas_reply(data_field):
    """Threat identity data of a resolution advisory report, by threat type indicator."""
    offset = 61
    dtypes = {0: _layout("tti:61:2"),
              1: _layout("tti:61:2 tid:63:26"),
              2: _layout("tti:61:2 tida:63:13 tidr:76:7 tidb:83:6")}

    def get_type(self):
        return self.get_bits(61, 2)

    def get_numbits(self):
        return 28


class mb_reply(data_field):
    """The 56-bit MB field of a Comm-B reply (DF20/21), by BDS1 code (BDS2 must be 0)."""
    offset = 33
    dtypes = {0: _layout("bds1:33:4 bds2:37:4"),
              1: _layout("bds1:33:4 bds2:37:4 cfs:41:4 acs:45:20 bcs:65:16 ecs:81:8"),
              2: _layout("bds1:33:4 bds2:37:4 ais:41:48"),
              3: _layout("bds1:33:4 bds2:37:4 ara:41:14 rac:55:4 rat:59:1 mte:60:1 tcas:61:28:tcas",
                         {"tcas": tcas_reply})}

    def get_type(self):
        bds1, bds2 = self.get_bits(33, 4), self.get_bits(37, 4)
        if bds1 > 3 or bds2 != 0:
            raise NoHandlerError(bds1)
        return int(bds1)

    def get_numbits(self):
        return 56


class mv_reply(data_field):
    """The MV field of a long air-air reply (DF16).  As in the reference (parse.py:198-213) the
    table is not keyed by type, so no instance can be built; DF16 keeps MV as plain bits."""
    offset = 33
    dtypes = _layout("ara:41:14 mte:60:1 rac:55:4 rat:59:1 vds:33:8 vds1:33:4 vds2:37:4")

    def get_type(self):
        vds1, vds2 = self.get_bits(33, 4), self.get_bits(37, 4)
        if vds1 != 3 or vds2 != 0:
            raise NoHandlerError(vds1)
        return int(vds1)

    def get_numbits(self):
        return 56


class modes_reply(data_field):
    """A whole 56- or 112-bit reply, by downlink format."""
    _surv = "df:1:5 fs:6:3 dr:9:5 um:14:6"
    dtypes = {0: _layout("df:1:5 vs:6:1 cc:7:1 sl:9:3 ri:14:4 ac:20:13 ap:33:24"),
              4: _layout(_surv + " ac:20:13 ap:33:24"),
              5: _layout(_surv + " id:20:13 ap:33:24"),
              11: _layout("df:1:5 ca:6:3 aa:9:24 pi:33:24"),
              16: _layout("df:1:5 vs:6:1 sl:9:3 ri:14:4 ac:20:13 mv:33:56 ap:88:24"),
              17: _layout("df:1:5 ca:6:3 aa:9:24 me:33:56:me pi:88:24", {"me": me_reply}),
              20: _layout(_surv + " ac:20:13 mb:33:56:mb ap:88:24", {"mb": mb_reply}),
              21: _layout(_surv + " id:20:13 mb:33:56:mb ap:88:24", {"mb": mb_reply}),
              24: _layout("df:1:5 ke:6:1 nd:7:4 md:11:80 ap:88:24")}

    def is_long(self):
        return self.data > (1 << 56)

    def get_numbits(self):
        return 112 if self.is_long() else 56

    def get_type(self):
        return self.get_bits(1, 5)


# ---------------------------------------------------------------------------------------------
# field interpreters
# ---------------------------------------------------------------------------------------------
def decode_id(id):
    """Mode A identity code (squawk) as a decimal number with the digits A B C D
    (parse.py:236-259).  The 13 bits arrive as C1 A1 C2 A2 C4 A4 X B1 D1 B2 D2 B4 D4.  The D digit
    weights its pulses D1, D2, D4 as 4, 2, 4 there (not 1, 2, 4); this mirrors it, because the
    printed ident must be the same."""
    def digit(weighted):
        return sum(w for mask, w in weighted if id & mask)
    a = digit(((0x0800, 1), (0x0200, 2), (0x0080, 4)))
    b = digit(((0x0020, 1), (0x0008, 2), (0x0002, 4)))
    c = digit(((0x1000, 1), (0x0400, 2), (0x0100, 4)))
    d = digit(((0x0010, 4), (0x0004, 2), (0x0001, 4)))
    return a * 1000 + b * 100 + c * 10 + d


def charmap(d):
    """One 6-bit character of the aircraft identification alphabet (parse.py:262-272)."""
    if 0 < d < 27:
        return chr(ord("A") + d - 1)
    if 47 < d < 58:
        return chr(ord("0") + d - 48)
    return " "


def _ident_text(bits48):
    return "".join(charmap((bits48 >> (42 - 6 * i)) & 0x3F) for i in range(8))


_CATEGORIES = (
    ("NO INFO", "RESERVED", "RESERVED", "RESERVED", "RESERVED", "RESERVED", "RESERVED", "RESERVED"),
    ("NO INFO", "SURFACE EMERGENCY VEHICLE", "SURFACE SERVICE VEHICLE", "FIXED OBSTRUCTION", "CLUSTER OBSTRUCTION",
     "LINE OBSTRUCTION", "RESERVED"),
    ("NO INFO", "GLIDER", "BALLOON/BLIMP", "PARACHUTE", "ULTRALIGHT", "RESERVED", "UAV", "SPACECRAFT"),
    ("NO INFO", "LIGHT", "SMALL", "LARGE", "LARGE HIGH VORTEX", "HEAVY", "HIGH PERFORMANCE", "ROTORCRAFT"))


def parseBDS08(data):
    """(callsign, emitter category text) of an identification squitter (parse.py:274-285)."""
    return (_ident_text(data["ident"]), _CATEGORIES[data["ftc"] - 1][data["cat"]])


def parseBDS05(data, cprdec):
    """[altitude ft, lat, lon, range, bearing] of an airborne position; needs the stateful CPR
    decoder (parse.py:288-291)."""
    altitude = decode_alt(data["alt"], False)
    return [altitude] + cprdec.decode(data["aa"], data["lat"], data["lon"], data["cpr"], 0)


def parseBDS06(data, cprdec):
    """[ground track deg, lat, lon, range, bearing] of a surface position (parse.py:294-297)."""
    ground_track = data["gtk"] * 360. / 128
    return [ground_track] + cprdec.decode(data["aa"], data["lat"], data["lon"], data["cpr"], 1)


def _signed(value, negative):
    return 0 - value if negative else value


def parseBDS09_0(data):
    """[speed kt, heading deg, vertical rate ft/min, turn rate] of a subtype-0 velocity report
    (parse.py:299-324)."""
    vert_spd = _signed(data["vr"] * 32, bool(data["dvr"]))
    turn_rate = _signed(data["tr"] * 15 / 62, data["str"])
    ns_vel, ew_vel = data["vns"] - 1, data["vew"] - 1
    velocity = math.hypot(ns_vel, ew_vel)
    ew_vel = _signed(ew_vel, bool(data["dew"]))
    ns_vel = _signed(ns_vel, bool(data["dns"]))
    heading = math.atan2(ew_vel, ns_vel) * (180.0 / math.pi)
    if heading < 0:
        heading += 360
    return [velocity, heading, vert_spd, turn_rate]


def parseBDS09_1(data):
    """[speed kt, heading deg, vertical rate ft/min] of a ground-velocity report, subtypes 1-2
    (parse.py:326-361)."""
    vert_spd = _signed(float(data["vr"] - 1) * 64, bool(data["dvr"]))
    ns_vel, ew_vel = float(data["vns"]), float(data["vew"])
    if data["sub"] == 0x02:
        ns_vel *= 4
        ew_vel *= 4
    velocity = math.hypot(ns_vel, ew_vel)
    ew_vel = _signed(ew_vel, bool(data["dew"]))
    heading = 0 if ns_vel == 0 else math.atan(float(ew_vel) / float(ns_vel)) * (180.0 / math.pi)
    if bool(data["dns"]):
        heading = 180 - heading
    if heading < 0:
        heading += 360
    return [velocity, heading, vert_spd]


def parseBDS09_3(data):
    """[magnetic heading, 'TAS'|'IAS', airspeed kt, vertical rate, geometric-baro difference ft]
    of an airspeed report, subtypes 3-4 (parse.py:363-376; the heading is formed from the
    heading STATUS bit there, and so it is here)."""
    mag_hdg = data["mhs"] * 360. / 1024
    vel = data["spd"] * 4 if data["sub"] == 4 else data["spd"]
    vert_spd = _signed(float(data["vr"] - 1) * 64, data["dvr"] == 1)
    return [mag_hdg, "TAS" if data["ast"] == 1 else "IAS", vel, vert_spd, float(data["hd"] - 1) * 25]


_EMERGENCY = ("NO EMERGENCY", "GENERAL EMERGENCY", "LIFEGUARD/MEDICAL", "FUEL EMERGENCY", "NO COMMUNICATIONS",
              "UNLAWFUL INTERFERENCE", "RESERVED", "RESERVED")


def parseBDS62(data):
    return _EMERGENCY[data["eps"]]


def parseMB_id(data):
    """Callsign from a BDS2,0 Comm-B reply."""
    return _ident_text(data["ais"])


_ARA_TEXT = ("CLIMB", "DON'T DESCEND", "DON'T DESCEND >500FPM", "DON'T DESCEND >1000FPM", "DON'T DESCEND >2000FPM",
             "DESCEND", "DON'T CLIMB", "DON'T CLIMB >500FPM", "DON'T CLIMB >1000FPM", "DON'T CLIMB >2000FPM",
             "TURN LEFT", "TURN RIGHT", "DON'T TURN LEFT", "DON'T TURN RIGHT")        # MB bits 41..54
_RAC_TEXT = ("DON'T DESCEND", "DON'T CLIMB", "DON'T TURN LEFT", "DON'T TURN RIGHT")  # MB bits 55..58


def parseMB_TCAS_resolutions(data):
    """Active resolution advisories and complements as ' A B' strings (parse.py:387-406)."""
    def listed(value, texts):
        top = len(texts) - 1
        return "".join(" " + t for i, t in enumerate(texts) if value & (1 << (top - i)))
    return (listed(data["ara"], _ARA_TEXT), listed(data["rac"], _RAC_TEXT))


def parseMB_TCAS_threatid(data):
    """TTI = 1: the threat is named by its address."""
    res, comp = parseMB_TCAS_resolutions(data)
    return (res, comp, data["rat"], data["mte"], data["tid"])


def parseMB_TCAS_threatloc(data):
    """TTI = 2: the threat is given as altitude / range / bearing."""
    res, comp = parseMB_TCAS_resolutions(data)
    return (res, comp, data["rat"], data["mte"], decode_alt(data["tida"], True), data["tidr"], data["tidb"])


def parse_TCAS_CRM(data):
    """DF16 coordination reply message."""
    res, comp = parseMB_TCAS_resolutions(data)
    return (res, comp, data["rat"], data["mte"])


def make_parser(pub):
    """Returns the subscriber for the slicer's text messages ("<hex> <syndrome> <level> <secs>
    <frac>", slicer_impl.cc:176-193): parses one message and publishes the modes_report under
    "modes_dl" and "type<DF>_dl" (parse.py:422-436).  Anything wrong with the frame is dropped."""
    def publish(message):
        data, ecc, reference, int_timestamp, frac_timestamp = message.split()
        try:
            report = modes_report(modes_reply(int(data, 16)), int(ecc, 16),
                                  10.0 * math.log10(max(1e-8, float(reference))),
                                  stamp(int(int_timestamp), float(frac_timestamp)))
            pub["modes_dl"] = report
            pub["type%i_dl" % report.data.get_type()] = report
        except ADSBError:
            pass
    return publish
