"""Human-readable report lines, the format `modes_rx` prints (reference: python/msprint.py:29-236).

output_print subscribes to the parser's "type<DF>_dl" topics and emits one line per report it can
describe.  Which reports produce NO line is part of the format (flight status 0 in DF4/5, DF16,
metric altitudes, unresolved positions ...): those paths raise ADSBError, which either the handler
or make_parser() swallows, exactly as in the reference.
"""
from .altitude import decode_alt
from .exceptions import ADSBError
from . import parse as _p

_RI_TEXT = {0: " (No TCAS)", 2: " (TCAS resolution inhibited)", 3: " (Vertical TCAS resolution only)",
            4: " (Full TCAS resolution)", 9: " (speed <75kt)"}
_FS_TEXT = {1: " (aircraft is on the ground)", 2: " (AIRBORNE ALERT)", 3: " (GROUND ALERT)", 4: " (SPI ALERT)",
            5: " (SPI)"}


class output_print:
    HANDLED = (0, 4, 5, 11, 16, 17, 20, 21)

    def __init__(self, cpr, publisher, callback=None):
        self._cpr = cpr
        self._callback = callback
        self._fns = list(self.HANDLED)
        for df in self.HANDLED:
            publisher.subscribe("type%i_dl" % df, getattr(self, "handle%i" % df))
        publisher.subscribe("modes_dl", self.catch_nohandler)

    @staticmethod
    def prefix(msg):
        return "(%i %.8f) " % (msg.rssi, msg.timestamp)

    def _print(self, msg):
        if self._callback is None:
            print(msg)
        else:
            self._callback(msg)

    def catch_nohandler(self, msg):
        df = msg.data.get_type()
        if df in self._fns:
            return
        who = msg.data["aa"] if "aa" in msg.data.fields else msg.ecc
        self._print(self.prefix(msg) + "No handler for message type %i" % df + " from %.6x" % who)

    @staticmethod
    def fs_text(fs):
        if fs not in _FS_TEXT:
            raise ADSBError
        return _FS_TEXT[fs]

    def handle0(self, msg):
        try:
            line = self.prefix(msg) + "Type 0 (short A-A surveillance) from %x at %ift" % (
                msg.ecc, decode_alt(msg.data["ac"], True))
            ri = msg.data["ri"]
            if ri in _RI_TEXT:
                line += _RI_TEXT[ri]
            elif ri > 9:
                line += " (speed %i-%ikt)" % (75 * (1 << (ri - 10)), 75 * (1 << (ri - 9)))
            else:
                raise ADSBError
        except ADSBError:
            return
        if msg.data["vs"] == 1:
            line += " (aircraft is on the ground)"
        self._print(line)

    def handle4(self, msg):
        try:
            line = self.prefix(msg) + "Type 4 (short surveillance altitude reply) from %x at %ift" % (
                msg.ecc, decode_alt(msg.data["ac"], True))
            line += self.fs_text(msg.data["fs"])
        except ADSBError:
            return
        self._print(line)

    def handle5(self, msg):
        try:
            line = self.prefix(msg) + "Type 5 (short surveillance ident reply) from %x with ident %i" % (
                msg.ecc, _p.decode_id(msg.data["id"]))
            line += self.fs_text(msg.data["fs"])
        except ADSBError:
            return
        self._print(line)

    def handle11(self, msg):
        try:
            line = self.prefix(msg) + \
                "Type 11 (all call reply) from %x in reply to interrogator %i with capability level %i" % (
                    msg.data["aa"], msg.ecc & 0xF, msg.data["ca"] + 1)
        except ADSBError:
            return
        self._print(line)

    def handle17(self, msg):
        icao24 = msg.data["aa"]
        bdsreg = msg.data["me"].get_type()
        line = self.prefix(msg)
        try:
            if bdsreg == 0x08:
                ident, typestring = _p.parseBDS08(msg.data)
                line += "Type 17 BDS0,8 (ident) from %x type %s ident %s" % (icao24, typestring, ident)
            elif bdsreg == 0x06:
                track, lat, lon, rnge, bearing = _p.parseBDS06(msg.data, self._cpr)
                line += "Type 17 BDS0,6 (surface report) from %x at (%.6f, %.6f) ground track %i" % (
                    icao24, lat, lon, track)
                if rnge is not None and bearing is not None:
                    line += " (%.2f @ %.0f)" % (rnge, bearing)
            elif bdsreg == 0x05:
                altitude, lat, lon, rnge, bearing = _p.parseBDS05(msg.data, self._cpr)
                line += "Type 17 BDS0,5 (position report) from %x at (%.6f, %.6f)" % (icao24, lat, lon)
                if rnge is not None and bearing is not None:
                    line += " (%.2f @ %.0f)" % (rnge, bearing)
                line += " at " + str(altitude) + "ft"
            elif bdsreg == 0x09:
                subtype = msg.data["bds09"].get_type()
                if subtype == 0:
                    line += "Type 17 BDS0,9-%i (track report) from %x with velocity %.0fkt heading %.0f VS %.0f " \
                            "turn rate %.0f" % ((subtype, icao24) + tuple(_p.parseBDS09_0(msg.data)))
                elif subtype == 1:
                    line += "Type 17 BDS0,9-%i (track report) from %x with velocity %.0fkt heading %.0f VS %.0f" % (
                        (subtype, icao24) + tuple(_p.parseBDS09_1(msg.data)))
                elif subtype == 3:
                    mag_hdg, vel_src, vel, vert_spd, geo_diff = _p.parseBDS09_3(msg.data)
                    line += "Type 17 BDS0,9-%i (air course report) from %x with %s %.0fkt magnetic heading %.0f " \
                            "VS %.0f geo. diff. from baro. alt. %.0fft" % (
                                subtype, icao24, vel_src, vel, mag_hdg, vert_spd, geo_diff)
                else:
                    line += "Type 17 BDS0,9-%i from %x not implemented" % (subtype, icao24)
            else:
                # (emergency status, register 0x61, lands here as well: the reference tests for 0x62)
                line += "Type 17 with FTC=%i from %x not implemented" % (msg.data["ftc"], icao24)
        except ADSBError:
            return
        self._print(line)

    def printTCAS(self, msg):
        df = msg.data["df"]
        if df == 16:
            bds1, bds2 = msg.data["vds1"], msg.data["vds2"]    # not in the DF16 table: FieldNotInPacket
        else:
            bds1, bds2 = msg.data["bds1"], msg.data["bds2"]
        line = self.prefix(msg)
        if bds2 != 0:
            line += "No handler in type %i for BDS2 == %i from %x" % (df, bds2, msg.ecc)
        elif bds1 == 0:
            line += "No handler in type %i for BDS1 == 0 from %x" % (df, msg.ecc)
        elif bds1 == 1:
            line += "Type %i link capability report from %x: ACS: 0x%x, BCS: 0x%x, ECS: 0x%x, continues %i" % (
                df, msg.ecc, msg.data["acs"], msg.data["bcs"], msg.data["ecs"], msg.data["cfs"])
        elif bds1 == 2:
            line += "Type %i identification from %x with text %s" % (df, msg.ecc, _p.parseMB_id(msg.data))
        elif bds1 == 3:
            line += "Type %i TCAS report from %x: " % (df, msg.ecc)
            tti = msg.data["tti"]
            if df == 16:
                res, comp, rat, mte = _p.parse_TCAS_CRM(msg.data)
                line += "advised: %s complement: %s" % (res, comp)
            elif tti == 1:
                res, comp, rat, mte, threat_id = _p.parseMB_TCAS_threatid(msg.data)
                line += "threat ID: %x advised: %s complement: %s" % (threat_id, res, comp)
            elif tti == 2:
                res, comp, rat, mte, alt, rng, brg = _p.parseMB_TCAS_threatloc(msg.data)
                line += "range: %i bearing: %i alt: %i advised: %s complement: %s" % (rng, brg, alt, res, comp)
            else:
                rat = mte = 0
                line += " (no handler for TTI=%i)" % tti
            if mte == 1:
                line += " (multiple threats)"
            if rat == 1:
                line += " (resolved)"
        else:
            line += "No handler for type %i, BDS1 == %i from %x" % (df, bds1, msg.ecc)
        if df == 20 or df == 16:
            line += " at %ift" % decode_alt(msg.data["ac"], True)
        else:
            line += " ident %x" % _p.decode_id(msg.data["id"])
        self._print(line)

    handle16 = printTCAS
    handle20 = printTCAS
    handle21 = printTCAS
