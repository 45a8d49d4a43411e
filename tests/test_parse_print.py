"""Message consumers (parser + printer + CPR + altitude) against vectors produced by the reference's
own Python modules (tools/gen_parse_golden.py -> tests/golden/parse_print.json): same printed line
(or no line) for every message, and the same escaping exception where the reference has a bug."""
import json
import os

import pytest

from conftest import GOLDEN

from air_modes import altitude, cpr, msprint, parse
from air_modes.exceptions import CPRBoundaryStraddleError, CPRNoPositionError, MetricAltError
from air_modes.pubsub import pubsub


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "parse_print.json")) as f:
        return json.load(f)


def replay(records, location):
    pub = pubsub()
    lines = []
    msprint.output_print(cpr.cpr_decoder(location), pub, callback=lines.append)
    feed = parse.make_parser(pub)
    for rec in records:
        del lines[:]
        exc = None
        try:
            feed(rec["msg"])
        except Exception as e:
            exc = type(e).__name__
        yield rec, list(lines), exc


@pytest.mark.parametrize("run", [0, 1])
def test_printed_lines_match_reference(golden, run):
    doc = golden["runs"][run]
    n_lines = 0
    for rec, lines, exc in replay(doc["records"], doc["location"]):
        assert lines == rec["out"], rec["msg"]
        assert exc == rec["exc"], rec["msg"]
        n_lines += len(lines)
    assert n_lines > 400                      # the corpus does exercise the printer


def test_corpus_covers_every_line_kind(golden):
    text = "\n".join(l for r in golden["runs"] for x in r["records"] for l in x["out"])
    for needle in ("Type 0 ", "Type 4 ", "Type 5 ", "Type 11 ", "BDS0,8", "BDS0,6", "BDS0,5", "BDS0,9-0", "BDS0,9-1",
                   "BDS0,9-3", "not implemented", "link capability", "identification from", "TCAS report",
                   "threat ID", "range:", "No handler for message type 24", "No handler in type", " @ "):
        assert needle in text, needle


def test_altitude_round_trip_and_metric():
    # the reference's own self-test (altitude.py:129-144), with the encoder made integer-safe
    for b13 in (False, True):
        for alt in range(-1000, 101400, 25):
            if alt >= 50175 and not b13:
                break                                        # 11 bits of 25 ft steps
            assert altitude.decode_alt(altitude.encode_alt_modes(alt, b13), b13) == alt
    with pytest.raises(MetricAltError):
        altitude.decode_alt(0x0040 | 0x0010, True)
    assert altitude.gray2bin(0b110) == 0b100


def test_cpr_round_trip():
    # the reference's own self-test (cpr.py:260-332), thinned out: global decode of an even/odd pair
    # and a local decode land within 1e-3 degrees of the encoded position
    rounds, straddles = 2001, 0
    for i in range(rounds):
        even = (i / (rounds / 170.) - 85, i / (rounds / 360.) - 180)
        odd = (even[0] + 1e-3, min(even[1] + 1e-3, 180))
        dec = cpr.cpr_decoder([odd[0], odd[1]])
        with pytest.raises(CPRNoPositionError):
            dec.decode(i, *cpr.cpr_encode(even[0], even[1], False, False), False, False)
        try:
            lat, lon, rng, brg = dec.decode(i, *cpr.cpr_encode(odd[0], odd[1], True, False), True, False)
        except CPRBoundaryStraddleError:
            straddles += 1
            continue
        assert abs(lat - odd[0]) < 1e-3 and abs(lon - odd[1]) < 1e-3
        nxt = (odd[0] + 1e-3, min(odd[1] + 1e-3, 180))
        llat, llon = cpr.cpr_resolve_local(list(even), list(cpr.cpr_encode(nxt[0], nxt[1], False, False)), False, False)
        assert abs(llat - nxt[0]) < 1e-3 and abs(llon - nxt[1]) < 1e-3
    assert straddles < rounds // 20


def test_parser_surface():
    r = parse.modes_reply(int("8D4840D6202CC371C32CE0576098", 16))      # DF17 identification, KLM1023
    assert r.get_type() == 17 and r["aa"] == 0x4840D6 and r["me"].get_type() == 0x08
    assert parse.parseBDS08(r)[0] == "KLM1023 "
    with pytest.raises(parse.FieldNotInPacket):
        r["ac"]
    with pytest.raises(parse.NoHandlerError):
        parse.modes_reply(int("C0000000000000", 16))                    # DF24 is known, DF 22 is not
        parse.modes_reply(22 << 51)


def test_modes_rx_cli_end_to_end(emu_lib, oracle_mod, tmp_path, monkeypatch):
    """apps/modes_rx, file source: IQ file -> (emulated) GPU path -> parser -> printed lines.
    The raw messages must equal the oracle's, the parsed lines what the consumers make of them."""
    import io
    import numpy as np
    import synth
    from conftest import EMU_LIB
    from air_modes import modes_rx
    rate = 2e6
    iq, _ = synth.synth_capture(rate, 400000, 600.0, seed=4242)
    path = tmp_path / "cap.cf32"
    np.asarray(iq, dtype=np.complex64).tofile(path)
    monkeypatch.setenv("AIRMODES_HIP_LIB", EMU_LIB)          # test infrastructure: kernels on CPU fibers
    raw = io.StringIO()
    assert modes_rx.main(["-s", str(path), "-r", "2e6", "--raw", "--chunk", "150000", "--no-resample"], out=raw) == 0
    want = oracle_mod.format_messages(oracle_mod.demod(iq, rate, 7.0, True), rate)
    got = raw.getvalue().splitlines()
    assert got == want and len(got) > 20
    parsed = io.StringIO()
    assert modes_rx.main(["-s", str(path), "-r", "2e6", "-l", "37.7,-122.4", "--no-resample"], out=parsed) == 0
    pub = pubsub()
    lines = []
    msprint.output_print(cpr.cpr_decoder([37.7, -122.4]), pub, callback=lines.append)
    feed = parse.make_parser(pub)
    for m in want:
        try:
            feed(m)
        except IndexError:              # a reference table bug the command line survives (see modes_rx.py)
            pass
    assert parsed.getvalue().splitlines() == lines and len(lines) > 5
