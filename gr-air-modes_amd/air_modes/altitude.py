"""Mode S / Mode C altitude codes (reference: python/altitude.py:28-129; RTCA DO-181D 2.2.13.1.2).

decode_alt(code, bit13): `code` is the 13-bit AC field of DF0/4/16/20 (bit13=True) or the 12-bit
altitude of an airborne-position squitter (bit13=False).  Returns feet.
"""
from .exceptions import MetricAltError

_M_BIT = 0x0040
_Q_BIT = 0x0010

# 13-bit AC field, MSB first: C1 A1 C2 A2 C4 A4 (M) B1 D1/Q B2 D2 B4 D4
_PULSE = {"C1": 0x1000, "A1": 0x0800, "C2": 0x0400, "A2": 0x0200, "C4": 0x0100, "A4": 0x0080,
          "B1": 0x0020, "D1": 0x0010, "B2": 0x0008, "D2": 0x0004, "B4": 0x0002, "D4": 0x0001}
# Gillham code: the 500 ft part is the Gray code D2 D4 A1 A2 A4 B1 B2 B4 (MSB first), the 100 ft
# part the Gray code C1 C2 C4
_GRAY500 = ("D2", "D4", "A1", "A2", "A4", "B1", "B2", "B4")
_GRAY100 = ("C1", "C2", "C4")


def gray2bin(gray):
    out = gray
    shift = gray >> 1
    while shift:
        out ^= shift
        shift >>= 1
    return out


def _gather(code, names):
    v = 0
    for n in names:
        v = (v << 1) | (1 if code & _PULSE[n] else 0)
    return v


def decode_alt(alt, bit13):
    if (alt & _M_BIT) and bit13:
        raise MetricAltError            # altitude.py:32-43: metric replies are discarded
    if alt & _Q_BIT:
        # 25 ft increments: the remaining bits, M and Q squeezed out, are a plain binary number
        if bit13:
            n = ((alt & 0x3F80) >> 2) | ((alt & 0x0020) >> 1) | (alt & 0x000F)
        else:
            n = ((alt & 0x1FE0) >> 1) | (alt & 0x000F)
        return n * 25 - 1000
    # Gillham (Mode C) code, 100 ft increments
    if bit13 is False:
        # altitude.py:66-67 as written (`&` binds looser than `<<`): the upper bits are masked with
        # 0x1F80 in place, not moved up past the missing M bit
        alt = (alt & 0x003F) | (alt & 0x1F80)
    n500 = gray2bin(_gather(alt, _GRAY500))
    n100 = gray2bin(_gather(alt, _GRAY100))
    if n100 == 7:
        n100 = 5
    if n500 & 1:
        n100 = 6 - n100                 # the 100 ft code runs backwards in odd 500 ft steps
    return n500 * 500 + n100 * 100 - 1300


def encode_alt_modes(alt, bit13):
    """25 ft (Q = 1) encoding, the inverse of decode_alt for that branch (altitude.py:114-127;
    integer division here -- the reference's `/` only worked under Python 2)."""
    n = (int(alt) + 1000) // 25
    if bit13 is True:
        hi, mid = (n & 0xFE0) << 2, (n & 0x010) << 1
    else:
        hi, mid = (n & 0xFF8) << 1, 0
    return (n & 0x0F) | hi | mid | _Q_BIT
