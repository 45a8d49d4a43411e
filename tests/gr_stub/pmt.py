"""pmt stand-in for tests: the handful of calls INTEGRATION.md's binding makes (interned symbols compare by value)."""


def intern(s):
    return ("sym", s)


string_to_symbol = intern


def from_uint64(v):
    return ("u64", int(v))


def from_double(v):
    return ("f64", float(v))


def to_uint64(p):
    assert p[0] == "u64"
    return p[1]


def to_double(p):
    assert p[0] == "f64"
    return p[1]


def make_tuple(*items):
    return ("tuple", tuple(items))


def tuple_ref(p, k):
    assert p[0] == "tuple"
    return p[1][k]
