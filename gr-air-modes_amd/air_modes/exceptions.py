"""Error types of the message consumers (reference: python/exceptions.py:22-47).

Everything a malformed or unsupported Mode S message can raise derives from ADSBError, which is
what make_parser() swallows -- one bad frame must never stop the receiver.
"""


class ADSBError(Exception):
    """Base class: any problem with one received message."""


class MetricAltError(ADSBError):
    """Altitude field with the M bit set (metric altitude: treated as a spurious reply)."""


class ParserError(ADSBError):
    """The message could be framed but a field could not be obtained."""


class NoHandlerError(ADSBError):
    """No field table for this message (sub)type."""

    def __init__(self, msgtype=None):
        super().__init__(msgtype)
        self.msgtype = msgtype


class MlatNonConvergeError(ADSBError):
    pass


class CPRNoPositionError(ADSBError):
    """A compact position report that cannot (yet) be resolved to a position."""


class CPRBoundaryStraddleError(CPRNoPositionError):
    """Even and odd reports lie in different longitude zones."""


class FieldNotInPacket(ParserError):
    def __init__(self, item):
        super().__init__(item)
        self.item = item
