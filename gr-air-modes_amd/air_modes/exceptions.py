"""Error types of the message consumers (reference: python/exceptions.py:22-47).

Everything a malformed or unsupported Mode S message can raise derives from ADSBError, which is
what make_parser() swallows -- one bad frame must never stop the receiver.  The hierarchy:

    ADSBError
      +-- MetricAltError             altitude field with the M bit set (metric: treated as spurious)
      +-- ParserError                the message could be framed but a field could not be obtained
      |     +-- FieldNotInPacket     (.item = the field name)
      +-- NoHandlerError             no field table for this message (sub)type (.msgtype)
      +-- MlatNonConvergeError
      +-- CPRNoPositionError         a compact position report that cannot (yet) be resolved
            +-- CPRBoundaryStraddleError   even and odd reports lie in different longitude zones
"""


class ADSBError(Exception): """Base class: any problem with one received message."""


class _WithPayload(ADSBError):
    _field = "value"

    def __init__(self, payload=None):
        ADSBError.__init__(self, payload)
        setattr(self, self._field, payload)


class NoHandlerError(_WithPayload): _field = "msgtype"


class ParserError(ADSBError): """A field could not be obtained."""


class FieldNotInPacket(ParserError, _WithPayload): _field = "item"


class MetricAltError(ADSBError): """Altitude code with the M bit set."""


class CPRNoPositionError(ADSBError): """No position can be given for this report (yet)."""


class CPRBoundaryStraddleError(CPRNoPositionError): """Even / odd reports in different zones."""


class MlatNonConvergeError(ADSBError): """The multilateration solver did not converge."""
