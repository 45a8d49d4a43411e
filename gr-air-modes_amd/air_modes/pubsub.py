"""Minimal publish/subscribe dictionary.

The reference wires its parser and output plugins together with gnuradio.gr.pubsub
(apps/modes_rx:27,68-70; python/parse.py:422-436; python/msprint.py:33-39).  GNU Radio is not a
dependency here, so this is the part of that contract the message consumers use: assigning
pub[key] = value stores the value and calls every subscriber of key with it, in subscription
order, in the caller's thread.
"""


class pubsub(dict):
    def __init__(self):
        super().__init__()
        self._subscribers = {}

    def subscribe(self, key, subscriber):
        self._subscribers.setdefault(key, []).append(subscriber)
        if key not in self:
            dict.__setitem__(self, key, None)

    def unsubscribe(self, key, subscriber):
        self._subscribers.get(key, []).remove(subscriber)

    def __setitem__(self, key, value):
        dict.__setitem__(self, key, value)
        for fn in tuple(self._subscribers.get(key, ())):
            fn(value)
