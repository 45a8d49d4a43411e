"""gnuradio.gr stand-in for tests (see the package docstring)."""
import collections

import numpy as np


class tag_t(object):
    def __init__(self, offset, key, value, srcid=None):
        self.offset, self.key, self.value, self.srcid = int(offset), key, value, srcid


class message(object):
    def __init__(self, text):
        self._text = text

    def to_string(self):
        return self._text


def message_from_string(text):
    return message(text)


class msg_queue(object):
    def __init__(self, limit=0):
        self._q = collections.deque()

    def insert_tail(self, msg):
        self._q.append(msg)

    handle = insert_tail

    def delete_head(self):
        return self._q.popleft()

    def count(self):
        return len(self._q)

    def empty_p(self):
        return not self._q


class sync_block(object):
    def __init__(self, name, in_sig, out_sig):
        self._name, self._in_sig, self._out_sig = name, in_sig, out_sig
        self._nitems_read = 0
        self._tags = []                 # tag_t on input 0, absolute offsets

    def name(self):
        return self._name

    def nitems_read(self, which_input):
        return self._nitems_read

    def get_tags_in_window(self, which_input, rel_start, rel_end, key=None):
        lo, hi = self._nitems_read + rel_start, self._nitems_read + rel_end
        return [t for t in self._tags if lo <= t.offset < hi and (key is None or t.key == key)]

    def start(self):
        return True

    def stop(self):
        return True


def run_sink(block, samples, chunk_sizes, tags=()):
    """What the scheduler does for a sink: start(), work() over successive chunks (it consumes what work returns),
    stop() at the end of the stream."""
    block._tags = sorted(tags, key=lambda t: t.offset)
    block._nitems_read = 0
    block.start()
    pos, k = 0, 0
    n = len(samples)
    while pos < n:
        size = chunk_sizes[k % len(chunk_sizes)]
        k += 1
        chunk = np.ascontiguousarray(samples[pos:pos + size])
        used = block.work([chunk], [])
        assert 0 < used <= len(chunk)
        block._nitems_read += used
        pos += used
    block.stop()
