"""Stand-ins for gr.message / gr.msg_queue: the hand-off the reference's slicer uses
(lib/slicer_impl.cc:193-194 `d_queue->handle(msg)`; consumers: python/radio.py:84-87
`gru.msgq_runner(queue, cb)` -> `msg.to_string()`, python/mlat_client.py:81 `insert_tail`).
A real gr.msg_queue can be passed to rx_path/slicer instead when GNU Radio is installed:
only handle()/insert_tail() are called on it."""
import collections
import threading


class message(object):
    def __init__(self, text, type_=0, arg1=0.0, arg2=0.0):
        self._text = text
        self._type = type_
        self._arg1 = arg1
        self._arg2 = arg2

    @staticmethod
    def make_from_string(text, type_=0, arg1=0.0, arg2=0.0):
        return message(text, type_, arg1, arg2)

    def to_string(self):
        return self._text

    def type(self):
        return self._type

    def arg1(self):
        return self._arg1

    def arg2(self):
        return self._arg2

    def length(self):
        return len(self._text)


def message_from_string(text, type_=0, arg1=0.0, arg2=0.0):
    return message(text, type_, arg1, arg2)


class msg_queue(object):
    """Thread-safe FIFO of messages; limit 0 = unbounded (gr.msg_queue semantics)."""

    def __init__(self, limit=0):
        self._limit = limit
        self._q = collections.deque()
        self._cv = threading.Condition()

    def handle(self, msg):
        self.insert_tail(msg)

    def insert_tail(self, msg):
        with self._cv:
            while self._limit and len(self._q) >= self._limit:
                self._cv.wait()
            self._q.append(msg)
            self._cv.notify_all()

    def delete_head(self):
        with self._cv:
            while not self._q:
                self._cv.wait()
            m = self._q.popleft()
            self._cv.notify_all()
            return m

    def delete_head_nowait(self):
        with self._cv:
            if not self._q:
                return None
            m = self._q.popleft()
            self._cv.notify_all()
            return m

    def flush(self):
        with self._cv:
            self._q.clear()
            self._cv.notify_all()

    def empty_p(self):
        return not self._q

    def full_p(self):
        return bool(self._limit) and len(self._q) >= self._limit

    def count(self):
        return len(self._q)

    def limit(self):
        return self._limit
