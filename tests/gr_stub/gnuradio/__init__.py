"""TEST STUB of the slice of GNU Radio's Python API that INTEGRATION.md's binding touches (GNU Radio is not installed in
this image): gr.sync_block with work() / stop() / get_tags_in_window, gr.message_from_string, gr.msg_queue, gr.tag_t,
and a scheduler-like driver (run_sink) that hands a sink block its input in chunks.  Test infrastructure only."""
from . import gr  # noqa: F401
