#!/usr/bin/env python3
"""Thin launcher for `python -m air_modes.modes_rx` from a source checkout (see that module)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gr-air-modes_amd"))

from air_modes.modes_rx import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
