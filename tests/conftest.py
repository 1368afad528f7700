import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def free_port() -> str:
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests).  The former scheme -- a fixed base
    plus pid % 2000 -- sat inside Linux' ephemeral range (32768-60999): a connection of an earlier test still in TIME_WAIT,
    or any other process of the box, could hold the number (EADDRINUSE in the round-6 measurement pass)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])
