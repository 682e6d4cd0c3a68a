"""Per-kernel-class device timing with HIP events recorded on the launch stream.

    with timing.scope("lookup"):
        ops.geo_corr_lookup(...)

Disabled (the default) a scope costs one bool test.  Enabled, a start/stop event pair is recorded on
torch's current stream — the stream every launcher of libanystereo_hip.so is given — so the elapsed
time is that kernel's device duration, not host time.  bench.py enables it over the timed steps.
"""
from __future__ import annotations

import torch

_enabled = False
_events = []  # (name, start, stop)


def enable(flag: bool) -> None:
    global _enabled
    _enabled = bool(flag)
    if flag:
        _events.clear()


def enabled() -> bool:
    return _enabled


class scope:
    __slots__ = ("name", "start")

    def __init__(self, name: str):
        self.name = name
        self.start = None

    def __enter__(self):
        if _enabled and not torch.cuda.is_current_stream_capturing():
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record()
            _events.append((self.name, self.start, stop))
            self.start = None
        return False


def collect():
    """-> {name: {"count": n, "total_ms": t}} ; synchronises the device."""
    torch.cuda.synchronize()
    out = {}
    for name, s, e in _events:
        d = out.setdefault(name, {"count": 0, "total_ms": 0.0})
        d["count"] += 1
        d["total_ms"] += s.elapsed_time(e)
    _events.clear()
    return out
