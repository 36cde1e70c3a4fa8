"""Side HIP streams that run beside the main stream.

The HIP runtime multiplexes streams onto a small set of hardware queues (GPU_MAX_HW_QUEUES, 4 by default; the package
raises it to 8, see __init__.py); two streams that land on one queue execute their kernels strictly one after the other.
side_stream() takes a few candidates from torch's pool and keeps the first one that is MEASURED to overlap with the
current stream (and with the side streams handed out before) with a pair of spin kernels.  This removes the static
collisions (every 4th..5th pool stream shares the main stream's queue, tools/debug/stream_probe.py); the runtime can
still re-assign queues later, which is what the larger queue count is for.
"""
import os

import torch

_CACHE = {}      # (device index, role) -> stream
_PROBE_CYCLES = 400_000
_NCAND = 8
stats = {"probes": 0, "picked": {}}


def _spin(stream, cycles):
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cycles)


def _pair_ms(s0, s1, cycles):
    """time of one spin on s0 and one on s1, started together (s1 == s0: serial reference)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if s1 is not s0:
        s1.wait_stream(s0)
    e0.record(s0)
    _spin(s0, cycles)
    _spin(s1, cycles)
    if s1 is not s0:
        s0.wait_stream(s1)
    e1.record(s0)
    e1.synchronize()
    return e0.elapsed_time(e1)


def overlaps(s0, s1):
    """True when kernels on s0 and s1 execute concurrently (measured)."""
    if s0 == s1:
        return False
    torch.cuda.synchronize(s0.device)
    cycles = _PROBE_CYCLES
    serial = _pair_ms(s0, s0, cycles)
    while serial < 0.2 and cycles < (1 << 28):   # keep the probe well above launch latency
        cycles *= 4
        serial = _pair_ms(s0, s0, cycles)
    stats["probes"] += 1
    return min(_pair_ms(s0, s1, cycles), _pair_ms(s0, s1, cycles)) < 0.75 * serial


def side_stream(role, main=None):
    """Cached side stream for `role` ("text", "wgrad", "comm") on the current device, concurrent with `main` (default: the
    current stream) and, where possible, with the side streams of the other roles."""
    main = main or torch.cuda.current_stream()
    key = (main.device.index, role)
    st = _CACHE.get(key)
    if st is not None:
        return st
    others = [s for (d, r), s in _CACHE.items() if d == main.device.index]
    prio = os.environ.get("SEGCLIP_WGRAD_PRIO") if role == "wgrad" else None   # experiment: low-priority gap filler
    cands = [torch.cuda.Stream(device=main.device, priority=int(prio)) if prio is not None else torch.cuda.Stream(device=main.device)
             for _ in range(_NCAND)]
    cands = [c for c in cands if c != main and all(c != o for o in others)]
    good = [c for c in cands if overlaps(main, c)]
    best = next((c for c in good if all(overlaps(o, c) for o in others)), None)
    st = best or (good[0] if good else cands[0])
    stats["picked"][role] = {"concurrent_with_main": bool(good), "concurrent_with_all": best is not None}
    _CACHE[key] = st
    return st


def reset():
    _CACHE.clear()
