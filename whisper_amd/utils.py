"""Small host-side helpers shared by the decoding / transcribe mirrors (reference: whisper/utils.py:24-82).
The file writers of the reference (utils.py:85-318) are outside the hot-path scope."""
from __future__ import annotations

import os
import sys
import zlib
from typing import List, Optional

_ENCODING = sys.getdefaultencoding()


def make_safe(text: str) -> str:
    """Replace characters the console encoding cannot represent (utils.py:10-22)."""
    if _ENCODING == "utf-8":
        return text
    return text.encode(_ENCODING, errors="replace").decode(_ENCODING)


def exact_div(x: int, y: int) -> int:
    assert x % y == 0
    return x // y


def compression_ratio(text: str) -> float:
    """utf-8 length over zlib length (utils.py:45-47) — the repetition detector of transcribe()."""
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


def format_timestamp(seconds: float, always_include_hours: bool = False, decimal_marker: str = ".") -> str:
    assert seconds >= 0, "non-negative timestamp expected"
    ms = round(seconds * 1000.0)
    hours, ms = divmod(ms, 3_600_000)
    minutes, ms = divmod(ms, 60_000)
    secs, ms = divmod(ms, 1_000)
    head = f"{hours:02d}:" if always_include_hours or hours > 0 else ""
    return f"{head}{minutes:02d}:{secs:02d}{decimal_marker}{ms:03d}"


def get_start(segments: List[dict]) -> Optional[float]:
    for s in segments:
        for w in s["words"]:
            return w["start"]
    return segments[0]["start"] if segments else None


def get_end(segments: List[dict]) -> Optional[float]:
    for s in reversed(segments):
        for w in reversed(s["words"]):
            return w["end"]
    return segments[-1]["end"] if segments else None


def usable_cores() -> int:
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota.  A container that
    sees 256 CPUs under a 16-CPU quota is throttled ~20x when a thread pool is sized by the visible count; the
    CPU-side timing helpers (bench.py cpu_baseline, the CPU oracle in tests) size their pools with this."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:                                                           # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n
