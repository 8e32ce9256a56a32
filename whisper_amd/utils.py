"""Host-side helpers shared by the decoding / transcribe mirrors (reference: whisper/utils.py:24-82) and the result
writers `get_writer` / `WriteTXT` / `WriteVTT` / `WriteSRT` / `WriteTSV` / `WriteJSON` (reference utils.py:85-318;
SURVEY.md §8f rank 4: byte-identical output, tests/test_writers.py).  The argparse CLI and `whisper.normalizers` are
deliberately not part of this package (README.md)."""
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


# ---------------------------------------------------------------------------------------------------------------
# result writers — the file formats of reference whisper/utils.py:85-318 (txt, vtt, srt, tsv, json), same class
# names, constructor / call signatures and option names (max_line_width, max_line_count, highlight_words,
# max_words_per_line), so `get_writer(fmt, out_dir)(result, audio_path, options)` drops in.  Own implementation:
# subtitle layout is a small state machine (`_CueLayout`) fed word by word.
# ---------------------------------------------------------------------------------------------------------------
import json  # noqa: E402
import re  # noqa: E402
from typing import Callable, Iterator, TextIO, Tuple  # noqa: E402


class ResultWriter:
    extension: str

    def __init__(self, output_dir: str):
        self.output_dir = output_dir

    def __call__(self, result: dict, audio_path: str, options: Optional[dict] = None, **kwargs):
        stem = os.path.splitext(os.path.basename(audio_path))[0]
        with open(os.path.join(self.output_dir, f"{stem}.{self.extension}"), "w", encoding="utf-8") as f:
            self.write_result(result, file=f, options=options, **kwargs)

    def write_result(self, result: dict, file: TextIO, options: Optional[dict] = None, **kwargs):
        raise NotImplementedError


class WriteTXT(ResultWriter):
    extension = "txt"

    def write_result(self, result: dict, file: TextIO, options: Optional[dict] = None, **kwargs):
        for segment in result["segments"]:
            file.write(segment["text"].strip() + "\n")
            file.flush()


class _CueLayout:
    """Packs timed words into subtitle cues.  Two regimes, as in the reference: without both a line width and a line
    count every segment (or every `max_words_per_line` chunk of it) is its own cue; with both, lines are filled up to
    the width, cues up to the line count, and a pause longer than 3 s always starts a new cue."""

    def __init__(self, width: Optional[int], lines: Optional[int], first_start: float):
        self.free_form = not (width is None or lines is None)      # lines flow across segment boundaries
        self.width = width or 1000
        self.lines = lines
        self.cue: List[dict] = []
        self.used = 0            # characters on the current line
        self.n_lines = 1
        self.previous_start = first_start

    def push(self, word: dict, starts_chunk: bool) -> Optional[List[dict]]:
        """add one word; returns a finished cue when this word opened a new one"""
        word = dict(word)
        finished = None
        paused = self.free_form and word["start"] - self.previous_start > 3.0
        forced = starts_chunk and bool(self.cue) and not self.free_form
        fits = self.used + len(word["word"]) <= self.width
        if self.used > 0 and fits and not paused and not forced:
            self.used += len(word["word"])
        else:
            word["word"] = word["word"].strip()
            cue_full = bool(self.cue) and self.lines is not None and (paused or self.n_lines >= self.lines)
            if cue_full or forced:
                finished, self.cue, self.n_lines = self.cue, [], 1
            elif self.used > 0:
                self.n_lines += 1
                word["word"] = "\n" + word["word"]
            self.used = len(word["word"].strip())
        self.cue.append(word)
        self.previous_start = word["start"]
        return finished


class SubtitlesWriter(ResultWriter):
    always_include_hours: bool
    decimal_marker: str

    def format_timestamp(self, seconds: float) -> str:
        return format_timestamp(seconds, self.always_include_hours, self.decimal_marker)

    def _cues(self, result: dict, width, lines, per_line) -> Iterator[List[dict]]:
        layout = _CueLayout(width, lines, get_start(result["segments"]) or 0.0)
        step = per_line or 1000
        for segment in result["segments"]:
            words = segment["words"]
            for at in range(0, len(words), step):
                for i, word in enumerate(words[at: at + step]):
                    done = layout.push(word, starts_chunk=(i == 0))
                    if done is not None:
                        yield done
        if layout.cue:
            yield layout.cue

    def iterate_result(self, result: dict, options: Optional[dict] = None, *, max_line_width: Optional[int] = None,
                       max_line_count: Optional[int] = None, highlight_words: bool = False,
                       max_words_per_line: Optional[int] = None) -> Iterator[Tuple[str, str, str]]:
        options = options or {}
        width = max_line_width or options.get("max_line_width")
        lines = max_line_count or options.get("max_line_count")
        highlight = highlight_words or options.get("highlight_words", False)
        per_line = max_words_per_line or options.get("max_words_per_line")
        segments = result["segments"]
        if not (segments and "words" in segments[0]):
            for segment in segments:
                yield (self.format_timestamp(segment["start"]), self.format_timestamp(segment["end"]),
                       segment["text"].strip().replace("-->", "->"))
            return
        for cue in self._cues(result, width, lines, per_line):
            begin, end = self.format_timestamp(cue[0]["start"]), self.format_timestamp(cue[-1]["end"])
            texts = [w["word"] for w in cue]
            plain = "".join(texts)
            if not highlight:
                yield begin, end, plain
                continue
            cursor = begin
            for i, w in enumerate(cue):                      # one sub-cue per word, the spoken word underlined
                w0, w1 = self.format_timestamp(w["start"]), self.format_timestamp(w["end"])
                if cursor != w0:
                    yield cursor, w0, plain
                marked = re.sub(r"^(\s*)(.*)$", r"\1<u>\2</u>", texts[i])
                yield w0, w1, "".join(texts[:i] + [marked] + texts[i + 1:])
                cursor = w1


class WriteVTT(SubtitlesWriter):
    extension = "vtt"
    always_include_hours = False
    decimal_marker = "."

    def write_result(self, result: dict, file: TextIO, options: Optional[dict] = None, **kwargs):
        file.write("WEBVTT\n\n")
        for begin, end, text in self.iterate_result(result, options, **kwargs):
            file.write(f"{begin} --> {end}\n{text}\n\n")
            file.flush()


class WriteSRT(SubtitlesWriter):
    extension = "srt"
    always_include_hours = True
    decimal_marker = ","

    def write_result(self, result: dict, file: TextIO, options: Optional[dict] = None, **kwargs):
        for number, (begin, end, text) in enumerate(self.iterate_result(result, options, **kwargs), start=1):
            file.write(f"{number}\n{begin} --> {end}\n{text}\n\n")
            file.flush()


class WriteTSV(ResultWriter):
    """start / end in integer milliseconds, tab separated — locale-proof and trivial to parse"""
    extension = "tsv"

    def write_result(self, result: dict, file: TextIO, options: Optional[dict] = None, **kwargs):
        file.write("start\tend\ttext\n")
        for segment in result["segments"]:
            text = segment["text"].strip().replace("\t", " ")
            file.write(f"{round(1000 * segment['start'])}\t{round(1000 * segment['end'])}\t{text}\n")
            file.flush()


class WriteJSON(ResultWriter):
    extension = "json"

    def write_result(self, result: dict, file: TextIO, options: Optional[dict] = None, **kwargs):
        json.dump(result, file)


_WRITERS = {"txt": WriteTXT, "vtt": WriteVTT, "srt": WriteSRT, "tsv": WriteTSV, "json": WriteJSON}


def get_writer(output_format: str, output_dir: str) -> Callable[..., None]:
    """one writer, or with "all" a callable that writes every format (reference utils.py:294-318)"""
    if output_format != "all":
        return _WRITERS[output_format](output_dir)
    every = [cls(output_dir) for cls in _WRITERS.values()]

    def write_all(result: dict, file, options: Optional[dict] = None, **kwargs):
        for writer in every:
            writer(result, file, options, **kwargs)
    return write_all
