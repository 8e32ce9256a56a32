"""The result writers (txt / vtt / srt / tsv / json) against the reference's on randomly
generated transcripts with word timings, for every combination of the subtitle layout options.  CPU only; needs the
reference checkout (build container)."""
import io
import itertools
import json
import os
import random
import sys

import pytest

from whisper_amd import utils as U

REF = "/root/reference"


def _result(seed, with_words=True):
    rnd = random.Random(seed)
    vocab = ["the", "a", "quick", "brown", "fox", "jumps", "over", "lazy", "dog,", "and", "then", "-->", "it's", "über",
             "naïve", "tab\there", "end."]
    t, segments = rnd.uniform(0, 2), []
    for _ in range(rnd.randint(1, 7)):
        words = []
        if rnd.random() < 0.3:
            t += rnd.uniform(3.1, 6.0)                     # long pause between segments
        for _ in range(rnd.randint(0 if rnd.random() < 0.2 else 1, 14)):
            d = rnd.uniform(0.05, 0.6)
            words.append({"word": " " + rnd.choice(vocab), "start": round(t, 2), "end": round(t + d, 2),
                          "probability": rnd.random()})
            t += d + (rnd.uniform(3.2, 4.0) if rnd.random() < 0.05 else rnd.uniform(0, 0.2))
        seg = {"id": len(segments), "seek": 0, "start": words[0]["start"] if words else round(t, 2),
               "end": words[-1]["end"] if words else round(t, 2), "text": "".join(w["word"] for w in words),
               "tokens": [1, 2, 3], "temperature": 0.0, "avg_logprob": -0.3, "compression_ratio": 1.2,
               "no_speech_prob": 0.01}
        if with_words:
            seg["words"] = words
        segments.append(seg)
    return {"text": "".join(s["text"] for s in segments), "segments": segments, "language": "en"}


@pytest.fixture(scope="module")
def ref_utils():
    if not os.path.isdir(os.path.join(REF, "whisper")):
        pytest.skip("reference checkout not present")
    sys.path[:0] = [os.path.join(os.path.dirname(__file__), "shims"), REF]
    import whisper.utils as RU
    return RU


@pytest.mark.reference
def test_writers_match_reference(ref_utils, tmp_path):
    grids = list(itertools.product([None, 12, 28, 60], [None, 1, 2, 3], [False, True], [None, 1, 3, 5]))
    for seed in range(12):
        for with_words in (True, False):
            res = _result(seed, with_words)
            for fmt in ("txt", "vtt", "srt", "tsv", "json"):
                for width, lines, hl, per_line in (grids if fmt in ("vtt", "srt") else grids[:1]):
                    kw = dict(max_line_width=width, max_line_count=lines, highlight_words=hl, max_words_per_line=per_line)
                    mine, theirs = io.StringIO(), io.StringIO()
                    U.get_writer(fmt, str(tmp_path)).write_result(res, file=mine, options=dict(kw))
                    ref_utils.get_writer(fmt, str(tmp_path)).write_result(res, file=theirs, options=dict(kw))
                    assert mine.getvalue() == theirs.getvalue(), (seed, with_words, fmt, kw)
    # keyword arguments instead of the options dict, and the file-naming __call__ / "all" paths
    res = _result(99)
    a, b = io.StringIO(), io.StringIO()
    U.WriteSRT(str(tmp_path)).write_result(res, file=a, max_line_width=20, max_line_count=2, highlight_words=True)
    ref_utils.WriteSRT(str(tmp_path)).write_result(res, file=b, max_line_width=20, max_line_count=2, highlight_words=True)
    assert a.getvalue() == b.getvalue()
    (tmp_path / "mine").mkdir()
    (tmp_path / "ref").mkdir()
    U.get_writer("all", str(tmp_path / "mine"))(res, "/some/dir/clip.flac", {"max_line_width": 30, "max_line_count": 2})
    ref_utils.get_writer("all", str(tmp_path / "ref"))(res, "/some/dir/clip.flac", {"max_line_width": 30, "max_line_count": 2})
    for ext in ("txt", "vtt", "srt", "tsv", "json"):
        assert (tmp_path / "mine" / f"clip.{ext}").read_text(encoding="utf-8") == (tmp_path / "ref" / f"clip.{ext}").read_text(encoding="utf-8")


def test_writers_known_output():
    """runs anywhere: a hand-checked transcript"""
    res = {"text": " Hello world. Bye", "language": "en", "segments": [
        {"start": 0.0, "end": 1.5, "text": " Hello world.", "words": [
            {"word": " Hello", "start": 0.0, "end": 0.6, "probability": 0.9},
            {"word": " world.", "start": 0.7, "end": 1.5, "probability": 0.8}]},
        {"start": 5.0, "end": 5.4, "text": " Bye", "words": [{"word": " Bye", "start": 5.0, "end": 5.4, "probability": 0.7}]}]}
    out = io.StringIO()
    U.WriteVTT(".").write_result(res, file=out)
    assert out.getvalue() == "WEBVTT\n\n00:00.000 --> 00:01.500\nHello world.\n\n00:05.000 --> 00:05.400\nBye\n\n"
    out = io.StringIO()
    U.WriteSRT(".").write_result(res, file=out, highlight_words=True)
    assert out.getvalue().startswith("1\n00:00:00,000 --> 00:00:00,600\n<u>Hello</u> world.\n\n2\n00:00:00,600 --> 00:00:00,700\nHello world.\n\n"
                                     "3\n00:00:00,700 --> 00:00:01,500\nHello <u>world.</u>\n\n4\n00:00:05,000 --> 00:00:05,400\n<u>Bye</u>\n\n")
    out = io.StringIO()
    U.WriteTSV(".").write_result(res, file=out)
    assert out.getvalue() == "start\tend\ttext\n0\t1500\tHello world.\n5000\t5400\tBye\n"
    out = io.StringIO()
    U.WriteTXT(".").write_result(res, file=out)
    assert out.getvalue() == "Hello world.\nBye\n"
    out = io.StringIO()
    U.WriteJSON(".").write_result(res, file=out)
    assert json.loads(out.getvalue()) == res
    with pytest.raises(KeyError):
        U.get_writer("docx", ".")


@pytest.mark.reference
def test_small_helpers_match_reference(ref_utils):
    """exact_div, compression_ratio, format_timestamp, get_start / get_end (utils.py:24-82) on random inputs"""
    RU = ref_utils
    rnd = random.Random(9)
    for _ in range(200):
        y, k = rnd.randint(1, 50), rnd.randint(0, 99)
        assert U.exact_div(y * k, y) == RU.exact_div(y * k, y) == k
    for x, y in ((480000, 160), (3000, 2), (0, 7)):
        assert U.exact_div(x, y) == RU.exact_div(x, y)
    for bad in ((10, 3), (7, 2)):
        with pytest.raises(AssertionError):
            U.exact_div(*bad)
        with pytest.raises(AssertionError):
            RU.exact_div(*bad)
    words = ["the", "quick", "über", "naïve", "日本語", "a", " ", "\n", "and and and and"]
    for _ in range(100):
        text = " ".join(rnd.choice(words) for _ in range(rnd.randint(1, 60)))
        assert U.compression_ratio(text) == RU.compression_ratio(text)
    for _ in range(300):
        s = rnd.choice([0.0, rnd.uniform(0, 70), rnd.uniform(3500, 3700), rnd.uniform(0, 1e5), 59.9996, 3599.9999])
        for hours, mark in ((False, "."), (True, ","), (False, ","), (True, ".")):
            assert U.format_timestamp(s, hours, mark) == RU.format_timestamp(s, hours, mark)
    with pytest.raises(AssertionError):
        U.format_timestamp(-1.0)
    for seed in range(40):
        segs = _result(seed)["segments"]                  # both index s["words"]: word-timed segments only
        assert U.get_start(segs) == RU.get_start(segs) and U.get_end(segs) == RU.get_end(segs)
        empty = [dict(s, words=[]) for s in segs]
        assert U.get_start(empty) == RU.get_start(empty) and U.get_end(empty) == RU.get_end(empty)
    assert U.get_start([]) == RU.get_start([]) and U.get_end([]) == RU.get_end([])
