"""Word-level timestamps: cross-attention alignment + dynamic time warping, on the HIP path.

Mirror of the reference's `whisper/timing.py`: `median_filter` (:19), `dtw` (:141), `WordTiming` (:154),
`find_alignment` (:163), `merge_punctuations` (:245), `add_word_timestamps` (:279) keep their signatures.
Differences in *how*: the reference re-runs encoder + decoder with SDPA disabled and hooks every
cross-attention module to grab QK^T (timing.py:186-197); here one teacher-forced prefill keeps the per-layer
queries (wh_task created with WH_TASK_CAPTURE_Q) and `wh_task_cross_qk` evaluates QK^T only for the alignment
heads.  softmax / z-norm / median / head-mean are one C call (`wh_align_matrix`), DTW runs as an anti-diagonal
wavefront kernel with dtw_cpu's tie rule (timing.py:95-100) and the back-trace walk as one lane per clip on an
LDS-packed copy of the trace (`wh_dtw_backtrace_batch`): per clip only the frame at which each token is first reached
travels to the host.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import TYPE_CHECKING, List, Optional

import numpy as np
import torch

from . import hip
from .audio import HOP_LENGTH, SAMPLE_RATE, TOKENS_PER_SECOND
from .tokenizer import Tokenizer

if TYPE_CHECKING:
    from .model import Whisper


def median_filter(x: torch.Tensor, filter_width: int) -> torch.Tensor:
    """Sliding median of odd width along the last axis with reflect padding; input returned unchanged when it
    is not longer than the padding (reference timing.py:19-54)."""
    if x.shape[-1] <= filter_width // 2:
        return x
    assert filter_width > 0 and filter_width % 2 == 1, "`filter_width` should be an odd number"
    return hip.median_filter(x, filter_width).to(x.dtype)


def backtrace(trace: np.ndarray) -> np.ndarray:
    """walk the trace matrix from (N, M) to the origin: code 0 = diagonal, 1 = up, 2 = left
    (reference timing.py:57-79; row 0 / column 0 are forced to 2 / 1)"""
    i, j = trace.shape[0] - 1, trace.shape[1] - 1
    path = []
    while i > 0 or j > 0:
        path.append((i - 1, j - 1))
        step = 2 if i == 0 else 1 if j == 0 else trace[i, j]
        if step == 0:
            i, j = i - 1, j - 1
        elif step == 1:
            i -= 1
        elif step == 2:
            j -= 1
        else:
            raise ValueError("Unexpected trace[i, j]")
    return np.array(path)[::-1, :].T


def dtw(x: torch.Tensor) -> np.ndarray:
    """(2, path_len) array of (row, column) indices of the cheapest monotone path through cost matrix x
    (reference timing.py:141-151).  Cost fill and the back-trace walk both run on the device; only the path comes back."""
    _, path = hip.dtw_backtrace(hip.dtw_trace(x.to(torch.float32)))
    return path.cpu().numpy().astype(np.int64)


@dataclass
class WordTiming:
    word: str
    tokens: List[int]
    start: float
    end: float
    probability: float


def find_alignment(model: "Whisper", tokenizer: Tokenizer, text_tokens: List[int], mel: torch.Tensor,
                   num_frames: int, *, medfilt_width: int = 7, qk_scale: float = 1.0,
                   audio_features: Optional[torch.Tensor] = None) -> List[WordTiming]:
    """reference timing.py:163-242.  `audio_features` (extension): the encoder output of this window, (1500, D), as
    returned in DecodingResult.audio_features — the reference re-runs the encoder on `mel` here (timing.py:199); passing
    the features the decode already computed skips that pass, with identical results."""
    if len(text_tokens) == 0:
        return []
    n_sot = len(tokenizer.sot_sequence)
    tokens = torch.tensor([*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot],
                          device=model.device)
    with torch.no_grad():
        features = model.encoder(mel.unsqueeze(0)) if audio_features is None else audio_features.reshape(
            1, model.dims.n_audio_ctx, model.dims.n_audio_state)
        engine = model.engine(features.dtype)
        task = engine.acquire_task(1, 1, max(int(tokens.numel()), 8), capture_q=True)
        try:
            task.set_audio(features.contiguous())
            text_positions = list(range(n_sot, n_sot + len(text_tokens)))
            logits = task.prefill(tokens[None].contiguous(), sel=text_positions)[0]       # (n_text, vocab)
            text_token_probs = _token_probs(logits[None], torch.tensor([list(text_tokens)], device=logits.device),
                                            tokenizer.eot)[0].tolist()
            heads = model.alignment_heads.indices().T.tolist()
            qk = task.cross_qk(0, [h[0] for h in heads], [h[1] for h in heads], 0, int(tokens.numel()))
        finally:
            task.close()
        # softmax over frames -> z-norm over tokens -> median filter -> -mean over heads, rows [n_sot, -1)
        matrix = hip.align_matrix(qk, num_frames // 2, medfilt_width, n_sot, int(tokens.numel()) - 1, qk_scale)
        jumps, _ = hip.dtw_backtrace(hip.dtw_trace(matrix), want_path=False)

    return _words_from_jumps(tokenizer, list(text_tokens), text_token_probs, jumps.cpu().numpy())


def _token_probs(logits: torch.Tensor, tokens: torch.Tensor, eot: int) -> torch.Tensor:
    """logits (rows, n, vocab) at the positions that predict `tokens` (rows, n): softmax over the text vocabulary
    [0, eot) and the probability of each token (reference timing.py:218-221); padded token slots hold token 0's."""
    return logits[:, :, :eot].softmax(dim=-1).gather(2, tokens[:, :, None])[:, :, 0]


def _words_from_path(tokenizer: Tokenizer, text_tokens: List[int], text_token_probs: List[float],
                     text_indices: np.ndarray, time_indices: np.ndarray) -> List[WordTiming]:
    """token / frame path -> word boundaries (reference timing.py:218-242)"""
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    return _words_from_jumps(tokenizer, text_tokens, text_token_probs, time_indices[jumps])


def _words_from_jumps(tokenizer: Tokenizer, text_tokens: List[int], text_token_probs: List[float],
                      jump_frames: np.ndarray) -> List[WordTiming]:
    """`jump_frames[i]` = frame at which the alignment path first reaches text row i (= `time_indices[jumps]`,
    reference timing.py:226-228; computed on the device by wh_dtw_backtrace_batch) -> word boundaries (timing.py:218-242)"""
    words, word_tokens = tokenizer.split_to_word_tokens(text_tokens + [tokenizer.eot])
    if len(word_tokens) <= 1:
        return []          # only EOT: nothing to align
    word_boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jump_times = np.asarray(jump_frames) / TOKENS_PER_SECOND
    start_times = jump_times[word_boundaries[:-1]]
    end_times = jump_times[word_boundaries[1:]]
    word_probabilities = [np.mean(text_token_probs[i:j]) for i, j in zip(word_boundaries[:-1], word_boundaries[1:])]
    return [WordTiming(w, t, s, e, p)
            for w, t, s, e, p in zip(words, word_tokens, start_times, end_times, word_probabilities)]


ALIGN_BATCH_SCRATCH_BYTES = 6 << 30      # bound on the QK / softmax slabs of one find_alignment_batch chunk


def find_alignment_batch(model: "Whisper", tokenizer: Tokenizer, text_tokens: List[List[int]], mel: torch.Tensor,
                         num_frames: List[int], *, medfilt_width: int = 7, qk_scale: float = 1.0,
                         audio_features: Optional[torch.Tensor] = None, stats: Optional[dict] = None) -> List[List[WordTiming]]:
    """`find_alignment` for every clip of a batch — mel (B, n_mels, 3000), one token list and frame count per clip —
    with identical results, clip by clip (BASELINE configs[4]: word timestamps over a batch).  One encoder pass (none
    when `audio_features` (B, 1500, D), the DecodingResult.audio_features of the same windows, is passed), ONE
    teacher-forced decoder pass over all clips (rows padded on the right; causal attention keeps the padding out of the
    real positions), one launch each for the alignment heads' QK, softmax, z-norm, median, head mean and DTW
    (wh_task_align_batch, a workgroup per clip for the DTW wavefront) and for the back-trace walk
    (wh_dtw_backtrace_batch: a lane per clip on the LDS-packed trace); what reaches the host per clip is the frame at
    which each token is first reached and the token probabilities.  Host code = splitting tokens into words.
    `stats` (dict, optional): receives "device_s" (wall clock until those results are on the host, i.e. encoder +
    teacher-forced pass + alignment kernels + the copy) and "host_s" (the word split after it) of this call."""
    import time
    t_start = time.perf_counter()
    if stats is not None:
        stats.clear()
    B = len(text_tokens)
    out: List[List[WordTiming]] = [[] for _ in range(B)]
    live = [i for i in range(B) if len(text_tokens[i]) > 0]
    if not live:
        return out
    n_sot = len(tokenizer.sot_sequence)
    heads = model.alignment_heads.indices().T.tolist()
    dims = model.dims
    # chunk so that the fp32 score slabs stay bounded: (n_audio_ctx + 2 F) floats per (clip, pair, token)
    tmax_all = max(len(text_tokens[i]) for i in live) + n_sot + 2
    per_clip = len(heads) * tmax_all * (dims.n_audio_ctx + 2 * (max(num_frames) // 2)) * 4
    chunk = max(1, min(len(live), ALIGN_BATCH_SCRATCH_BYTES // max(per_clip, 1)))
    with torch.no_grad():
        for c0 in range(0, len(live), chunk):
            ids = live[c0: c0 + chunk]
            rows = [[*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens[i], tokenizer.eot] for i in ids]
            n_tok = [len(r) for r in rows]
            Tmax = max(n_tok)
            tokens = torch.tensor([r + [tokenizer.eot] * (Tmax - len(r)) for r in rows], device=model.device)
            features = model.encoder(mel[ids]) if audio_features is None else audio_features[ids]
            engine = model.engine(features.dtype)
            task = engine.acquire_task(len(ids), 1, max(Tmax, 8), capture_q=True)
            try:
                task.set_audio(features.contiguous())
                n_text_max = Tmax - n_sot - 2
                logits = task.prefill(tokens.contiguous(), sel=list(range(n_sot, n_sot + n_text_max)))   # (rows, n_text_max, V)
                cost, jumps, plen = task.align_batch([h[0] for h in heads], [h[1] for h in heads], n_tok,
                                               [int(num_frames[i]) // 2 for i in ids], medfilt_width, n_sot, qk_scale)
            finally:
                task.close()
            # probabilities of all text tokens of all clips in one gather; ONE copy to the host (the paths' jump frames
            # and the probabilities: a few KB per clip — the trace matrices stay on the device)
            padded = torch.tensor([list(text_tokens[i]) + [0] * (n_text_max - len(text_tokens[i])) for i in ids],
                                  device=model.device)
            probs = _token_probs(logits, padded, tokenizer.eot)
            jumps_h, probs_h = jumps.cpu().numpy(), probs.cpu().numpy()
            if bool((plen.cpu() < 0).any()):
                raise ValueError("Unexpected trace[i, j]")          # reference timing.py:77
            if stats is not None:
                stats["device_s"] = stats.get("device_s", 0.0) + time.perf_counter() - t_start
                t_start = time.perf_counter()
            for k, i in enumerate(ids):
                tt = list(text_tokens[i])
                out[i] = _words_from_jumps(tokenizer, tt, probs_h[k, : len(tt)].tolist(), jumps_h[k, : len(tt) + 1])
            if stats is not None:
                stats["host_s"] = stats.get("host_s", 0.0) + time.perf_counter() - t_start
                t_start = time.perf_counter()
    return out


def merge_punctuations(alignment: List[WordTiming], prepended: str, appended: str):
    """glue opening punctuation to the following word and closing punctuation to the preceding one, in place"""
    i, j = len(alignment) - 2, len(alignment) - 1
    while i >= 0:
        prev, nxt = alignment[i], alignment[j]
        if prev.word.startswith(" ") and prev.word.strip() in prepended:
            nxt.word = prev.word + nxt.word
            nxt.tokens = prev.tokens + nxt.tokens
            prev.word, prev.tokens = "", []
        else:
            j = i
        i -= 1
    i, j = 0, 1
    while j < len(alignment):
        prev, nxt = alignment[i], alignment[j]
        if not prev.word.endswith(" ") and nxt.word in appended:
            prev.word = prev.word + nxt.word
            prev.tokens = prev.tokens + nxt.tokens
            nxt.word, nxt.tokens = "", []
        else:
            i = j
        j += 1


def alignment_text_tokens(segments: List[dict], tokenizer: Tokenizer) -> List[int]:
    """the token list add_word_timestamps aligns: the text tokens of all segments of the window (timing.py:293-297)"""
    return list(itertools.chain.from_iterable([t for t in seg["tokens"] if t < tokenizer.eot] for seg in segments))


def add_word_timestamps(*, segments: List[dict], model: "Whisper", tokenizer: Tokenizer, mel: torch.Tensor,
                        num_frames: int, prepend_punctuations: str = "\"'“¿([{-",
                        append_punctuations: str = "\"'.。,，!！?？:：”)]}、", last_speech_timestamp: float,
                        alignment: Optional[List[WordTiming]] = None, **kwargs):
    """attach a "words" list to every segment (reference timing.py:279-388, same clipping heuristics).
    `alignment` (extension): the result of find_alignment for this window when the caller already has it —
    transcribe_batch aligns the windows of many files in one find_alignment_batch pass."""
    if len(segments) == 0:
        return
    per_segment = [[t for t in seg["tokens"] if t < tokenizer.eot] for seg in segments]
    text_tokens = list(itertools.chain.from_iterable(per_segment))
    if alignment is None:
        alignment = find_alignment(model, tokenizer, text_tokens, mel, num_frames, **kwargs)

    durations = np.array([w.end - w.start for w in alignment])
    durations = durations[durations.nonzero()]
    median_duration = min(0.7, float(np.median(durations) if len(durations) > 0 else 0.0))
    max_duration = median_duration * 2

    if len(durations) > 0:
        # words at sentence boundaries must not be longer than twice the median word
        marks = ".。!！?？"
        for i in range(1, len(alignment)):
            if alignment[i].end - alignment[i].start > max_duration:
                if alignment[i].word in marks:
                    alignment[i].end = alignment[i].start + max_duration
                elif alignment[i - 1].word in marks:
                    alignment[i].start = alignment[i].end - max_duration

    merge_punctuations(alignment, prepend_punctuations, append_punctuations)

    time_offset = segments[0]["seek"] * HOP_LENGTH / SAMPLE_RATE
    cursor = 0
    for segment, seg_tokens in zip(segments, per_segment):
        consumed, words = 0, []
        while cursor < len(alignment) and consumed < len(seg_tokens):
            timing = alignment[cursor]
            if timing.word:
                words.append(dict(word=timing.word, start=round(time_offset + timing.start, 2),
                                  end=round(time_offset + timing.end, 2), probability=timing.probability))
            consumed += len(timing.tokens)
            cursor += 1

        if len(words) > 0:
            # first / second word after a pause must not be longer than twice the median word
            if words[0]["end"] - last_speech_timestamp > median_duration * 4 and (
                    words[0]["end"] - words[0]["start"] > max_duration
                    or (len(words) > 1 and words[1]["end"] - words[0]["start"] > max_duration * 2)):
                if len(words) > 1 and words[1]["end"] - words[1]["start"] > max_duration:
                    boundary = max(words[1]["end"] / 2, words[1]["end"] - max_duration)
                    words[0]["end"] = words[1]["start"] = boundary
                words[0]["start"] = max(0, words[0]["end"] - max_duration)
            # prefer the segment-level start when the first word is too long
            if segment["start"] < words[0]["end"] and segment["start"] - 0.5 > words[0]["start"]:
                words[0]["start"] = max(0, min(words[0]["end"] - median_duration, segment["start"]))
            else:
                segment["start"] = words[0]["start"]
            # prefer the segment-level end when the last word is too long
            if segment["end"] > words[-1]["start"] and segment["end"] + 0.5 < words[-1]["end"]:
                words[-1]["end"] = max(words[-1]["start"] + median_duration, segment["end"])
            else:
                segment["end"] = words[-1]["end"]
            last_speech_timestamp = segment["end"]
        segment["words"] = words
