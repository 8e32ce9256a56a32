"""Long-form transcription driver: 30-second sliding window over a file, temperature fallback, prompt
carry-over, timestamp-based seeking, optional word timestamps and hallucination skipping.

Behavioural mirror of the reference's `whisper/transcribe.py:38-514` — same keyword arguments, same result
dictionary (`text`, `segments[...]`, `language`), same seek arithmetic — organised as a small state machine
(`_Transcriber`) instead of one long function.  The heavy lifting of every window (`model.decode`,
`add_word_timestamps`) runs on the HIP path; the spectrogram of the whole file is computed once on the GPU.
The argparse CLI of the reference (transcribe.py:517-623) is out of the hot-path scope.
"""
from __future__ import annotations

import warnings
from typing import TYPE_CHECKING, List, Optional, Tuple, Union

import numpy as np
import torch
import tqdm

from .audio import (FRAMES_PER_SECOND, HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE, load_audio, log_mel_spectrogram,
                    pad_or_trim)
from dataclasses import replace

from .decoding import DecodingOptions, DecodingResult, DecodingTask
from .timing import add_word_timestamps, alignment_text_tokens, find_alignment, find_alignment_batch
from .tokenizer import LANGUAGES, get_tokenizer
from .utils import exact_div, format_timestamp, get_end, make_safe

if TYPE_CHECKING:
    from .model import Whisper

_PUNCTUATION = "\"'“¿([{-\"'.。,，!！?？:：”)]}、"


def _word_anomaly_score(word: dict) -> float:
    """very short / very long / improbable words look like hallucinations"""
    duration = word["end"] - word["start"]
    score = 1.0 if word.get("probability", 0.0) < 0.15 else 0.0
    if duration < 0.133:
        score += (0.133 - duration) * 15
    if duration > 2.0:
        score += duration - 2.0
    return score


def _is_segment_anomaly(segment: Optional[dict]) -> bool:
    if segment is None or not segment["words"]:
        return False
    words = [w for w in segment["words"] if w["word"] not in _PUNCTUATION][:8]
    score = sum(_word_anomaly_score(w) for w in words)
    return score >= 3 or score + 0.01 >= len(words)


def _first_segment_with_words(segments: List[dict]) -> Optional[dict]:
    for s in segments:
        if s["words"]:
            return s
    return None


class _AlignRequest:
    """what a file's state machine yields when it needs the word alignment of the window it just decoded; the driver
    answers with find_alignment's result (one file) or one row of find_alignment_batch (many files)"""

    def __init__(self, text_tokens: List[int], mel: torch.Tensor, num_frames: int, tokenizer, language: str, task: str,
                 audio_features: Optional[torch.Tensor] = None):
        self.text_tokens, self.mel, self.num_frames = text_tokens, mel, num_frames
        self.tokenizer, self.language, self.task = tokenizer, language, task
        self.audio_features = audio_features       # encoder output of this window from the decode (no second pass)


class _Transcriber:
    def __init__(self, model: "Whisper", verbose, temperature, compression_ratio_threshold, logprob_threshold,
                 no_speech_threshold, condition_on_previous_text, initial_prompt, carry_initial_prompt,
                 word_timestamps, prepend_punctuations, append_punctuations, clip_timestamps,
                 hallucination_silence_threshold, decode_options: dict):
        self.model = model
        self.verbose = verbose
        self.temperatures = [temperature] if isinstance(temperature, (int, float)) else list(temperature)
        self.compression_ratio_threshold = compression_ratio_threshold
        self.logprob_threshold = logprob_threshold
        self.no_speech_threshold = no_speech_threshold
        self.condition_on_previous_text = condition_on_previous_text
        self.carry_initial_prompt = carry_initial_prompt
        self.word_timestamps = word_timestamps
        self.prepend_punctuations = prepend_punctuations
        self.append_punctuations = append_punctuations
        self.hallucination_silence_threshold = hallucination_silence_threshold
        self.decode_options = decode_options
        self.initial_prompt = initial_prompt
        self.clip_timestamps = clip_timestamps

    # ---- one window ------------------------------------------------------------------------------------
    def decode_with_fallback(self, segment: torch.Tensor, first: int = 0) -> DecodingResult:
        """retry at increasing temperature while the output is too repetitive or too improbable
        (reference transcribe.py:184-224); `first` skips temperatures already tried by a batched pass"""
        result = None
        for t in self.temperatures[first:]:
            result = self.model.decode(segment, self._options_for(t))
            if not self._needs_retry(result):
                break
            feats = getattr(result, "audio_features", None)
            if torch.is_tensor(feats) and feats.shape[-2:] == (self.model.dims.n_audio_ctx, self.model.dims.n_audio_state):
                segment = feats             # the retry decodes the same window: hand it the encoder output (decode()
                                            # accepts encoded features, reference decoding.py:655-662) — no second pass
        return result

    def _needs_retry(self, result: DecodingResult) -> bool:
        """the fallback criteria of reference transcribe.py:202-222"""
        retry = False
        if (self.compression_ratio_threshold is not None
                and result.compression_ratio > self.compression_ratio_threshold):
            retry = True
        if self.logprob_threshold is not None and result.avg_logprob < self.logprob_threshold:
            retry = True
        if (self.no_speech_threshold is not None and result.no_speech_prob > self.no_speech_threshold
                and self.logprob_threshold is not None and result.avg_logprob < self.logprob_threshold):
            retry = False                      # it is silence, not a failure
        return retry

    def _options_for(self, t: float) -> DecodingOptions:
        kwargs = {**self.decode_options}
        if t > 0:
            kwargs.pop("beam_size", None)      # sampling: no beams
            kwargs.pop("patience", None)
        else:
            kwargs.pop("best_of", None)        # greedy / beam: no best-of
        return DecodingOptions(**kwargs, temperature=t)

    def run(self, audio) -> dict:
        """one file, windows decoded one at a time (the reference's control flow)"""
        walk = self._walk(audio)
        try:
            request = next(walk)
            while True:
                if isinstance(request, _AlignRequest):
                    request = walk.send(find_alignment(self.model, request.tokenizer, request.text_tokens, request.mel,
                                                       request.num_frames, audio_features=request.audio_features))
                else:
                    request = walk.send(self.decode_with_fallback(request))
        except StopIteration as stop:
            return stop.value

    def _walk(self, audio, mel: Optional[torch.Tensor] = None):
        """The per-file state machine of reference transcribe.py:126-514 as a generator: yields the next 30 s mel
        window to decode (with `self.decode_options["prompt"]` already set for it) and is sent its DecodingResult;
        returns the result dict.  `run` drives one file; `transcribe_batch` drives many files in lock-step so that
        their windows share one batched decode."""
        model, opts = self.model, self.decode_options
        dtype = torch.float16 if opts.get("fp16", True) else torch.float32
        if model.device == torch.device("cpu"):
            if torch.cuda.is_available():
                warnings.warn("Performing inference on CPU when CUDA is available")
            if dtype == torch.float16:
                warnings.warn("FP16 is not supported on CPU; using FP32 instead")
                dtype = torch.float32
        if dtype == torch.float32:
            opts["fp16"] = False

        # whole-file spectrogram with 30 s of trailing silence so every window can be sliced
        if mel is None:
            mel = log_mel_spectrogram(audio, model.dims.n_mels, padding=N_SAMPLES, device=model.device)
        content_frames = mel.shape[-1] - N_FRAMES
        content_duration = float(content_frames * HOP_LENGTH / SAMPLE_RATE)

        if opts.get("language", None) is None:
            if not model.is_multilingual:
                opts["language"] = "en"
            else:
                if self.verbose:
                    print("Detecting language using up to the first 30 seconds. Use `--language` to specify the language")
                head = pad_or_trim(mel, N_FRAMES).to(model.device).to(dtype)
                _, probs = model.detect_language(head)
                opts["language"] = max(probs, key=probs.get)
                if self.verbose is not None:
                    print(f"Detected language: {LANGUAGES[opts['language']].title()}")

        language: str = opts["language"]
        task: str = opts.get("task", "transcribe")
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language, task=task)

        clips = self.clip_timestamps
        if isinstance(clips, str):
            clips = [float(ts) for ts in (clips.split(",") if clips else [])]
        seek_points: List[int] = [round(ts * FRAMES_PER_SECOND) for ts in clips]
        if len(seek_points) == 0:
            seek_points.append(0)
        if len(seek_points) % 2 == 1:
            seek_points.append(content_frames)
        seek_clips: List[Tuple[int, int]] = list(zip(seek_points[::2], seek_points[1::2]))

        if self.word_timestamps and task == "translate":
            warnings.warn("Word-level timestamps on translations may not be reliable.")

        input_stride = exact_div(N_FRAMES, model.dims.n_audio_ctx)           # mel frames per timestamp token (2)
        time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE             # 0.02 s
        all_tokens: List[int] = []
        all_segments: List[dict] = []
        prompt_reset_since = 0
        remaining_prompt_length = model.dims.n_text_ctx // 2 - 1
        if self.initial_prompt is not None:
            initial_prompt_tokens = tokenizer.encode(" " + self.initial_prompt.strip())
            all_tokens.extend(initial_prompt_tokens)
            remaining_prompt_length -= len(initial_prompt_tokens)
        else:
            initial_prompt_tokens = []

        clip_idx = 0
        seek = seek_clips[clip_idx][0]
        last_speech_timestamp = 0.0
        threshold = self.hallucination_silence_threshold

        def make_segment(*, seek_at: int, start: float, end: float, tokens: torch.Tensor, result: DecodingResult) -> dict:
            ids = tokens.tolist()
            return {
                "seek": seek_at, "start": start, "end": end,
                "text": tokenizer.decode([t for t in ids if t < tokenizer.eot]),
                "tokens": ids, "temperature": result.temperature, "avg_logprob": result.avg_logprob,
                "compression_ratio": result.compression_ratio, "no_speech_prob": result.no_speech_prob,
            }

        with tqdm.tqdm(total=content_frames, unit="frames", disable=self.verbose is not False) as pbar:
            while clip_idx < len(seek_clips):
                clip_start, clip_end = seek_clips[clip_idx]
                if seek < clip_start:
                    seek = clip_start
                if seek >= clip_end:
                    clip_idx += 1
                    if clip_idx < len(seek_clips):
                        seek = seek_clips[clip_idx][0]
                    continue
                time_offset = float(seek * HOP_LENGTH / SAMPLE_RATE)
                window_end_time = float((seek + N_FRAMES) * HOP_LENGTH / SAMPLE_RATE)
                segment_size = min(N_FRAMES, content_frames - seek, clip_end - seek)
                segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
                mel_segment = pad_or_trim(mel[:, seek: seek + segment_size], N_FRAMES).to(model.device).to(dtype)

                if self.carry_initial_prompt:
                    ignored = max(len(initial_prompt_tokens), prompt_reset_since)
                    tail = all_tokens[ignored:][-remaining_prompt_length:]
                    opts["prompt"] = initial_prompt_tokens + tail
                else:
                    opts["prompt"] = all_tokens[prompt_reset_since:]

                result = yield mel_segment
                tokens = torch.tensor(result.tokens)

                if self.no_speech_threshold is not None:
                    skip = result.no_speech_prob > self.no_speech_threshold
                    if self.logprob_threshold is not None and result.avg_logprob > self.logprob_threshold:
                        skip = False           # confident text wins over the no-speech probability
                    if skip:
                        seek += segment_size
                        continue

                previous_seek = seek
                current_segments: List[dict] = []

                is_ts = tokens.ge(tokenizer.timestamp_begin)
                single_timestamp_ending = is_ts[-2:].tolist() == [False, True]
                pair_ends = (torch.where(is_ts[:-1] & is_ts[1:])[0] + 1).tolist()
                if len(pair_ends) > 0:
                    # consecutive timestamp pairs delimit segments
                    cuts = pair_ends + ([len(tokens)] if single_timestamp_ending else [])
                    begin = 0
                    for cut in cuts:
                        piece = tokens[begin:cut]
                        t_start = piece[0].item() - tokenizer.timestamp_begin
                        t_end = piece[-1].item() - tokenizer.timestamp_begin
                        current_segments.append(make_segment(
                            seek_at=seek, start=time_offset + t_start * time_precision,
                            end=time_offset + t_end * time_precision, tokens=piece, result=result))
                        begin = cut
                    if single_timestamp_ending:
                        seek += segment_size      # nothing spoken after the last timestamp
                    else:
                        last_pos = tokens[begin - 1].item() - tokenizer.timestamp_begin
                        seek += last_pos * input_stride      # drop the unfinished tail, resume at its start
                else:
                    duration = segment_duration
                    stamps = tokens[is_ts.nonzero().flatten()]
                    if len(stamps) > 0 and stamps[-1].item() != tokenizer.timestamp_begin:
                        duration = (stamps[-1].item() - tokenizer.timestamp_begin) * time_precision
                    current_segments.append(make_segment(seek_at=seek, start=time_offset, end=time_offset + duration,
                                                         tokens=tokens, result=result))
                    seek += segment_size

                if self.word_timestamps:
                    # the alignment itself (encoder + teacher-forced pass + DTW) is the driver's job: it may batch the
                    # requests of many files
                    alignment = yield _AlignRequest(alignment_text_tokens(current_segments, tokenizer), mel_segment,
                                                    segment_size, tokenizer, language, task,
                                                    audio_features=getattr(result, "audio_features", None))
                    add_word_timestamps(
                        segments=current_segments, model=model, tokenizer=tokenizer, mel=mel_segment,
                        num_frames=segment_size, prepend_punctuations=self.prepend_punctuations,
                        append_punctuations=self.append_punctuations, last_speech_timestamp=last_speech_timestamp,
                        alignment=alignment)

                    if not single_timestamp_ending:
                        last_word_end = get_end(current_segments)
                        if last_word_end is not None and last_word_end > time_offset:
                            seek = round(last_word_end * FRAMES_PER_SECOND)

                    if threshold is not None:
                        # skip silence around probable hallucinations
                        if not single_timestamp_ending:
                            last_word_end = get_end(current_segments)
                            if last_word_end is not None and last_word_end > time_offset:
                                if window_end_time - last_word_end > threshold:
                                    seek = round(last_word_end * FRAMES_PER_SECOND)
                                else:
                                    seek = previous_seek + segment_size

                        first = _first_segment_with_words(current_segments)
                        if first is not None and _is_segment_anomaly(first):
                            gap = first["start"] - time_offset
                            if gap > threshold:
                                seek = previous_seek + round(gap * FRAMES_PER_SECOND)
                                continue

                        hal_last_end = last_speech_timestamp
                        for si in range(len(current_segments)):
                            segment = current_segments[si]
                            if not segment["words"]:
                                continue
                            if _is_segment_anomaly(segment):
                                nxt = _first_segment_with_words(current_segments[si + 1:])
                                hal_next_start = nxt["words"][0]["start"] if nxt is not None else time_offset + segment_duration
                                silence_before = (segment["start"] - hal_last_end > threshold
                                                  or segment["start"] < threshold
                                                  or segment["start"] - time_offset < 2.0)
                                silence_after = (hal_next_start - segment["end"] > threshold
                                                 or _is_segment_anomaly(nxt)
                                                 or window_end_time - segment["end"] < 2.0)
                                if silence_before and silence_after:
                                    seek = round(max(time_offset + 1, segment["start"]) * FRAMES_PER_SECOND)
                                    if content_duration - segment["end"] < threshold:
                                        seek = content_frames
                                    current_segments[si:] = []
                                    break
                            hal_last_end = segment["end"]

                    last_word_end = get_end(current_segments)
                    if last_word_end is not None:
                        last_speech_timestamp = last_word_end

                if self.verbose:
                    for segment in current_segments:
                        line = (f"[{format_timestamp(segment['start'])} --> {format_timestamp(segment['end'])}] "
                                f"{segment['text']}")
                        print(make_safe(line))

                for segment in current_segments:      # instantaneous or empty segments carry no text
                    if segment["start"] == segment["end"] or segment["text"].strip() == "":
                        segment["text"] = ""
                        segment["tokens"] = []
                        segment["words"] = []

                base = len(all_segments)
                all_segments.extend({"id": base + i, **segment} for i, segment in enumerate(current_segments))
                all_tokens.extend(token for segment in current_segments for token in segment["tokens"])

                if not self.condition_on_previous_text or result.temperature > 0.5:
                    prompt_reset_since = len(all_tokens)      # do not condition on a high-temperature output

                pbar.update(min(content_frames, seek) - previous_seek)

        return dict(text=tokenizer.decode(all_tokens[len(initial_prompt_tokens):]), segments=all_segments,
                    language=language)


def transcribe(
    model: "Whisper",
    audio: Union[str, np.ndarray, torch.Tensor],
    *,
    verbose: Optional[bool] = None,
    temperature: Union[float, Tuple[float, ...]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
    compression_ratio_threshold: Optional[float] = 2.4,
    logprob_threshold: Optional[float] = -1.0,
    no_speech_threshold: Optional[float] = 0.6,
    condition_on_previous_text: bool = True,
    initial_prompt: Optional[str] = None,
    carry_initial_prompt: bool = False,
    word_timestamps: bool = False,
    prepend_punctuations: str = "\"'“¿([{-",
    append_punctuations: str = "\"'.。,，!！?？:：”)]}、",
    clip_timestamps: Union[str, List[float]] = "0",
    hallucination_silence_threshold: Optional[float] = None,
    **decode_options,
):
    """Transcribe an audio file / waveform.  Arguments and return value as in the reference
    (whisper/transcribe.py:38-125): returns {"text", "segments", "language"}; `decode_options` are forwarded to
    `DecodingOptions`; `temperature` may be a tuple of fallback temperatures; `clip_timestamps` selects
    start,end,start,end,... ranges in seconds; `word_timestamps` adds per-word timing from cross-attention + DTW.
    """
    return _Transcriber(
        model, verbose, temperature, compression_ratio_threshold, logprob_threshold, no_speech_threshold,
        condition_on_previous_text, initial_prompt, carry_initial_prompt, word_timestamps, prepend_punctuations,
        append_punctuations, clip_timestamps, hallucination_silence_threshold, decode_options,
    ).run(audio)


def _options_key(opts: dict):
    """hashable identity of a DecodingOptions kwargs dict without its prompt: windows may share one batched decode
    only if every other option is the same (the prompts go to DecodingTask as one token list per row)"""
    def freeze(v):
        if isinstance(v, (list, tuple)):
            return tuple(freeze(x) for x in v)
        if isinstance(v, torch.Tensor):
            return tuple(v.flatten().tolist())
        return v
    return tuple(sorted((k, freeze(v)) for k, v in opts.items() if k != "prompt"))


def _prompt_batches(model: "Whisper", options: DecodingOptions, prompts: List[Optional[List[int]]], members: List[int],
                    batch_size: int) -> List[List[int]]:
    """split `members` (windows with the same options and the prompts `prompts[i]`) into batches one DecodingTask
    can take: rows whose initial sequences have different lengths may share a call in the device-side modes (greedy,
    sampling, beam search with the stock decoder / filters) below the length limit (DecodingTask.ragged_limit: no row
    may reach the context limit before the step budget ends); any rows of EQUAL length may always share one"""
    probe = DecodingTask(model, options)
    limit = probe.ragged_limit()
    classes = {}
    for i in members:
        n = len(probe._get_initial_tokens(prompts[i]))
        classes.setdefault("ragged" if limit is not None and n <= limit else n, []).append(i)
    return [rows[at: at + batch_size] for rows in classes.values() for at in range(0, len(rows), batch_size)]


def _load_all(audios) -> list:
    """decode the inputs that are file paths concurrently (ffmpeg subprocesses / the native FLAC decoder release the
    GIL) instead of one after the other at the start of every file's state machine; arrays pass through untouched"""
    paths = [i for i, a in enumerate(audios) if isinstance(a, str)]
    out = list(audios)
    if len(paths) == 0:
        return out
    if len(paths) == 1:
        out[paths[0]] = load_audio(out[paths[0]])
        return out
    from concurrent.futures import ThreadPoolExecutor
    from .utils import usable_cores
    with ThreadPoolExecutor(max(1, min(len(paths), usable_cores(), 16))) as pool:
        for i, samples in zip(paths, pool.map(lambda i: load_audio(out[i]), paths)):
            out[i] = samples
    return out


def transcribe_batch(model: "Whisper", audios, *, batch_size: int = 24, max_active_files: Optional[int] = None,
                     in_flight: int = 1, **kwargs) -> List[dict]:
    """Transcribe several files at once (no counterpart in the reference, which is strictly
    one file at a time).  Every file keeps its own seek / prompt / fallback state machine exactly as `transcribe`;
    the driver advances them in lock-step and decodes the windows that are pending at the same moment — and whose
    decoding options other than the prompt are identical — as batches of up to `batch_size` rows (default 24: the widest decode
    chain the row-tiled projection kernels take, 110 us per clip and step on large-v3 against 130 at 16 rows and 180 at 8), each row
    conditioned on its own file's previous text (`DecodingTask(..., prompts=...)`: rows of different prompt lengths
    share a call in every device-side mode — greedy, sampling and beam search with the stock decoder and filters; the
    beams of a segment share its prompt.  With ragged rows the beam loop copies whole cache rows when beams are
    reordered instead of only the part after the shared history).  Results are the same dicts `transcribe`
    returns, in input order.  Windows of one file stay sequential (seek and prompt depend on the previous window).
    Windows whose result trips the temperature-fallback criteria climb the temperature ladder together: the next
    rung decodes them as a batch again (sampling runs on the device, `best_of` rows per window).  With
    `word_timestamps` the alignments of the windows just decoded are computed together (find_alignment_batch).
    At most `max_active_files` (default 2 * batch_size) files are in flight: only those are decoded to PCM and keep
    their whole-file spectrogram on the device; the next file starts when one finishes.
    `in_flight` > 1: the files are dealt round-robin into that many groups and every group is driven as described above on a
    host thread and HIP stream of its own (decoding.run_in_lanes): the groups' decode chains overlap on the GPU — each is a
    chain of dependent launches that leaves the chip idle between them — while every file's result at temperature 0 stays
    exactly what it is with in_flight = 1 (a file's windows never depend on another file; the draws of the temperature ladder
    are repeatable under torch.manual_seed but belong to the group, so they differ from the in_flight = 1 draws).  `batch_size`
    applies per group; an explicit `max_active_files` is divided between the groups, so that bound on resident spectrograms holds
    for the call as a whole (the default is 2 * batch_size per group), while each group keeps its own cached decoding tasks (workspaces scale with the number of groups).  Prefer
    a larger `batch_size` (rows per decode chain, up to 24) over more groups when there are enough files: a wider chain streams
    the decoder's weights once for all its rows.  Worth it from about 2 * batch_size files on."""
    audios = list(audios)
    if in_flight > 1 and len(audios) > 1:
        from .decoding import run_in_lanes
        n = min(int(in_flight), len(audios))
        groups = [list(range(k, len(audios), n)) for k in range(n)]
        fp16 = kwargs.get("fp16", True)
        # an explicit bound holds for the call as a whole; the default (None) stays 2 * batch_size PER GROUP, so that every
        # group can fill its batches
        per_group = None if max_active_files is None else max(1, int(max_active_files) // n)

        def job(ids):
            return lambda: transcribe_batch(model, [audios[i] for i in ids], batch_size=batch_size,
                                            max_active_files=per_group, in_flight=1, **kwargs)
        parts = run_in_lanes(model, [job(ids) for ids in groups], n, torch.float16 if fp16 else torch.float32)
        merged: List[Optional[dict]] = [None] * len(audios)
        for ids, part in zip(groups, parts):
            for i, r in zip(ids, part):
                merged[i] = r
        return merged
    names = ("verbose", "temperature", "compression_ratio_threshold", "logprob_threshold", "no_speech_threshold",
             "condition_on_previous_text", "initial_prompt", "carry_initial_prompt", "word_timestamps",
             "prepend_punctuations", "append_punctuations", "clip_timestamps", "hallucination_silence_threshold")
    defaults = dict(verbose=None, temperature=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), compression_ratio_threshold=2.4,
                    logprob_threshold=-1.0, no_speech_threshold=0.6, condition_on_previous_text=True,
                    initial_prompt=None, carry_initial_prompt=False, word_timestamps=False,
                    prepend_punctuations="\"'“¿([{-", append_punctuations="\"'.。,，!！?？:：”)]}、",
                    clip_timestamps="0", hallucination_silence_threshold=None)
    fixed = {k: kwargs.pop(k, defaults[k]) for k in names}
    audios = list(audios)
    if batch_size < 1:
        raise ValueError(f"batch_size must be at least 1 (got {batch_size})")
    if max_active_files is None:
        max_active_files = 2 * batch_size
    if max_active_files < 1:
        raise ValueError(f"max_active_files must be at least 1 (got {max_active_files})")
    workers = [_Transcriber(model, *[fixed[k] for k in names], dict(kwargs)) for _ in audios]
    detect = kwargs.get("language") is None and model.is_multilingual and len(audios) > 1
    walks: List[Optional[object]] = [None] * len(audios)
    results: List[Optional[dict]] = [None] * len(audios)
    pending = {}
    waiting = list(range(len(audios)))          # files not started yet: neither decoded nor on the device

    def advance(i: int, value):
        try:
            pending[i] = walks[i].send(value) if value is not None else next(walks[i])
        except StopIteration as stop:
            pending.pop(i, None)
            walks[i] = None                      # drops the file's whole-file mel
            results[i] = stop.value

    def admit():
        """start state machines until `max_active_files` are running: only those files are decoded to PCM and hold a
        whole-file spectrogram on the device, so memory follows the window, not the total audio duration"""
        room = max_active_files - len(pending)
        if room <= 0 or not waiting:
            return
        take = waiting[:room]
        del waiting[:room]
        loaded = _load_all([audios[i] for i in take])
        mels = [None] * len(take)
        if detect:
            # language identification (transcribe.py:139-152) of the admitted files in batched passes instead of one
            # encoder pass + one decoder step per file; every state machine then starts with its language set
            dtype = torch.float16 if kwargs.get("fp16", True) and model.device != torch.device("cpu") else torch.float32
            mels = [log_mel_spectrogram(a, model.dims.n_mels, padding=N_SAMPLES, device=model.device) for a in loaded]
            for at in range(0, len(mels), batch_size):
                heads = torch.stack([pad_or_trim(m, N_FRAMES).to(model.device).to(dtype) for m in mels[at: at + batch_size]])
                _, probs = model.detect_language(heads)
                for i, p in zip(take[at: at + batch_size], probs):
                    workers[i].decode_options["language"] = max(p, key=p.get)
                    if fixed["verbose"] is not None:
                        print(f"Detected language: {LANGUAGES[workers[i].decode_options['language']].title()}")
        for i, a, m in zip(take, loaded, mels):
            walks[i] = workers[i]._walk(a, m)
            advance(i, None)

    admit()
    while pending or waiting:
        if not pending:
            admit()
            continue
        # word alignment requests of the windows just decoded: one find_alignment_batch pass per (language, task)
        aligns = {i: r for i, r in pending.items() if isinstance(r, _AlignRequest)}
        if aligns:
            groups = {}
            for i, r in aligns.items():
                groups.setdefault((r.language, r.task), []).append(i)
            for members in groups.values():
                for at in range(0, len(members), batch_size):
                    chunk = members[at: at + batch_size]
                    reqs = [aligns[i] for i in chunk]
                    if len(chunk) == 1:          # alone: exactly the call `transcribe` makes
                        answers = [find_alignment(model, reqs[0].tokenizer, reqs[0].text_tokens, reqs[0].mel, reqs[0].num_frames,
                                                  audio_features=reqs[0].audio_features)]
                    else:
                        feats = None
                        if all(r.audio_features is not None for r in reqs):
                            feats = torch.stack([r.audio_features for r in reqs])
                        answers = find_alignment_batch(model, reqs[0].tokenizer, [r.text_tokens for r in reqs],
                                                       torch.stack([r.mel for r in reqs]), [r.num_frames for r in reqs],
                                                       audio_features=feats)
                    for i, answer in zip(chunk, answers):
                        advance(i, answer)
            admit()
            continue
        # one round = the windows pending right now, taken up the temperature ladder together: every rung decodes,
        # in batches, the windows that still fail the fallback criteria of transcribe.py:202-222 at the rung below
        rung = {i: 0 for i in pending}
        todo = sorted(pending)
        answers = {}
        encoded = {}
        while todo:
            groups = {}
            for i in todo:
                t = workers[i].temperatures[rung[i]]
                groups.setdefault((_options_key(workers[i].decode_options), t), []).append(i)
            todo = []
            for (_, t), members in groups.items():
                shared = replace(workers[members[0]]._options_for(t), prompt=None)
                prompts = {i: workers[i].decode_options.get("prompt") for i in members}
                for chunk in _prompt_batches(model, shared, prompts, members, batch_size):
                    # a window on a higher rung has been encoded already: its retry takes the features
                    inputs = [encoded.get(i, pending[i]) for i in chunk]
                    if len({tuple(x.shape) for x in inputs}) > 1:
                        inputs = [pending[i] for i in chunk]
                    if len(chunk) == 1:        # alone: exactly the call `transcribe` makes
                        decoded = [model.decode(inputs[0], workers[chunk[0]]._options_for(t))]
                    else:
                        decoded = model.decode(torch.stack(inputs), shared, prompts=[prompts[i] for i in chunk])
                    for i, result in zip(chunk, decoded):
                        if workers[i]._needs_retry(result) and rung[i] + 1 < len(workers[i].temperatures):
                            rung[i] += 1
                            todo.append(i)
                            feats = getattr(result, "audio_features", None)
                            if torch.is_tensor(feats) and feats.shape[-2:] == (model.dims.n_audio_ctx, model.dims.n_audio_state):
                                encoded[i] = feats
                        else:
                            answers[i] = result
            todo.sort()
        for i, result in sorted(answers.items()):
            advance(i, result)
        admit()
    return results
