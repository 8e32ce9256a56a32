"""Decoding of 30-second windows on the HIP path.

Host-side mirror of the reference's `whisper/decoding.py`: the same public names (DecodingOptions,
DecodingResult, Inference, TokenDecoder, LogitFilter, DecodingTask, decode, detect_language), the same option
semantics and error behaviour, so code written against the reference runs unchanged.  What differs is where the
work happens:

* `HipInference` implements the `Inference` seam (reference decoding.py:130-141) on a `wh_task`: pre-allocated
  self/cross KV caches, GEMM prefill, hipGraph-replayed single-token steps, in-place beam reordering.
* the logit filters are vectorised over rows (no per-row `.tolist()` round trips, reference :458-505).
* greedy decoding at temperature 0 with the stock filters runs as ONE C call (`wh_task_greedy`): the filters,
  arg-max and log-prob accumulation are a device kernel (csrc/sampling.hip) and the host never syncs per token.
* batched beam search works for n_audio > 1 (the reference raises there): cross-attention K/V
  is indexed by row // beam_size inside the kernels.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field, replace
from typing import TYPE_CHECKING, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor
from torch.distributions import Categorical

from . import hip
from .audio import CHUNK_LENGTH
from .tokenizer import Tokenizer, get_tokenizer
from .utils import compression_ratio

if TYPE_CHECKING:
    from .model import Whisper

import threading

# the generator the calling thread's sampling seeds are drawn from: None = torch's process-wide CPU generator (so that
# torch.manual_seed makes a run repeatable, as in the reference); `run_in_lanes` / `run_interleaved` give every JOB a generator of
# its own, seeded on the caller's thread from the process-wide one plus the job's index — several lanes must not race for draws
_SEEDS = threading.local()


def _draw_seed() -> int:
    gen = getattr(_SEEDS, "gen", None)
    return int(torch.randint(0, 2 ** 62, (1,), generator=gen).item())


def _job_generator(base: int, index: int) -> torch.Generator:
    return torch.Generator().manual_seed((base + 0x9E3779B97F4A7C15 * (index + 1)) % (2 ** 63))


@contextlib.contextmanager
def _job_seeds(gen: torch.Generator):
    """seeds drawn by the calling thread inside the block come from `gen` (one generator per job, kept across resumes)"""
    prev = getattr(_SEEDS, "gen", None)
    _SEEDS.gen = gen
    try:
        yield
    finally:
        _SEEDS.gen = prev


# ---------------------------------------------------------------------------------------------------------------
# language identification (reference decoding.py:18-77)
# ---------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def detect_language(model: "Whisper", mel: Tensor, tokenizer: Tokenizer = None) -> Tuple[Tensor, List[dict]]:
    """Most probable language token per audio and the probability of every language.  `mel` may be a
    spectrogram (n_mels, 3000) / (B, n_mels, 3000) or already-encoded audio features."""
    if tokenizer is None:
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages)
    if tokenizer.language is None or tokenizer.language_token not in tokenizer.sot_sequence:
        raise ValueError("This model doesn't have language tokens so it can't perform lang id")

    single = mel.ndim == 2
    if single:
        mel = mel.unsqueeze(0)
    if mel.shape[-2:] != (model.dims.n_audio_ctx, model.dims.n_audio_state):
        mel = model.encoder(mel)

    n_audio = mel.shape[0]
    x = torch.tensor([[tokenizer.sot]] * n_audio, device=mel.device)
    logits = model.logits(x, mel)[:, 0]

    keep = torch.zeros(logits.shape[-1], dtype=torch.bool, device=logits.device)
    keep[list(tokenizer.all_language_tokens)] = True
    logits = logits.masked_fill(~keep, -np.inf)
    language_tokens = logits.argmax(dim=-1)
    probs = logits.softmax(dim=-1).cpu()
    language_probs = [
        {c: probs[i, j].item() for j, c in zip(tokenizer.all_language_tokens, tokenizer.all_language_codes)}
        for i in range(n_audio)
    ]
    if single:
        language_tokens = language_tokens[0]
        language_probs = language_probs[0]
    return language_tokens, language_probs


# ---------------------------------------------------------------------------------------------------------------
# options / results: field-for-field the reference's dataclasses (decoding.py:80-127)
# ---------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"                 # "transcribe" (X->X) or "translate" (X->English)
    language: Optional[str] = None           # detected when None
    temperature: float = 0.0
    sample_len: Optional[int] = None         # max tokens to sample
    best_of: Optional[int] = None            # independent samples when temperature > 0
    beam_size: Optional[int] = None          # beams when temperature == 0
    patience: Optional[float] = None         # beam-search patience (arxiv:2204.05424)
    length_penalty: Optional[float] = None   # Google-NMT alpha, None = length normalisation
    prompt: Optional[Union[str, List[int]]] = None   # previous context
    prefix: Optional[Union[str, List[int]]] = None   # forced start of the current context
    suppress_tokens: Optional[Union[str, Iterable[int]]] = "-1"   # "-1" = tokenizer.non_speech_tokens
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True


@dataclass(frozen=True)
class DecodingResult:
    audio_features: Tensor
    language: str
    language_probs: Optional[Dict[str, float]] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = np.nan
    no_speech_prob: float = np.nan
    temperature: float = np.nan
    compression_ratio: float = np.nan


# ---------------------------------------------------------------------------------------------------------------
# the Inference seam
# ---------------------------------------------------------------------------------------------------------------
class Inference:
    def logits(self, tokens: Tensor, audio_features: Tensor) -> Tensor:
        """Forward pass of the decoder: per-token logits"""
        raise NotImplementedError

    def rearrange_kv_cache(self, source_indices) -> None:
        """Re-order the cached keys/values after the beams were re-ranked"""
        raise NotImplementedError

    def cleanup_caching(self) -> None:
        """Release per-task state"""
        pass


class HipInference(Inference):
    """`Inference` on a wh_task (C ABI).  Contract of reference decoding.py:155-176: the first `logits` call
    sees all initial tokens and returns logits for them; later calls look at `tokens[:, -1]` only."""

    def __init__(self, model: "Whisper", initial_token_length: int, n_group: int = 1, capture_q: bool = False):
        self.model = model
        self.initial_token_length = initial_token_length
        self.n_group = n_group
        self.capture_q = capture_q
        self.task: Optional[hip.HipTask] = None
        self.positions: Optional[List[int]] = None      # restrict first-call logits to these positions
        self.two_launch = False                         # keep the task off the fused step kernels (csrc/xattn.hip)
        self._timeouts_at_start = 0

    def _ensure_task(self, tokens: Tensor, audio_features: Tensor) -> hip.HipTask:
        if self.task is None:
            engine = self.model.engine(audio_features.dtype)
            n_rows = tokens.shape[0]
            n_audio = audio_features.shape[0]
            group = n_rows // n_audio if n_rows % n_audio == 0 and n_rows >= n_audio else None
            if group is None:
                raise ValueError(f"rows ({n_rows}) must be a multiple of audio segments ({n_audio})")
            if self.two_launch:      # a task of its own, outside the engine's cache (whose tasks are keyed by shape only)
                self.task = hip.HipTask(engine, n_audio, group, max(tokens.shape[1], 8), capture_q=self.capture_q,
                                        two_launch_self=True, two_launch_cross=True)
            else:
                self.task = engine.acquire_task(n_audio, group, max(tokens.shape[1], 8), capture_q=self.capture_q)
            self.task.set_audio(audio_features.contiguous())
            fused = self.task.fused_cross_attention or self.task.fused_self_attention
            self._timeouts_at_start = self.task.handoff_timeouts() if fused else 0
        return self.task

    def handoff_timed_out(self) -> bool:
        """host-driven steps (wh_task_step) on the fused step kernels: did a bounded hand-off spin run out since this
        task was taken?  (The device-side loops check and recover inside wh_task_greedy / wh_task_beam.)  Synchronises."""
        t = self.task
        if t is None or not (t.fused_cross_attention or t.fused_self_attention):
            return False
        return t.handoff_timeouts() != self._timeouts_at_start

    def logits(self, tokens: Tensor, audio_features: Tensor) -> Tensor:
        task = self._ensure_task(tokens, audio_features)
        if task.position == 0:
            return task.prefill(tokens.contiguous(), sel=self.positions)
        if tokens.shape[-1] > self.initial_token_length or task.position == tokens.shape[-1] - 1:
            return task.step(tokens[:, -1])[:, None]
        return task.prefill(tokens[:, task.position:].contiguous())

    def rearrange_kv_cache(self, source_indices) -> None:
        if self.task is not None:
            self.task.rearrange([int(i) for i in source_indices])

    def cleanup_caching(self) -> None:
        if self.task is not None:
            self.task.close()
            self.task = None


# ---------------------------------------------------------------------------------------------------------------
# ranking
# ---------------------------------------------------------------------------------------------------------------
class SequenceRanker:
    def rank(self, tokens: List[List[Tensor]], sum_logprobs: List[List[float]]) -> List[int]:
        """index of the chosen sample in every group"""
        raise NotImplementedError


class MaximumLikelihoodRanker(SequenceRanker):
    """Highest log-probability after length normalisation (length_penalty None) or the Google-NMT penalty
    ((5 + len) / 6) ** alpha (reference decoding.py:190-213)."""

    def __init__(self, length_penalty: Optional[float]):
        self.length_penalty = length_penalty

    def rank(self, tokens: List[List[Tensor]], sum_logprobs: List[List[float]]):
        picks = []
        for group, logprobs in zip(tokens, sum_logprobs):
            scores = []
            for seq, lp in zip(group, logprobs):
                n = len(seq)
                penalty = n if self.length_penalty is None else ((5 + n) / 6) ** self.length_penalty
                scores.append(lp / penalty)
            picks.append(int(np.argmax(scores)))
        return picks


# ---------------------------------------------------------------------------------------------------------------
# token decoders
# ---------------------------------------------------------------------------------------------------------------
class TokenDecoder:
    def reset(self):
        """forget state from a previous sequence"""

    def update(self, tokens: Tensor, logits: Tensor, sum_logprobs: Tensor) -> Tuple[Tensor, bool]:
        """tokens (n_batch, len), logits (n_batch, vocab), sum_logprobs (n_batch) ->
        (tokens with one more column, True when every sequence is finished)"""
        raise NotImplementedError

    def finalize(self, tokens: Tensor, sum_logprobs: Tensor) -> Tuple[Sequence[Sequence[Tensor]], List[List[float]]]:
        """tokens (n_audio, n_group, len), sum_logprobs (n_audio, n_group) -> candidate sequences and their
        cumulative log-probabilities per audio"""
        raise NotImplementedError


class GreedyDecoder(TokenDecoder):
    """arg-max (temperature 0) or categorical sampling; rows that already produced EOT keep producing EOT and
    stop accumulating log-probability (reference decoding.py:272-298)."""

    def __init__(self, temperature: float, eot: int):
        self.temperature = temperature
        self.eot = eot
        self._buf: Optional[Tensor] = None
        # True only while DecodingTask._host_loop drives this decoder: it hands every returned `tokens` straight back, so the new
        # column can be written in place.  Any other caller gets the reference's semantics — a fresh tensor per update
        # (torch.cat, decoding.py:290) that never aliases an earlier result.
        self._in_place = False

    def reset(self):
        self._buf = None

    def _append(self, tokens: Tensor, picked: Tensor) -> Tensor:
        """`torch.cat([tokens, picked[:, None]], -1)` (reference decoding.py:290) without a reallocation + full copy per step:
        the sequences live in a buffer with spare columns; as long as the caller hands back the view this method returned,
        the new column is written in place and a one-column-wider view of the same storage goes out.  Any other `tokens`
        (first call, a caller that built its own tensor) is copied into a fresh buffer — same values either way."""
        if not self._in_place:
            return torch.cat([tokens, picked[:, None]], dim=-1)
        n = tokens.shape[1]
        buf = self._buf
        if (buf is None or buf.shape[0] != tokens.shape[0] or buf.dtype != tokens.dtype or buf.device != tokens.device
                or n + 1 > buf.shape[1] or tokens.data_ptr() != buf.data_ptr() or (tokens.shape[0] > 1 and tokens.stride(0) != buf.stride(0))
                or tokens.stride(1) != 1):
            buf = torch.empty(tokens.shape[0], max(2 * (n + 1), 64), dtype=tokens.dtype, device=tokens.device)
            buf[:, :n] = tokens
            self._buf = buf
        buf[:, n] = picked
        return buf[:, : n + 1]

    def update(self, tokens: Tensor, logits: Tensor, sum_logprobs: Tensor) -> Tuple[Tensor, bool]:
        if self.temperature == 0:
            picked = logits.argmax(dim=-1)
        else:
            picked = Categorical(logits=logits / self.temperature).sample()
        logprobs = F.log_softmax(logits.float(), dim=-1)
        chosen = logprobs.gather(1, picked[:, None])[:, 0]
        ended = tokens[:, -1] == self.eot
        sum_logprobs += chosen * (~ended)
        picked = torch.where(ended, torch.full_like(picked, self.eot), picked)
        tokens = self._append(tokens, picked)
        return tokens, bool((tokens[:, -1] == self.eot).all())

    def finalize(self, tokens: Tensor, sum_logprobs: Tensor):
        return F.pad(tokens, (0, 1), value=self.eot), sum_logprobs.tolist()   # at least one EOT per sequence


class BeamSearchDecoder(TokenDecoder):
    """Beam search with patience, same candidate bookkeeping as reference decoding.py:301-404 (candidate
    sequences keyed by their token tuple, later duplicates overwrite, stable descending sort), but with one
    device->host transfer per step instead of one per candidate."""

    def __init__(self, beam_size: int, eot: int, inference: Inference, patience: Optional[float] = None):
        self.beam_size = beam_size
        self.eot = eot
        self.inference = inference
        self.patience = patience or 1.0
        self.max_candidates: int = round(beam_size * self.patience)
        self.finished_sequences = None
        assert self.max_candidates > 0, f"Invalid beam size ({beam_size}) or patience ({patience})"

    def reset(self):
        self.finished_sequences = None

    def update(self, tokens: Tensor, logits: Tensor, sum_logprobs: Tensor) -> Tuple[Tensor, bool]:
        G = self.beam_size
        if tokens.shape[0] % G != 0:
            raise ValueError(f"{tokens.shape}[0] % {G} != 0")
        n_audio = tokens.shape[0] // G
        if self.finished_sequences is None:
            self.finished_sequences = [{} for _ in range(n_audio)]

        logprobs = F.log_softmax(logits.float(), dim=-1)
        top_lp, top_tok = logprobs.topk(G + 1, dim=-1)
        cand_scores = (sum_logprobs[:, None] + top_lp).cpu().tolist()     # fp32 sums, read back as Python floats
        cand_tokens = top_tok.cpu().tolist()
        prefixes = tokens.cpu().tolist()

        next_tokens, source_indices, newly_finished, new_sums = [], [], [], []
        for a in range(n_audio):
            scores, sources = {}, {}
            for j in range(G):
                row = a * G + j
                for s, t in zip(cand_scores[row], cand_tokens[row]):
                    seq = tuple(prefixes[row] + [t])
                    scores[seq] = s
                    sources[seq] = row
            finished, kept = {}, 0
            for seq in sorted(scores, key=scores.get, reverse=True):
                if seq[-1] == self.eot:
                    finished[seq] = scores[seq]
                else:
                    new_sums.append(scores[seq])
                    next_tokens.append(seq)
                    source_indices.append(sources[seq])
                    kept += 1
                    if kept == G:
                        break
            newly_finished.append(finished)

        sum_logprobs[: len(new_sums)] = torch.tensor(new_sums, dtype=sum_logprobs.dtype, device=sum_logprobs.device)
        tokens = torch.tensor(next_tokens, device=tokens.device)
        self.inference.rearrange_kv_cache(source_indices)

        assert len(self.finished_sequences) == len(newly_finished)
        for done, new in zip(self.finished_sequences, newly_finished):
            for seq in sorted(new, key=new.get, reverse=True):
                if len(done) >= self.max_candidates:
                    break
                done[seq] = new[seq]
        completed = all(len(s) >= self.max_candidates for s in self.finished_sequences)
        return tokens, completed

    def finalize(self, preceding_tokens: Tensor, sum_logprobs: Tensor):
        sum_logprobs = sum_logprobs.cpu()
        for i, sequences in enumerate(self.finished_sequences):
            if len(sequences) < self.beam_size:      # top up with the best unfinished beams
                for j in list(np.argsort(sum_logprobs[i]))[::-1]:
                    sequences[tuple(preceding_tokens[i, j].tolist() + [self.eot])] = sum_logprobs[i][j].item()
                    if len(sequences) >= self.beam_size:
                        break
        tokens = [[torch.tensor(seq) for seq in s.keys()] for s in self.finished_sequences]
        sums = [list(s.values()) for s in self.finished_sequences]
        return tokens, sums


# ---------------------------------------------------------------------------------------------------------------
# logit filters (vectorised over rows; semantics of reference decoding.py:407-505)
# ---------------------------------------------------------------------------------------------------------------
class LogitFilter:
    def apply(self, logits: Tensor, tokens: Tensor) -> None:
        """mask `logits` (n_batch, vocab) in place given the context `tokens` (n_batch, len)"""
        raise NotImplementedError


class SuppressBlank(LogitFilter):
    def __init__(self, tokenizer: Tokenizer, sample_begin: int):
        self.tokenizer = tokenizer
        self.sample_begin = sample_begin

    def apply(self, logits: Tensor, tokens: Tensor):
        if tokens.shape[1] == self.sample_begin:
            logits[:, self.tokenizer.encode(" ") + [self.tokenizer.eot]] = -np.inf


class SuppressTokens(LogitFilter):
    def __init__(self, suppress_tokens: Sequence[int]):
        self.suppress_tokens = list(suppress_tokens)

    def apply(self, logits: Tensor, tokens: Tensor):
        logits[:, self.suppress_tokens] = -np.inf


class ApplyTimestampRules(LogitFilter):
    def __init__(self, tokenizer: Tokenizer, sample_begin: int, max_initial_timestamp_index: Optional[int]):
        self.tokenizer = tokenizer
        self.sample_begin = sample_begin
        self.max_initial_timestamp_index = max_initial_timestamp_index

    def apply(self, logits: Tensor, tokens: Tensor):
        tk = self.tokenizer
        TB, eot = tk.timestamp_begin, tk.eot
        V = logits.shape[1]
        if tk.no_timestamps is not None:
            logits[:, tk.no_timestamps] = -np.inf

        sampled = tokens[:, self.sample_begin:]
        L = sampled.shape[1]
        vocab = torch.arange(V, device=logits.device)[None, :]
        if L >= 1:
            is_ts = sampled >= TB
            last_ts = is_ts[:, -1]
            pen_ts = is_ts[:, -2] if L >= 2 else torch.ones_like(last_ts)
            # timestamps come in pairs (except right before EOT)
            banned = (last_ts & pen_ts)[:, None] & (vocab >= TB)
            banned |= (last_ts & ~pen_ts)[:, None] & (vocab < eot)
            # timestamps never decrease; a closed segment must have non-zero length
            has_ts = is_ts.any(dim=1)
            pos = (is_ts * torch.arange(1, L + 1, device=tokens.device)[None, :]).amax(dim=1) - 1
            newest = sampled.gather(1, pos.clamp(min=0)[:, None])[:, 0]
            floor = torch.where(last_ts & ~pen_ts, newest, newest + 1)
            banned |= has_ts[:, None] & (vocab >= TB) & (vocab < floor[:, None])
            logits.masked_fill_(banned, -np.inf)
        else:
            # first sampled token: must be a timestamp, and not later than max_initial_timestamp
            logits[:, :TB] = -np.inf
            if self.max_initial_timestamp_index is not None:
                logits[:, TB + self.max_initial_timestamp_index + 1:] = -np.inf

        # if the timestamps together are more probable than any single text token, force a timestamp
        logprobs = F.log_softmax(logits.float(), dim=-1)
        ts_mass = logprobs[:, TB:].logsumexp(dim=-1)
        best_text = logprobs[:, :TB].max(dim=-1).values
        logits[:, :TB] = logits[:, :TB].masked_fill((ts_mass > best_text)[:, None], -np.inf)


# ---------------------------------------------------------------------------------------------------------------
# the task
# ---------------------------------------------------------------------------------------------------------------
_FROM_OPTIONS = object()


class DecodingTask:
    inference: Inference
    sequence_ranker: SequenceRanker
    decoder: TokenDecoder
    logit_filters: List[LogitFilter]

    def __init__(self, model: "Whisper", options: DecodingOptions, prompts: Optional[Sequence[Sequence[int]]] = None):
        """`prompts` (no counterpart in the reference, whose task shares ONE initial_tokens tuple between all rows,
        decoding.py:719): one previous-text token list per audio segment, each used exactly as
        `options.prompt` would be for that segment alone.  Prompts of different lengths ("ragged") are decoded by the
        device-side loops (greedy, sampling, beam search: wh_task_set_lag), every row at its own positions;
        `options.prompt` must then be unset."""
        self.model = model
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages,
                                  language=options.language or "en", task=options.task)
        self.tokenizer: Tokenizer = tokenizer
        self.options: DecodingOptions = self._verify_options(options)

        self.n_group: int = options.beam_size or options.best_of or 1
        self.n_ctx: int = model.dims.n_text_ctx
        self.sample_len: int = options.sample_len or model.dims.n_text_ctx // 2

        self.sot_sequence: Tuple[int] = tokenizer.sot_sequence
        if self.options.without_timestamps:
            self.sot_sequence = tokenizer.sot_sequence_including_notimestamps

        # per-segment prompts: `initial_tokens`, `sample_begin` and `sot_index` describe the LONGEST row; row r is
        # that layout shifted left by row_lag[r] tokens
        self.row_tokens: Optional[List[Tuple[int]]] = None
        self.row_lag: Optional[List[int]] = None
        if prompts is not None:
            if options.prompt:
                raise ValueError("per-segment prompts and options.prompt can't be given together")
            self.row_tokens = [self._get_initial_tokens(list(p) if p is not None else None) for p in prompts]
            self.initial_tokens: Tuple[int] = max(self.row_tokens, key=len)
            self.row_lag = [len(self.initial_tokens) - len(r) for r in self.row_tokens]
        else:
            self.initial_tokens: Tuple[int] = self._get_initial_tokens()
        self.sample_begin: int = len(self.initial_tokens)
        self.sot_index: int = self.initial_tokens.index(tokenizer.sot)

        self.inference = HipInference(model, len(self.initial_tokens), self.n_group)
        self.sequence_ranker = MaximumLikelihoodRanker(options.length_penalty)
        if options.beam_size is not None:
            self.decoder = BeamSearchDecoder(options.beam_size, tokenizer.eot, self.inference, options.patience)
        else:
            self.decoder = GreedyDecoder(options.temperature, tokenizer.eot)

        self.logit_filters = []
        self._suppress: Tuple[int, ...] = ()
        self._max_initial_ts: Optional[int] = None
        if self.options.suppress_blank:
            self.logit_filters.append(SuppressBlank(self.tokenizer, self.sample_begin))
        if self.options.suppress_tokens:
            self._suppress = self._get_suppress_tokens()
            self.logit_filters.append(SuppressTokens(self._suppress))
        if not options.without_timestamps:
            precision = CHUNK_LENGTH / model.dims.n_audio_ctx      # 0.02 s per timestamp token
            if options.max_initial_timestamp:
                self._max_initial_ts = round(self.options.max_initial_timestamp / precision)
            self.logit_filters.append(ApplyTimestampRules(tokenizer, self.sample_begin, self._max_initial_ts))
        self._stock_filters = list(self.logit_filters)

    def _verify_options(self, options: DecodingOptions) -> DecodingOptions:
        if options.beam_size is not None and options.best_of is not None:
            raise ValueError("beam_size and best_of can't be given together")
        if options.temperature == 0 and options.best_of is not None:
            raise ValueError("best_of with greedy sampling (T=0) is not compatible")
        if options.patience is not None and options.beam_size is None:
            raise ValueError("patience requires beam_size to be given")
        if options.length_penalty is not None and not (0 <= options.length_penalty <= 1):
            raise ValueError("length_penalty (alpha) should be a value between 0 and 1")
        return options

    def _get_initial_tokens(self, prompt=_FROM_OPTIONS) -> Tuple[int]:
        tokens = list(self.sot_sequence)
        prefix = self.options.prefix
        if prefix:
            prefix_tokens = self.tokenizer.encode(" " + prefix.strip()) if isinstance(prefix, str) else prefix
            if self.sample_len is not None:
                prefix_tokens = prefix_tokens[-(self.n_ctx // 2 - self.sample_len):]
            tokens = tokens + prefix_tokens
        if prompt is _FROM_OPTIONS:
            prompt = self.options.prompt
        if prompt:
            prompt_tokens = self.tokenizer.encode(" " + prompt.strip()) if isinstance(prompt, str) else prompt
            tokens = [self.tokenizer.sot_prev] + prompt_tokens[-(self.n_ctx // 2 - 1):] + tokens
        return tuple(tokens)

    def _get_suppress_tokens(self) -> Tuple[int]:
        suppress = self.options.suppress_tokens
        if isinstance(suppress, str):
            suppress = [int(t) for t in suppress.split(",")]
        if -1 in suppress:
            suppress = [t for t in suppress if t >= 0]
            suppress.extend(self.tokenizer.non_speech_tokens)
        elif suppress is None or len(suppress) == 0:
            suppress = []
        else:
            assert isinstance(suppress, list), "suppress_tokens must be a list"
        tk = self.tokenizer
        suppress.extend([tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm])
        if tk.no_speech is not None:
            suppress.append(tk.no_speech)      # its probability is read separately
        return tuple(sorted(set(suppress)))

    def _get_audio_features(self, mel: Tensor):
        if self.options.fp16:
            mel = mel.half()
        if mel.shape[-2:] == (self.model.dims.n_audio_ctx, self.model.dims.n_audio_state):
            audio_features = mel               # already encoded
        else:
            audio_features = self.model.encoder(mel)
        want = torch.float16 if self.options.fp16 else torch.float32
        if audio_features.dtype != want:
            raise TypeError(f"audio_features has an incorrect dtype: {audio_features.dtype}")
        return audio_features

    def _detect_language(self, audio_features: Tensor, tokens: Tensor):
        languages = [self.options.language] * audio_features.shape[0]
        lang_probs = None
        if self.options.language is None or self.options.task == "lang_id":
            lang_tokens, lang_probs = self.model.detect_language(audio_features, self.tokenizer)
            languages = [max(probs, key=probs.get) for probs in lang_probs]
            if self.options.language is None:
                lang_tokens = lang_tokens.to(tokens.device)
                if self.row_lag is None:
                    tokens[:, self.sot_index + 1] = lang_tokens
                else:
                    for r, lag in enumerate(self.row_lag):
                        tokens[r, self.sot_index - lag + 1] = lang_tokens[r]
        return languages, lang_probs

    # -- sampling loops -------------------------------------------------------------------------------------
    def _fused_greedy_ok(self, tokens: Tensor) -> bool:
        """device-side loop (wh_task_greedy): stock decoder / filters / inference.  Temperature 0 is the exact arg-max
        path; temperature > 0 (any best_of) draws every token on the device from the same categorical distribution
        the reference samples with torch's generator."""
        return (type(self.decoder) is GreedyDecoder and self.options.temperature >= 0
                and (self.n_group == 1 or self.options.temperature > 0)
                and type(self.inference) is HipInference and self.logit_filters == self._stock_filters
                and all(type(f) in (SuppressBlank, SuppressTokens, ApplyTimestampRules) for f in self.logit_filters)
                and self.sample_begin + self.sample_len <= 2 * self.n_ctx)

    def _beam_shape_ok(self) -> bool:
        return (type(self.decoder) is BeamSearchDecoder and self.options.beam_size == self.n_group
                and 2 <= self.n_group <= 8 and type(self.inference) is HipInference
                and self.logit_filters == self._stock_filters
                and all(type(f) in (SuppressBlank, SuppressTokens, ApplyTimestampRules) for f in self.logit_filters))

    def _fused_beam_ok(self) -> bool:
        """device-side beam search (wh_task_beam): stock decoder / filters / inference, 2..8 beams; rows of different
        prompt lengths (every row at its own positions, the beams of a segment sharing its prompt) as long as no row can
        reach the context limit before the step budget ends"""
        return (type(self.decoder) is BeamSearchDecoder and self.options.beam_size == self.n_group
                and 2 <= self.n_group <= 8 and type(self.inference) is HipInference
                and self.logit_filters == self._stock_filters
                and all(type(f) in (SuppressBlank, SuppressTokens, ApplyTimestampRules) for f in self.logit_filters)
                and (not self._ragged() or self.sample_begin + self.sample_len <= self.n_ctx))

    def _sampling_rules(self, T0: int, dev) -> Tuple["hip.GreedyParams", Tensor]:
        """the stock logit filters as the parameter block of the device-side loops (the mask tensor must stay alive)"""
        tk = self.tokenizer
        mask = torch.zeros(self.model.dims.n_vocab, dtype=torch.uint8)
        if self._suppress:
            mask[list(self._suppress)] = 1
        mask = mask.to(dev)
        with_ts = not self.options.without_timestamps
        params = hip.GreedyParams(
            sample_begin=T0, max_steps=self.sample_len, n_ctx=self.n_ctx, eot=tk.eot,
            timestamp_begin=tk.timestamp_begin if with_ts else -1,
            no_timestamps=tk.no_timestamps if tk.no_timestamps is not None else -1,
            max_initial_timestamp_index=self._max_initial_ts if self._max_initial_ts is not None else -1,
            suppress_blank=int(bool(self.options.suppress_blank)), blank_token=tk.encode(" ")[0],
            suppress_mask=mask.data_ptr())
        if type(self.decoder) is GreedyDecoder and self.options.temperature > 0:
            params.temperature = float(self.options.temperature)
            params.seed = _draw_seed()                                    # torch.manual_seed makes runs repeatable
        return params, mask

    # The device-side loops are written as GENERATORS: with `wait` they make the blocking C call (wh_task_greedy / wh_task_beam)
    # and never yield; without it they begin the loop (wh_task_*_begin), then yield while `wh_task_poll` says it is still running —
    # the caller (`run_interleaved`) resumes other tasks' generators meanwhile, all from one host thread.
    def _await(self, pend: "hip.PendingLoop"):
        try:
            while True:
                res = pend.poll()
                if res is not None:
                    return res
                yield
        except GeneratorExit:            # abandoned mid-loop: the task must not go back to the cache with a loop running
            pend.wait()
            raise

    def _main_loop_beam_fused(self, audio_features: Tensor, tokens: Tensor, wait: bool = True):
        """BeamSearchDecoder.update for every step on the device; leaves the decoder's finished_sequences as the
        host loop would (same dict order) for finalize()"""
        tk = self.tokenizer
        dev = audio_features.device
        n_rows, T0 = tokens.shape
        try:
            task = self.inference._ensure_task(tokens, audio_features)
            buf = torch.zeros(2, n_rows, T0 + self.sample_len + 1, dtype=torch.int64, device=dev)
            buf[0, :, :T0] = tokens
            rules, mask = self._sampling_rules(T0, dev)
            params = hip.BeamParams(rules=rules, beam_size=self.n_group, max_candidates=self.decoder.max_candidates)
            no_speech = tk.no_speech if tk.no_speech is not None else -1
            ragged = self._ragged()
            row_lag = [lag for lag in self.row_lag for _ in range(self.n_group)] if ragged else None
            if ragged:
                task.set_lag(row_lag)
            if wait:
                res = task.beam(buf, params, self.sot_index, no_speech)
            else:
                res = yield from self._await(task.beam_begin(buf, params, self.sot_index, no_speech))
            n, sum_logprobs, nsp, (fin_tok, fin_len, fin_score, fin_count) = res
            no_speech_probs = nsp.tolist() if nsp is not None else [np.nan] * n_rows
            fin_tok, fin_len, fin_score, fin_count = fin_tok.cpu(), fin_len.tolist(), fin_score.tolist(), fin_count.tolist()
            # rows with a shorter prompt hold fewer tokens: left-pad (as the greedy path does) so that every sequence's
            # sampled part starts at sample_begin, where finalize() / the ranker slice it
            pad = (lambda a: (tk.sot,) * self.row_lag[a]) if ragged else (lambda a: ())
            self.decoder.finished_sequences = [
                {pad(a) + tuple(fin_tok[a, i, : fin_len[a][i]].tolist()): fin_score[a][i] for i in range(fin_count[a])}
                for a in range(n_rows // self.n_group)]
            if not ragged:
                return buf[0, :, :n], sum_logprobs, no_speech_probs
            out = torch.full((n_rows, n), tk.sot, dtype=torch.int64, device=dev)
            for r, lag in enumerate(row_lag):
                out[r, lag:] = buf[0, r, : n - lag]
            return out, sum_logprobs, no_speech_probs
        finally:
            self.inference.cleanup_caching()

    def _main_loop_fused(self, audio_features: Tensor, tokens: Tensor, wait: bool = True):
        tk = self.tokenizer
        dev = audio_features.device
        n_rows, T0 = tokens.shape
        try:
            task = self.inference._ensure_task(tokens, audio_features)
            buf = torch.zeros(n_rows, T0 + self.sample_len + 1, dtype=torch.int64, device=dev)
            buf[:, :T0] = tokens
            params, mask = self._sampling_rules(T0, dev)
            no_speech = tk.no_speech if tk.no_speech is not None else -1
            ragged = self._ragged()
            row_lag = [lag for lag in self.row_lag for _ in range(self.n_group)] if ragged else None
            if ragged:
                task.set_lag(row_lag)
            if wait:
                res = task.greedy(buf, params, self.sot_index, no_speech)
            else:
                res = yield from self._await(task.greedy_begin(buf, params, self.sot_index, no_speech))
            n, sum_logprobs, nsp = res
            no_speech_probs = nsp.tolist() if nsp is not None else [np.nan] * n_rows
            if not ragged:
                return buf[:, :n], sum_logprobs, no_speech_probs
            # row r holds n - lag tokens: right-align so that every row's sampled part starts at sample_begin
            out = torch.full((n_rows, n), tk.sot, dtype=torch.int64, device=dev)
            for r, lag in enumerate(row_lag):
                out[r, lag:] = buf[r, : n - lag]
            return out, sum_logprobs, no_speech_probs
        finally:
            self.inference.cleanup_caching()

    def _ragged(self) -> bool:
        return self.row_lag is not None and any(self.row_lag)

    def ragged_limit(self) -> Optional[int]:
        """longest initial sequence that rows of different lengths may have in one call (None: this task can only
        take rows of equal length).  Rows share one step counter on the device, so none may reach the context
        limit (reference decoding.py:705) before the `sample_len` budget runs out."""
        fused = self._fused_greedy_ok(None) or (type(self.decoder) is BeamSearchDecoder and self._beam_shape_ok())
        return self.n_ctx - self.sample_len if fused else None

    def _main_loop(self, audio_features: Tensor, tokens: Tensor):
        """decoding.py:680-710: (tokens, sum_logprobs, no_speech_probs), waiting for the device-side loops (the reference's name and
        signature; `_main_loop_steps` is the same as a generator)"""
        steps = self._main_loop_steps(audio_features, tokens, wait=True)
        try:
            while True:
                next(steps)                      # with wait=True nothing ever yields
        except StopIteration as done:
            return done.value

    def _main_loop_steps(self, audio_features: Tensor, tokens: Tensor, wait: bool = True):
        """generator (see `_await`); returns (tokens, sum_logprobs, no_speech_probs)"""
        if self._ragged():
            limit = self.ragged_limit()
            if limit is None:
                raise ValueError("prompts of different lengths need a device-side loop (greedy / sampling / beam search "
                                 "with the stock decoder and logit filters)")
            if self.sample_begin > limit:
                raise ValueError(f"prompts of different lengths: the longest initial sequence ({self.sample_begin}) "
                                 f"+ sample_len ({self.sample_len}) exceeds n_text_ctx ({self.n_ctx})")
        if self._fused_greedy_ok(tokens):
            return (yield from self._main_loop_fused(audio_features, tokens, wait))
        if self._fused_beam_ok():
            return (yield from self._main_loop_beam_fused(audio_features, tokens, wait))
        if type(self.inference) is not HipInference:
            return self._host_loop(audio_features, tokens)[:3]
        first = tokens.clone()
        out = self._host_loop(audio_features, tokens)
        if out[3]:
            # a hand-off spin of the fused step kernels ran out under host-driven steps (never seen on an unshared device):
            # the logits of some step were not valid.  Once more, on a task that uses the two-launch kernels.
            self.decoder.reset()
            self.inference.two_launch = True
            out = self._host_loop(audio_features, first)
        return out[:3]

    def _host_loop(self, audio_features: Tensor, tokens: Tensor):
        """the reference's loop (decoding.py:683-712), one wh_task_step per token; -> (tokens, sum_logprobs,
        no_speech_probs, hand-off timed out)"""
        n_batch = tokens.shape[0]
        sum_logprobs: Tensor = torch.zeros(n_batch, device=audio_features.device)
        no_speech_probs = [np.nan] * n_batch
        timed_out = False
        if type(self.inference) is HipInference:   # only two positions of the first pass are ever read
            self.inference.positions = sorted({self.sot_index, tokens.shape[1] - 1})
        in_place = type(self.decoder) is GreedyDecoder
        if in_place:
            self.decoder._in_place = True           # this loop hands every returned `tokens` straight back (see _append)
        try:
            for i in range(self.sample_len):
                logits = self.inference.logits(tokens, audio_features)
                if i == 0 and self.tokenizer.no_speech is not None:
                    at_sot = 0 if getattr(self.inference, "positions", None) else self.sot_index
                    probs_at_sot = logits[:, at_sot].float().softmax(dim=-1)
                    no_speech_probs = probs_at_sot[:, self.tokenizer.no_speech].tolist()
                logits = logits[:, -1]
                for logit_filter in self.logit_filters:
                    logit_filter.apply(logits, tokens)
                hip_steps = type(self.inference) is HipInference
                try:
                    tokens, completed = self.decoder.update(tokens, logits, sum_logprobs)
                except ValueError:
                    # Categorical(logits=NaN) at temperature > 0: garbage logits after a hand-off time-out must reach the
                    # retry below instead of raising here; anything else is the caller's error
                    if hip_steps and self.inference.handoff_timed_out():
                        timed_out = True
                        break
                    raise
                if completed or tokens.shape[-1] > self.n_ctx:
                    break
                # a time-out invalidates every later step of the window: look every 16 tokens (the loop synchronises with
                # the device at every token anyway) instead of ranking candidates on garbage until the window ends
                if hip_steps and i % 16 == 15 and self.inference.handoff_timed_out():
                    timed_out = True
                    break
            if type(self.inference) is HipInference and not timed_out:
                timed_out = self.inference.handoff_timed_out()
        finally:
            if in_place:
                self.decoder._in_place = False
            self.inference.cleanup_caching()
        return tokens, sum_logprobs, no_speech_probs, timed_out

    @torch.no_grad()
    def run(self, mel: Tensor) -> List[DecodingResult]:
        """decoding.py:713-789.  (`run_steps(mel, wait=False)` is the same as a generator that yields wherever this would
        wait for a device-side loop: `run_interleaved` drives several of those from one thread.)"""
        steps = self.run_steps(mel, wait=True)
        try:
            while True:
                next(steps)                      # with wait=True nothing ever yields
        except StopIteration as done:
            return done.value

    def run_steps(self, mel: Tensor, wait: bool = False):
        """`run` as a generator (resume it under torch.no_grad(), as `run_interleaved` does: a grad-mode context must not be
        held across a yield, several generators share the thread)"""
        self.decoder.reset()
        tokenizer: Tokenizer = self.tokenizer
        n_audio: int = mel.shape[0]

        audio_features: Tensor = self._get_audio_features(mel)
        if self.row_tokens is None:
            tokens: Tensor = torch.tensor([self.initial_tokens]).repeat(n_audio, 1)
        else:
            if len(self.row_tokens) != n_audio:
                raise ValueError(f"{len(self.row_tokens)} prompts for {n_audio} audio segments")
            # shorter rows are padded on the right: causal attention keeps the padding out of the real positions
            tokens = torch.tensor([list(r) + [tokenizer.eot] * lag for r, lag in zip(self.row_tokens, self.row_lag)])

        languages, language_probs = self._detect_language(audio_features, tokens)
        if self.options.task == "lang_id":
            return [DecodingResult(audio_features=f, language=lang, language_probs=probs)
                    for f, lang, probs in zip(audio_features, languages, language_probs)]

        # one row per (audio, beam / sample); the kernels map row -> audio as row // n_group
        tokens = tokens.repeat_interleave(self.n_group, dim=0).to(audio_features.device)
        tokens, sum_logprobs, no_speech_probs = yield from self._main_loop_steps(audio_features, tokens, wait)

        no_speech_probs = no_speech_probs[:: self.n_group]
        assert audio_features.shape[0] == len(no_speech_probs) == n_audio

        tokens = tokens.reshape(n_audio, self.n_group, -1)
        sum_logprobs = sum_logprobs.reshape(n_audio, self.n_group)
        tokens, sum_logprobs = self.decoder.finalize(tokens, sum_logprobs)
        if torch.is_tensor(tokens):
            tokens = tokens.cpu()
        tokens: List[List[Tensor]] = [
            [t[self.sample_begin: (t == tokenizer.eot).nonzero()[0, 0]] for t in s] for s in tokens
        ]

        selected = self.sequence_ranker.rank(tokens, sum_logprobs)
        tokens: List[List[int]] = [t[i].tolist() for i, t in zip(selected, tokens)]
        texts: List[str] = [tokenizer.decode(t).strip() for t in tokens]
        sum_logprobs: List[float] = [lp[i] for i, lp in zip(selected, sum_logprobs)]
        avg_logprobs: List[float] = [lp / (len(t) + 1) for t, lp in zip(tokens, sum_logprobs)]

        fields = (texts, languages, tokens, audio_features, avg_logprobs, no_speech_probs)
        if len(set(map(len, fields))) != 1:
            raise RuntimeError(f"inconsistent result lengths: {list(map(len, fields))}")
        return [
            DecodingResult(audio_features=features, language=language, tokens=tokens, text=text,
                           avg_logprob=avg_logprob, no_speech_prob=no_speech_prob,
                           temperature=self.options.temperature, compression_ratio=compression_ratio(text))
            for text, language, tokens, features, avg_logprob, no_speech_prob in zip(*fields)
        ]


@torch.no_grad()
def decode(model: "Whisper", mel: Tensor, options: DecodingOptions = DecodingOptions(),
           prompts: Optional[Sequence[Sequence[int]]] = None, **kwargs) -> Union[DecodingResult, List[DecodingResult]]:
    """Decode 30-second segment(s) given as (n_mels, 3000) or (*, n_mels, 3000) log-mel spectrograms.
    `prompts`: optional previous-text token list per segment (see DecodingTask)."""
    single = mel.ndim == 2
    if single:
        mel = mel.unsqueeze(0)
    if kwargs:
        options = replace(options, **kwargs)
    result = DecodingTask(model, options, prompts).run(mel)
    return result[0] if single else result


def run_in_lanes(model: "Whisper", jobs: Sequence, in_flight: int = 3, dtype: Optional[torch.dtype] = None) -> list:
    """Call every job — a zero-argument callable that drives this model (log-mel, encoder, a decode loop, ...) — with up to
    `in_flight` of them running at once, each on a host thread and a HIP stream of its own (`HipModel.lane`); results in job
    order.  No counterpart in the reference.  Why it pays: a decode chain of few rows is ~190 dependent launches per token and
    leaves the chip idle between them, so independent chains fill each other's gaps (large-v3, 8 clips per job, greedy, 224
    tokens: 692 audio-s/s one after the other, 917 / 1025 with 2 / 3 in flight, identical tokens; base x 1 clip: 564 -> 1330).
    Three is the useful maximum: the lanes' streams and the engine's encoder stream then occupy the GPU's four hardware queues.
    The threads SLEEP while their loop runs on the device (the library waits on blocking events), so a lane costs no host core;
    `run_interleaved` / `decode_many` need no threads at all.  When there are enough clips, WIDER chains beat more chains:
    `decode_many(chain_rows=24)`.  Jobs must be independent of each other; sampling seeds (temperature > 0) are a function of
    torch's generator state at the call and the job's INDEX, not of which lane runs it when.  An exception in a job — or in a
    lane's own set-up — is re-raised here after the other lanes have stopped; every job either has its result or the call raises."""
    jobs = list(jobs)
    n = max(1, min(int(in_flight), len(jobs)))
    if n <= 1:
        return [job() for job in jobs]
    # the engine is built here, once, before the threads start (packing weights is not something to race on)
    engine = model.engine(dtype if dtype is not None else torch.float16)
    caller = torch.cuda.current_stream(engine.device)        # what the jobs' inputs were produced on
    base = _draw_seed()                                       # ONE draw on the caller's thread; job i derives its own from (base, i)
    missing = object()
    results, errors, streams = [missing] * len(jobs), [], []
    nxt = [0]
    lock = threading.Lock()

    def worker():
        try:
            with engine.lane() as st:
                st.wait_stream(caller)
                with lock:
                    streams.append(st)
                while True:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(jobs) or errors:
                        return
                    with _job_seeds(_job_generator(base, i)):
                        results[i] = jobs[i]()
        except BaseException as e:      # noqa: BLE001 — a job's or the lane's own (stream creation, out of memory): to the caller's thread
            errors.append(e)
    threads = [threading.Thread(target=worker, name=f"whisper-lane-{k}") for k in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for st in streams:                          # device tensors in the results (audio features, ...) are safe to use from here on
        caller.wait_stream(st)
    if errors:
        raise errors[0]
    if any(r is missing for r in results):
        raise RuntimeError("run_in_lanes: a lane ended without running its jobs")
    return results


def run_interleaved(model: "Whisper", jobs: Sequence, in_flight: int = 3, dtype: Optional[torch.dtype] = None,
                    sleep_s: float = 2e-4) -> list:
    """`run_in_lanes` without threads: every job is a GENERATOR (e.g. `DecodingTask(...).run_steps(mel)`) that yields wherever it
    would otherwise wait for a device-side loop; up to `in_flight` of them are kept going from THIS thread, each inside a lane
    (HIP stream) of its own, by resuming them in turn — a resumed job asks its loop whether it has ended (`wh_task_poll`, which
    also queues the next decode steps and never waits) and yields again.  Returns the generators' return values in job order.
    The device sees the same streams and launches as with one host thread per lane; the host spends one thread, mostly asleep
    (`sleep_s` between rounds in which nothing ended; a task has ~10 decode steps queued at any time, so 0.2 ms is early enough
    even for the smallest model).  An exception in a job is re-raised here after the others have been closed (a job abandoned
    mid-loop drains its loop first)."""
    import time
    jobs = list(jobs)
    if not jobs:
        return []
    n = max(1, min(int(in_flight), len(jobs)))
    engine = model.engine(dtype if dtype is not None else torch.float16)
    caller = torch.cuda.current_stream(engine.device)
    base = _draw_seed()
    with engine._lock:
        streams = [engine._lane_pool.pop() if engine._lane_pool else None for _ in range(n)]
    streams = [st if st is not None else torch.cuda.Stream(device=engine.device) for st in streams]
    results = [None] * len(jobs)
    active: Dict[int, Tuple[int, object, torch.Generator]] = {}          # lane -> (job index, generator, its seed generator)
    nxt, error = 0, None

    def resume(lane: int) -> bool:
        """one turn of the job in `lane`; True when it has ended"""
        i, gen, seeds = active[lane]
        with engine.lane(streams[lane]), torch.no_grad(), _job_seeds(seeds):
            try:
                next(gen)
                return False
            except StopIteration as done:
                results[i] = done.value
                return True
    # A new job starts only when every job already running is past its FRONT part (log-mel, encoder, prompt pass and the first
    # decode steps it queued before it first yielded — an event recorded on its stream at that point): jobs started in the same
    # instant would run their encoders back to back with nothing to overlap them, then decode in lock-step and all end together;
    # staggered, one chain's encoder runs under the others' decode steps from the first round on.
    front: Dict[int, torch.cuda.Event] = {}
    try:
        while nxt < len(jobs) or active:
            ended = False
            for lane in range(n):
                if lane not in active and nxt < len(jobs) and all(ev.query() for ev in front.values()):
                    streams[lane].wait_stream(caller)
                    active[lane] = (nxt, jobs[nxt], _job_generator(base, nxt))
                    nxt += 1
                    if resume(lane):
                        del active[lane]
                        ended = True
                    else:
                        front[lane] = torch.cuda.Event()
                        front[lane].record(streams[lane])
            for lane in list(active):
                if resume(lane):
                    del active[lane]
                    front.pop(lane, None)
                    ended = True
            if not ended and active:
                time.sleep(sleep_s)
    except BaseException as e:      # noqa: BLE001
        error = e
    finally:
        for lane, (i, gen, _) in list(active.items()):
            try:
                with engine.lane(streams[lane]):
                    gen.close()
            except BaseException:   # noqa: BLE001 — the first error is the one reported
                pass
        for st in streams:
            caller.wait_stream(st)
        engine.adopt_lane_streams(streams)
    if error is not None:
        raise error
    return results


def coalesce_batches(rows: Sequence[int], chain_rows: Optional[int]) -> List[List[int]]:
    """Indices of consecutive batches grouped into decode chains: a batch joins the chain before it while the chain's rows
    (clips x beams / samples) stay within `chain_rows`; a batch that is wider than `chain_rows` by itself is a chain of its own;
    None / 0 = no coalescing.  Order is kept, every index appears exactly once."""
    chains: List[List[int]] = []
    total = 0
    for i, r in enumerate(rows):
        if chains and chain_rows and total + r <= chain_rows:
            chains[-1].append(i)
            total += r
        else:
            chains.append([i])
            total = r
    return chains


def decode_many(model: "Whisper", mels: Sequence[Tensor], options: DecodingOptions = DecodingOptions(), in_flight: int = 3,
                chain_rows: Optional[int] = 24, **kwargs) -> List[List[DecodingResult]]:
    """`decode(model, mel, options)` for every batch of `mels` — each a (B, n_mels, 3000) tensor, or raw (B, 480000) audio (its
    log-mel is then taken here, per batch: audio.py:155 clamps against the maximum over the tensor it is given) — scheduled for
    throughput; returns the per-batch result lists in order.  No counterpart in the reference.  Two levers:
      * `chain_rows`: consecutive batches are COALESCED into one decode chain while its rows (clips x beams / samples) stay within
        this bound — one task, one prompt pass, one decode step per token for all of them.  The decoder's weights are then
        streamed once per step for up to 24 rows instead of once per batch (large-v3, three batches of 8 clips, greedy: 1.6 GB x 3
        per step become 1.6 GB; 24 rows is where the row-tiled projection kernels of csrc/gemv.hip end).  Every clip's result is
        what `decode` gives it in its own batch: exactly in the fp32 engine; in the fp16 engine up to the order of fp32 partial
        sums (the number of cross-attention key splits depends on the row count), which can only move a decision that was a
        tie to within rounding.  None / 0: every batch is its own chain.
      * `in_flight`: up to this many chains are decoded at once, each on a HIP stream of its own, all driven from the calling
        thread (`run_interleaved`): chains of few rows leave the chip idle between their dependent launches and fill each other's
        gaps.  Chains of 16+ rows of a large model gain little from it."""
    if kwargs:
        options = replace(options, **kwargs)
    dtype = torch.float16 if options.fp16 else torch.float32
    mels = list(mels)
    group = options.beam_size or options.best_of or 1

    def rows_of(m: Tensor) -> int:
        return (m.shape[0] if m.dim() >= 2 else 1) * group

    for m in mels:
        if m.dim() < 2:
            raise ValueError("decode_many takes batches: (B, n_mels, 3000) spectrograms or (B, 480000) audio")
    chains = coalesce_batches([rows_of(m) for m in mels], chain_rows)

    def chain_steps(idx: List[int]):
        parts = []
        for i in idx:
            x = mels[i]
            if x.dim() == 2 and x.shape[-1] == 480000:
                from .audio import log_mel_spectrogram
                x = log_mel_spectrogram(x, model.dims.n_mels)
            parts.append(x.to(dtype))
        res = yield from DecodingTask(model, options).run_steps(parts[0] if len(parts) == 1 else torch.cat(parts), wait=False)
        out, at = [], 0
        for p_ in parts:
            out.append(res[at: at + p_.shape[0]])
            at += p_.shape[0]
        return out
    done = run_interleaved(model, [chain_steps(c) for c in chains], in_flight, dtype)
    return [r for chain in done for r in chain]
