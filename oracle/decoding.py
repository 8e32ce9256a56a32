"""Sampling loop on CPU — restates DecodingTask._main_loop and friends (whisper/decoding.py:272-505,
680-789) row by row, following SURVEY.md Appendix B.2.  Token-id rules are passed in as plain numbers
(SamplingRules) so this file needs no tokenizer."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .model import OracleModel


@dataclass
class SamplingRules:
    sample_begin: int                  # len(initial_tokens), decoding.py:536
    sot_index: int                     # decoding.py:537
    eot: int
    n_ctx: int = 448
    timestamp_begin: Optional[int] = None     # None = without_timestamps (no ApplyTimestampRules, decoding.py:559)
    no_timestamps: Optional[int] = None
    max_initial_timestamp_index: Optional[int] = 50     # round(1.0 / 0.02), decoding.py:561-565
    suppress_blank: bool = True
    blank_token: int = 220                              # tokenizer.encode(" ")[0]
    suppress_tokens: List[int] = field(default_factory=list)   # decoding.py:615-642 (already expanded)
    no_speech: Optional[int] = None


def apply_filters(logits: torch.Tensor, sampled: List[int], r: SamplingRules, mass_rule: bool = True) -> None:
    """In-place on one row of fp32 logits.  decoding.py:423-505 in filter order :554-570.
    mass_rule=False stops before the "timestamp mass" rule (:498-505) — oracle/condition.py needs the two halves."""
    L = len(sampled)
    ninf = -np.inf
    if r.suppress_blank and L == 0:                      # SuppressBlank :428-430
        logits[[r.blank_token, r.eot]] = ninf
    if r.suppress_tokens:                                # SuppressTokens :437-438
        logits[r.suppress_tokens] = ninf
    TB = r.timestamp_begin
    if TB is None:
        return
    if r.no_timestamps is not None:                      # :454-455
        logits[r.no_timestamps] = ninf
    last_ts = L >= 1 and sampled[-1] >= TB               # :461-466
    pen_ts = L < 2 or sampled[-2] >= TB
    if last_ts:
        if pen_ts:
            logits[TB:] = ninf                           # :470
        else:
            logits[: r.eot] = ninf                       # :472
    stamps = [t for t in sampled if t >= TB]
    if stamps:                                           # :477-484
        last = stamps[-1] if (last_ts and not pen_ts) else stamps[-1] + 1
        logits[TB:last] = ninf
    if L == 0:                                           # :486-495
        logits[:TB] = ninf
        if r.max_initial_timestamp_index is not None:
            logits[TB + r.max_initial_timestamp_index + 1:] = ninf
    if not mass_rule:
        return
    lp = F.log_softmax(logits.float(), dim=-1)           # :498-505
    if lp[TB:].logsumexp(dim=-1) > lp[:TB].max():
        logits[:TB] = ninf


def _first_logits(model: OracleModel, feats: torch.Tensor, tokens: torch.Tensor, r: SamplingRules, cache: dict):
    logits = model.decoder(tokens, feats, cache)         # decoding.py:155-163 first call: all positions
    nsp = None
    if r.no_speech is not None:                          # decoding.py:689-693
        nsp = logits[:, r.sot_index].float().softmax(dim=-1)[:, r.no_speech].tolist()
    return logits[:, -1], nsp


def greedy_decode(model: OracleModel, feats: torch.Tensor, initial_tokens: List[int], sample_len: int,
                  r: SamplingRules, keep_logits: bool = False) -> Dict:
    """GreedyDecoder at temperature 0 (decoding.py:277-293) inside _main_loop (:680-710).
    Returns {"tokens": (R, n) int64 incl. the initial tokens, "sum_logprobs": [R], "no_speech_probs": [R]} and, with
    keep_logits, "step_logits": the filtered fp32 logits (R, V) every arg-max was taken from (test helper: margins) and
    "rule_margins": per step and row, timestamp mass minus best text log-probability where decoding.py:498-505 had a
    choice to make (None elsewhere) — the distance by which that rule was decided."""
    kept, kept_rule = [], []
    R = feats.shape[0]
    tokens = torch.tensor([list(initial_tokens)] * R, dtype=torch.int64)
    sum_lp = torch.zeros(R)
    cache = model.new_cache()
    nsp = [float("nan")] * R
    for i in range(sample_len):
        if i == 0:
            logits, ns = _first_logits(model, feats, tokens, r, cache)
            if ns is not None:
                nsp = ns
        else:
            logits = model.decoder(tokens[:, -1:], feats, cache)[:, -1]
        logits = logits.clone()
        nxt = torch.empty(R, dtype=torch.int64)
        for k in range(R):
            if keep_logits and r.timestamp_begin is not None:
                apply_filters(logits[k], tokens[k, r.sample_begin:].tolist(), r, mass_rule=False)
                txt, ts = logits[k, : r.timestamp_begin], logits[k, r.timestamp_begin:]
                both = bool(torch.isfinite(txt).any()) and bool(torch.isfinite(ts).any())
                kept_rule.append(float(ts.float().logsumexp(0) - txt.max()) if both else None)
            apply_filters(logits[k], tokens[k, r.sample_begin:].tolist(), r)
            nxt[k] = int(logits[k].argmax())
            lp = F.log_softmax(logits[k].float(), dim=-1)[nxt[k]]
            if tokens[k, -1] != r.eot:
                sum_lp[k] += lp
            else:
                nxt[k] = r.eot
        if keep_logits:
            kept.append(logits)
        tokens = torch.cat([tokens, nxt[:, None]], dim=-1)
        if bool((tokens[:, -1] == r.eot).all()) or tokens.shape[-1] > r.n_ctx:      # :292, :705
            break
    out = {"tokens": tokens, "sum_logprobs": sum_lp.tolist(), "no_speech_probs": nsp}
    if keep_logits:
        out["step_logits"] = kept
        out["rule_margins"] = kept_rule
    return out


def filtered_logits(model: OracleModel, feats: torch.Tensor, initial_tokens: List[int], sampled: List[int],
                    r: SamplingRules) -> torch.Tensor:
    """Filtered fp32 logits of ONE row after `sampled` tokens have been emitted (teacher-forced, no cache): the
    vector GreedyDecoder.update takes the arg-max of (decoding.py:277-283).  Test helper for the fp16 engine: where
    its token ids leave the oracle's, the margin between the two candidates here tells a rounding-level near-tie
    from an error."""
    toks = torch.tensor([list(initial_tokens) + list(sampled)], dtype=torch.int64)
    with torch.no_grad():
        logits = model.decoder(toks, feats[None] if feats.dim() == 2 else feats)[0, -1].clone()
    apply_filters(logits, list(sampled), r)
    return logits


def first_divergence(got: List[int], want: List[int]) -> Optional[int]:
    """index of the first differing token, or None if one list is a prefix of the other and lengths agree"""
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w:
            return i
    return None if len(got) == len(want) else min(len(got), len(want))


def beam_decode(model: OracleModel, feats: torch.Tensor, initial_tokens: List[int], sample_len: int,
                r: SamplingRules, beam_size: int, patience: Optional[float] = None) -> Dict:
    """BeamSearchDecoder (decoding.py:301-404) inside _main_loop.  The reference's batched beam path is
    broken for n_audio > 1, so — as the survey prescribes — the oracle for a batch is each
    audio decoded on its own; rows here are (audio, beam) with cross-KV indexed by row // beam_size.
    Returns {"candidates": per audio list of (token list incl. initial tokens and EOT, sum_logprob),
             "no_speech_probs": [n_audio]}."""
    n_audio = feats.shape[0]
    G = beam_size
    max_cand = round(G * (patience or 1.0))
    R = n_audio * G
    tokens = torch.tensor([list(initial_tokens)] * R, dtype=torch.int64)
    sum_lp = torch.zeros(R)
    finished: List[Dict[Tuple[int, ...], float]] = [{} for _ in range(n_audio)]
    cache = model.new_cache()
    nsp = [float("nan")] * n_audio
    for i in range(sample_len):
        if i == 0:
            logits, ns = _first_logits(model, feats, tokens, r, cache)
            if ns is not None:
                nsp = ns[::G]
        else:
            logits = model.decoder(tokens[:, -1:], feats, cache)[:, -1]
        logits = logits.clone()
        for k in range(R):
            apply_filters(logits[k], tokens[k, r.sample_begin:].tolist(), r)
        logprobs = F.log_softmax(logits.float(), dim=-1)
        next_tokens, sources, newly = [], [], []
        for a in range(n_audio):
            scores: Dict[Tuple[int, ...], float] = {}
            src: Dict[Tuple[int, ...], int] = {}
            fin: Dict[Tuple[int, ...], float] = {}
            for j in range(G):                                        # :339-346
                idx = a * G + j
                prefix = tokens[idx].tolist()
                vals, toks = logprobs[idx].topk(G + 1)
                for lpv, tok in zip(vals, toks):
                    seq = tuple(prefix + [int(tok)])
                    scores[seq] = (sum_lp[idx] + lpv).item()
                    src[seq] = idx
            saved = 0
            for seq in sorted(scores, key=scores.get, reverse=True):  # :349-360
                if seq[-1] == r.eot:
                    fin[seq] = scores[seq]
                else:
                    sum_lp[len(next_tokens)] = scores[seq]
                    next_tokens.append(seq)
                    sources.append(src[seq])
                    saved += 1
                    if saved == G:
                        break
            newly.append(fin)
        tokens = torch.tensor(next_tokens, dtype=torch.int64)
        model.rearrange(cache, sources)                               # :365
        for prev, new in zip(finished, newly):                        # :368-375
            for seq in sorted(new, key=new.get, reverse=True):
                if len(prev) >= max_cand:
                    break
                prev[seq] = new[seq]
        completed = all(len(s) >= max_cand for s in finished)         # :378-381
        if completed or tokens.shape[-1] > r.n_ctx:
            break
    # finalize (:384-404)
    sum_lp2 = sum_lp.reshape(n_audio, G)
    tok3 = tokens.reshape(n_audio, G, -1)
    for a, seqs in enumerate(finished):
        if len(seqs) < G:
            for j in list(np.argsort(sum_lp2[a].numpy()))[::-1]:
                seqs[tuple(tok3[a, j].tolist() + [r.eot])] = sum_lp2[a][j].item()
                if len(seqs) >= G:
                    break
    return {"candidates": [[(list(k), v) for k, v in s.items()] for s in finished], "no_speech_probs": nsp}


def rank_candidates(cands, sample_begin: int, eot: int, length_penalty: Optional[float] = None):
    """MaximumLikelihoodRanker (decoding.py:190-213) after the [sample_begin : first EOT] slice (:749-752).
    Returns (tokens, sum_logprob) of the winner."""
    best, best_score = None, None
    for toks, lp in cands:
        body = toks[sample_begin:]
        body = body[: body.index(eot)] if eot in body else body
        n = len(body)
        pen = n if length_penalty is None else ((5 + n) / 6) ** length_penalty
        score = lp / pen if pen != 0 else (np.inf if lp > 0 else -np.inf if lp < 0 else np.nan)
        if best is None or score > best_score:
            best, best_score = (body, lp), score
    return best


def sequence_logprob(model: OracleModel, feats: torch.Tensor, initial_tokens: List[int], sampled: List[int],
                     r: SamplingRules) -> float:
    """Sum of the filtered log-probabilities the oracle assigns to `sampled` after `initial_tokens` on ONE audio
    (feats (1500, D) or (1, 1500, D)): what BeamSearchDecoder / GreedyDecoder accumulate in sum_logprobs
    (decoding.py:283-287, 341-345) when they emit exactly this sequence.  One teacher-forced pass; an EOT ends the sum
    (tokens after it carry no log-probability, decoding.py:285).  Test helper: scores the sequence an fp16 beam search
    chose under the fp32 model, so that a different winner can be told from a worse one."""
    toks = torch.tensor([list(initial_tokens) + list(sampled)], dtype=torch.int64)
    with torch.no_grad():
        logits = model.decoder(toks[:, :-1], feats[None] if feats.dim() == 2 else feats)[0]
    n0 = len(initial_tokens)
    total = 0.0
    for t, tok in enumerate(sampled):
        row = logits[n0 - 1 + t].clone()
        apply_filters(row, list(sampled[:t]), r)
        total += float(F.log_softmax(row.float(), dim=-1)[tok])
        if tok == r.eot:
            break
    return total
