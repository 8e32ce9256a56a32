"""Margin-conditioned synthetic checkpoints.  TEST INFRASTRUCTURE (like the rest of oracle/).

Why.  No released Whisper checkpoint exists offline, and on seeded random-init weights the greedy decision of a step
is the larger of two nearly equal numbers: the top two of ~50 000 i.i.d.-looking logits lie a few hundredths apart
somewhere in every couple of hundred steps, so "token ids equal to the fp32 reference" cannot hold for ANY reduced
precision engine over a 224-step decode, and a parity test has to fall back on a near-tie rule — which cannot tell
rounding from a defect.  A trained model does not look like that: its next-token distribution is peaked.  Scaling
the tied embedding or the final LayerNorm does not make random weights peaked — it multiplies margins and rounding
errors alike.  What does: the rows of the tied embedding `decoder.token_embedding.weight` (model.py:245-247) of the
tokens that the decode actually emits are moved along the hidden state that emits them — by exactly as much as it
takes for the oracle's arg-max (decoding.py:277-283) to win by a drawn margin, and for the "timestamp mass" rule
(decoding.py:498-505) to be decided by the same margin.

How.  One greedy pass of the oracle (fp32, reference operation order).  At step i, row k, with h = ln(x)[k] the hidden
state model.py:244 produces and E the embedding so far:
  1. hard filters (decoding.py:423-495) on  E h;
  2. the token class follows a speech-like script (a timestamp, runs of text tokens, timestamp pairs between them;
     the hard filters force most of it), the target y is the best-scoring admissible token of that class that no row
     has emitted yet (for timestamps: among the next few values, so that a row does not run to <|30.00|> at once);
  3. m ~ U[margin_lo, margin_hi];  E[y] += delta * d / (d . h), d = h minus its projection on the span of (a) every
     EARLIER hidden state of the same row and (b) the other rows' running mean hidden states.  (A row's hidden states
     share a large audio-dependent component, cos ~ 0.96 between steps, so the best tokens of one step are the
     runners-up of the row's other steps: an edit along h itself would raise y's logit at every other step of the row
     almost as much.  With (a) an edit leaves the logits of all earlier steps of its row exactly as they were when
     their margins were built; (b) keeps it from lifting y in the other rows.)  delta is the smallest value >= 0 such
     that, after rounding E[y] to fp16 (both engines and the oracle see the same weights),
         logit[y] >= (best other token of its class) + m     and
         logit[y] >= (logsumexp of the timestamps, if y is text | best text token, if y is a timestamp) + m;
  4. y is emitted; its embedding row is never touched again (it is an INPUT of the following steps).
A row of E changes by ~delta / |d| ~ 0.1-0.2 against a norm of 1.8; nothing else in the checkpoint changes.
Edits for later steps move earlier logits a little (hidden states are not orthogonal), so the caller always re-runs
the plain `greedy_decode` on the finished weights and asserts the margins there (`margins_of`).

The result is conditioned on (weights, audio features, prompt, rules): a synthetic checkpoint on which THESE clips
decode with real-model-like margins.  The same kernels run as on any other weights.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .decoding import SamplingRules, apply_filters
from .model import OracleModel

EMB = "decoder.token_embedding.weight"


def center_cross_values(om: OracleModel, feats: torch.Tensor) -> torch.Tensor:
    """A random-init ENCODER maps every audio to nearly the same output: its frames share one large constant vector c
    (|c| = 34 of |x| = 35.8 on bench.py's synthetic clips, cos 0.999 between clips — the "ordered phase" of a deep random
    network).  Cross-attention keys do not care (q . W_k c is the same for every frame and cancels in the softmax), but
    every value carries W_v c, so each decoder layer adds the same audio- and token-independent vector to the residual
    stream and the hidden states of different steps and rows become near-copies (cos 0.994 / 0.995): there is then no
    direction left along which a token could be told from its neighbours (see `condition_greedy`).  A trained encoder has
    no such DC term.  This removes it where the architecture allows: `decoder.blocks.*.cross_attn.value.bias` (model.py:
    88-89, 106-111) -= W_v c with c = the mean encoder output over the given clips and frames, rounded to fp16-exact
    values.  IN PLACE in om.sd; returns c.  After it the same clips give cos 0.93 between steps and 0.93 between rows."""
    c = feats.float().mean(dim=(0, 1))
    for i in range(om.dims.n_text_layer):
        p = f"decoder.blocks.{i}.cross_attn.value"
        b = om.sd[p + ".bias"]
        b.copy_((b - om.sd[p + ".weight"].float() @ c).half().float())
    return c


def time_code(n: int, n_freq: int = 32, longest: float = 3000.0, shortest: float = 6.0) -> torch.Tensor:
    """(n, 2 n_freq) code of the integers 0 .. n-1: cos / sin at n_freq geometrically spaced periods.  code(t) . code(s) =
    sum_j cos(w_j (t - s)): 2 n_freq / 2 at t = s, a few units elsewhere — a ridge a few frames wide."""
    t = torch.arange(n, dtype=torch.float64)[:, None]
    w = 2 * np.pi / torch.tensor(np.geomspace(longest, shortest, n_freq))[None, :]
    return torch.cat([torch.cos(t * w), torch.sin(t * w)], 1).float()


def condition_alignment(sd: Dict[str, torch.Tensor], dims, heads: List[Tuple[int, int]], seed: int = 0,
                        frames_per_token: float = 11.0, first_frame: float = 12.0, pos_gain: float = 40.0,
                        qk_gain: float = 0.7) -> Dict:
    """Alignment-conditioned synthetic checkpoint (TEST INFRASTRUCTURE).  Random-init cross attention has no ridge: the
    DTW of timing.py:141-151 then picks one of many nearly equally cheap monotone paths, and a rounding-level change of
    the cost matrix moves word boundaries by tenths of a second (the fp16-vs-fp32 figures of the random-init tests).
    A trained model's alignment heads attend along the diagonal of (token, time).  This builds that property into a seeded
    checkpoint, IN PLACE in `sd`, for the (layer, head) pairs `heads` (to be installed as the model's alignment heads):
      * `decoder.positional_embedding[i]` += pos_gain * U_d code(tau_i), tau_i = first_frame + frames_per_token * i: the
        residual stream of token position i carries where in the audio the token "is spoken";
      * rows of head h of `cross_attn.query` in layer l := qk_gain * U_d^T (LayerNorm affine divided out, bias cancelling
        its shift), rows of head h of `cross_attn.key` := qk_gain * U_a^T: q_i . k_t peaks where the frame's code
        matches the token's;
    U_d, U_a: seeded orthonormal (D, 64) bases.  The AUDIO side of the code cannot come from a random-init encoder; the
    caller builds the audio features as noise + feat_gain * code(t) U_a^T (`alignment_features`) and passes them as
    `audio_features`, which find_alignment / find_alignment_batch accept in place of the mel (transcribe() does the same
    with the features of the window it has just decoded).  Everything is rounded to fp16-exact values.
    frames_per_token is ODD on purpose: with an even spacing the frame midway between two tokens' target frames scores
    exactly the same for both (the code's kernel is symmetric) and rounding would decide the boundary.  Deep decoders dilute
    the position code (the residual stream grows with depth): 32 layers want pos_gain ~ 120, qk_gain ~ 0.5; 4 layers 40 / 0.7
    (chosen on the CPU with a crude all-fp16 oracle as the perturbation: no boundary moved).
    Returns {"U_a", "U_d", "tau": frame of every token position}."""
    D = dims.n_text_state
    g = torch.Generator().manual_seed(seed)
    U_d, _ = torch.linalg.qr(torch.randn(D, 64, generator=g))
    U_a, _ = torch.linalg.qr(torch.randn(D, 64, generator=g))
    tau = first_frame + frames_per_token * torch.arange(dims.n_text_ctx, dtype=torch.float64)
    code = time_code(int(tau.max().item()) + 2)
    # token side: code of the (rounded) target frame of every position
    pos_code = code[tau.round().long()]                                       # (n_ctx, 64)
    q16 = lambda t: t.half().float()
    sd["decoder.positional_embedding"] = q16(sd["decoder.positional_embedding"].float() + pos_gain * pos_code @ U_d.T)
    for l, h in heads:
        p = f"decoder.blocks.{l}"
        rows = slice(64 * h, 64 * h + 64)
        gamma = sd[p + ".cross_attn_ln.weight"].float()
        beta = sd[p + ".cross_attn_ln.bias"].float()
        wq = sd[p + ".cross_attn.query.weight"].float().clone()
        bq = sd[p + ".cross_attn.query.bias"].float().clone()
        wk = sd[p + ".cross_attn.key.weight"].float().clone()
        wq[rows] = q16(qk_gain * U_d.T / gamma[None, :])
        bq[rows] = q16(-(wq[rows] @ beta))
        wk[rows] = q16(qk_gain * U_a.T)
        sd[p + ".cross_attn.query.weight"], sd[p + ".cross_attn.query.bias"] = wq, bq
        sd[p + ".cross_attn.key.weight"] = wk
    return {"U_a": U_a, "U_d": U_d, "tau": tau}


def alignment_features(dims, n: int, U_a: torch.Tensor, seed: int = 0, feat_gain: float = 4.0, noise: float = 1.0) -> torch.Tensor:
    """(n, n_audio_ctx, D) fp16-exact audio features for an alignment-conditioned checkpoint: unit noise (its own per clip)
    plus feat_gain * code(frame) along U_a."""
    g = torch.Generator().manual_seed(seed)
    code = time_code(dims.n_audio_ctx)
    x = noise * torch.randn(n, dims.n_audio_ctx, dims.n_audio_state, generator=g) + feat_gain * (code @ U_a.T)[None]
    return x.half().float()


def _class_script(rng: np.random.Generator, n_steps: int, text_run: Tuple[int, int]) -> List[bool]:
    """want_timestamp[i] for the steps where the hard filters leave the choice: True where a timestamp pair starts"""
    want = [False] * n_steps
    i = 1 + int(rng.integers(text_run[0], text_run[1] + 1))
    while i < n_steps:
        want[i] = True
        i += 2 + int(rng.integers(text_run[0], text_run[1] + 1))
    return want


def condition_greedy(om: OracleModel, feats: torch.Tensor, initial_tokens: List[int], n_steps: int, r: SamplingRules,
                     seed: int = 0, margin: Tuple[float, float] = (0.3, 3.0), text_run: Tuple[int, int] = (4, 14),
                     ts_window: int = 12, log=None, passes: int = 3, share: bool = False,
                     head: Optional[Tuple[int, float, float]] = None) -> Dict:
    """Edits om.sd["decoder.token_embedding.weight"] IN PLACE (see the module docstring) for a greedy decode of `n_steps`
    tokens over `feats` (R, 1500, D).  Pass 1 chooses the tokens and builds the margins; an edit made for a later step can
    still lift its token in ANOTHER row's earlier steps (rows are only decorrelated through their mean hidden states), so
    passes 2.. walk the same token sequence again on the weights as they now stand and top the margins up where they fell
    short (second-order small: the top-ups are a fraction of a logit).  Returns the last pass's {"tokens" (R, T0 +
    n_steps), "rows": all edited token ids, "margins": per (step, row) the margin at the time the pass left the step,
    "deltas": logit boosts of that pass, "drawn": the margins asked for}.  The caller re-packs its engines from the state dict
    and ALWAYS re-decodes with the plain oracle to assert the final margins (`margins_of`).
    share: rows may emit the SAME token at the same step and follow one class script.  For inputs the model cannot tell
    apart — a random-init ENCODER maps every audio to nearly the same features (cos 0.999 between clips of bench.py's
    synthetic audio; the tests use feature tensors with a per-clip offset instead), so the rows' hidden states are
    near-copies (cos 0.995) — forcing each row onto a token of its own would mean lifting it over its neighbour's boosted
    token along the sliver of hidden state the two rows do not share: edits of norm >> the embedding's.  With `share` such
    rows simply decode alike, as they do on the unconditioned weights.
    head = (n, lo, hi): the first n steps draw their margins from [lo, hi] instead of `margin`.  A BEAM search over those steps
    is then decided by the model as well: with top-1 margins of a few tenths a runner-up hypothesis can end within 0.01 of the
    winner's sum_logprob (seen: 0.013 over 64 steps), and which of the two wins then differs between two hosts' fp32 oracles;
    with margins >= 2 the greedy path is the beam winner by >= 2 (bench.py's beam leg and the 64-step beam test)."""
    res = _condition_pass(om, feats, initial_tokens, n_steps, r, seed, margin, text_run, ts_window, log, None, share, head)
    rows = set(res["rows"])
    for p in range(1, passes):
        if log is not None:
            log(f"condition: pass {p + 1} (top-up along the built sequence)")
        res = _condition_pass(om, feats, initial_tokens, n_steps, r, seed, margin, text_run, ts_window, log,
                              (res["tokens"], res["drawn"]), share, head)
        rows |= set(res["rows"])
        if log is not None:
            nz = [d for d in res["deltas"] if d > 0]
            log(f"condition: pass {p + 1} topped up {len(nz)} of {len(res['deltas'])} decisions, largest {max(res['deltas']):.3f}")
        if max(res["deltas"]) == 0.0:
            break
    res["rows"] = sorted(rows)
    return res


def _condition_pass(om, feats, initial_tokens, n_steps, r, seed, margin, text_run, ts_window, log, target, share=False, head=None) -> Dict:
    """one walk; target = None (choose tokens, draw margins) or (tokens, drawn margins) of an earlier pass"""
    rng = np.random.default_rng(seed)
    E = om.sd[EMB]
    assert E.dtype == torch.float32 and E.is_contiguous()
    R = feats.shape[0]
    TB = r.timestamp_begin
    tokens = torch.tensor([list(initial_tokens)] * R, dtype=torch.int64)
    used = torch.zeros(E.shape[0], dtype=torch.bool)
    used[list(initial_tokens)] = True
    scripts = [_class_script(rng, n_steps, text_run) for _ in range(R)]
    if share:
        scripts = [scripts[0]] * R
    cache = om.new_cache()
    margins, deltas, rows, drawn = [], [], [], []
    ninf = -np.inf
    mean_sum, mean_n = None, 0
    hist = [torch.zeros(E.shape[1], 0) for _ in range(R)]         # per row: orthonormal basis of its earlier hidden states

    def grow(Q, v):
        """append v's component outside span(Q) to the orthonormal basis Q (two Gram-Schmidt passes)"""
        for _ in range(2):
            v = v - Q @ (Q.T @ v)
        n = float(v.norm())
        return torch.cat([Q, (v / n)[:, None]], 1) if n > 1e-3 else Q

    with torch.no_grad():
        for i in range(n_steps):
            h_seq = om.decoder_hidden(tokens if i == 0 else tokens[:, -1:], feats, cache).float()             # (R, T, D)
            h_all = h_seq[:, -1]                                                                               # (R, D)
            mean_sum = h_seq.sum(1) if mean_sum is None else mean_sum + h_all
            mean_n += h_seq.shape[1]
            means = mean_sum / mean_n                               # (R, D) running mean hidden state of every row
            for k in range(R):                                      # prompt positions count as earlier hidden states
                for t in range(h_seq.shape[1] - 1):
                    hist[k] = grow(hist[k], h_seq[k, t])
            logits_all = h_all @ E.T                                                                           # (R, V)
            nxt = torch.empty(R, dtype=torch.int64)
            step_edits: List[int] = []
            step_chosen: List[int] = []
            for k in range(R):
                h = h_all[k]
                lg = logits_all[k].clone()
                if step_edits:                                  # rows edited for earlier rows of this step
                    lg[step_edits] = E[step_edits] @ h
                sampled = tokens[k, r.sample_begin:].tolist()
                apply_filters(lg, sampled, r, mass_rule=False)
                free = lg.clone()
                free[used] = ninf
                if share and step_chosen:                       # tokens other rows chose at THIS step stay admissible
                    free[step_chosen] = lg[step_chosen]
                if TB is None:
                    text_ok, ts_ok = True, False
                    text_free = free
                else:
                    text_ok, ts_ok = bool(torch.isfinite(lg[:TB]).any()), bool(torch.isfinite(lg[TB:]).any())
                    text_free = free[:TB]
                y = int(target[0][k, len(initial_tokens) + i]) if target is not None else None
                pair_open = TB is not None and len(sampled) >= 2 and sampled[-1] >= TB and sampled[-2] < TB
                if y is not None:
                    pass
                elif ts_ok and (not text_ok or scripts[k][i] or pair_open):    # pair_open: timestamps come in pairs
                    cand = torch.nonzero(torch.isfinite(free[TB:]))[:ts_window, 0]
                    if len(cand):
                        y = TB + int(cand[free[TB:][cand].argmax()])
                    elif not text_ok:
                        raise RuntimeError(f"row {k} step {i}: no unused admissible timestamp left; shorten the decode, "
                                           f"lengthen text_run or shrink ts_window")
                if y is None:
                    assert text_ok and bool(torch.isfinite(text_free).any()), (k, i)
                    y = int(text_free.argmax())
                m = float(rng.uniform(*(margin if head is None or i >= head[0] else head[1:]))) if target is None else float(target[1][i * R + k])
                drawn.append(m)
                others = lg.clone()
                others[y] = ninf
                if TB is None:
                    need = float(others.max()) + m
                elif y < TB:
                    need = float(others[:TB].max()) + m
                    if ts_ok:
                        need = max(need, float(torch.logsumexp(others[TB:], 0)) + m)
                else:
                    need = float(others[TB:].max()) + m if bool(torch.isfinite(others[TB:]).any()) else ninf
                    if text_ok:
                        need = max(need, float(others[:TB].max()) + m)
                have = float(lg[y])
                boost = 0.0
                if need > have:
                    Qk = hist[k]
                    for j in range(R):
                        if j != k:
                            Qk = grow(Qk, means[j])
                    d = h
                    for _ in range(2):
                        d = d - Qk @ (Qk.T @ d)
                    u = d / float(d @ h)
                    base = E[y].clone()
                    boost = need - have
                    if float(d.norm()) < 0.2:
                        raise RuntimeError(f"row {k} step {i}: the hidden state has no direction of its own left (|d| = "
                                           f"{float(d.norm()):.3f}): the rows are near-copies — condition with share=True")
                    for _ in range(8):
                        E[y] = (base + boost * u).half().float()
                        got = float(E[y] @ h)
                        if got >= need:
                            break
                        boost += (need - got) + 1e-3
                    else:
                        raise RuntimeError(f"row {k} step {i}: fp16 rounding of an edited embedding row did not converge "
                                           f"(boost {boost:.2f}, |d| {float(d.norm()):.3f}, d.h {float(d @ h):.3f}, need {need:.2f}, "
                                           f"have {have:.2f}, got {got:.2f})")
                    if y not in step_edits:
                        step_edits.append(y)
                    rows.append(y)
                    have = float(E[y] @ h)
                hist[k] = grow(hist[k], h)
                step_chosen.append(y)
                used[y] = True
                nxt[k] = y
                margins.append(have - (need - m))
                deltas.append(boost)
                if log is not None and boost > 30:
                    log(f"condition: large boost {boost:.1f} at step {i} row {k} token {y} (|d| {float(d.norm()):.2f})")
            tokens = torch.cat([tokens, nxt[:, None]], dim=-1)
            if log is not None and (i % 32 == 31 or i == n_steps - 1):
                mm = np.asarray(margins[-32 * R:])
                log(f"condition: step {i + 1}/{n_steps}, margins of the last steps min {mm.min():.2f} median "
                    f"{np.median(mm):.2f}, boost median {np.median(deltas[-32 * R:]):.2f} max {max(deltas[-32 * R:]):.2f}")
    return {"tokens": tokens, "rows": sorted(set(rows)), "margins": margins, "deltas": deltas, "drawn": drawn}


def margins_of(dec: Dict) -> Dict:
    """Decision margins of a finished `greedy_decode(..., keep_logits=True)`: per (step, row) the winner's filtered
    logit minus the runner-up's in the vector the arg-max was taken from (decoding.py:277-283), and the distance by
    which the timestamp-mass rule (decoding.py:498-505) was decided wherever it had a choice to make."""
    top = []
    for lg in dec["step_logits"]:
        v, _ = lg.float().topk(2, dim=-1)
        top += (v[:, 0] - v[:, 1]).tolist()
    top = np.asarray(top)
    out = {"min": float(top.min()), "median": float(np.median(top)), "p05": float(np.quantile(top, 0.05)),
           "n": int(top.size)}
    rule = [abs(x) for x in dec.get("rule_margins", []) if x is not None]
    if rule:
        out["rule_min"] = float(min(rule))
        out["rule_n"] = len(rule)
    return out


def build_reference(dims, sd: Dict[str, torch.Tensor], audio: np.ndarray, initial_tokens: List[int], n_steps: int,
                    r: SamplingRules, seed: int = 0, margin: Tuple[float, float] = (0.35, 3.0),
                    text_run: Tuple[int, int] = (4, 14), passes: int = 2, log=None, sdpa: bool = True) -> Dict:
    """The checker's side of an END-TO-END parity statement on raw audio (what bench.py's `oracle_side` does for the headline,
    as one call for the other configurations and for tests): oracle log-mel (audio.py:110-157) -> oracle AudioEncoder
    (model.py:188-204) on `audio` (B, n_samples) fp32, `center_cross_values` + `condition_greedy` on those features — `sd` is
    edited IN PLACE (cross-attention value biases, tied-embedding rows): pack the engines from it AFTERWARDS — then the plain
    oracle's greedy decode of `n_steps` tokens with every step's filtered logits kept.
    Returns {"om", "mel", "feats", "dec" (greedy_decode(..., keep_logits=True)), "margins", "edited_rows", "consistent"}."""
    from .decoding import greedy_decode
    from .mel import log_mel_spectrogram, mel_filterbank
    om = OracleModel(dims, sd, sdpa=sdpa)
    filt = mel_filterbank(dims.n_mels)
    with torch.no_grad():
        mel = torch.stack([torch.as_tensor(log_mel_spectrogram(audio[b], filt)) for b in range(audio.shape[0])])
        feats = om.encoder(mel)
        center_cross_values(om, feats)
        built = condition_greedy(om, feats, initial_tokens, n_steps, r, seed=seed, margin=margin, text_run=text_run,
                                 log=log, passes=passes)
        dec = greedy_decode(om, feats, initial_tokens, n_steps, r, keep_logits=True)
    return {"om": om, "mel": mel, "feats": feats, "dec": dec, "margins": margins_of(dec), "edited_rows": len(built["rows"]),
            "consistent": bool(torch.equal(built["tokens"], dec["tokens"]))}
