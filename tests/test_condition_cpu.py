"""oracle/condition.py on the CPU (small dims, seconds): the margin-conditioned and alignment-conditioned synthetic
checkpoints the GPU parity tests and bench.py decode on.  Nothing here touches the HIP path."""
import numpy as np
import torch

import oracle
from oracle import condition
from oracle.decoding import apply_filters
from whisper_amd.tokenizer import get_tokenizer


def _setup(name, n_steps):
    dims = oracle.dims_for(name)
    tok = get_tokenizer(True, num_languages=dims.n_vocab - 51765 - 1, language="en", task="transcribe")
    init = list(tok.sot_sequence)
    suppress = sorted(set(list(tok.non_speech_tokens) + [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm,
                                                         tok.no_speech, tok.eot]))
    rules = oracle.SamplingRules(sample_begin=len(init), sot_index=0, eot=tok.eot, n_ctx=dims.n_text_ctx,
                                 timestamp_begin=tok.timestamp_begin, no_timestamps=tok.no_timestamps,
                                 suppress_tokens=suppress, blank_token=tok.encode(" ")[0], no_speech=tok.no_speech)
    return dims, tok, init, rules


def _offset_feats(dims, n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, dims.n_audio_ctx, dims.n_audio_state, generator=g)
            + 3.0 * torch.randn(n, 1, dims.n_audio_state, generator=g)).half().float()


def test_hard_filters_plus_mass_rule_equal_the_full_filter():
    """apply_filters(mass_rule=False) followed by the rule of decoding.py:498-505 == apply_filters: the split the conditioner
    relies on changes nothing"""
    dims, tok, init, rules = _setup("micro-v3", 4)
    g = torch.Generator().manual_seed(0)
    for sampled in ([], [tok.timestamp_begin + 3], [tok.timestamp_begin + 3, 500], [tok.timestamp_begin + 3, 500, tok.timestamp_begin + 9]):
        lg = torch.randn(dims.n_vocab, generator=g) * 2
        lg[tok.timestamp_begin:] += 1.0
        a, b = lg.clone(), lg.clone()
        apply_filters(a, sampled, rules)
        apply_filters(b, sampled, rules, mass_rule=False)
        lp = torch.log_softmax(b.float(), -1)
        if lp[tok.timestamp_begin:].logsumexp(-1) > lp[: tok.timestamp_begin].max():
            b[: tok.timestamp_begin] = -np.inf
        assert torch.equal(a, b), sampled


def test_margin_conditioning_builds_what_the_plain_oracle_then_decodes():
    """wide-v3 (D = 1280, 2 + 2 layers), 3 rows x 40 steps: the plain greedy decode of the conditioned weights emits exactly
    the built sequence, every decision distinct, margins (arg-max and timestamp-mass rule) at or above the drawn minimum;
    only embedding rows of emitted tokens changed, to fp16-exact values; the unconditioned weights have near-ties."""
    dims, tok, init, rules = _setup("wide-v3", 40)
    sd = oracle.synthetic_state_dict(dims, seed=0)
    before = sd[condition.EMB].clone()
    om = oracle.OracleModel(dims, sd, sdpa=True)
    feats = _offset_feats(dims, 3, seed=12)
    with torch.no_grad():
        plain = oracle.greedy_decode(om, feats, init, 40, rules, keep_logits=True)
    assert condition.margins_of(plain)["min"] < 0.05                        # random-init: near-ties within 120 decisions
    built = condition.condition_greedy(om, feats, init, 40, rules, seed=5, margin=(0.35, 3.0))
    with torch.no_grad():
        dec = oracle.greedy_decode(om, feats, init, 40, rules, keep_logits=True)
    mg = condition.margins_of(dec)
    assert torch.equal(dec["tokens"], built["tokens"])
    assert mg["min"] >= 0.3 and mg["median"] >= 1.0 and mg["rule_min"] >= 0.3, mg
    T0 = len(init)
    emitted = {int(x) for x in dec["tokens"][:, T0:].flatten()}
    assert len(emitted) == 3 * 40
    changed = torch.nonzero((sd[condition.EMB] != before).any(dim=1))[:, 0].tolist()
    assert set(changed) <= emitted and set(changed) == set(built["rows"])
    E = sd[condition.EMB]
    assert torch.equal(E, E.half().float())                                 # both engines and the oracle see the same weights
    ts = dec["tokens"][:, T0:] >= tok.timestamp_begin
    assert bool(ts[:, 0].all()) and 3 <= int(ts.sum()) < 3 * 20             # a timestamp first, pairs in between, mostly text


def test_value_centering_removes_the_encoders_dc_term():
    """center_cross_values: with features that are one large constant vector plus noise (what a random-init encoder
    produces), hidden states of different steps are near-copies; after cancelling the constant's contribution to the
    cross-attention values they are not, and the cancelled term is exactly W_v c."""
    dims, tok, init, rules = _setup("wide-v3", 8)
    sd = oracle.synthetic_state_dict(dims, seed=0)
    g = torch.Generator().manual_seed(3)
    c = 0.95 * torch.randn(dims.n_audio_state, generator=g)
    feats = (0.32 * torch.randn(2, dims.n_audio_ctx, dims.n_audio_state, generator=g) + c).half().float()
    toks = torch.randint(0, 50000, (2, len(init) + 16), generator=g)
    toks[:, : len(init)] = torch.tensor(init)

    def step_cos(om):
        with torch.no_grad():
            h = om.decoder_hidden(toks, feats)[0, len(init):]
        h = h / h.norm(dim=-1, keepdim=True)
        cc = h @ h.T
        return float(cc[~torch.eye(len(h), dtype=bool)].mean())
    om = oracle.OracleModel(dims, sd, sdpa=True)
    b0 = sd["decoder.blocks.1.cross_attn.value.bias"].clone()
    cos_before = step_cos(om)
    got_c = condition.center_cross_values(om, feats)
    cos_after = step_cos(om)
    assert torch.allclose(got_c, feats.mean(dim=(0, 1)))
    want = (b0 - sd["decoder.blocks.1.cross_attn.value.weight"] @ got_c).half().float()
    assert torch.equal(sd["decoder.blocks.1.cross_attn.value.bias"], want)
    assert cos_after < cos_before - 0.02, (cos_before, cos_after)


def test_alignment_conditioning_gives_a_diagonal():
    """condition_alignment on turbo dims: the installed heads attend along the (token, time) diagonal, the DTW recovers one
    word per 11 frames, and a crude all-fp16 oracle moves no boundary"""
    dims, tok, init, rules = _setup("turbo", 4)
    sd = oracle.synthetic_state_dict(dims, seed=4)
    L = dims.n_text_layer
    heads = sorted([(L - 1, 3), (L - 1, 11), (L - 2, 0), (L - 2, 7)])
    info = condition.condition_alignment(sd, dims, heads, seed=1, pos_gain=40.0, qk_gain=0.7)
    om = oracle.OracleModel(dims, sd)
    om16 = oracle.OracleModel(dims, sd, dtype=torch.float16)
    text = tok.encode(" the quick brown fox jumps over the lazy dog and keeps running")
    frames = 2 * int(12 + 11 * (len(text) + 6) + 6)
    feats = condition.alignment_features(dims, 1, info["U_a"], seed=3)
    with torch.no_grad():
        ws, we, _ = oracle.word_times(om, tok, text, feats, frames, heads)
        ws16, we16, _ = oracle.word_times(om16, tok, text, feats.half(), frames, heads)
    step = np.diff(ws[1:])                                  # (the first word owns the leading frames, as in the reference)
    assert len(ws) == len(text) and np.all(step > 0.15) and np.all(step < 0.30), ws      # 11 frames = 0.22 s per token
    assert np.array_equal(ws, ws16) and np.array_equal(we, we16)
