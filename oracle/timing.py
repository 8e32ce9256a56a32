"""Word-timestamp kernels on CPU — restates whisper/timing.py:19-54 (median_filter), :57-79 (backtrace),
:82-105 (dtw_cpu).  The C versions in timing_oracle.c are used when built (oracle/Makefile); the numpy
versions below are the definition and the fallback."""
import ctypes
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_timing.so")
_lib = None


def _clib():
    global _lib
    if _lib is None and os.path.isfile(_SO):
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_dtw_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.oracle_median_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int]
    return _lib


def median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """timing.py:19-54: reflect-pad by width//2 and take the sliding median along the last axis;
    input returned unchanged when shape[-1] <= width//2 (timing.py:22-24)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    assert width > 0 and width % 2 == 1
    lib = _clib()
    if lib is not None:
        out = np.empty_like(x)
        lib.oracle_median_filter(x.ctypes.data, out.ctypes.data, x.size // x.shape[-1], x.shape[-1], width)
        return out
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def dtw_trace(x: np.ndarray) -> np.ndarray:
    """timing.py:82-103: cost/trace fill with dtw_cpu's tie rule; fp32 cost; returns int8 trace
    (N+1, M+1) with the borders backtrace() forces (timing.py:61-62)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    N, M = x.shape
    trace = np.full((N + 1, M + 1), -1, dtype=np.int8)
    lib = _clib()
    if lib is not None:
        lib.oracle_dtw_trace(x.ctypes.data, N, M, trace.ctypes.data)
    else:
        cost = np.full((N + 1, M + 1), np.inf, dtype=np.float32)
        cost[0, 0] = 0
        for j in range(1, M + 1):
            for i in range(1, N + 1):
                c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
                if c0 < c1 and c0 < c2:
                    c, t = c0, 0
                elif c1 < c0 and c1 < c2:
                    c, t = c1, 1
                else:
                    c, t = c2, 2
                cost[i, j] = np.float32(np.float64(x[i - 1, j - 1]) + np.float64(c))
                trace[i, j] = t
    trace[0, :] = 2
    trace[:, 0] = 1
    return trace


def backtrace(trace: np.ndarray) -> np.ndarray:
    """timing.py:57-79."""
    i, j = trace.shape[0] - 1, trace.shape[1] - 1
    out = []
    while i > 0 or j > 0:
        out.append((i - 1, j - 1))
        t = trace[i, j] if (i > 0 and j > 0) else (2 if i == 0 else 1)
        if t == 0:
            i -= 1; j -= 1
        elif t == 1:
            i -= 1
        elif t == 2:
            j -= 1
        else:
            raise ValueError("Unexpected trace[i, j]")
    return np.array(out)[::-1, :].T


def dtw_path(x: np.ndarray) -> np.ndarray:
    """timing.py:141-151 (CPU branch): (2, path_len) int array of (text_index, time_index)."""
    return backtrace(dtw_trace(x))


def alignment_matrix(model, tokens, feats, num_frames: int, heads, n_sot: int, medfilt_width: int = 7,
                     qk_scale: float = 1.0):
    """whisper/timing.py:186-216 on the oracle model: teacher-forced pass keeping cross-attention QK of every
    layer, stack the alignment heads, crop to num_frames//2, softmax over frames, z-normalise over tokens
    (biased std), median filter, mean over heads, drop the sot rows and the last row.
    Returns (matrix [n_text+1, frames] float32, logits [T, V])."""
    import torch
    toks = torch.as_tensor(tokens, dtype=torch.int64)[None]
    logits = model.decoder(toks, feats, None, keep_qk=True)[0]
    qks = model.last_qk                                   # per layer (1, H, T, 1500)
    w = torch.stack([qks[l][0, h] for l, h in heads])     # (n_heads, T, 1500)
    w = w[:, :, : num_frames // 2]
    w = (w * qk_scale).softmax(dim=-1)
    std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
    w = (w - mean) / std
    w = torch.from_numpy(median_filter(w.numpy(), medfilt_width))
    matrix = w.mean(axis=0)[n_sot:-1]
    return matrix.numpy().astype(np.float32), logits


def word_times(model, tokenizer, text_tokens, feats, num_frames: int, heads, medfilt_width: int = 7):
    """whisper/timing.py:163-242 end to end: returns (starts, ends, probabilities) per word."""
    import torch
    n_sot = len(tokenizer.sot_sequence)
    tokens = [*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot]
    matrix, logits = alignment_matrix(model, tokens, feats, num_frames, heads, n_sot, medfilt_width)
    probs = logits[n_sot:, : tokenizer.eot].softmax(dim=-1)
    tprobs = probs[np.arange(len(text_tokens)), text_tokens].tolist()
    text_idx, time_idx = dtw_path(-matrix)
    words, word_tokens = tokenizer.split_to_word_tokens(list(text_tokens) + [tokenizer.eot])
    if len(word_tokens) <= 1:
        return np.array([]), np.array([]), np.array([])
    bounds = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jumps = np.pad(np.diff(text_idx), (1, 0), constant_values=1).astype(bool)
    jump_times = time_idx[jumps] / 50.0                   # TOKENS_PER_SECOND (audio.py:22)
    return (jump_times[bounds[:-1]], jump_times[bounds[1:]],
            np.array([np.mean(tprobs[i:j]) for i, j in zip(bounds[:-1], bounds[1:])]))
