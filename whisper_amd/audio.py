"""Audio front end of whisper_amd — the reference's `whisper/audio.py` surface (constants :13-22, load_audio :25,
pad_or_trim :65, mel_filters :91, log_mel_spectrogram :110) with the spectrogram computed by the fused HIP
kernel (csrc/mel.hip) instead of torch.stft + a dense matmul."""
from __future__ import annotations

import os
import struct
import subprocess
from functools import lru_cache
from typing import Optional, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import hip
from .utils import exact_div

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE          # 480000 samples per 30 s window
N_FRAMES = exact_div(N_SAMPLES, HOP_LENGTH)     # 3000 mel frames per window
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2            # conv2 has stride 2
FRAMES_PER_SECOND = exact_div(SAMPLE_RATE, HOP_LENGTH)
TOKENS_PER_SECOND = exact_div(SAMPLE_RATE, N_SAMPLES_PER_TOKEN)


def _read_wav(path: str, sr: int) -> Optional[np.ndarray]:
    """Minimal RIFF/WAVE reader (PCM 8/16/24/32-bit and float32) used when ffmpeg is not installed."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        return None
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos: pos + 4], struct.unpack("<I", data[pos + 4: pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if tag == b"fmt ":
            if len(body) < 16:
                return None                                   # truncated fmt chunk: not a file we can read
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE:                              # WAVE_FORMAT_EXTENSIBLE: the real format tag is the first
                if len(body) < 26:                            # two bytes of the SubFormat GUID (1 = PCM, 3 = IEEE float)
                    return None
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        return None
    code, channels, rate, _, _, bits = fmt
    if channels == 0 or rate == 0:
        return None
    if code == 3 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<f4").astype(np.float32)
    elif code == 3 and bits == 64:
        x = np.frombuffer(pcm[: len(pcm) // 8 * 8], "<f8").astype(np.float32)
    elif code == 1 and bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], "<i2").astype(np.float32) / 32768.0
    elif code == 1 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<i4").astype(np.float32) / 2147483648.0
    elif code == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif code == 1 and bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        return None
    x = x[: len(x) // channels * channels].reshape(-1, channels)
    return _to_mono_s16(x, rate, sr)


def _to_mono_s16(x: np.ndarray, rate: int, sr: int) -> np.ndarray:
    """[frames][channels] float in [-1, 1) -> what `ffmpeg -ac 1 -ar sr -f s16le` hands the reference: equal-weight
    down-mix, linear-phase polyphase resampling, 16-bit quantisation."""
    x = x.mean(axis=1)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sr))
        x = resample_poly(x, sr // g, rate // g).astype(np.float32)
    # ffmpeg's s16le round trip quantises to 16 bits; mirror it so both routes agree
    return (np.clip(np.round(x * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)


_AUDIO_LIB = None


def _audio_lib():
    """whisper_amd/libwhisper_audio.so (csrc/flac_decode.c, plain C built by `make -C whisper_amd/csrc`)"""
    global _AUDIO_LIB
    if _AUDIO_LIB is None:
        import ctypes as C
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwhisper_audio.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C whisper_amd/csrc`")
        lib = C.CDLL(path)
        lib.wh_flac_decode.restype = C.c_int
        lib.wh_flac_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.wh_flac_free.restype = None
        lib.wh_flac_free.argtypes = [C.POINTER(C.c_int32)]
        lib.wh_flac_error.restype = C.c_char_p
        lib.wh_flac_error.argtypes = [C.c_int]
        _AUDIO_LIB = lib
    return _AUDIO_LIB


def decode_flac(data: bytes):
    """FLAC bytes -> (int32 samples [frames][channels], sample_rate, bits_per_sample); CRC-8/16 of every frame and
    the STREAMINFO MD5 of the decoded audio are verified by the decoder (RuntimeError otherwise)."""
    import ctypes as C
    lib = _audio_lib()
    ptr, n, ch, rate, bps = C.POINTER(C.c_int32)(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.wh_flac_decode(data, len(data), C.byref(ptr), C.byref(n), C.byref(ch), C.byref(rate), C.byref(bps))
    if rc != 0:
        raise RuntimeError(f"Failed to load audio: {lib.wh_flac_error(rc).decode()}")
    try:
        pcm = np.ctypeslib.as_array(ptr, shape=(n.value * ch.value,)).reshape(n.value, ch.value).copy()
    finally:
        lib.wh_flac_free(ptr)
    return pcm, rate.value, bps.value


def _read_flac(path: str, sr: int) -> Optional[np.ndarray]:
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"fLaC" and data[:3] != b"ID3":
        return None
    pcm, rate, bps = decode_flac(data)
    return _to_mono_s16(pcm.astype(np.float32) / float(1 << (bps - 1)), rate, sr)


def load_audio(file: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """Decode `file` to mono float32 at `sr` Hz.  Same contract as the reference (audio.py:25-62): ffmpeg does
    the decoding / down-mixing / resampling; a RuntimeError is raised when it fails.  When the ffmpeg binary
    does not exist, RIFF/WAVE files are read natively and FLAC files through the native decoder of
    csrc/flac_decode.c (frame CRCs and the stream's MD5 signature are verified)."""
    cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", file, "-f", "s16le", "-ac", "1",
           "-acodec", "pcm_s16le", "-ar", str(sr), "-"]
    try:
        out = subprocess.run(cmd, capture_output=True, check=True).stdout
    except FileNotFoundError:
        wav = None
        if os.path.isfile(file):
            wav = _read_wav(file, sr)
            if wav is None:
                wav = _read_flac(file, sr)
        if wav is None:
            raise RuntimeError("Failed to load audio: ffmpeg is not installed and the file is neither RIFF/WAVE nor FLAC")
        return wav
    except subprocess.CalledProcessError as e:
        raise RuntimeError(f"Failed to load audio: {e.stderr.decode()}") from e
    return np.frombuffer(out, np.int16).flatten().astype(np.float32) / 32768.0


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Zero-pad or cut `array` along `axis` to exactly `length` entries (audio.py:65-88); numpy or torch."""
    n = array.shape[axis]
    if torch.is_tensor(array):
        if n > length:
            array = array.narrow(axis, 0, length)
        elif n < length:
            pads = [0, 0] * array.ndim
            pads[2 * (array.ndim - 1 - (axis % array.ndim)) + 1] = length - n
            array = F.pad(array, pads)
        return array
    if n > length:
        array = array.take(indices=range(length), axis=axis)
    elif n < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - n)
        array = np.pad(array, widths)
    return array


def _slaney_mel_matrix(n_mels: int) -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels=n_mels) recomputed (Slaney scale + area norm): the
    matrix the reference ships as assets/mel_filters.npz (audio.py:92-107).  Agrees with the asset to ~4e-9."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    bins = np.linspace(0, SAMPLE_RATE / 2, 1 + N_FFT // 2)
    edges = to_hz(np.linspace(to_mel(0.0), to_mel(SAMPLE_RATE / 2.0), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - bins[None, :]
    rising = -ramps[:-2] / width[:-1, None]
    falling = ramps[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri.astype(np.float32)


@lru_cache(maxsize=None)
def mel_filters(device, n_mels: int) -> torch.Tensor:
    assert n_mels in {80, 128}, f"Unsupported n_mels: {n_mels}"
    return torch.from_numpy(_slaney_mel_matrix(n_mels)).to(device)


def log_mel_spectrogram(audio: Union[str, np.ndarray, torch.Tensor], n_mels: int = 80, padding: int = 0,
                        device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
    """(n_mels, n_frames) log-mel spectrogram, computed on the GPU by the HIP kernel.

    Same arguments as the reference (audio.py:110-115).  `device=None` keeps a GPU tensor where it is and
    sends host audio to the current GPU; the result lives on that GPU (the reference leaves it on the
    audio's device).  The global `max - 8` clamp spans the whole input, batched input included, exactly
    like audio.py:155."""
    if not torch.is_tensor(audio):
        if isinstance(audio, str):
            audio = load_audio(audio)
        audio = torch.from_numpy(np.ascontiguousarray(audio))
    if device is not None:
        audio = audio.to(device)
    elif not audio.is_cuda:
        if not torch.cuda.is_available():
            raise hip.HipError("log_mel_spectrogram: no ROCm GPU visible (whisper_amd has no CPU path)")
        audio = audio.to("cuda")
    hip.require_gpu(audio.device)
    audio = audio.float()
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    lead = audio.shape[:-1]
    flat = audio.reshape(-1, audio.shape[-1]) if audio.dim() != 1 else audio
    out = hip.log_mel(flat, mel_filters(audio.device, n_mels))
    return out.reshape(*lead, n_mels, out.shape[-1]) if audio.dim() > 2 else out
