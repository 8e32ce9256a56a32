"""Log-mel front end — restates whisper/audio.py:110-157 (and the librosa call quoted at audio.py:96-100)."""
import numpy as np
import torch

N_FFT, HOP = 400, 160


def _hz_to_mel(f):
    # Slaney scale (librosa.filters.mel default htk=False): linear below 1 kHz, log above
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(n_mels: int, sr: int = 16000, n_fft: int = N_FFT) -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels=n) (slaney norm) — the matrix stored in
    whisper/assets/mel_filters.npz (audio.py:92-107).  float32 [n_mels][201]."""
    fftfreqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def log_mel_spectrogram(audio, filters, padding: int = 0, dtype=torch.float32) -> torch.Tensor:
    """audio.py:110-157 on CPU.  audio: (n,) or (B, n); filters: (n_mels, 201).  The max of audio.py:155
    is global over the whole (batched) tensor, as in the reference."""
    x = torch.as_tensor(np.asarray(audio) if not torch.is_tensor(audio) else audio).to(dtype)
    if padding > 0:
        x = torch.nn.functional.pad(x, (0, padding))                      # audio.py:145-146
    window = torch.hann_window(N_FFT, dtype=dtype)                        # audio.py:147 (periodic)
    single = x.dim() == 1
    xb = x[None] if single else x
    xp = torch.nn.functional.pad(xb[:, None], (N_FFT // 2, N_FFT // 2), mode="reflect")[:, 0]   # stft center=True
    frames = xp.unfold(-1, N_FFT, HOP)                                    # (B, 1 + n//160, 400)
    spec = torch.fft.rfft(frames * window, dim=-1)                        # audio.py:148
    mag = (spec.real ** 2 + spec.imag ** 2)[:, :-1].transpose(1, 2)       # audio.py:149: drop last frame, |.|^2
    mel = torch.as_tensor(filters).to(dtype) @ mag                        # audio.py:151-152
    log_spec = torch.clamp(mel, min=1e-10).log10()                        # audio.py:154
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)              # audio.py:155
    log_spec = (log_spec + 4.0) / 4.0                                     # audio.py:156
    return (log_spec[0] if single else log_spec).float()
