"""Audio ingest without ffmpeg: the native FLAC decoder (csrc/flac_decode.c ->
libwhisper_audio.so) and the RIFF/WAVE reader behind whisper_amd.load_audio.  CPU only.

The decoder verifies every frame's CRC-8 / CRC-16 and the STREAMINFO MD5 of the decoded PCM, so a successful decode
of a third-party file (tests/jfk.flac of the reference, 24-bit stereo 44.1 kHz, LPC + Rice coded) is self-validating;
the bit-level paths are additionally exercised with FLAC streams written by the small encoder below (VERBATIM,
CONSTANT and FIXED-predictor subframes with Rice residuals, independent / left-side / mid-side stereo)."""
import hashlib
import os
import struct
import wave

import numpy as np
import pytest

from whisper_amd import audio as A

JFK = "/root/reference/tests/jfk.flac"


# ---------------------------------------------------------------------------------------------------------------
# a tiny FLAC writer (test-side only)
# ---------------------------------------------------------------------------------------------------------------
class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, n):
        if n == 0:
            return
        self.acc = (self.acc << n) | (v & ((1 << n) - 1))
        self.n += n
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def _crc(data, poly, width):
    c, top = 0, 1 << (width - 1)
    for b in data:
        c ^= b << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) if c & top else (c << 1)
        c &= (1 << width) - 1
    return c


def _rice(bits, residual, k):
    for r in residual:
        u = (r << 1) if r >= 0 else ((-r << 1) - 1)
        bits.put(0, u >> k)
        bits.put(1, 1)
        bits.put(u & ((1 << k) - 1), k)


def _subframe(bits, x, bps, kind):
    x = [int(v) for v in x]
    if kind == "constant":
        bits.put(0, 1); bits.put(0, 6); bits.put(0, 1); bits.put(x[0], bps)
    elif kind == "verbatim":
        bits.put(0, 1); bits.put(1, 6); bits.put(0, 1)
        for v in x:
            bits.put(v, bps)
    else:                                     # fixed predictor of order `kind`, one Rice partition (k = 9)
        order = kind
        coef = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [x[i] - sum(c * x[i - 1 - j] for j, c in enumerate(coef)) for i in range(order, len(x))]
        bits.put(0, 1); bits.put(8 + order, 6); bits.put(0, 1)
        for v in x[:order]:
            bits.put(v, bps)
        bits.put(0, 2); bits.put(0, 4); bits.put(9, 4)
        _rice(bits, res, 9)


def write_flac(pcm, rate, bps, blocksize, plan):
    """pcm int [frames][channels]; plan(frame_index) -> (channel_assignment 0..10, [subframe kind per channel])"""
    frames, ch = pcm.shape
    nb = (bps + 7) // 8
    md5 = hashlib.md5(b"".join(int(v).to_bytes(nb, "little", signed=True) for v in pcm.reshape(-1))).digest()
    info = _Bits()
    info.put(blocksize, 16); info.put(blocksize, 16); info.put(0, 24); info.put(0, 24)
    info.put(rate, 20); info.put(ch - 1, 3); info.put(bps - 1, 5); info.put(frames, 36)
    out = bytearray(b"fLaC" + bytes([0x80]) + (34).to_bytes(3, "big") + bytes(info.out) + md5)
    for fi, start in enumerate(range(0, frames, blocksize)):
        blk = pcm[start: start + blocksize].astype(np.int64)
        n = len(blk)
        assign, kinds = plan(fi)
        b = _Bits()
        b.put(0x3FFE, 14); b.put(0, 1); b.put(0, 1)
        b.put(7, 4); b.put(0, 4); b.put(assign, 4)
        b.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3); b.put(0, 1)
        assert fi < 128
        b.put(fi, 8)
        b.put(n - 1, 16)
        b.put(_crc(bytes(b.out), 0x07, 8), 8)
        chans = [blk[:, c] for c in range(ch)]
        widths = [bps] * ch
        if assign == 8:
            chans, widths = [chans[0], chans[0] - chans[1]], [bps, bps + 1]
        elif assign == 9:
            chans, widths = [chans[0] - chans[1], chans[1]], [bps + 1, bps]
        elif assign == 10:
            chans, widths = [(chans[0] + chans[1]) >> 1, chans[0] - chans[1]], [bps, bps + 1]
        for c in range(ch):
            _subframe(b, chans[c], widths[c], kinds[c])
        b.align()
        crc = _crc(bytes(b.out), 0x8005, 16)
        out += bytes(b.out) + crc.to_bytes(2, "big")
    return bytes(out)


def _pcm(frames, ch, bps, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(frames)[:, None] / 8000.0
    x = 0.4 * np.sin(2 * np.pi * (200 + 90 * np.arange(ch)[None, :]) * t) + 0.01 * rng.standard_normal((frames, ch))
    return np.round(x * (1 << (bps - 1)) * 0.9).astype(np.int32)


@pytest.mark.parametrize("bps", [16, 24])
@pytest.mark.parametrize("ch", [1, 2])
def test_flac_roundtrip_all_subframe_kinds(ch, bps):
    pcm = _pcm(4096 * 3 + 777, ch, bps, seed=ch * 100 + bps)
    pcm[2 * 4096: 3 * 4096, 0] = 1234                       # block 2: what a CONSTANT subframe can carry

    def plan(fi):
        if ch == 1:
            return 0, [["verbatim", 2, "constant", 4][fi % 4]]
        return [1, 8, 1, 10][fi % 4], [["verbatim", 1], [3, 2], ["constant", 0], [2, 4]][fi % 4]
    data = write_flac(pcm, 22050, bps, 4096, plan)
    got, rate, got_bps = A.decode_flac(data)
    assert (rate, got_bps) == (22050, bps)
    assert np.array_equal(got, pcm)


def test_flac_right_side_and_errors():
    pcm = _pcm(2000, 2, 16, seed=5)
    data = write_flac(pcm, 16000, 16, 1000, lambda fi: (9, [1, 2]))
    got, rate, _ = A.decode_flac(data)
    assert np.array_equal(got, pcm) and rate == 16000
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x40
    with pytest.raises(RuntimeError, match="CRC|MD5|sync|header"):
        A.decode_flac(bytes(bad))
    with pytest.raises(RuntimeError, match="truncated"):
        A.decode_flac(data[: len(data) - 700])
    with pytest.raises(RuntimeError, match="not a FLAC"):
        A.decode_flac(b"RIFF" + data[4:])
    tampered = bytearray(data)
    tampered[4 + 4 + 18] ^= 0xFF                                # flip a byte of the stored MD5 signature
    with pytest.raises(RuntimeError, match="MD5"):
        A.decode_flac(bytes(tampered))


def test_load_audio_native_wav_and_flac(tmp_path, monkeypatch):
    """load_audio contract of the reference (mono float32 at 16 kHz) through the no-ffmpeg route"""
    import subprocess

    def no_ffmpeg(*a, **k):
        raise FileNotFoundError("ffmpeg")
    monkeypatch.setattr(subprocess, "run", no_ffmpeg)
    pcm = _pcm(32000, 2, 16, seed=9)
    wav_path = str(tmp_path / "a.wav")
    with wave.open(wav_path, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(32000)
        w.writeframes(pcm.astype("<i2").tobytes())
    flac_path = str(tmp_path / "a.flac")
    with open(flac_path, "wb") as f:
        f.write(write_flac(pcm, 32000, 16, 4000, lambda fi: (10, [2, 2])))
    a, b = A.load_audio(wav_path), A.load_audio(flac_path)
    assert a.dtype == np.float32 and a.ndim == 1 and a.shape == (16000,)
    assert np.array_equal(a, b)                                 # same PCM -> same mono 16 kHz s16 samples
    assert 0.05 < a.std() < 1
    with open(str(tmp_path / "junk.bin"), "wb") as f:
        f.write(b"\x00" * 64)
    with pytest.raises(RuntimeError):
        A.load_audio(str(tmp_path / "junk.bin"))


def _riff(fmt_body: bytes, data: bytes) -> bytes:
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_wav_extensible_and_malformed_headers(tmp_path, monkeypatch):
    """WAVE_FORMAT_EXTENSIBLE carries the real sample format in the SubFormat GUID: an extensible IEEE-float file must
    decode as float (not as int32 PCM), an extensible PCM file as PCM; truncated fmt chunks and zero channels make
    load_audio raise its RuntimeError instead of struct.error / ZeroDivisionError"""
    import subprocess

    def no_ffmpeg(*a, **k):
        raise FileNotFoundError("ffmpeg")
    monkeypatch.setattr(subprocess, "run", no_ffmpeg)
    t = np.arange(16000) / 16000.0
    sine = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    guid_tail = bytes.fromhex("000000001000800000aa00389b71")

    def ext(sub, bits, block):
        return struct.pack("<HHIIHH", 0xFFFE, 1, 16000, 16000 * block, block, bits) + struct.pack("<HHI", 22, bits, 4) + \
            struct.pack("<H", sub) + guid_tail
    p = str(tmp_path / "ext_float.wav")
    open(p, "wb").write(_riff(ext(3, 32, 4), sine.astype("<f4").tobytes()))
    got = A.load_audio(p)
    assert got.shape == (16000,) and np.abs(got - sine).max() < 1e-4        # 16-bit quantisation of the s16 round trip
    p = str(tmp_path / "ext_pcm16.wav")
    open(p, "wb").write(_riff(ext(1, 16, 2), np.round(sine * 32767).astype("<i2").tobytes()))
    assert np.abs(A.load_audio(p) - sine).max() < 1e-4
    plain = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)
    for name, blob in (("short_fmt.wav", _riff(plain[:10], b"\0" * 64)),
                       ("zero_ch.wav", _riff(struct.pack("<HHIIHH", 1, 0, 16000, 32000, 2, 16), b"\0" * 64)),
                       ("ext_short.wav", _riff(ext(3, 32, 4)[:20], b"\0" * 64))):
        q = str(tmp_path / name)
        open(q, "wb").write(blob)
        with pytest.raises(RuntimeError):
            A.load_audio(q)


@pytest.mark.reference
def test_jfk_flac_decodes_and_matches_reference_test_invariants():
    """tests/jfk.flac of the reference: decodes (CRCs + MD5 signature verified inside the decoder) to the PCM whose
    SHA-256 is pinned below, and load_audio meets the asserts of the reference's tests/test_audio.py:10-19"""
    if not os.path.isfile(JFK):
        pytest.skip("reference checkout not present")
    with open(JFK, "rb") as f:
        pcm, rate, bps = A.decode_flac(f.read())
    assert (pcm.shape, rate, bps) == ((485100, 2), 44100, 24)
    assert hashlib.sha256(pcm.astype("<i4").tobytes()).hexdigest() == JFK_PCM_SHA256
    audio = A.load_audio(JFK)
    assert audio.ndim == 1
    assert A.SAMPLE_RATE * 10 < audio.shape[0] < A.SAMPLE_RATE * 12
    assert 0 < audio.std() < 1


JFK_PCM_SHA256 = "c7741f92548994acaee42e83e128e4f1ec59e9fd47e2c4c886c37a33ec021521"
