/*
 * whisper_audio.h — C ABI of libwhisper_audio.so (whisper_amd/csrc/flac_decode.c): host-only audio ingest used by
 * whisper_amd.audio.load_audio when the ffmpeg CLI the reference shells out to (whisper/audio.py:25-62) is absent.
 * Plain C, no HIP, no torch types.  RIFF/WAVE needs no native code (read in Python).
 */
#ifndef WHISPER_AUDIO_H
#define WHISPER_AUDIO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Decode a complete FLAC stream held in memory.  On success (0) `*samples` is a malloc'ed array of
 * n_frames * channels interleaved int32 PCM samples (release with wh_flac_free); every frame's CRC-8 / CRC-16 and
 * the STREAMINFO MD5 signature of the decoded audio have been verified.  Negative return = error code. */
int wh_flac_decode(const uint8_t *data, size_t size, int32_t **samples, int64_t *n_frames, int *channels,
                   int *sample_rate, int *bits_per_sample);
void wh_flac_free(int32_t *samples);
const char *wh_flac_error(int code);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_AUDIO_H */
