#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "whisper_audio.h"
/* mutate a valid FLAC file in many ways and decode under ASan/UBSan */
static uint32_t rs = 12345;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  unsigned char* base = malloc(n); fread(base, 1, n, f); fclose(f);
  int iters = atoi(argv[2]); long ok = 0, bad = 0; long okm[6] = {0};
  for (int it = 0; it < iters; ++it) {
    long len = n;
    unsigned char* d = malloc(n + 64); memcpy(d, base, n);
    int mode = rnd() % 6;
    if (mode == 0) { len = rnd() % n; }                                   /* truncate */
    else if (mode == 1) { for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) d[rnd() % n] ^= 1u << (rnd() % 8); }   /* bit flips */
    else if (mode == 2) { long p = rnd() % n; long m = 1 + rnd() % 64; if (p + m > n) m = n - p; for (long k = 0; k < m; ++k) d[p + k] = rnd(); }
    else if (mode == 3) { for (int k = 0; k < 4; ++k) d[4 + rnd() % 60] = rnd(); }     /* header area */
    else if (mode == 4) { long p = 42 + rnd() % 4000; if (p < n) for (int k = 0; k < 16 && p + k < n; ++k) d[p + k] = 0xff; }
    else { len = 42 + rnd() % 200; if (len > n) len = n; }
    int32_t* pcm = NULL; int64_t ns = 0; int ch = 0, rate = 0, bps = 0;
    int rc = wh_flac_decode((const char*)d, (size_t)len, &pcm, &ns, &ch, &rate, &bps);
    if (rc == 0) { ok++; okm[mode]++; volatile int64_t s = 0; for (int64_t i = 0; i < ns * ch; i += 997) s += pcm[i]; wh_flac_free(pcm); }
    else { bad++; (void)wh_flac_error(rc); }
    free(d);
  }
  printf("decoded ok %ld, rejected %ld; ok per mode %ld %ld %ld %ld %ld %ld\n", ok, bad, okm[0], okm[1], okm[2], okm[3], okm[4], okm[5]);
  return 0;
}
