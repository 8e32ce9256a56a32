/*
 * flac_decode.c — a small native FLAC decoder for whisper_amd.audio.load_audio.
 *
 * The reference decodes every input through the ffmpeg CLI (whisper/audio.py:25-62).  When ffmpeg is not
 * installed, whisper_amd reads RIFF/WAVE natively in Python and FLAC through this file (built by gcc into
 * whisper_amd/libwhisper_audio.so — host code, no HIP).  It implements the FLAC stream format as published
 * (xiph.org FLAC format / RFC 9639): STREAMINFO, frame headers with CRC-8 and CRC-16, CONSTANT / VERBATIM /
 * FIXED / LPC subframes, partitioned Rice residuals (4- and 5-bit parameters, escape partitions), wasted bits,
 * all stereo decorrelation modes, 4..32 bits per sample, and checks the decoded audio against the MD5
 * signature stored in STREAMINFO.  Output: interleaved int32 samples.
 *
 *   int  wh_flac_decode(const uint8_t* data, size_t size, int32_t** samples, int64_t* n_frames,
 *                       int* channels, int* sample_rate, int* bits_per_sample);   // 0 = ok, < 0 = error code
 *   void wh_flac_free(int32_t* samples);
 *   const char* wh_flac_error(int code);
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { FLAC_OK = 0, FLAC_E_MAGIC = -1, FLAC_E_TRUNC = -2, FLAC_E_HEADER = -3, FLAC_E_CRC = -4, FLAC_E_UNSUP = -5,
       FLAC_E_MD5 = -6, FLAC_E_MEM = -7, FLAC_E_SYNC = -8 };

const char* wh_flac_error(int code) {
  switch (code) {
    case FLAC_OK: return "ok";
    case FLAC_E_MAGIC: return "not a FLAC stream (missing fLaC marker)";
    case FLAC_E_TRUNC: return "truncated FLAC stream";
    case FLAC_E_HEADER: return "invalid FLAC frame or metadata header";
    case FLAC_E_CRC: return "FLAC frame CRC mismatch";
    case FLAC_E_UNSUP: return "unsupported FLAC feature";
    case FLAC_E_MD5: return "decoded audio does not match the MD5 signature in STREAMINFO";
    case FLAC_E_MEM: return "out of memory";
    case FLAC_E_SYNC: return "lost FLAC frame sync";
    default: return "unknown FLAC error";
  }
}

/* ---------------------------------------------------------------- bit reader (MSB first) */
typedef struct { const uint8_t* p; size_t n, pos; uint64_t acc; int bits; int eof; } BR;

static void br_init(BR* b, const uint8_t* p, size_t n, size_t pos) { b->p = p; b->n = n; b->pos = pos; b->acc = 0; b->bits = 0; b->eof = 0; }
static void br_fill(BR* b) {
  while (b->bits <= 56) {
    uint64_t byte = 0;
    if (b->pos < b->n) byte = b->p[b->pos];
    else if (b->pos >= b->n + 8) { b->eof = 1; }
    b->pos++;
    b->acc |= byte << (56 - b->bits);
    b->bits += 8;
  }
}
static uint32_t br_u(BR* b, int n) {            /* n in 0..32 */
  if (n == 0) return 0;
  if (b->bits < n) br_fill(b);
  uint32_t v = (uint32_t)(b->acc >> (64 - n));
  b->acc <<= n; b->bits -= n;                      /* n <= 32 */
  return v;
}
static int32_t br_s(BR* b, int n) {             /* signed, n in 1..32 */
  uint32_t v = br_u(b, n);
  if (n < 32 && (v >> (n - 1))) v |= ~0u << n;
  return (int32_t)v;
}
static int64_t br_s64(BR* b, int n) {           /* signed up to 33 bits (side channel of 32-bit audio) */
  if (n <= 32) return br_s(b, n);
  int64_t hi = br_s(b, n - 32);
  return (hi << 32) | br_u(b, 32);
}
static uint32_t br_unary(BR* b) {               /* number of 0 bits before the next 1 bit */
  uint32_t q = 0;
  for (;;) {
    if (b->bits == 0) br_fill(b);
    if (b->acc == 0) { q += b->bits; b->bits = 0; if (b->eof) return q; continue; }
    int z = __builtin_clzll(b->acc);
    if (z >= b->bits) { q += b->bits; b->acc = 0; b->bits = 0; continue; }
    q += z;
    b->acc = z == 63 ? 0 : b->acc << (z + 1);      /* a 64-bit shift would be undefined */
    b->bits -= z + 1;
    return q;
  }
}
static size_t br_bytepos(const BR* b) { return b->pos - (size_t)(b->bits / 8); }   /* valid when byte aligned */
static void br_align(BR* b) { int r = b->bits & 7; b->acc <<= r; b->bits -= r; }

/* ---------------------------------------------------------------- CRCs */
static uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) { c ^= p[i]; for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1); }
  return c;
}
static uint16_t crc16(const uint8_t* p, size_t n) {
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) { c ^= (uint16_t)(p[i] << 8); for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1); }
  return c;
}

/* ---------------------------------------------------------------- MD5 (RFC 1321) */
typedef struct { uint32_t h[4]; uint64_t len; uint8_t buf[64]; size_t fill; } MD5;
static const uint32_t md5_k[64] = {
  0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,
  0x6b901122,0xfd987193,0xa679438e,0x49b40821,0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,
  0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,
  0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
  0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,
  0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391};
static const uint8_t md5_r[64] = {7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22,5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                                  4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23,6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21};
static void md5_block(MD5* m, const uint8_t* p) {
  uint32_t w[16], a = m->h[0], b = m->h[1], c = m->h[2], d = m->h[3];
  for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4*i] | (uint32_t)p[4*i+1] << 8 | (uint32_t)p[4*i+2] << 16 | (uint32_t)p[4*i+3] << 24;
  for (int i = 0; i < 64; ++i) {
    uint32_t f; int g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    uint32_t t = a + f + md5_k[i] + w[g];
    a = d; d = c; c = b; b = b + ((t << md5_r[i]) | (t >> (32 - md5_r[i])));
  }
  m->h[0] += a; m->h[1] += b; m->h[2] += c; m->h[3] += d;
}
static void md5_init(MD5* m) { m->h[0] = 0x67452301; m->h[1] = 0xefcdab89; m->h[2] = 0x98badcfe; m->h[3] = 0x10325476; m->len = 0; m->fill = 0; }
static void md5_update(MD5* m, const uint8_t* p, size_t n) {
  m->len += n;
  while (n) {
    size_t k = 64 - m->fill; if (k > n) k = n;
    memcpy(m->buf + m->fill, p, k); m->fill += k; p += k; n -= k;
    if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
  }
}
static void md5_final(MD5* m, uint8_t out[16]) {
  uint64_t bits = m->len * 8;
  uint8_t pad = 0x80; md5_update(m, &pad, 1);
  uint8_t z = 0; while (m->fill != 56) md5_update(m, &z, 1);
  uint8_t l[8]; for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
  md5_update(m, l, 8);
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) out[4*i+k] = (uint8_t)(m->h[i] >> (8 * k));
}

/* ---------------------------------------------------------------- subframes */
static int read_residual(BR* b, int64_t* s, int blocksize, int order) {
  int method = (int)br_u(b, 2);
  if (method > 1) return FLAC_E_UNSUP;
  int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  int porder = (int)br_u(b, 4);
  int nparts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) return FLAC_E_HEADER;
  int i = order;
  for (int part = 0; part < nparts; ++part) {
    int count = (blocksize >> porder) - (part == 0 ? order : 0);
    if (count < 0) return FLAC_E_HEADER;
    int k = (int)br_u(b, pbits);
    if (k == esc) {
      int nb = (int)br_u(b, 5);
      for (int j = 0; j < count; ++j) s[i++] = nb ? br_s(b, nb) : 0;
    } else {
      for (int j = 0; j < count; ++j) {
        uint32_t q = br_unary(b);
        uint64_t u = ((uint64_t)q << k) | br_u(b, k);
        s[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
      }
    }
    if (b->eof) return FLAC_E_TRUNC;
  }
  return FLAC_OK;
}

static int read_subframe(BR* b, int64_t* s, int blocksize, int bps) {
  if (br_u(b, 1)) return FLAC_E_HEADER;
  int type = (int)br_u(b, 6);
  int wasted = 0;
  if (br_u(b, 1)) wasted = (int)br_unary(b) + 1;
  bps -= wasted;
  if (bps <= 0) return FLAC_E_HEADER;
  if (type == 0) {
    int64_t v = br_s64(b, bps);
    for (int i = 0; i < blocksize; ++i) s[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) s[i] = br_s64(b, bps);
  } else if (type >= 8 && type <= 12) {
    int order = type - 8;
    if (order > blocksize) return FLAC_E_HEADER;
    for (int i = 0; i < order; ++i) s[i] = br_s64(b, bps);
    int rc = read_residual(b, s, blocksize, order);
    if (rc) return rc;
    uint64_t* u = (uint64_t*)s;                /* wrapping arithmetic, as in the LPC branch below */
    for (int i = order; i < blocksize; ++i) {
      switch (order) {
        case 0: break;
        case 1: u[i] += u[i-1]; break;
        case 2: u[i] += 2 * u[i-1] - u[i-2]; break;
        case 3: u[i] += 3 * u[i-1] - 3 * u[i-2] + u[i-3]; break;
        case 4: u[i] += 4 * u[i-1] - 6 * u[i-2] + 4 * u[i-3] - u[i-4]; break;
      }
    }
  } else if (type >= 32) {
    int order = (type & 31) + 1;
    if (order > blocksize) return FLAC_E_HEADER;
    for (int i = 0; i < order; ++i) s[i] = br_s64(b, bps);
    int prec = (int)br_u(b, 4) + 1;
    if (prec == 16) return FLAC_E_HEADER;
    int shift = br_s(b, 5);
    if (shift < 0) return FLAC_E_UNSUP;
    int32_t coef[32];
    for (int i = 0; i < order; ++i) coef[i] = br_s(b, prec);
    int rc = read_residual(b, s, blocksize, order);
    if (rc) return rc;
    for (int i = order; i < blocksize; ++i) {
      /* wrapping arithmetic: a valid stream never overflows 64 bits here, a corrupt one must not be undefined
         behaviour (its frame CRC / the MD5 reject it afterwards) */
      uint64_t sum = 0;
      for (int j = 0; j < order; ++j) sum += (uint64_t)(int64_t)coef[j] * (uint64_t)s[i - 1 - j];
      s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)sum >> shift));
    }
  } else {
    return FLAC_E_UNSUP;      /* reserved subframe types */
  }
  if (wasted) for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
  return b->eof ? FLAC_E_TRUNC : FLAC_OK;
}

/* ---------------------------------------------------------------- stream */
int wh_flac_decode(const uint8_t* data, size_t size, int32_t** samples_out, int64_t* n_frames_out, int* channels_out,
                   int* sample_rate_out, int* bps_out) {
  if (!data || !samples_out || !n_frames_out || !channels_out || !sample_rate_out || !bps_out) return FLAC_E_HEADER;
  *samples_out = NULL;
  size_t pos = 0;
  if (size >= 10 && !memcmp(data, "ID3", 3)) {        /* skip an ID3v2 tag */
    size_t tag = ((size_t)(data[6] & 127) << 21) | ((size_t)(data[7] & 127) << 14) | ((size_t)(data[8] & 127) << 7) | (data[9] & 127);
    pos = 10 + tag;
  }
  if (pos + 4 > size || memcmp(data + pos, "fLaC", 4)) return FLAC_E_MAGIC;
  pos += 4;
  int have_info = 0, sr = 0, ch = 0, bps = 0, max_block = 0;
  uint64_t total = 0;
  uint8_t md5_want[16];
  for (;;) {
    if (pos + 4 > size) return FLAC_E_TRUNC;
    int last = data[pos] >> 7, type = data[pos] & 127;
    size_t len = ((size_t)data[pos+1] << 16) | ((size_t)data[pos+2] << 8) | data[pos+3];
    pos += 4;
    if (pos + len > size) return FLAC_E_TRUNC;
    if (type == 0) {
      if (len < 34) return FLAC_E_HEADER;
      const uint8_t* q = data + pos;
      max_block = (q[2] << 8) | q[3];
      sr = (q[10] << 12) | (q[11] << 4) | (q[12] >> 4);
      ch = ((q[12] >> 1) & 7) + 1;
      bps = (((q[12] & 1) << 4) | (q[13] >> 4)) + 1;
      total = ((uint64_t)(q[13] & 15) << 32) | ((uint64_t)q[14] << 24) | ((uint64_t)q[15] << 16) | ((uint64_t)q[16] << 8) | q[17];
      memcpy(md5_want, q + 18, 16);
      have_info = 1;
    }
    pos += len;
    if (last) break;
  }
  if (!have_info || sr == 0 || max_block < 16 || bps < 4 || bps > 32) return FLAC_E_HEADER;

  /* STREAMINFO's sample count sizes the output only as far as the file could possibly hold that much: a frame of
     constant subframes packs 65535 samples per channel into ~16 bytes, nothing packs more; the buffer grows anyway */
  size_t cap = (size_t)1 << 20;
  if (total && total / 4096 <= (uint64_t)size) cap = (size_t)total;
  int32_t* out = (int32_t*)malloc(cap * ch * sizeof(int32_t));
  int64_t* work = (int64_t*)malloc((size_t)65536 * 8 * sizeof(int64_t));
  if (!out || !work) { free(out); free(work); return FLAC_E_MEM; }
  size_t done = 0;
  int rc = FLAC_OK;

  while (pos + 2 <= size) {
    if (!(data[pos] == 0xFF && (data[pos+1] & 0xFE) == 0xF8)) {
      /* trailing bytes (e.g. an ID3v1 tag) end the stream; garbage in the middle is an error */
      if (total && done >= total) break;
      if (size - pos <= 128) break;
      rc = FLAC_E_SYNC; break;
    }
    BR b; br_init(&b, data, size, pos);
    br_u(&b, 15);
    int variable = (int)br_u(&b, 1); (void)variable;
    int bs_code = (int)br_u(&b, 4), sr_code = (int)br_u(&b, 4), ch_code = (int)br_u(&b, 4), ss_code = (int)br_u(&b, 3);
    if (br_u(&b, 1)) { rc = FLAC_E_HEADER; break; }
    /* UTF-8 style coded frame / sample number */
    uint32_t first = br_u(&b, 8);
    int ones = 0, extra = 0;
    while (ones < 8 && (first & (0x80u >> ones))) ++ones;
    if (ones == 1 || ones == 8) { rc = FLAC_E_HEADER; break; }
    if (ones) extra = ones - 1;
    for (int i = 0; i < extra; ++i) if ((br_u(&b, 8) & 0xC0) != 0x80) { rc = FLAC_E_HEADER; }
    if (rc) break;
    int blocksize;
    if (bs_code == 0) { rc = FLAC_E_HEADER; break; }
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br_u(&b, 8) + 1;
    else if (bs_code == 7) blocksize = (int)br_u(&b, 16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br_u(&b, 8); else if (sr_code == 13 || sr_code == 14) br_u(&b, 16); else if (sr_code == 15) { rc = FLAC_E_HEADER; break; }
    int fbps = bps;
    switch (ss_code) { case 0: break; case 1: fbps = 8; break; case 2: fbps = 12; break; case 4: fbps = 16; break;
                       case 5: fbps = 20; break; case 6: fbps = 24; break; case 7: fbps = 32; break; default: rc = FLAC_E_HEADER; }
    if (rc) break;
    int nch = ch_code < 8 ? ch_code + 1 : 2;
    if (ch_code > 10 || nch != ch || fbps != bps || blocksize > 65535) { rc = ch_code > 10 ? FLAC_E_HEADER : FLAC_E_UNSUP; break; }
    size_t hdr_end = br_bytepos(&b);
    if (hdr_end + 1 > size) { rc = FLAC_E_TRUNC; break; }
    if (crc8(data + pos, hdr_end - pos) != data[hdr_end]) { rc = FLAC_E_CRC; break; }
    br_u(&b, 8);
    for (int c = 0; c < nch && !rc; ++c) {
      int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
      rc = read_subframe(&b, work + (size_t)c * 65536, blocksize, bps + side);
    }
    if (rc) break;
    br_align(&b);
    size_t body_end = br_bytepos(&b);
    if (body_end + 2 > size) { rc = FLAC_E_TRUNC; break; }
    if (crc16(data + pos, body_end - pos) != (uint16_t)((data[body_end] << 8) | data[body_end + 1])) { rc = FLAC_E_CRC; break; }
    pos = body_end + 2;

    if (done + (size_t)blocksize > cap) {
      size_t ncap = cap * 2 > done + blocksize ? cap * 2 : done + blocksize;
      int32_t* n = (int32_t*)realloc(out, ncap * ch * sizeof(int32_t));
      if (!n) { rc = FLAC_E_MEM; break; }
      out = n; cap = ncap;
    }
    int64_t* w0 = work; int64_t* w1 = work + 65536;
    for (int i = 0; i < blocksize; ++i) {
      if (ch_code == 8) w1[i] = w0[i] - w1[i];                       /* left, side  -> right = left - side */
      else if (ch_code == 9) w0[i] = w0[i] + w1[i];                  /* side, right -> left = right + side */
      else if (ch_code == 10) {                                      /* mid, side */
        int64_t side = w1[i], mid = (int64_t)((uint64_t)w0[i] << 1) | (side & 1);
        w0[i] = (mid + side) >> 1; w1[i] = (mid - side) >> 1;
      }
    }
    for (int c = 0; c < nch; ++c) {
      const int64_t* w = work + (size_t)c * 65536;
      for (int i = 0; i < blocksize; ++i) out[(done + i) * ch + c] = (int32_t)w[i];
    }
    done += blocksize;
  }
  free(work);
  if (!rc && total && done < total) rc = FLAC_E_TRUNC;
  if (!rc) {
    if (total && done > total) done = total;
    int nonzero = 0;
    for (int i = 0; i < 16; ++i) nonzero |= md5_want[i];
    if (nonzero) {                                                    /* an all-zero signature means "not computed" */
      MD5 m; md5_init(&m);
      const int nb = (bps + 7) / 8;
      uint8_t buf[4096]; size_t fill = 0;
      for (size_t i = 0; i < done * (size_t)ch; ++i) {
        uint32_t v = (uint32_t)out[i];
        for (int k = 0; k < nb; ++k) buf[fill++] = (uint8_t)(v >> (8 * k));
        if (fill + 4 > sizeof buf) { md5_update(&m, buf, fill); fill = 0; }
      }
      md5_update(&m, buf, fill);
      uint8_t got[16]; md5_final(&m, got);
      if (memcmp(got, md5_want, 16)) rc = FLAC_E_MD5;
    }
  }
  if (rc) { free(out); return rc; }
  *samples_out = out; *n_frames_out = (int64_t)done; *channels_out = ch; *sample_rate_out = sr; *bps_out = bps;
  return FLAC_OK;
}

void wh_flac_free(int32_t* samples) { free(samples); }
