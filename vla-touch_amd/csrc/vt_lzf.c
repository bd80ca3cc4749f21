/* vt_lzf.c — LZF block codec (the format of Marc Lehmann's liblzf, which h5py's HDF5 filter 32000 wraps) for the episode
 * reader / writer (vlatouch/h5lite.py).  Host-side C, own statement of the published format:
 *   control byte c < 32 : c + 1 literal bytes follow
 *   control byte c >= 32: back reference, length = (c >> 5) (+ next byte when 7) + 2,
 *                         distance = ((c & 31) << 8 | next byte) + 1 before the write position (may overlap).
 * Any encoder output obeying this is decodable by liblzf; the encoder below is a greedy 3-byte-hash matcher. */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "../../include/vtlzf.h"

long vt_lzf_decompress(const uint8_t* in, long in_len, uint8_t* out, long out_len) {
  long ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      ctrl++;
      if (op + ctrl > out_len || ip + ctrl > in_len) return -1;
      memcpy(out + op, in + ip, ctrl);
      ip += ctrl; op += ctrl;
    } else {
      long len = ctrl >> 5;
      long ref = op - (long)((ctrl & 0x1f) << 8) - 1;
      if (ip >= in_len) return -1;
      if (len == 7) { len += in[ip++]; if (ip >= in_len) return -1; }
      ref -= in[ip++];
      len += 2;
      if (ref < 0 || op + len > out_len) return -1;
      for (long k = 0; k < len; ++k) out[op + k] = out[ref + k];     /* byte-wise: the ranges may overlap */
      op += len;
    }
  }
  return op;
}

#define HLOG 16
#define MAX_LIT 32
#define MAX_OFF (1 << 13)
#define MAX_REF ((1 << 8) + (1 << 3))

/* returns the compressed size, or 0 when the output would not fit in out_cap (caller stores the block raw) */
long vt_lzf_compress(const uint8_t* in, long in_len, uint8_t* out, long out_cap) {
  static __thread long htab[1 << HLOG];
  if (in_len < 4 || out_cap < 4) return 0;
  for (long i = 0; i < (1 << HLOG); ++i) htab[i] = -1;
  long ip = 0, op = 0, lit_start = 0;
#define FLUSH_LITERALS(end)                                               \
  do {                                                                    \
    long s_ = lit_start, e_ = (end);                                      \
    while (s_ < e_) {                                                     \
      long n_ = e_ - s_ > MAX_LIT ? MAX_LIT : e_ - s_;                    \
      if (op + 1 + n_ > out_cap) return 0;                                \
      out[op++] = (uint8_t)(n_ - 1);                                      \
      memcpy(out + op, in + s_, n_); op += n_; s_ += n_;                  \
    }                                                                     \
  } while (0)
  while (ip + 2 < in_len) {
    const uint32_t v = ((uint32_t)in[ip] << 16) | ((uint32_t)in[ip + 1] << 8) | in[ip + 2];
    const uint32_t h = ((v * 2654435761u) >> (32 - HLOG)) & ((1u << HLOG) - 1);
    const long ref = htab[h];
    htab[h] = ip;
    long off;
    if (ref >= 0 && (off = ip - ref - 1) < MAX_OFF && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] && in[ref + 2] == in[ip + 2]) {
      long len = 3;
      const long maxlen = in_len - ip > MAX_REF ? MAX_REF : in_len - ip;
      while (len < maxlen && in[ref + len] == in[ip + len]) ++len;
      FLUSH_LITERALS(ip);
      const long l2 = len - 2;
      if (op + 3 > out_cap) return 0;
      if (l2 < 7) out[op++] = (uint8_t)((l2 << 5) | (off >> 8));
      else { out[op++] = (uint8_t)((7 << 5) | (off >> 8)); out[op++] = (uint8_t)(l2 - 7); }
      out[op++] = (uint8_t)(off & 0xff);
      ip += len;
      lit_start = ip;
    } else {
      ++ip;
    }
  }
  FLUSH_LITERALS(in_len);
  return op;
}
