// Host front half of the JPEG decoder (include/vitcap_jpeg.h): marker parsing + baseline Huffman entropy decoding into coefficient
// blocks.  Plain C++ (g++), no HIP: the loader's worker processes load libvitcap_jpeg.so through ctypes.  What the stream means is
// ITU-T T.81; which streams are accepted and which colour space a 3-component stream has follow libjpeg's rules
// (jdapimin.c default_decompress_parms), because the parity target is Pillow = libjpeg-turbo with default settings.
#include "../../include/vitcap_jpeg.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// zigzag position -> natural (row-major) position inside a block
const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int LOOK = 11;      // bits resolved by one table look-up

struct Huff {
  bool present = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // look-up by the next LOOK bits: (code length << 8) | symbol, 0 = longer than LOOK bits
  uint16_t look[1 << LOOK];
  int32_t maxcode[18];       // largest code of each length (-1 if none), maxcode[17] = sentinel
  int32_t valoff[17];        // vals index of the first code of a length minus that code
  bool build() {
    int code = 0, k = 0;
    memset(look, 0, sizeof(look));
    for (int l = 1; l <= 16; ++l) {
      valoff[l] = k - code;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (k >= 256) return false;
        if (l <= LOOK) {
          const int lo = code << (LOOK - l), n = 1 << (LOOK - l);
          if (lo + n > (1 << LOOK)) return false;
          for (int j = 0; j < n; ++j) look[lo + j] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      maxcode[l] = bits[l] ? code - 1 : -1;
      if (code > (1 << l)) return false;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    return true;
  }
};

struct Parsed {
  vitcap_jpeg_info info;
  int comp_id[3], comp_tq[3], comp_td[3], comp_ta[3];
  Huff dc[4], ac[4];
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  int restart_interval = 0;
  size_t scan_offset = 0;      // first entropy-coded byte
  int mcus_x = 0, mcus_y = 0;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int parse(const uint8_t* d, size_t n, Parsed& P) {
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return fail(VITCAP_JPEG_EINVAL, "not a JPEG (no SOI)");
  memset(&P.info, 0, sizeof(P.info));
  P.info.abi = VITCAP_JPEG_ABI;
  bool saw_jfif = false, saw_adobe = false, saw_sof = false;
  int adobe_transform = 0;
  size_t p = 2;
  while (true) {
    // next marker (fill bytes 0xff may precede it)
    if (p + 4 > n) return fail(VITCAP_JPEG_EINVAL, "truncated before the scan");
    if (d[p] != 0xff) return fail(VITCAP_JPEG_EINVAL, "marker expected at byte %zu", p);
    while (p < n && d[p] == 0xff) ++p;
    if (p >= n) return fail(VITCAP_JPEG_EINVAL, "truncated marker");
    const int m = d[p++];
    if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;      // stand-alone markers
    if (m == 0xd9) return fail(VITCAP_JPEG_EINVAL, "EOI before any scan");
    if (p + 2 > n) return fail(VITCAP_JPEG_EINVAL, "truncated segment");
    const int len = be16(d + p);
    if (len < 2 || p + len > n) return fail(VITCAP_JPEG_EINVAL, "bad segment length");
    const uint8_t* s = d + p + 2;
    const int sl = len - 2;
    switch (m) {
      case 0xe0:
        if (sl >= 5 && !memcmp(s, "JFIF\0", 5)) saw_jfif = true;
        break;
      case 0xee:
        if (sl >= 12 && !memcmp(s, "Adobe", 5)) {
          saw_adobe = true;
          adobe_transform = s[11];
        }
        break;
      case 0xdb: {
        int q = 0;
        while (q < sl) {
          const int pq = s[q] >> 4, tq = s[q] & 15;
          ++q;
          if (tq > 3 || pq > 1 || q + 64 * (pq + 1) > sl) return fail(VITCAP_JPEG_EINVAL, "bad DQT");
          for (int i = 0; i < 64; ++i) {
            P.qt[tq][ZZ[i]] = (uint16_t)(pq ? be16(s + q + 2 * i) : s[q + i]);
          }
          P.qt_present[tq] = true;
          q += 64 * (pq + 1);
        }
        break;
      }
      case 0xc4: {
        int q = 0;
        while (q < sl) {
          if (q + 17 > sl) return fail(VITCAP_JPEG_EINVAL, "bad DHT");
          const int tc = s[q] >> 4, th = s[q] & 15;
          if (tc > 1 || th > 3) return fail(VITCAP_JPEG_EINVAL, "bad DHT id");
          Huff& h = tc ? P.ac[th] : P.dc[th];
          int cnt = 0;
          h.bits[0] = 0;
          for (int i = 1; i <= 16; ++i) cnt += (h.bits[i] = s[q + i]);
          q += 17;
          if (cnt > 256 || q + cnt > sl) return fail(VITCAP_JPEG_EINVAL, "bad DHT counts");
          memset(h.vals, 0, sizeof(h.vals));
          memcpy(h.vals, s + q, cnt);
          q += cnt;
          if (!h.build()) return fail(VITCAP_JPEG_EINVAL, "inconsistent Huffman table");
          h.present = true;
        }
        break;
      }
      case 0xdd:
        if (sl < 2) return fail(VITCAP_JPEG_EINVAL, "bad DRI");
        P.restart_interval = be16(s);
        break;
      case 0xc0:
      case 0xc1: {
        if (saw_sof) return fail(VITCAP_JPEG_EINVAL, "two frame headers");
        saw_sof = true;
        if (sl < 6) return fail(VITCAP_JPEG_EINVAL, "bad SOF");
        if (s[0] != 8) return fail(VITCAP_JPEG_EUNSUPPORTED, "%d-bit samples", s[0]);
        P.info.height = be16(s + 1);
        P.info.width = be16(s + 3);
        P.info.ncomp = s[5];
        if (P.info.height <= 0 || P.info.width <= 0) return fail(VITCAP_JPEG_EUNSUPPORTED, "frame without a size (DNL)");
        if ((long long)P.info.height * P.info.width > (1ll << 26)) return fail(VITCAP_JPEG_EUNSUPPORTED, "more than 64 Mpixel");
        if (P.info.ncomp != 1 && P.info.ncomp != 3) return fail(VITCAP_JPEG_EUNSUPPORTED, "%d components", P.info.ncomp);
        if (sl < 6 + 3 * P.info.ncomp) return fail(VITCAP_JPEG_EINVAL, "bad SOF length");
        for (int c = 0; c < P.info.ncomp; ++c) {
          P.comp_id[c] = s[6 + 3 * c];
          P.info.hs[c] = s[7 + 3 * c] >> 4;
          P.info.vs[c] = s[7 + 3 * c] & 15;
          P.comp_tq[c] = s[8 + 3 * c];
          if (P.comp_tq[c] > 3 || P.info.hs[c] < 1 || P.info.hs[c] > 4 || P.info.vs[c] < 1 || P.info.vs[c] > 4)
            return fail(VITCAP_JPEG_EINVAL, "bad component");
        }
        break;
      }
      case 0xc2: case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
        return fail(VITCAP_JPEG_EUNSUPPORTED, "frame type SOF%d (progressive / lossless / arithmetic)", m - 0xc0);
      case 0xda: {
        if (!saw_sof) return fail(VITCAP_JPEG_EINVAL, "scan before the frame header");
        vitcap_jpeg_info& I = P.info;
        if (sl < 1 || s[0] != I.ncomp || sl < 4 + 2 * I.ncomp)
          return fail(VITCAP_JPEG_EUNSUPPORTED, "a scan of %d of the %d components (several scans)", sl >= 1 ? s[0] : -1, I.ncomp);
        for (int c = 0; c < I.ncomp; ++c) {
          if (s[1 + 2 * c] != P.comp_id[c]) return fail(VITCAP_JPEG_EUNSUPPORTED, "scan components out of frame order");
          P.comp_td[c] = s[2 + 2 * c] >> 4;
          P.comp_ta[c] = s[2 + 2 * c] & 15;
          if (P.comp_td[c] > 3 || P.comp_ta[c] > 3 || !P.dc[P.comp_td[c]].present || !P.ac[P.comp_ta[c]].present)
            return fail(VITCAP_JPEG_EINVAL, "scan names a missing Huffman table");
          if (!P.qt_present[P.comp_tq[c]]) return fail(VITCAP_JPEG_EINVAL, "component names a missing quantisation table");
          memcpy(I.qt[c], P.qt[P.comp_tq[c]], sizeof(I.qt[c]));
        }
        const uint8_t* t = s + 1 + 2 * I.ncomp;
        if (t[0] != 0 || t[1] != 63 || t[2] != 0) return fail(VITCAP_JPEG_EUNSUPPORTED, "spectral selection / successive approximation");
        // colour space of a 3-component stream: libjpeg jdapimin.c default_decompress_parms
        if (I.ncomp == 3) {
          bool ycc = true;
          if (saw_jfif) ycc = true;
          else if (saw_adobe) ycc = adobe_transform != 0;
          else if (P.comp_id[0] == 'R' && P.comp_id[1] == 'G' && P.comp_id[2] == 'B') ycc = false;
          if (!ycc) return fail(VITCAP_JPEG_EUNSUPPORTED, "RGB colour space");
          if (I.hs[1] != 1 || I.vs[1] != 1 || I.hs[2] != 1 || I.vs[2] != 1 ||
              !((I.hs[0] == 1 && I.vs[0] == 1) || (I.hs[0] == 2 && I.vs[0] == 1) || (I.hs[0] == 2 && I.vs[0] == 2)))
            return fail(VITCAP_JPEG_EUNSUPPORTED, "sampling %dx%d,%dx%d,%dx%d", I.hs[0], I.vs[0], I.hs[1], I.vs[1], I.hs[2], I.vs[2]);
        } else {
          I.hs[0] = I.vs[0] = 1;       // a single-component scan is never interleaved: one block per MCU whatever the factors say
        }
        const int hmax = I.hs[0], vmax = I.vs[0];
        P.mcus_x = (I.width + 8 * hmax - 1) / (8 * hmax);
        P.mcus_y = (I.height + 8 * vmax - 1) / (8 * vmax);
        int nb = 0;
        for (int c = 0; c < I.ncomp; ++c) {
          I.blocks_w[c] = P.mcus_x * I.hs[c];
          I.blocks_h[c] = P.mcus_y * I.vs[c];
          I.samp_w[c] = (I.width * I.hs[c] + hmax - 1) / hmax;
          I.samp_h[c] = (I.height * I.vs[c] + vmax - 1) / vmax;
          I.block0[c] = nb;
          nb += I.blocks_w[c] * I.blocks_h[c];
          // jdsample.c: fancy upsampling only for planes wider than 2 samples (narrower ones are replicated)
          if (c > 0 && I.hs[0] == 2 && I.samp_w[c] <= 2) return fail(VITCAP_JPEG_EUNSUPPORTED, "chroma plane %d samples wide", I.samp_w[c]);
        }
        I.nblocks = nb;
        P.scan_offset = p + len;
        return VITCAP_JPEG_OK;
      }
      default:
        break;      // APPn, COM, ...: skipped
    }
    p += len;
  }
}

// ---- entropy-coded data.  A first pass removes the byte stuffing (0xff00 -> 0xff) and cuts the scan at its restart markers, so the bit
// reader itself never looks for markers: it refills 32 bits at a time from a clean, zero-padded buffer (libjpeg feeds zeros after a
// premature end of data as well).
struct Segments {
  std::vector<uint8_t> bytes;          // unstuffed data of all intervals, each followed by 16 zero bytes
  std::vector<size_t> start;           // first byte of every restart interval
};

bool unstuff(const uint8_t* d, size_t n, size_t p, Segments& S) {
  S.bytes.clear();
  S.bytes.reserve(n - p + 64);
  S.start.assign(1, 0);
  int expect = 0;
  while (p < n) {
    const uint8_t* f = (const uint8_t*)memchr(d + p, 0xff, n - p);
    const size_t q = f ? (size_t)(f - d) : n;
    S.bytes.insert(S.bytes.end(), d + p, d + q);
    if (!f || q + 1 >= n) break;
    const int m = d[q + 1];
    if (m == 0) {
      S.bytes.push_back(0xff);
      p = q + 2;
    } else if (m == 0xff) {
      p = q + 1;                         // fill byte
    } else if (m >= 0xd0 && m <= 0xd7) {
      if (m != 0xd0 + expect) return false;
      expect = (expect + 1) & 7;
      S.bytes.insert(S.bytes.end(), 16, 0);
      S.start.push_back(S.bytes.size());
      p = q + 2;
    } else {
      break;                             // EOI or any other marker: end of the scan
    }
  }
  S.bytes.insert(S.bytes.end(), 16, 0);
  return true;
}

struct Bits {
  const uint8_t* p;
  const uint8_t* end;                    // reads past `end` return zeros
  uint64_t acc = 0;
  int cnt = 0;                           // valid bits in the low end of acc
  inline void fill() {
    if (cnt <= 32) {
      uint32_t w;
      if (p + 4 <= end) {
        w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        p += 4;
      } else {
        w = 0;
        for (int i = 0; i < 4; ++i) w = (w << 8) | (p < end ? *p++ : 0);
      }
      acc = (acc << 32) | w;
      cnt += 32;
    }
  }
  inline int peek(int k) const { return (int)((acc >> (cnt - k)) & ((1u << k) - 1)); }
  inline void skip(int k) { cnt -= k; }
  inline int get(int k) {
    const int v = peek(k);
    cnt -= k;
    return v;
  }
};

inline int decode_sym(Bits& b, const Huff& h) {
  const int e = h.look[b.peek(LOOK)];
  if (e) {
    b.skip(e >> 8);
    return e & 0xff;
  }
  int l = LOOK + 1;
  int code = b.peek(l);
  while (l <= 16 && code > h.maxcode[l]) {
    ++l;
    code = b.peek(l);
  }
  if (l > 16) {
    b.skip(16);
    return 0;         // corrupt data: libjpeg warns and returns zero
  }
  b.skip(l);
  return h.vals[(code + h.valoff[l]) & 0xff];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// AC coefficients whose code AND value bits fit the look-ahead window decode with one table access:
// entry = (value << 8) | (run << 4) | total bits, 0 = take the general path
struct FastAC {
  int32_t t[1 << LOOK];
  void build(const Huff& h) {
    for (int i = 0; i < (1 << LOOK); ++i) {
      t[i] = 0;
      const int e = h.look[i];
      if (!e) continue;
      const int len = e >> 8, rs = e & 0xff, run = rs >> 4, sz = rs & 15;
      if (sz == 0 || len + sz > LOOK) continue;
      const int v = extend((i >> (LOOK - len - sz)) & ((1 << sz) - 1), sz);
      t[i] = (int32_t)(((uint32_t)v << 8) | (uint32_t)(run << 4) | (uint32_t)(len + sz));
    }
  }
};

}  // namespace

extern "C" int vitcap_jpeg_abi(void) { return VITCAP_JPEG_ABI; }
extern "C" const char* vitcap_jpeg_last_error(void) { return g_err; }

extern "C" int vitcap_jpeg_parse(const uint8_t* data, size_t n, vitcap_jpeg_info* info) {
  if (!data || !info) return fail(VITCAP_JPEG_EINVAL, "null argument");
  Parsed P;
  const int rc = parse(data, n, P);
  if (rc == VITCAP_JPEG_OK) *info = P.info;
  return rc;
}

extern "C" int vitcap_jpeg_decode_coefs(const uint8_t* data, size_t n, const vitcap_jpeg_info* info, int16_t* coefs) {
  if (!data || !info || !coefs) return fail(VITCAP_JPEG_EINVAL, "null argument");
  Parsed P;
  const int rc = parse(data, n, P);
  if (rc != VITCAP_JPEG_OK) return rc;
  const vitcap_jpeg_info& I = P.info;
  if (info->abi != VITCAP_JPEG_ABI || info->nblocks != I.nblocks || info->width != I.width || info->height != I.height)
    return fail(VITCAP_JPEG_EINVAL, "info does not belong to this stream");
  memset(coefs, 0, (size_t)I.nblocks * 64 * sizeof(int16_t));
  thread_local Segments S;
  if (!unstuff(data, n, P.scan_offset, S)) return fail(VITCAP_JPEG_EINVAL, "restart markers out of sequence");
  FastAC fast[4];
  bool fast_built[4] = {false, false, false, false};
  for (int c = 0; c < I.ncomp; ++c)
    if (!fast_built[P.comp_ta[c]]) {
      fast[P.comp_ta[c]].build(P.ac[P.comp_ta[c]]);
      fast_built[P.comp_ta[c]] = true;
    }
  const size_t total_mcus = (size_t)P.mcus_x * P.mcus_y;
  const size_t per_seg = P.restart_interval ? (size_t)P.restart_interval : total_mcus;
  const size_t need_segs = (total_mcus + per_seg - 1) / per_seg;
  if (S.start.size() < need_segs) return fail(VITCAP_JPEG_EINVAL, "%zu restart intervals found, %zu needed", S.start.size(), need_segs);
  size_t mcu = 0;
  for (size_t seg = 0; seg < need_segs; ++seg) {
    const uint8_t* sb = S.bytes.data() + S.start[seg];
    const uint8_t* se = seg + 1 < S.start.size() ? S.bytes.data() + S.start[seg + 1] : S.bytes.data() + S.bytes.size();
    Bits b{sb, se};
    int pred[3] = {0, 0, 0};
    const size_t mcu_end = mcu + per_seg < total_mcus ? mcu + per_seg : total_mcus;
    for (; mcu < mcu_end; ++mcu) {
      const int my = (int)(mcu / P.mcus_x), mx = (int)(mcu % P.mcus_x);
      for (int c = 0; c < I.ncomp; ++c) {
        const Huff& hd = P.dc[P.comp_td[c]];
        const Huff& ha = P.ac[P.comp_ta[c]];
        const int32_t* fa = fast[P.comp_ta[c]].t;
        for (int by = 0; by < I.vs[c]; ++by) {
          for (int bx = 0; bx < I.hs[c]; ++bx) {
            int16_t* blk = coefs + ((size_t)I.block0[c] + (size_t)(my * I.vs[c] + by) * I.blocks_w[c] + (mx * I.hs[c] + bx)) * 64;
            b.fill();
            int s = decode_sym(b, hd);
            if (s) {
              if (s > 15) s = 15;
              pred[c] += extend(b.get(s), s);
            }
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              b.fill();      // >= 33 bits: a code (<= 16) and its value bits (<= 15)
              const int32_t f = fa[b.peek(LOOK)];
              if (f) {
                k += (f >> 4) & 15;
                b.skip(f & 15);
                blk[ZZ[k & 63]] = (int16_t)(f >> 8);
                ++k;
                continue;
              }
              const int rs = decode_sym(b, ha);
              const int r = rs >> 4, sz = rs & 15;
              if (sz) {
                k += r;
                blk[ZZ[k & 63]] = (int16_t)extend(b.get(sz), sz);
                ++k;
              } else {
                if (r != 15) break;      // EOB
                k += 16;
              }
            }
          }
        }
      }
    }
  }
  return VITCAP_JPEG_OK;
}
