// lk_image_io.cpp - decoded frames for headless runs (include/lk_tracker.h: lk_load_image, lk_decode_image).
//
// What the reference hands its engines is cv::imread(path, IMREAD_GRAYSCALE) (manager_class.cpp:102-107,174,211,250;
// cuda_class.cu:484-510): an 8-bit single-channel image whatever the file holds.  OpenCV is not in this image, so the
// containers its sample data comes in (PNG: mainapp.cpp:384-408) and the simple ones are decoded here, with the
// conversions OpenCV's decoders apply for that flag:
//   PNG   every colour type and bit depth, Adam7 interlacing; zlib inflates, the rest is here.  16 bits -> the high byte
//         (png_set_strip_16), 1/2/4-bit grey -> bit replication (expand_gray_1_2_4_to_8), alpha dropped, palette -> RGB,
//         RGB -> grey with libpng's 15-bit coefficients of png_set_rgb_to_gray(0.299, 0.587): (9797 R + 19234 G + 3737 B
//         + 16384) >> 15
//   BMP   uncompressed 1 / 4 / 8-bit palette, 24 and 32 bits, bottom-up or top-down
//   PNM   P1-P6 (maxval <= 255 as stored, 16-bit samples -> the high byte)
//   TIFF  the container most cameras for image correlation write: either byte order, strips or tiles, 8 / 16-bit grey (either
//         polarity), grey + alpha, RGB(A), 8-bit palette; uncompressed, PackBits, LZW, Deflate; horizontal differencing; first page
//   BMP, PPM and TIFF colour -> grey with OpenCV's 14-bit coefficients (4899 R + 9617 G + 1868 B + 8192) >> 14
// Grey files decode exactly (tests/test_image_io.py compares with PIL's decoders); the two colour conversions and the 16 -> 8
// bit reduction (the high byte) are restated from the libraries' documented behaviour and are NOT pinned against OpenCV (it
// cannot be built or run here).  JPEG needs libjpeg's headers, which this image lacks (and lossy frames are no input for
// image correlation): LK_ERROR_BAD_DOMAIN, like BigTIFF, planar or float TIFF and every malformed or truncated file.
// Every length in a file is checked against the bytes that are there before it is used (ASan/UBSan run of the corpus and
// of truncated / bit-flipped copies: tests/test_image_io.py).
#include "../../include/lk_tracker.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Bytes {
  const uint8_t *p;
  size_t n;
  bool has(size_t at, size_t len) const { return at <= n && len <= n - at; }
  uint32_t be32(size_t at) const { return (uint32_t)p[at] << 24 | (uint32_t)p[at + 1] << 16 | (uint32_t)p[at + 2] << 8 | p[at + 3]; }
  uint32_t le32(size_t at) const { return (uint32_t)p[at + 3] << 24 | (uint32_t)p[at + 2] << 16 | (uint32_t)p[at + 1] << 8 | p[at]; }
  uint32_t le16(size_t at) const { return (uint32_t)p[at + 1] << 8 | p[at]; }
};

const long long kMaxPixels = 1ll << 30; // (the engine's own bound on a frame is far below)

inline uint8_t grey_libpng(unsigned r, unsigned g, unsigned b) {
  // (libpng leaves r == g == b pixels alone; the formula gives the same value for them: the coefficients sum to 32768)
  return (uint8_t)((9797u * r + 19234u * g + 3737u * b + 16384u) >> 15);
}
inline uint8_t grey_opencv(unsigned r, unsigned g, unsigned b) { return (uint8_t)((4899u * r + 9617u * g + 1868u * b + 8192u) >> 14); }

struct Image {
  uint8_t *px = nullptr;
  int rows = 0, cols = 0;
  bool alloc(long long w, long long h) {
    if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || w * h > kMaxPixels)
      return false;
    px = (uint8_t *)std::malloc((size_t)(w * h));
    rows = (int)h;
    cols = (int)w;
    return px != nullptr;
  }
};

// ---- PNG ---------------------------------------------------------------------------------------------------------
bool decode_png(const Bytes &f, Image &out) {
  size_t at = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool seen_end = false;
  while (!seen_end) {
    if (!f.has(at, 12))
      return false;
    const uint32_t len = f.be32(at);
    if (len > 0x7fffffffu || !f.has(at + 8, (size_t)len + 4))
      return false;
    const uint8_t *type = f.p + at + 4, *data = f.p + at + 8;
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, (uInt)(len + 4)) != f.be32(at + 8 + len))
      return false;
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13 || ctype >= 0)
        return false;
      w = f.be32(at + 8), h = f.be32(at + 12);
      depth = data[8], ctype = data[9], interlace = data[12];
      if (data[10] != 0 || data[11] != 0 || interlace > 1)
        return false;
    } else if (ctype < 0) {
      return false; // IHDR comes first
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (len % 3 != 0 || len > 768)
        return false;
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      seen_end = true;
    } else if (!(type[0] & 0x20)) {
      return false; // an unknown critical chunk
    }
    at += 12 + (size_t)len;
  }
  int channels = 0;
  switch (ctype) {
  case 0: channels = 1; break;
  case 2: channels = 3; break;
  case 3: channels = 1; break;
  case 4: channels = 2; break;
  case 6: channels = 4; break;
  default: return false;
  }
  const bool depth_ok = ctype == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (depth == 8 || depth == 16);
  if (!depth_ok || (ctype == 3 && plte.empty()) || idat.empty() || w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24))
    return false;
  const int bits = depth * channels;      // per pixel
  const size_t bpp = bits >= 8 ? (size_t)bits / 8 : 1; // the filters' "corresponding byte to the left"
  // the reduced images of the stream: one (the image itself) or Adam7's seven
  static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
  struct Pass {
    size_t pw, ph, row_bytes;
    int x0, y0, dx, dy;
  };
  std::vector<Pass> passes;
  size_t raw = 0;
  for (int k = 0; k < (interlace ? 7 : 1); ++k) {
    Pass p;
    p.x0 = interlace ? x0[k] : 0, p.y0 = interlace ? y0[k] : 0, p.dx = interlace ? dx[k] : 1, p.dy = interlace ? dy[k] : 1;
    p.pw = ((size_t)w - (size_t)p.x0 + (size_t)p.dx - 1) / (size_t)p.dx;
    p.ph = ((size_t)h - (size_t)p.y0 + (size_t)p.dy - 1) / (size_t)p.dy;
    if ((size_t)p.x0 >= w || (size_t)p.y0 >= h)
      p.pw = p.ph = 0;
    p.row_bytes = (p.pw * (size_t)bits + 7) / 8;
    if (p.pw && p.ph)
      raw += p.ph * (1 + p.row_bytes);
    passes.push_back(p);
  }
  // deflate expands by at most 1032 : 1 - a header that promises more than its data can hold is refused before anything is allocated
  if (raw > 0xffffffffu || idat.size() > 0xffffffffu || raw / 1032 > idat.size() + 1 || !out.alloc(w, h))
    return false;
  std::vector<uint8_t> data(raw);
  {
    z_stream z;
    std::memset(&z, 0, sizeof(z));
    if (inflateInit(&z) != Z_OK)
      return false;
    z.next_in = idat.data();
    z.avail_in = (uInt)idat.size();
    z.next_out = data.data();
    z.avail_out = (uInt)raw;
    const int rc = inflate(&z, Z_FINISH);
    const bool full = z.avail_out == 0;
    inflateEnd(&z);
    if (!(rc == Z_STREAM_END || rc == Z_BUF_ERROR || rc == Z_OK) || !full) // (trailing bytes after the image are tolerated, missing ones are not)
      return false;
  }
  // grey value of one decoded sample group
  uint8_t pal_grey[256];
  std::memset(pal_grey, 0, sizeof(pal_grey));
  for (size_t i = 0; i + 2 < plte.size(); i += 3)
    pal_grey[i / 3] = grey_libpng(plte[i], plte[i + 1], plte[i + 2]);
  const size_t pal_n = plte.size() / 3;
  size_t src = 0;
  std::vector<uint8_t> prev, cur;
  for (const Pass &p : passes) {
    if (!p.pw || !p.ph)
      continue;
    prev.assign(p.row_bytes, 0);
    cur.resize(p.row_bytes);
    for (size_t r = 0; r < p.ph; ++r) {
      const int filter = data[src++];
      const uint8_t *in = data.data() + src;
      src += p.row_bytes;
      for (size_t i = 0; i < p.row_bytes; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
        int v = in[i];
        switch (filter) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
        }
        cur[i] = (uint8_t)v;
      }
      uint8_t *dst = out.px + ((size_t)p.y0 + r * (size_t)p.dy) * (size_t)w;
      for (size_t x = 0; x < p.pw; ++x) {
        uint8_t g;
        if (depth < 8) {
          const size_t bit = x * (size_t)depth;
          const unsigned v = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u);
          if (ctype == 3) {
            if (v >= pal_n)
              return false;
            g = pal_grey[v];
          } else {
            g = (uint8_t)(v * (255u / ((1u << depth) - 1u)));
          }
        } else {
          const uint8_t *s = cur.data() + x * bpp; // 16-bit samples: big endian, the high byte is the first
          const size_t step = depth == 16 ? 2 : 1;
          if (ctype == 3) {
            if (s[0] >= pal_n)
              return false;
            g = pal_grey[s[0]];
          } else if (channels <= 2) {
            g = s[0];
          } else {
            g = grey_libpng(s[0], s[step], s[2 * step]);
          }
        }
        dst[(size_t)p.x0 + x * (size_t)p.dx] = g;
      }
      prev.swap(cur);
    }
  }
  return true;
}

// ---- BMP ---------------------------------------------------------------------------------------------------------
bool decode_bmp(const Bytes &f, Image &out) {
  if (!f.has(0, 14 + 12))
    return false;
  const size_t data_at = f.le32(10), hdr = f.le32(14);
  long long w, h;
  int bpp, compression = 0;
  size_t pal_entry = 4, pal_at = 14 + hdr;
  uint32_t pal_count = 0;
  if (hdr == 12) {
    w = (long long)f.le16(18), h = (long long)f.le16(20), bpp = (int)f.le16(24);
    pal_entry = 3;
  } else if (hdr >= 40 && f.has(14, hdr)) {
    w = (long long)(int32_t)f.le32(18), h = (long long)(int32_t)f.le32(22), bpp = (int)f.le16(28);
    compression = (int)f.le32(30);
    pal_count = f.le32(46);
  } else {
    return false;
  }
  const bool top_down = h < 0;
  if (top_down)
    h = -h;
  if (!(compression == 0 || (compression == 3 && bpp == 32)) || !(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32))
    return false;
  uint8_t pal_grey[256];
  std::memset(pal_grey, 0, sizeof(pal_grey));
  if (bpp <= 8) {
    if (pal_count == 0 || pal_count > (1u << bpp))
      pal_count = 1u << bpp;
    if (!f.has(pal_at, (size_t)pal_count * pal_entry))
      return false;
    for (uint32_t i = 0; i < pal_count; ++i) {
      const uint8_t *e = f.p + pal_at + (size_t)i * pal_entry; // B, G, R(, reserved)
      pal_grey[i] = grey_opencv(e[2], e[1], e[0]);
    }
  }
  if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24))
    return false;
  const size_t stride = (((size_t)w * (size_t)bpp + 31) / 32) * 4;
  if (!f.has(data_at, stride * (size_t)h - (stride - ((size_t)w * (size_t)bpp + 7) / 8)) || !out.alloc(w, h)) // (the last row's padding may be missing)
    return false;
  for (long long y = 0; y < h; ++y) {
    const uint8_t *s = f.p + data_at + (size_t)(top_down ? y : h - 1 - y) * stride;
    uint8_t *d = out.px + (size_t)y * (size_t)w;
    for (long long x = 0; x < w; ++x) {
      switch (bpp) {
      case 1: d[x] = pal_grey[(s[x >> 3] >> (7 - (x & 7))) & 1]; break;
      case 4: d[x] = pal_grey[(s[x >> 1] >> ((x & 1) ? 0 : 4)) & 15]; break;
      case 8: d[x] = pal_grey[s[x]]; break;
      case 24: d[x] = grey_opencv(s[3 * x + 2], s[3 * x + 1], s[3 * x]); break;
      default: d[x] = grey_opencv(s[4 * x + 2], s[4 * x + 1], s[4 * x]); break;
      }
    }
  }
  return true;
}

// ---- TIFF --------------------------------------------------------------------------------------------------------
// Baseline TIFF 6.0 plus what scientific cameras and PIL / libtiff write: either byte order, strips or tiles, 8 or 16 bits
// per sample, grey (either polarity), grey + alpha, RGB(A), 8-bit palette; uncompressed, PackBits, LZW, Deflate; horizontal
// differencing.  The first page only (as cv::imread).  Not: BigTIFF, planar RGB, JPEG-in-TIFF, float samples, 1 / 4-bit.
struct TiffChunk {
  std::vector<uint8_t> px; // decoded rows: `rows` x `row_bytes`
};

bool tiff_unpack_bits(const uint8_t *s, size_t n, uint8_t *d, size_t want) {
  size_t i = 0, o = 0;
  while (o < want && i < n) {
    const int c = (int8_t)s[i++];
    if (c >= 0) {
      const size_t len = (size_t)c + 1;
      if (i + len > n)
        return false;
      const size_t take = len < want - o ? len : want - o;
      std::memcpy(d + o, s + i, take);
      o += take;
      i += len;
    } else if (c != -128) {
      if (i >= n)
        return false;
      const size_t len = (size_t)(1 - c), take = len < want - o ? len : want - o;
      std::memset(d + o, s[i++], take);
      o += take;
    }
  }
  return o == want;
}

bool tiff_lzw(const uint8_t *s, size_t n, uint8_t *d, size_t want) {
  struct Code {
    uint16_t prefix;
    uint8_t suffix, first;
    uint32_t length;
  };
  std::vector<Code> t(4096);
  for (int i = 0; i < 256; ++i)
    t[(size_t)i] = Code{0, (uint8_t)i, (uint8_t)i, 1};
  int next = 258, width = 9, prev = -1;
  uint32_t acc = 0;
  int have = 0;
  size_t i = 0, o = 0;
  while (o < want) {
    while (have < width && i < n) {
      acc = (acc << 8) | s[i++];
      have += 8;
    }
    if (have < width)
      break;
    const int code = (int)((acc >> (have - width)) & ((1u << width) - 1u));
    have -= width;
    if (code == 256) {
      next = 258, width = 9, prev = -1;
      continue;
    }
    if (code == 257)
      break;
    if (prev < 0) {
      if (code > 255)
        return false;
      d[o++] = (uint8_t)code;
      prev = code;
      continue;
    }
    if (code > next || (code >= 256 && code < 258))
      return false;
    // the string of `code`, or (code == next) the previous string + its own first byte
    const int src = code < next ? code : prev;
    const uint32_t len = t[(size_t)src].length + (code < next ? 0u : 1u);
    const uint8_t first = t[(size_t)src].first;
    const size_t end = o + len;
    size_t at = end;
    if (code == next) {
      --at;
      if (at < want)
        d[at] = first;
    }
    for (int c = src; at > o; c = t[(size_t)c].prefix) { // back to front
      --at;
      if (at < want)
        d[at] = t[(size_t)c].suffix;
      if (t[(size_t)c].length == 1)
        break;
    }
    o = end < want ? end : want;
    if (next < 4096) {
      t[(size_t)next] = Code{(uint16_t)prev, first, t[(size_t)prev].first, t[(size_t)prev].length + 1};
      ++next;
      if (next >= (1 << width) - 1 && width < 12) // (TIFF's early change)
        ++width;
    }
    prev = code;
  }
  return o == want;
}

bool decode_tiff(const Bytes &f, Image &out) {
  if (!f.has(0, 8))
    return false;
  const bool le = f.p[0] == 'I';
  auto u16 = [&](size_t at) { return le ? f.le16(at) : ((uint32_t)f.p[at] << 8 | f.p[at + 1]); };
  auto u32 = [&](size_t at) { return le ? f.le32(at) : f.be32(at); };
  if (u16(2) != 42)
    return false; // (43: BigTIFF)
  const size_t ifd = u32(4);
  if (!f.has(ifd, 2))
    return false;
  const size_t n_entries = u16(ifd);
  if (!f.has(ifd + 2, n_entries * 12))
    return false;
  struct Entry {
    uint32_t type = 0, count = 0;
    size_t at = 0; // where the values are
  };
  auto find = [&](uint32_t tag, Entry &e) {
    for (size_t i = 0; i < n_entries; ++i) {
      const size_t at = ifd + 2 + 12 * i;
      if (u16(at) != tag)
        continue;
      e.type = u16(at + 2);
      e.count = u32(at + 4);
      const size_t unit = e.type == 3 ? 2 : e.type == 4 ? 4 : e.type == 1 ? 1 : 0;
      if (!unit || e.count == 0 || e.count > (1u << 28))
        return false;
      const size_t bytes = unit * (size_t)e.count;
      e.at = bytes <= 4 ? at + 8 : (size_t)u32(at + 8);
      return f.has(e.at, bytes);
    }
    return false;
  };
  auto value = [&](const Entry &e, size_t k) -> uint32_t { return e.type == 3 ? u16(e.at + 2 * k) : e.type == 4 ? u32(e.at + 4 * k) : f.p[e.at + k]; };
  auto scalar = [&](uint32_t tag, uint32_t fallback) {
    Entry e;
    return find(tag, e) ? value(e, 0) : fallback;
  };
  const long long w = scalar(256, 0), h = scalar(257, 0);
  const uint32_t spp = scalar(277, 1), compression = scalar(259, 1), photometric = scalar(262, 1), predictor = scalar(317, 1);
  uint32_t bps = 1;
  {
    Entry e;
    if (find(258, e)) {
      bps = value(e, 0);
      for (size_t k = 1; k < e.count && k < 8; ++k)
        if (value(e, k) != bps)
          return false;
    }
  }
  if ((bps != 8 && bps != 16) || spp < 1 || spp > 4 || (spp > 1 && scalar(284, 1) != 1) || scalar(339, 1) != 1 || scalar(266, 1) != 1 ||
      !(compression == 1 || compression == 5 || compression == 8 || compression == 32946 || compression == 32773) || predictor > 2 ||
      photometric > 3 || (photometric == 2 && spp < 3) || (photometric == 3 && (spp != 1 || bps != 8)) || (photometric < 2 && spp > 2))
    return false;
  uint8_t pal_grey[256];
  if (photometric == 3) {
    Entry e;
    if (!find(320, e) || e.type != 3 || e.count < 3 * 256)
      return false;
    for (size_t i = 0; i < 256; ++i)
      pal_grey[i] = grey_opencv(value(e, i) >> 8, value(e, 256 + i) >> 8, value(e, 512 + i) >> 8);
  }
  if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || w * h > kMaxPixels)
    return false;
  // strips (full-width chunks of rows_per_strip rows) or tiles
  Entry offs, counts;
  long long cw = w, chh;
  const bool tiled = find(324, offs);
  if (tiled) {
    cw = scalar(322, 0), chh = scalar(323, 0);
    if (!find(325, counts))
      return false;
  } else {
    chh = scalar(278, 0xffffffffu);
    if (!find(273, offs) || !find(279, counts))
      return false;
  }
  if (chh > h && !tiled)
    chh = h;
  if (cw <= 0 || chh <= 0 || cw > (1 << 24) || chh > (1 << 24))
    return false;
  const long long across = (w + cw - 1) / cw, down = (h + chh - 1) / chh;
  if ((long long)offs.count < across * down || (long long)counts.count < across * down)
    return false;
  const size_t sample_bytes = bps / 8, px_bytes = sample_bytes * spp, row_bytes = (size_t)cw * px_bytes;
  // a file cannot hold more than its compression allows (LZW at most ~4096 : 1): refuse before allocating
  if ((double)w * (double)h * (double)px_bytes > 4096.0 * (double)f.n + 65536.0 || !out.alloc(w, h))
    return false;
  std::vector<uint8_t> chunk;
  for (long long cy = 0; cy < down; ++cy)
    for (long long cx = 0; cx < across; ++cx) {
      const size_t k = (size_t)(cy * across + cx);
      const size_t at = value(offs, k), len = value(counts, k);
      const long long rows = tiled ? chh : (chh < h - cy * chh ? chh : h - cy * chh);
      const size_t want = (size_t)rows * row_bytes;
      // (what `len` bytes can expand to: nothing / PackBits 64 : 1 / Deflate 1032 : 1 / LZW's longest strings)
      const double most = compression == 1 ? 1.0 : compression == 32773 ? 64.0 : compression == 5 ? 4096.0 : 1032.0;
      if (!f.has(at, len) || (double)want > most * (double)len + 64.0)
        return false;
      chunk.resize(want);
      bool ok;
      switch (compression) {
      case 1:
        ok = len >= want;
        if (ok)
          std::memcpy(chunk.data(), f.p + at, want);
        break;
      case 5: ok = tiff_lzw(f.p + at, len, chunk.data(), want); break;
      case 32773: ok = tiff_unpack_bits(f.p + at, len, chunk.data(), want); break;
      default: {
        uLongf got = (uLongf)want;
        const int rc = uncompress(chunk.data(), &got, f.p + at, (uLong)len);
        ok = (rc == Z_OK || rc == Z_BUF_ERROR) && got == want;
        break;
      }
      }
      if (!ok)
        return false;
      for (long long r = 0; r < rows; ++r) {
        const long long y = cy * chh + r;
        if (y >= h)
          break;
        uint8_t *row = chunk.data() + (size_t)r * row_bytes;
        if (predictor == 2) { // horizontal differencing, sample by sample, in the file's byte order
          if (bps == 8) {
            for (size_t i = px_bytes; i < row_bytes; ++i)
              row[i] = (uint8_t)(row[i] + row[i - px_bytes]);
          } else {
            for (size_t i = px_bytes; i + 1 < row_bytes; i += 2) {
              const size_t j = i - px_bytes;
              const uint32_t a = le ? (uint32_t)row[i] | (uint32_t)row[i + 1] << 8 : (uint32_t)row[i] << 8 | row[i + 1];
              const uint32_t b = le ? (uint32_t)row[j] | (uint32_t)row[j + 1] << 8 : (uint32_t)row[j] << 8 | row[j + 1];
              const uint32_t v = (a + b) & 0xffffu;
              row[i + (le ? 0 : 1)] = (uint8_t)v;
              row[i + (le ? 1 : 0)] = (uint8_t)(v >> 8);
            }
          }
        }
        const size_t hi = bps == 16 && le ? 1 : 0; // where a sample's high byte sits
        uint8_t *dst = out.px + (size_t)y * (size_t)w;
        for (long long x = 0; x < cw && cx * cw + x < w; ++x) {
          const uint8_t *p = row + (size_t)x * px_bytes + hi;
          uint8_t g;
          if (photometric == 3)
            g = pal_grey[p[0]];
          else if (photometric == 2)
            g = grey_opencv(p[0], p[sample_bytes], p[2 * sample_bytes]);
          else
            g = photometric == 0 ? (uint8_t)(255 - p[0]) : p[0];
          dst[cx * cw + x] = g;
        }
      }
    }
  return true;
}

// ---- PNM ---------------------------------------------------------------------------------------------------------
bool decode_pnm(const Bytes &f, Image &out) {
  const int kind = f.p[1] - '0';
  size_t at = 2;
  auto token = [&](long &v) { // whitespace / comment separated decimal; consumes the one whitespace after it
    for (;;) {
      while (at < f.n && (f.p[at] == ' ' || f.p[at] == '\t' || f.p[at] == '\n' || f.p[at] == '\r'))
        ++at;
      if (at < f.n && f.p[at] == '#') {
        while (at < f.n && f.p[at] != '\n')
          ++at;
        continue;
      }
      break;
    }
    if (at >= f.n || f.p[at] < '0' || f.p[at] > '9')
      return false;
    long acc = 0;
    while (at < f.n && f.p[at] >= '0' && f.p[at] <= '9') {
      acc = acc * 10 + (f.p[at++] - '0');
      if (acc > 1 << 30)
        return false;
    }
    if (at < f.n)
      ++at;
    v = acc;
    return true;
  };
  long w = 0, h = 0, maxval = 1;
  if (!token(w) || !token(h) || ((kind != 1 && kind != 4) && !token(maxval)) || maxval < 1 || maxval > 65535 || !out.alloc(w, h))
    return false;
  const size_t n = (size_t)w * (size_t)h, ch = (kind == 3 || kind == 6) ? 3 : 1;
  const bool wide = maxval > 255;
  auto to8 = [&](long v) { return (unsigned)(wide ? v >> 8 : v) & 255u; };
  if (kind == 1) { // ascii bitmap: digits need no separator
    for (size_t i = 0; i < n; ++i) {
      while (at < f.n && f.p[at] != '0' && f.p[at] != '1') {
        if (f.p[at] == '#')
          while (at < f.n && f.p[at] != '\n')
            ++at;
        else
          ++at;
      }
      if (at >= f.n)
        return false;
      out.px[i] = f.p[at++] == '1' ? 0 : 255;
    }
  } else if (kind == 2 || kind == 3) {
    for (size_t i = 0; i < n; ++i) {
      long v[3] = {0, 0, 0};
      for (size_t c = 0; c < ch; ++c)
        if (!token(v[c]) || v[c] > maxval)
          return false;
      out.px[i] = ch == 1 ? (uint8_t)to8(v[0]) : grey_opencv(to8(v[0]), to8(v[1]), to8(v[2]));
    }
  } else if (kind == 4) {
    const size_t stride = ((size_t)w + 7) / 8;
    if (!f.has(at, stride * (size_t)h))
      return false;
    for (size_t y = 0; y < (size_t)h; ++y)
      for (size_t x = 0; x < (size_t)w; ++x)
        out.px[y * (size_t)w + x] = ((f.p[at + y * stride + (x >> 3)] >> (7 - (x & 7))) & 1) ? 0 : 255;
  } else {
    const size_t bytes = wide ? 2 : 1;
    if (!f.has(at, n * ch * bytes))
      return false;
    const uint8_t *s = f.p + at;
    for (size_t i = 0; i < n; ++i) // (16-bit samples are big endian: the high byte is the first)
      out.px[i] = ch == 1 ? s[i * bytes] : grey_opencv(s[3 * i * bytes], s[(3 * i + 1) * bytes], s[(3 * i + 2) * bytes]);
  }
  return true;
}

} // namespace

extern "C" {

int lk_decode_image(const uint8_t *bytes, size_t n, uint8_t **pixels, int *rows, int *cols) {
  if (!bytes || !pixels || !rows || !cols)
    return LK_ERROR_BAD_DOMAIN;
  *pixels = nullptr;
  *rows = *cols = 0;
  const Bytes f{bytes, n};
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  Image img;
  bool ok = false;
  try {
    if (n >= 8 && !std::memcmp(bytes, png_sig, 8))
      ok = decode_png(f, img);
    else if (n >= 2 && bytes[0] == 'B' && bytes[1] == 'M')
      ok = decode_bmp(f, img);
    else if (n >= 3 && bytes[0] == 'P' && bytes[1] >= '1' && bytes[1] <= '6')
      ok = decode_pnm(f, img);
    else if (n >= 4 && ((bytes[0] == 'I' && bytes[1] == 'I') || (bytes[0] == 'M' && bytes[1] == 'M')))
      ok = decode_tiff(f, img);
  } catch (...) { // (out of memory in a scratch vector)
    ok = false;
  }
  if (!ok) {
    std::free(img.px);
    return LK_ERROR_BAD_DOMAIN;
  }
  *pixels = img.px;
  *rows = img.rows;
  *cols = img.cols;
  return LK_ERROR_NONE;
}

int lk_load_image(const char *path, uint8_t **pixels, int *rows, int *cols) {
  if (!path || !pixels || !rows || !cols)
    return LK_ERROR_BAD_DOMAIN;
  *pixels = nullptr;
  FILE *fp = std::fopen(path, "rb");
  if (!fp)
    return LK_ERROR_BAD_DOMAIN;
  std::vector<uint8_t> bytes;
  bool ok = std::fseek(fp, 0, SEEK_END) == 0;
  const long size = ok ? std::ftell(fp) : -1;
  ok = ok && size > 0 && std::fseek(fp, 0, SEEK_SET) == 0;
  try {
    if (ok) {
      bytes.resize((size_t)size);
      ok = std::fread(bytes.data(), 1, bytes.size(), fp) == bytes.size();
    }
  } catch (...) {
    ok = false;
  }
  std::fclose(fp);
  return ok ? lk_decode_image(bytes.data(), bytes.size(), pixels, rows, cols) : LK_ERROR_BAD_DOMAIN;
}

} // extern "C"
