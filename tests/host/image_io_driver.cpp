// image_io_driver.cpp - csrc/lk_image_io.cpp under ASan + UBSan (tests/test_image_io.py).
//   image_io_driver <dir>   every <name> in <dir> that has a <name>.expect beside it ({rows, cols} + pixels):
//     1. the file decodes to exactly that;
//     2. every prefix of it (all short ones, then strided) and 400 copies with 1-4 bytes changed are decoded - PNG chunk CRCs
//        repaired first, so that the damage gets past the CRC check into the inflater, the filters and the pixel loops.
//        The outcome of those is free (an error, or some image); what is checked is that the sanitizers stay silent and that
//        an error leaves *pixels NULL.
#include "lk_tracker.h"

#include <dirent.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::vector<uint8_t> slurp(const std::string &path) {
  std::vector<uint8_t> v;
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f)
    return v;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
    v.insert(v.end(), buf, buf + n);
  std::fclose(f);
  return v;
}

static void repair_png_crcs(std::vector<uint8_t> &v) {
  if (v.size() < 8 || std::memcmp(v.data(), "\x89PNG\r\n\x1a\n", 8) != 0)
    return;
  size_t at = 8;
  while (at + 12 <= v.size()) {
    const size_t len = (size_t)v[at] << 24 | (size_t)v[at + 1] << 16 | (size_t)v[at + 2] << 8 | v[at + 3];
    if (len > v.size() || at + 12 + len > v.size())
      return;
    const uint32_t c = (uint32_t)crc32(crc32(0L, Z_NULL, 0), v.data() + at + 4, (uInt)(len + 4));
    v[at + 8 + len] = (uint8_t)(c >> 24), v[at + 9 + len] = (uint8_t)(c >> 16), v[at + 10 + len] = (uint8_t)(c >> 8), v[at + 11 + len] = (uint8_t)c;
    at += 12 + len;
  }
}

static bool try_decode(const std::vector<uint8_t> &v, long &decoded) {
  uint8_t *px = (uint8_t *)(uintptr_t)1;
  int rows = -1, cols = -1;
  // (a copy of exactly v.size() bytes on the heap: a read one past the end is a read outside the allocation)
  uint8_t *exact = (uint8_t *)std::malloc(v.size() ? v.size() : 1);
  if (!v.empty())
    std::memcpy(exact, v.data(), v.size());
  const int rc = lk_decode_image(exact, v.size(), &px, &rows, &cols);
  std::free(exact);
  if (rc != LK_ERROR_NONE) {
    if (px != nullptr) {
      std::printf("an error left *pixels set\n");
      return false;
    }
    return true;
  }
  if (!px || rows <= 0 || cols <= 0) {
    std::printf("success without an image\n");
    return false;
  }
  unsigned sum = 0; // touch every pixel of the result
  for (size_t i = 0; i < (size_t)rows * (size_t)cols; ++i)
    sum += px[i];
  decoded += 1 + (long)(sum & 0);
  std::free(px); // (malloc'ed: lk_free_image, which lives with the tracker, is free())
  return true;
}

int main(int argc, char **argv) {
  if (argc < 2)
    return 2;
  const std::string dir = argv[1];
  DIR *d = opendir(dir.c_str());
  if (!d)
    return 2;
  std::vector<std::string> names;
  while (dirent *e = readdir(d)) {
    const std::string n = e->d_name;
    if (n.size() > 7 && n.substr(n.size() - 7) == ".expect")
      names.push_back(n.substr(0, n.size() - 7));
  }
  closedir(d);
  long files = 0, damaged = 0, decoded = 0;
  uint32_t lcg = 12345u;
  auto rnd = [&]() { return (lcg = lcg * 1664525u + 1013904223u) >> 8; };
  for (const std::string &name : names) {
    const std::vector<uint8_t> v = slurp(dir + "/" + name), want = slurp(dir + "/" + name + ".expect");
    uint8_t *px = nullptr;
    int rows = 0, cols = 0;
    if (lk_load_image((dir + "/" + name).c_str(), &px, &rows, &cols) != LK_ERROR_NONE) {
      std::printf("%s does not decode\n", name.c_str());
      return 1;
    }
    int32_t shape[2];
    std::memcpy(shape, want.data(), 8);
    if (rows != shape[0] || cols != shape[1] || want.size() != 8 + (size_t)rows * (size_t)cols || std::memcmp(px, want.data() + 8, (size_t)rows * (size_t)cols) != 0) {
      std::printf("%s decodes to something else\n", name.c_str());
      return 1;
    }
    std::free(px);
    ++files;
    const size_t step = v.size() > 600 ? v.size() / 150 : 1;
    for (size_t n = 0; n < v.size(); n += (n < 200 ? 1 : step)) {
      std::vector<uint8_t> t(v.begin(), v.begin() + (long)n);
      if (!try_decode(t, decoded))
        return 1;
      ++damaged;
    }
    for (int k = 0; k < 400; ++k) {
      std::vector<uint8_t> t = v;
      const int flips = 1 + (int)(rnd() % 4);
      for (int i = 0; i < flips; ++i) {
        // half of the damage goes to the first 64 bytes (headers), the rest anywhere
        const size_t at = (rnd() & 1) ? rnd() % (t.size() < 64 ? t.size() : 64) : rnd() % t.size();
        t[at] = (rnd() & 1) ? (uint8_t)rnd() : (uint8_t)(t[at] ^ (1u << (rnd() % 8)));
      }
      if (k & 1)
        repair_png_crcs(t);
      if (!try_decode(t, decoded))
        return 1;
      ++damaged;
    }
  }
  std::printf("image_io_driver ok: %ld files, %ld damaged copies (%ld of them still decode)\n", files, damaged, decoded);
  return files > 0 ? 0 : 1;
}
