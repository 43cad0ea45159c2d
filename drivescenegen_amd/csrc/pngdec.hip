// Host-side batch PNG decoder for the training feeder -- no device code in this file.
//
// What it replaces: the reference decodes one file at a time on the training thread,
//   DriveSceneGen/utils/datasets/dataset.py:43-45   sample = ToTensor(Image.open(f))         (PIL, single thread)
//   DriveSceneGen/scripts/train.py:35               DataLoader(..., num_workers=0)
// 145-326 images/s for 512x512 RGB files -- below the 249 (fp32) / 699 (bf16) images/s one MI355X training step consumes, and
// a Python thread pool over PIL does not scale (121 -> 127 -> 182 -> 346 images/s at 1 / 2 / 4 / 7 threads on 8 cores: the chunk
// loop of ImageFile.load and the array export hold the GIL).  Here a batch of files is decoded by `threads` native threads
// straight into the rows of the caller's PINNED staging buffer [n][h][w][c] (ctypes releases the GIL for the whole call):
// read the file, walk the chunks, inflate the concatenated IDAT stream with zlib, undo the five scan-line filters in place.
//
// Scope: 8-bit, non-interlaced, colour types 0 (grey), 2 (RGB), 4 (grey + alpha), 6 (RGBA) -- what PIL / matplotlib write for
// the scene rasters.  Anything else (16-bit, palette, Adam7, a damaged file) gets a per-file status and the Python side reads
// THAT file with PIL, so the result is always the array `np.asarray(Image.open(f))` gives (tests/test_pngdec_cpu.py: bit-exact
// on every mode PIL writes).  CRCs of the chunks are not checked (zlib's adler32 over the pixel stream is).
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "dsg_common.h"

namespace dsg {
namespace png {

enum : int32_t { OK = 0, ERR_OPEN = 1, ERR_NOT_PNG = 2, ERR_UNSUPPORTED = 3, ERR_SHAPE = 4, ERR_CORRUPT = 5 };

static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Header {
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0, channels = 0;
};

static bool read_file(const char* path, std::vector<uint8_t>& buf) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) {
    fclose(f);
    return false;
  }
  buf.resize((size_t)n);
  const size_t got = n ? fread(buf.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)n;
}

static int32_t parse_header(const std::vector<uint8_t>& buf, Header& hd) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (buf.size() < 8 + 25 || memcmp(buf.data(), sig, 8) != 0) return ERR_NOT_PNG;
  const uint8_t* p = buf.data() + 8;
  if (be32(p) != 13 || memcmp(p + 4, "IHDR", 4) != 0) return ERR_NOT_PNG;
  hd.w = be32(p + 8);
  hd.h = be32(p + 12);
  hd.depth = p[16];
  hd.ctype = p[17];
  hd.interlace = p[20];
  if (p[18] != 0 || p[19] != 0) return ERR_UNSUPPORTED;
  if (hd.depth != 8 || hd.interlace != 0) return ERR_UNSUPPORTED;
  switch (hd.ctype) {
    case 0: hd.channels = 1; break;
    case 2: hd.channels = 3; break;
    case 4: hd.channels = 2; break;
    case 6: hd.channels = 4; break;
    default: return ERR_UNSUPPORTED;   // 3 = palette: PIL hands out indices; left to PIL
  }
  if (hd.w == 0 || hd.h == 0 || hd.w > (1u << 16) || hd.h > (1u << 16)) return ERR_UNSUPPORTED;
  return OK;
}

static inline uint8_t paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// raw: h rows of (1 filter byte + stride bytes); out: h rows of stride bytes
static int32_t unfilter(const uint8_t* raw, uint8_t* out, uint32_t h, size_t stride, int bpp) {
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t ft = raw[y * (stride + 1)];
    const uint8_t* in = raw + y * (stride + 1) + 1;
    uint8_t* cur = out + y * stride;
    const uint8_t* up = y ? cur - stride : nullptr;
    switch (ft) {
      case 0: memcpy(cur, in, stride); break;
      case 1:
        for (size_t i = 0; i < (size_t)bpp && i < stride; ++i) cur[i] = in[i];
        for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(in[i] + cur[i - bpp]);
        break;
      case 2:
        if (up) for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(in[i] + up[i]);
        else memcpy(cur, in, stride);
        break;
      case 3:
        for (size_t i = 0; i < stride; ++i) {
          const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
          cur[i] = (uint8_t)(in[i] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < stride; ++i) {
          const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
          cur[i] = (uint8_t)(in[i] + paeth(a, b, c));
        }
        break;
      default: return ERR_CORRUPT;
    }
  }
  return OK;
}

static int32_t decode_one(const char* path, uint8_t* out, int32_t h, int32_t w, int32_t c, std::vector<uint8_t>& file,
                          std::vector<uint8_t>& idat, std::vector<uint8_t>& raw) {
  if (!read_file(path, file)) return ERR_OPEN;
  Header hd;
  const int32_t st = parse_header(file, hd);
  if (st != OK) return st;
  if ((int32_t)hd.h != h || (int32_t)hd.w != w || hd.channels != c) return ERR_SHAPE;
  idat.clear();
  size_t pos = 8;
  bool end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(file.data() + pos);
    const uint8_t* type = file.data() + pos + 4;
    if (pos + 12 + (size_t)len > file.size()) return ERR_CORRUPT;
    if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), type + 4, type + 4 + len);
    else if (memcmp(type, "IEND", 4) == 0) end = true;
    pos += 12 + (size_t)len;
  }
  if (idat.empty()) return ERR_CORRUPT;
  const size_t stride = (size_t)w * c;
  raw.resize((size_t)h * (stride + 1));
  uLongf got = (uLongf)raw.size();
  if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return ERR_CORRUPT;
  return unfilter(raw.data(), out, (uint32_t)h, stride, c);
}

// Per-thread scratch (file image, IDAT stream, filtered scan lines: ~2x the decoded image) is kept across calls on a free
// list: a fresh 800-KB vector per thread and call costs its page faults, which serialise on the process's mm lock -- measured
// here: 16-file calls did not scale with threads at all (170 images/s at 1, 2, 4, 8 threads) until the buffers were reused.
struct Scratch {
  std::vector<uint8_t> file, idat, raw;
};
static std::mutex g_scratch_mu;
static std::vector<std::unique_ptr<Scratch>> g_scratch_free;

static std::unique_ptr<Scratch> take_scratch() {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_free.empty()) return std::unique_ptr<Scratch>(new Scratch());
  std::unique_ptr<Scratch> s = std::move(g_scratch_free.back());
  g_scratch_free.pop_back();
  return s;
}
static void give_scratch(std::unique_ptr<Scratch> s) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_free.size() < 64) g_scratch_free.push_back(std::move(s));
}

}  // namespace png
}  // namespace dsg

DSG_API int dsg_png_probe(const char* path, int32_t* h, int32_t* w, int32_t* c) {
  DSG_CHECK_ARG(path && h && w && c, "dsg_png_probe: NULL pointer");
  FILE* f = fopen(path, "rb");
  if (!f) return dsg::fail(DSG_ERR_INVALID_ARG, "dsg_png_probe: cannot open %s", path);
  std::vector<uint8_t> head(33);
  const size_t got = fread(head.data(), 1, head.size(), f);
  fclose(f);
  head.resize(got);
  dsg::png::Header hd;
  const int32_t st = dsg::png::parse_header(head, hd);
  if (st != dsg::png::OK) {
    *h = *w = *c = 0;
    return dsg::fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_png_probe: %s is not an 8-bit non-interlaced grey / RGB / RGBA PNG (code %d)",
                     path, (int)st);
  }
  *h = (int32_t)hd.h;
  *w = (int32_t)hd.w;
  *c = hd.channels;
  return DSG_OK;
}

DSG_API int dsg_png_decode_batch(const char* const* paths, int32_t n, uint8_t* out, int32_t h, int32_t w, int32_t c,
                                 int32_t threads, int32_t* status) {
  DSG_CHECK_ARG(paths && out && status, "dsg_png_decode_batch: NULL pointer");
  DSG_CHECK_ARG(n > 0 && h > 0 && w > 0 && c >= 1 && c <= 4, "dsg_png_decode_batch: bad dims");
  for (int32_t i = 0; i < n; ++i) DSG_CHECK_ARG(paths[i], "dsg_png_decode_batch: NULL path");
  const int nt = threads < 1 ? 1 : (threads > n ? n : threads);
  std::atomic<int32_t> next{0};
  const size_t per = (size_t)h * w * c;
  auto worker = [&]() {
    std::unique_ptr<dsg::png::Scratch> sc = dsg::png::take_scratch();
    for (;;) {
      const int32_t i = next.fetch_add(1);
      if (i >= n) break;
      status[i] = dsg::png::decode_one(paths[i], out + (size_t)i * per, h, w, c, sc->file, sc->idat, sc->raw);
    }
    dsg::png::give_scratch(std::move(sc));
  };
  if (nt == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 0; t < nt - 1; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  return DSG_OK;
}
