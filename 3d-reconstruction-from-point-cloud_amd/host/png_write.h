// png_write.h -- PNG writer for the baked texture (the reference calls cv::imwrite("texture.png", padded),
// src/pointsTransfer.cpp:613-615; OpenCV is not available, zlib is enough).
//
// 8-bit RGBA, no interlace, filter type 0 on every row.  The 8192 x 8192 atlas is 268 MB raw, so the deflate runs in parallel:
// the rows are cut into bands, every band is compressed on its own thread as a raw deflate stream ended with a sync flush (byte
// aligned, not final), the pieces are concatenated behind one zlib header and closed with the Adler-32 of the whole image
// (adler32_combine) -- the result is one ordinary zlib stream, readable by any PNG decoder.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace png {

inline void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back((x >> 16) & 255); v.push_back((x >> 8) & 255); v.push_back(x & 255); }
inline bool write_chunk(FILE* f, const char type[4], const unsigned char* data, size_t len) {
  unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len, (unsigned char)type[0],
                          (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
  uLong crc = crc32(0L, hdr + 4, 4);
  if (len) crc = crc32_z(crc, data, len);
  const unsigned char tail[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
  return std::fwrite(hdr, 1, 8, f) == 8 && (!len || std::fwrite(data, 1, len, f) == len) && std::fwrite(tail, 1, 4, f) == 4;
}

// bgra: height x width x 4 bytes in OpenCV's channel order (B, G, R, A), as the bake produces them.  level: zlib level (1 = what
// OpenCV's imwrite uses by default).  threads 0 = one per hardware thread (at most 64, at least 64 rows per band).
inline bool write_bgra(const std::string& path, const uint8_t* bgra, int width, int height, int level = 1, int threads = 0) {
  if (width < 1 || height < 1) return false;
  int nb = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nb = std::max(1, std::min({nb, 64, (height + 63) / 64}));
  struct Band { std::vector<unsigned char> z; uLong adler = 1; size_t raw = 0; bool ok = false; };
  std::vector<Band> bands((size_t)nb);
  const size_t stride = (size_t)width * 4 + 1;
  auto work = [&](int b) {
    const int r0 = (int)((long long)height * b / nb), r1 = (int)((long long)height * (b + 1) / nb);
    Band& B = bands[(size_t)b];
    std::vector<unsigned char> raw((size_t)(r1 - r0) * stride);
    for (int r = r0; r < r1; ++r) {
      unsigned char* o = raw.data() + (size_t)(r - r0) * stride;
      const uint8_t* s = bgra + (size_t)r * (size_t)width * 4;
      *o++ = 0;                                                   // filter type 0 (None)
      for (int x = 0; x < width; ++x, s += 4, o += 4) { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = s[3]; }      // BGRA -> RGBA
    }
    B.raw = raw.size();
    B.adler = adler32_z(1L, raw.data(), raw.size());
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;      // raw deflate: header and checksum are written once, below
    B.z.resize(deflateBound(&zs, (uLong)raw.size()) + 64);
    size_t in_off = 0, out_off = 0;
    const bool last = b == nb - 1;
    int rc = Z_OK;
    do {                                                          // (z_stream counts in 32-bit uInt: feed it in pieces)
      const size_t in_now = std::min<size_t>(raw.size() - in_off, 1u << 30), out_now = std::min<size_t>(B.z.size() - out_off, 1u << 30);
      zs.next_in = raw.data() + in_off; zs.avail_in = (uInt)in_now;
      zs.next_out = B.z.data() + out_off; zs.avail_out = (uInt)out_now;
      const bool final_piece = in_off + in_now == raw.size();
      rc = deflate(&zs, final_piece ? (last ? Z_FINISH : Z_SYNC_FLUSH) : Z_NO_FLUSH);
      in_off += in_now - zs.avail_in; out_off += out_now - zs.avail_out;
      if (rc == Z_STREAM_ERROR || rc == Z_BUF_ERROR) break;
      if (final_piece && zs.avail_in == 0 && (last ? rc == Z_STREAM_END : zs.avail_out != 0)) { B.ok = true; break; }
    } while (true);
    deflateEnd(&zs);
    B.z.resize(out_off);
  };
  std::vector<std::thread> th;
  for (int b = 1; b < nb; ++b) th.emplace_back(work, b);
  work(0);
  for (auto& t : th) t.join();
  for (const Band& B : bands) if (!B.ok) return false;

  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  bool ok = std::fwrite(sig, 1, 8, f) == 8;
  std::vector<unsigned char> ihdr;
  put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
  ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);      // 8 bits, RGBA, deflate, filter 0, no interlace
  ok = ok && write_chunk(f, "IHDR", ihdr.data(), ihdr.size());
  // one zlib stream cut into IDAT chunks: header, the bands, the Adler-32 of everything
  uLong adler = 1;
  bool first = true;
  for (const Band& B : bands) { adler = first ? B.adler : adler32_combine(adler, B.adler, (z_off_t)B.raw); first = false; }
  const unsigned char zhdr[2] = {0x78, 0x01};
  ok = ok && write_chunk(f, "IDAT", zhdr, 2);
  for (const Band& B : bands)
    for (size_t off = 0; off < B.z.size() && ok; off += (size_t)1 << 26)
      ok = write_chunk(f, "IDAT", B.z.data() + off, std::min<size_t>(B.z.size() - off, (size_t)1 << 26));
  const unsigned char zend[4] = {(unsigned char)(adler >> 24), (unsigned char)(adler >> 16), (unsigned char)(adler >> 8), (unsigned char)adler};
  ok = ok && write_chunk(f, "IDAT", zend, 4);
  ok = ok && write_chunk(f, "IEND", nullptr, 0);
  return (std::fclose(f) == 0) && ok;
}

}  // namespace png
