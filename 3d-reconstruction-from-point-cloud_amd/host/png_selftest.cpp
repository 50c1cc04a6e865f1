// png_selftest <out.png> <width> <height> <threads> -- writes a deterministic BGRA pattern through host/png_write.h
// (tests/test_boundary.py decodes the file with Python's zlib and compares; no GPU involved)
#include <cstdlib>
#include <vector>

#include "png_write.h"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), t = std::atoi(argv[4]);
  std::vector<uint8_t> img((size_t)w * h * 4);
  uint32_t s = 12345u;
  for (size_t i = 0; i < img.size(); ++i) {
    s = s * 1664525u + 1013904223u;
    img[i] = (i / 4 / (size_t)w) % 7 == 0 ? (uint8_t)(s >> 24) : (uint8_t)((i * 31) >> 3);      // noisy rows between smooth ones
  }
  return png::write_bgra(argv[1], img.data(), w, h, 1, t) ? 0 : 1;
}
