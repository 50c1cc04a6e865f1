// ply_bench -- ingest rate of the two PLY readers on a synthetic ASCII cloud (SURVEY.md 8 f2 measurement).
//   ply_bench <scratch-file> [points=5000000] [threads ...]
// Writes `points` records in the reference's cloud grammar (x y z nx ny nz r g b), then times ply::read_cloud (serial,
// strtod) and ply::read_cloud_fast at the given thread counts (default: 1, 8, all); checks that every run returns the
// same records.  Host only -- no GPU involved.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ply_fast.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: ply_bench <scratch-file> [points] [threads ...]\n"); return 2; }
  const std::string path = argv[1];
  const long n = argc > 2 ? std::atol(argv[2]) : 5000000;
  std::vector<int> threads;
  for (int i = 3; i < argc; ++i) threads.push_back(std::atoi(argv[i]));
  if (threads.empty()) threads = {1, 8, 0};
  {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::printf("cannot write %s\n", path.c_str()); return 1; }
    std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %ld\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                    "property float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (long i = 0; i < n; ++i) {
      const uint64_t a = next(), b = next(), c = next();
      std::fprintf(f, "%.6f %.6f %.6f %.4f %.4f %.4f %u %u %u\n", (double)(a >> 40) / 16777216.0, (double)(b >> 40) / 16777216.0,
                   (double)(c >> 40) / 16777216.0, (double)((a >> 8) & 0xFFFF) / 65535.0, (double)((b >> 8) & 0xFFFF) / 65535.0,
                   (double)((c >> 8) & 0xFFFF) / 65535.0, (unsigned)(a & 255), (unsigned)(b & 255), (unsigned)(c & 255));
    }
    std::fclose(f);
  }
  struct stat st;
  stat(path.c_str(), &st);
  const double mb = (double)st.st_size / 1e6;
  std::vector<Point> want;
  long declared = 0;
  double t = now();
  ply::read_cloud(path, want, declared);
  const double ts = now() - t;
  std::printf("file %.1f MB, %ld points\nserial  reader (ply_io.h, strtod):          %8.3f s  %8.1f MB/s  %7.2f M points/s\n", mb, (long)want.size(), ts, mb / ts,
              (double)want.size() / ts / 1e6);
  int bad = 0;
  for (int th : threads) {
    ply::RecordBuffer got;
    long d2 = 0;
    t = now();
    ply::read_cloud_fast(path, got, d2, th);
    const double tf = now() - t;
    bool same = got.size() == want.size();
    for (size_t i = 0; same && i < want.size(); ++i)
      same = got[i].ver[0] == want[i].ver[0] && got[i].ver[1] == want[i].ver[1] && got[i].ver[2] == want[i].ver[2] && got[i].normal[0] == want[i].normal[0] &&
             got[i].normal[1] == want[i].normal[1] && got[i].normal[2] == want[i].normal[2] && got[i].color[0] == want[i].color[0] &&
             got[i].color[1] == want[i].color[1] && got[i].color[2] == want[i].color[2];
    std::printf("parallel reader (ply_fast.h), %3d threads:     %8.3f s  %8.1f MB/s  %7.2f M points/s  x%.1f  %s\n",
                th ? th : (int)std::thread::hardware_concurrency(), tf, mb / tf, (double)got.size() / tf / 1e6, ts / tf, same ? "identical" : "DIFFERENT");
    bad += !same;
  }
  std::remove(path.c_str());
  return bad ? 1 : 0;
}
