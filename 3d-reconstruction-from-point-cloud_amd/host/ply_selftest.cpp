// Self-test of the PLY readers (host/ply_io.h, and the parallel host/ply_fast.h against it) on the grammar the reference accepts (reference
// src/pointsTransfer.cpp:134-253, :266-455) and on malformed input.  Built with -fsanitize=address,undefined by the CPU
// test-suite; exit code 0 = all good.
#include <cstdio>
#include <fstream>
#include <random>
#include <string>

#include "ply_fast.h"
#include "ply_io.h"

static int fails = 0;
#define CHECK(cond)                                                                       \
  do {                                                                                    \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++fails; } \
  } while (0)

static std::string write(const std::string& dir, const char* name, const std::string& body) {
  const std::string p = dir + "/" + name;
  std::ofstream(p, std::ios::binary) << body;
  return p;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  const std::string hdr_c = "ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nend_header\n";
  std::vector<Point> pts;
  long declared = 0;
  // well-formed cloud, tabs and CRLF tolerated, last record without a trailing newline kept
  CHECK(ply::read_cloud(write(dir, "a.ply", hdr_c + "1 2 3 0 0 1 10 20 30\r\n4\t5 6 0 1 0 255 0 7"), pts, declared));
  CHECK(declared == 2 && pts.size() == 2 && pts[1].x() == 4 && pts[1].z() == 6 && pts[1].ny() == 1 && pts[1].r() == 255 && pts[1].b() == 7);
  // colour given as a float is truncated like atof -> int (reference :236-246)
  CHECK(ply::read_cloud(write(dir, "b.ply", hdr_c + "0 0 0 0 0 1 12.9 0 0\n1 1 1 0 0 1 1 2 3\n"), pts, declared) && pts[0].r() == 12);
  // fewer records than declared: keeps what is there, no over-read
  CHECK(ply::read_cloud(write(dir, "c.ply", hdr_c + "1 2 3 0 0 1 1 2"), pts, declared) && declared == 2 && pts.empty());
  // no header at all / empty file / garbage
  CHECK(ply::read_cloud(write(dir, "d.ply", ""), pts, declared) && pts.empty());
  CHECK(ply::read_cloud(write(dir, "e.ply", "vertex"), pts, declared) && pts.empty());
  CHECK(ply::read_cloud(write(dir, "f.ply", "element vertex 3\nend_header\nx y z\n"), pts, declared) && pts.empty());
  CHECK(!ply::read_cloud(dir + "/does-not-exist.ply", pts, declared));
  // mesh: 11 numbers per vertex in file order x y z nx ny nz u v r g b, faces `n i j k`
  ply::Mesh m;
  const std::string hdr_m = "ply\nformat ascii 1.0\nelement vertex 3\nelement face 1\nend_header\n";
  CHECK(ply::read_mesh(write(dir, "m.ply", hdr_m + "0 0 0 0 0 1 0.1 0.2 1 2 3\n1 0 0 0 0 1 0.3 0.4 4 5 6\n0 1 0 0 0 1 0.5 0.6 7 8 9\n3 0 1 2\n"), m));
  CHECK(m.vertices.size() == 3 && m.faces.size() == 3 && m.vertices[1].u() == 0.3 && m.vertices[1].v() == 0.4 && m.vertices[2].g() == 8 && m.faces[2] == 2);
  CHECK(ply::read_mesh(write(dir, "n.ply", hdr_m + "0 0 0 0 0 1 0.1 0.2 1 2 3\n"), m) && m.vertices.size() == 1 && m.faces.empty());   // truncated
  CHECK(ply::read_mesh(write(dir, "o.ply", "element vertex -5\nelement face 99999999\nend_header\n"), m) && m.vertices.empty() && m.faces.empty());
  // ---- the parallel reader must agree with the serial one: every case above, at several forced range counts ----
  auto eq = [](double a, double b) { return a == b || (a != a && b != b); };          // "nan" tokens parse to NaN in both
  auto same_point = [&](const Point& a, const Point& b, bool uv) {
    return eq(a.ver[0], b.ver[0]) && eq(a.ver[1], b.ver[1]) && eq(a.ver[2], b.ver[2]) && eq(a.normal[0], b.normal[0]) &&
           eq(a.normal[1], b.normal[1]) && eq(a.normal[2], b.normal[2]) && a.color[0] == b.color[0] && a.color[1] == b.color[1] &&
           a.color[2] == b.color[2] && (!uv || (eq(a.U, b.U) && eq(a.V, b.V)));
  };
  auto check_cloud = [&](const std::string& path) {
    std::vector<Point> want;
    long dw = 0;
    const bool okw = ply::read_cloud(path, want, dw);
    for (int th : {1, -2, -3, -7, -64, 0}) {
      ply::RecordBuffer got;
      long dg = 0;
      const bool okg = ply::read_cloud_fast(path, got, dg, th);
      CHECK(okg == okw);
      if (!okw) continue;
      CHECK(dg == dw && got.size() == want.size());
      for (size_t i = 0; i < want.size() && i < got.size(); ++i)
        if (!same_point(got[i], want[i], false)) { CHECK(!"cloud record differs"); break; }
    }
  };
  auto check_mesh = [&](const std::string& path) {
    ply::Mesh want;
    const bool okw = ply::read_mesh(path, want);
    for (int th : {1, -2, -3, -7, -64, 0}) {
      ply::FastMesh got;
      const bool okg = ply::read_mesh_fast(path, got, th);
      CHECK(okg == okw);
      if (!okw) continue;
      CHECK(got.vertex_count == want.vertex_count && got.face_count == want.face_count);
      CHECK(got.vertices.size() == want.vertices.size() && got.faces == want.faces);
      for (size_t i = 0; i < want.vertices.size() && i < got.vertices.size(); ++i)
        if (!same_point(got.vertices[i], want.vertices[i], true)) { CHECK(!"mesh vertex differs"); break; }
    }
  };
  for (const char* f : {"a.ply", "b.ply", "c.ply", "d.ply", "e.ply", "f.ply", "does-not-exist.ply"}) check_cloud(dir + "/" + f);
  for (const char* f : {"m.ply", "n.ply", "o.ply", "a.ply", "d.ply"}) check_mesh(dir + "/" + f);
  check_mesh(write(dir, "p.ply", "element vertex 0\nelement face 2\nend_header\n3 0 1 2 3 4 5 6\n3 9"));     // faces without vertices
  check_cloud(write(dir, "q.ply", hdr_c + "+1 0x10 1e 12abc nan inf -inf .5 5.\n1e400 -1e-400 " + std::string(200, '7') + " x y z 1 2 3"));   // odd tokens
  {   // a larger random file: ragged whitespace, records spanning lines, junk tokens, more and fewer records than declared
    std::mt19937_64 rng(7);
    auto body = [&](int records, int per) {
      std::string b;
      char num[64];
      for (int i = 0; i < records * per; ++i) {
        const int kind = (int)(rng() % 16);
        if (kind == 0) std::snprintf(num, sizeof num, "%d", (int)(rng() % 512) - 256);
        else if (kind == 1) std::snprintf(num, sizeof num, "%.17g", (double)(rng() % 1000003) * 1e-9);
        else if (kind == 2) std::snprintf(num, sizeof num, "junk%d", (int)(rng() % 9));
        else std::snprintf(num, sizeof num, "%.9g", (double)(int64_t)(rng() % 2000001 - 1000000) * 1e-4);
        b += num;
        static const char* seps[] = {" ", " ", " ", "\n", "\t", "  ", " \r\n", "\n\n"};
        b += seps[rng() % 8];
      }
      return b;
    };
    check_cloud(write(dir, "r.ply", "ply\nelement vertex 20000\nend_header\n" + body(20000, 9)));
    check_cloud(write(dir, "s.ply", "ply\nelement vertex 20000\nend_header\n" + body(19999, 9) + "1 2 3"));    // one short
    check_cloud(write(dir, "t.ply", "ply\nelement vertex 500\nend_header\n" + body(20000, 9)));                 // more than declared
    check_mesh(write(dir, "u.ply", "ply\nelement vertex 7000\nelement face 9000\nend_header\n" + body(7000, 11) + body(9000, 4)));
    check_mesh(write(dir, "v.ply", "ply\nelement vertex 7000\nelement face 9000\nend_header\n" + body(7000, 11) + body(100, 4) + "3 1"));
  }
  std::printf(fails ? "ply selftest: %d failure(s)\n" : "ply selftest ok\n", fails);
  return fails ? 1 : 0;
}
