// Self-test of the PLY reader (host/ply_io.h) on the grammar the reference accepts (reference
// src/pointsTransfer.cpp:134-253, :266-455) and on malformed input.  Built with -fsanitize=address,undefined by the CPU
// test-suite; exit code 0 = all good.
#include <cstdio>
#include <fstream>
#include <string>

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
  std::printf(fails ? "ply selftest: %d failure(s)\n" : "ply selftest ok\n", fails);
  return fails ? 1 : 0;
}
