// Self-test of the PLY readers (host/ply_io.h, and the parallel host/ply_fast.h against it) on the grammar the reference accepts (reference
// src/pointsTransfer.cpp:134-253, :266-455) and on malformed input.  Built with -fsanitize=address,undefined by the CPU
// test-suite; exit code 0 = all good.
#include <array>
#include <cstdio>
#include <algorithm>
#include <fstream>
#include <mutex>
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
  {   // ---- planar (SoA) output and binary little-endian files: same values as the record readers, range callbacks cover every record once ----
    auto soa_equals = [&](const std::string& path, const std::vector<Point>& want, long want_declared, bool floats) {
      for (int th : {1, -3, -16, 0}) {
        ply::CloudSoA got;
        std::vector<void*> mem;
        std::vector<int> seen;
        std::mutex mu;
        long dg = 0;
        const bool ok = ply::read_cloud_soa(path, got, dg, [&](size_t bytes) { void* q = std::malloc(bytes ? bytes : 1); mem.push_back(q); return q; },
                                            [&](uint64_t n) { seen.assign((size_t)n, 0); return true; },
                                            [&](uint64_t first, uint64_t count) { std::lock_guard<std::mutex> lk(mu); for (uint64_t i = first; i < first + count; ++i) ++seen[(size_t)i]; }, th);
        CHECK(ok && dg == want_declared && got.n == want.size());
        for (size_t i = 0; i < want.size() && i < got.n; ++i) {
          const Point& w = want[i];
          auto f = [&](double v) { return floats ? (double)(float)v : v; };
          const bool same = eq(got.x[i], f(w.ver[0])) && eq(got.y[i], f(w.ver[1])) && eq(got.z[i], f(w.ver[2])) && eq(got.nrm[3 * i], (float)w.normal[0]) &&
                            eq(got.nrm[3 * i + 1], (float)w.normal[1]) && eq(got.nrm[3 * i + 2], (float)w.normal[2]) &&
                            got.rgb[3 * i] == std::min(std::max(w.color[0], 0), 255) && got.rgb[3 * i + 1] == std::min(std::max(w.color[1], 0), 255) &&
                            got.rgb[3 * i + 2] == std::min(std::max(w.color[2], 0), 255);
          if (!same) { std::printf("   planar mismatch in %s record %zu: got x %.9g y %.9g z %.9g n %.9g %.9g %.9g rgb %d %d %d\n", path.c_str(), i, got.x[i], got.y[i], got.z[i], got.nrm[3*i], got.nrm[3*i+1], got.nrm[3*i+2], got.rgb[3*i], got.rgb[3*i+1], got.rgb[3*i+2]); CHECK(!"planar record differs"); break; }
        }
        for (int c : seen) if (c != 1) { CHECK(!"a record was reported by the range callback zero or several times"); break; }
        // the same file read by `parts` cooperating readers (read_cloud_soa_part: what the ranks of pointsTransfer --gpus N do): the parts
        // tile the records of the whole-file read exactly, field for field
        for (int parts : {1, 2, 3, 7}) {
          std::vector<uint64_t> counts((size_t)parts, 0);
          std::vector<void*> pm;
          auto al = [&](size_t bytes) { void* q = std::malloc(bytes ? bytes : 1); pm.push_back(q); return q; };
          for (int r = 0; r < parts; ++r) {                        // sweep 1: every reader's token count (binary files never ask)
            ply::CloudSoA tmp; uint64_t f0 = 0, tot = 0; long d2 = 0;
            (void)ply::read_cloud_soa_part(path, r, parts, tmp, f0, tot, d2, al, [&](uint64_t mine, uint64_t&, uint64_t&) { counts[(size_t)r] = mine; return false; }, th);
          }
          uint64_t total = 0, next = 0;
          for (uint64_t c : counts) total += c;
          for (int r = 0; r < parts; ++r) {
            ply::CloudSoA pc; uint64_t f0 = 0, tot = 0; long d2 = 0;
            const bool pok = ply::read_cloud_soa_part(path, r, parts, pc, f0, tot, d2, al, [&](uint64_t mine, uint64_t& before, uint64_t& all) {
              before = 0; for (int q = 0; q < r; ++q) before += counts[(size_t)q];
              all = total; return mine == counts[(size_t)r]; }, th);
            CHECK(pok && d2 == want_declared && tot == got.n && f0 == next);
            for (size_t i = 0; i < pc.n && f0 + i < got.n; ++i) {
              const size_t g = (size_t)f0 + i;
              const bool same = eq(pc.x[i], got.x[g]) && eq(pc.y[i], got.y[g]) && eq(pc.z[i], got.z[g]) && eq(pc.nrm[3 * i], got.nrm[3 * g]) && eq(pc.nrm[3 * i + 1], got.nrm[3 * g + 1]) &&
                                eq(pc.nrm[3 * i + 2], got.nrm[3 * g + 2]) && pc.rgb[3 * i] == got.rgb[3 * g] && pc.rgb[3 * i + 1] == got.rgb[3 * g + 1] && pc.rgb[3 * i + 2] == got.rgb[3 * g + 2];
              if (!same) { CHECK(!"a record of a part differs from the whole-file read"); break; }
            }
            next = f0 + pc.n;
          }
          CHECK(next == got.n);
          for (void* q : pm) std::free(q);
        }
        for (void* q : mem) std::free(q);
      }
    };
    std::vector<Point> want;
    long dw = 0;
    for (const char* f : {"a.ply", "b.ply", "q.ply", "r.ply", "s.ply", "t.ply"}) {
      CHECK(ply::read_cloud(dir + "/" + f, want, dw));
      soa_equals(dir + "/" + f, want, dw, false);
    }
    // binary: the records of r.ply written as (double xyz, float normals, uchar colours) and as all-float properties + two extra ones
    CHECK(ply::read_cloud(dir + "/r.ply", want, dw));
    for (Point& q : want) for (int a = 0; a < 3; ++a) { q.color[a] = std::min(std::max(q.color[a], 0), 255); if (q.ver[a] != q.ver[a]) q.ver[a] = 0; if (q.normal[a] != q.normal[a]) q.normal[a] = 0; }
    {
      std::string b = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(want.size()) +
                      "\nproperty double x\nproperty double y\nproperty double z\nproperty float nx\nproperty float ny\nproperty float nz\n"
                      "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
      for (const Point& q : want) {
        b.append(reinterpret_cast<const char*>(q.ver), 24);
        for (int a = 0; a < 3; ++a) { const float v = (float)q.normal[a]; b.append(reinterpret_cast<const char*>(&v), 4); }
        for (int a = 0; a < 3; ++a) b.push_back((char)(unsigned char)q.color[a]);
      }
      soa_equals(write(dir, "bin_d.ply", b), want, (long)want.size(), false);
      b.resize(b.size() - 17);                                      // truncated: the last record is incomplete
      std::vector<Point> fewer(want.begin(), want.end() - 1);
      soa_equals(write(dir, "bin_t.ply", b), fewer, (long)want.size(), false);
    }
    {
      std::string b = "ply\nformat binary_little_endian 1.0\ncomment all floats\nelement vertex " + std::to_string(want.size()) +
                      "\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
                      "property ushort red\nproperty int green\nproperty uchar blue\nproperty float quality\nproperty uchar flag\nend_header\r\n";
      for (const Point& q : want) {
        for (int a = 0; a < 3; ++a) { const float v = (float)q.ver[a]; b.append(reinterpret_cast<const char*>(&v), 4); }
        for (int a = 0; a < 3; ++a) { const float v = (float)q.normal[a]; b.append(reinterpret_cast<const char*>(&v), 4); }
        const uint16_t r = (uint16_t)q.color[0]; const int32_t g = q.color[1]; const unsigned char bl = (unsigned char)q.color[2];
        b.append(reinterpret_cast<const char*>(&r), 2); b.append(reinterpret_cast<const char*>(&g), 4); b.push_back((char)bl);
        const float qual = 0.5f; b.append(reinterpret_cast<const char*>(&qual), 4); b.push_back((char)7);
      }
      soa_equals(write(dir, "bin_f.ply", b), want, (long)want.size(), true);
    }
    {   // fields BY NAME: the nine cloud fields declared under their usual names in another order, with other properties beside them -- text
        // and binary -- give the records of the plain positional file; a header that only renames (nine properties, odd names) stays positional
      std::mt19937_64 rg(99);
      const int nrec = 5003;
      std::vector<std::array<double, 9>> recs((size_t)nrec);
      std::string pos = "ply\nformat ascii 1.0\nelement vertex " + std::to_string(nrec) + "\nend_header\n";
      std::string hdr_named = "element vertex " + std::to_string(nrec) + "\nproperty float quality\nproperty uchar red\nproperty double z\nproperty double x\nproperty float nx\n"
                              "property uchar alpha\nproperty double y\nproperty uchar green\nproperty float nz\nproperty float ny\nproperty uchar blue\nend_header\n";
      std::string named = "ply\nformat ascii 1.0\n" + hdr_named, bin = "ply\nformat binary_little_endian 1.0\n" + hdr_named;
      char num[64];
      for (auto& r : recs) {
        for (int f = 0; f < 6; ++f) r[(size_t)f] = (double)(int64_t)(rg() % 1600001 - 800000) * 0.125;     // (values text, float and double all hold exactly)
        for (int f = 6; f < 9; ++f) r[(size_t)f] = (double)(rg() % 256);
        for (int f = 0; f < 9; ++f) { std::snprintf(num, sizeof num, f < 6 ? "%.9g" : "%.0f", r[(size_t)f]); pos += num; pos += f == 8 ? "\n" : " "; }
        const int order[11] = {-1, 6, 2, 0, 3, -2, 1, 7, 5, 4, 8};             // -1: quality, -2: alpha
        for (int p = 0; p < 11; ++p) {
          const int f = order[p];
          if (f == -1) std::snprintf(num, sizeof num, "0.25"); else if (f == -2) std::snprintf(num, sizeof num, "255");
          else std::snprintf(num, sizeof num, f < 6 ? "%.9g" : "%.0f", r[(size_t)f]);
          named += num; named += p == 10 ? "\n" : (p % 3 ? " " : "\t");
          if (f == -1) { const float v = 0.25f; bin.append(reinterpret_cast<const char*>(&v), 4); }
          else if (f == -2) bin.push_back((char)255);
          else if (f >= 6) bin.push_back((char)(unsigned char)r[(size_t)f]);
          else if (f < 3) bin.append(reinterpret_cast<const char*>(&r[(size_t)f]), 8);
          else { const float v = (float)r[(size_t)f]; bin.append(reinterpret_cast<const char*>(&v), 4); }
        }
      }
      std::vector<Point> wantn;
      long dn = 0;
      CHECK(ply::read_cloud(write(dir, "pos9.ply", pos), wantn, dn) && wantn.size() == (size_t)nrec);
      soa_equals(dir + "/pos9.ply", wantn, nrec, false);
      soa_equals(write(dir, "named_a.ply", named), wantn, nrec, false);
      soa_equals(write(dir, "named_b.ply", bin), wantn, nrec, false);
      // nine properties with the usual names in the usual order, and nine with unknown names: positional, as ever
      std::string odd = "ply\nformat ascii 1.0\nelement vertex " + std::to_string(nrec) + "\nproperty float a\nproperty float b\nproperty float c\nproperty float d\nproperty float e\n"
                        "property float f\nproperty uchar g\nproperty uchar h\nproperty uchar i\nend_header\n" + pos.substr(pos.find("end_header\n") + 11);
      soa_equals(write(dir, "odd9.ply", odd), wantn, nrec, false);
    }
    {   // binary mesh against the text mesh u.ply (cleaned of NaNs), faces with a quad in between (first three indices kept)
      ply::Mesh tm;
      CHECK(ply::read_mesh(dir + "/u.ply", tm));
      std::string b = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(tm.vertices.size()) +
                      "\nproperty double x\nproperty double y\nproperty double z\nproperty double nx\nproperty double ny\nproperty double nz\n"
                      "property double s\nproperty double t\nproperty int red\nproperty int green\nproperty int blue\nelement face " +
                      std::to_string(tm.faces.size() / 3) + "\nproperty list uchar int vertex_indices\nend_header\n";
      for (const Point& q : tm.vertices) {
        b.append(reinterpret_cast<const char*>(q.ver), 24); b.append(reinterpret_cast<const char*>(q.normal), 24);
        b.append(reinterpret_cast<const char*>(&q.U), 8); b.append(reinterpret_cast<const char*>(&q.V), 8);
        b.append(reinterpret_cast<const char*>(q.color), 12);
      }
      for (size_t f = 0; f < tm.faces.size() / 3; ++f) {
        const bool quad = f % 5 == 2;
        b.push_back(quad ? 4 : 3);
        b.append(reinterpret_cast<const char*>(&tm.faces[3 * f]), 12);
        if (quad) { const int extra = 123; b.append(reinterpret_cast<const char*>(&extra), 4); }
      }
      const std::string path = write(dir, "bin_m.ply", b);
      for (int th : {1, -5, 0}) {
        ply::FastMesh got;
        CHECK(ply::read_mesh_any(path, got, th));
        CHECK(got.vertices.size() == tm.vertices.size() && got.faces == tm.faces);
        for (size_t i = 0; i < tm.vertices.size() && i < got.vertices.size(); ++i)
          if (!same_point(got.vertices[i], tm.vertices[i], true)) { CHECK(!"binary mesh vertex differs"); break; }
      }
      ply::FastMesh viaany;
      CHECK(ply::read_mesh_any(dir + "/u.ply", viaany, 0) && viaany.faces == tm.faces);     // text files take the text path
    }
  }
  std::printf(fails ? "ply selftest: %d failure(s)\n" : "ply selftest ok\n", fails);
  return fails ? 1 : 0;
}
