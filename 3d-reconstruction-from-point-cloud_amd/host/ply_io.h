// ply_io.h -- ASCII PLY ingest for the pointsTransfer CLI.
//
// Accepts the grammar the reference's hand-rolled tokenizer accepts (reference
// src/pointsTransfer.cpp:134-253 for the cloud, :266-455 for the mesh):
//   header  : whitespace-separated tokens; the token `vertex` is followed (same line) by the vertex count,
//             `face` by the face count, `end_header` ends it.  Property declarations are not interpreted.
//   cloud   : 9 numbers per vertex   x y z nx ny nz r g b          (colour read as a number, stored as int)
//   mesh    : 11 numbers per vertex  x y z nx ny nz u v r g b, then per face `n i j k` (n ignored, triangles)
// Deliberate supersets: '\r' counts as whitespace, and a last record without a trailing newline is kept
// (the reference drops it).  The whole file is read in one go and parsed with strtod -- ingest speed is a
// "next" row, not part of this path.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Point.h"

namespace ply {

struct Mesh {
  std::vector<Point> vertices;
  std::vector<int> faces;   // 3 per face
  long vertex_count = 0, face_count = 0;
};

inline bool slurp(const std::string& path, std::string& out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize(sz > 0 ? (size_t)sz : 0);
  const size_t got = sz > 0 ? std::fread(&out[0], 1, (size_t)sz, f) : 0;
  std::fclose(f);
  out.resize(got);
  return true;
}

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

// Reads header tokens from `p`; returns the body start.  Counts are -1 when absent.
inline const char* parse_header(const char* p, const char* end, long& vertex_count, long& face_count) {
  vertex_count = face_count = -1;
  while (p < end) {
    while (p < end && is_ws(*p)) ++p;
    const char* t = p;
    while (p < end && !is_ws(*p)) ++p;
    const size_t len = (size_t)(p - t);
    auto rest_of_line_count = [&](long& dst) {
      const char* q = p;
      while (p < end && *p != '\n') ++p;
      dst = std::atol(std::string(q, p).c_str());
    };
    if (len == 6 && !std::memcmp(t, "vertex", 6)) rest_of_line_count(vertex_count);
    else if (len == 4 && !std::memcmp(t, "face", 4)) rest_of_line_count(face_count);
    else if (len == 10 && !std::memcmp(t, "end_header", 10)) return p;
  }
  return p;
}

inline bool next_number(const char*& p, const char* end, double& v) {
  while (p < end && is_ws(*p)) ++p;
  if (p >= end) return false;
  char* q = nullptr;
  v = std::strtod(p, &q);
  if (q == p) {   // not a number: skip the token, value 0 (atof semantics)
    while (p < end && !is_ws(*p)) ++p;
    v = 0.0;
    return true;
  }
  p = q;
  while (p < end && !is_ws(*p)) ++p;   // trailing junk of the token
  return true;
}

// cloud: reference src/pointsTransfer.cpp:134-253
inline bool read_cloud(const std::string& path, std::vector<Point>& points, long& declared) {
  std::string buf;
  if (!slurp(path, buf)) return false;
  const char* end = buf.data() + buf.size();
  long faces;
  const char* p = parse_header(buf.data(), end, declared, faces);
  points.clear();
  if (declared > 0) points.reserve((size_t)declared);
  double v[9];
  while ((long)points.size() < declared) {
    int got = 0;
    while (got < 9 && next_number(p, end, v[got])) ++got;
    if (got < 9) break;
    points.emplace_back(v[0], v[1], v[2], v[3], v[4], v[5], (int)v[6], (int)v[7], (int)v[8]);
  }
  return true;
}

// mesh: reference src/pointsTransfer.cpp:266-455
inline bool read_mesh(const std::string& path, Mesh& mesh) {
  std::string buf;
  if (!slurp(path, buf)) return false;
  const char* end = buf.data() + buf.size();
  const char* p = parse_header(buf.data(), end, mesh.vertex_count, mesh.face_count);
  mesh.vertices.clear();
  mesh.faces.clear();
  double v[11];
  while ((long)mesh.vertices.size() < mesh.vertex_count) {
    int got = 0;
    while (got < 11 && next_number(p, end, v[got])) ++got;
    if (got < 11) break;
    // file order x y z nx ny nz u v r g b  ->  Point(x,y,z,nx,ny,nz,r,g,b,u,v)   (reference :394)
    mesh.vertices.emplace_back(v[0], v[1], v[2], v[3], v[4], v[5], (int)v[8], (int)v[9], (int)v[10], v[6], v[7]);
  }
  for (long f = 0; f < mesh.face_count; ++f) {
    double q[4];
    int got = 0;
    while (got < 4 && next_number(p, end, q[got])) ++got;
    if (got < 4) break;
    mesh.faces.push_back((int)q[1]); mesh.faces.push_back((int)q[2]); mesh.faces.push_back((int)q[3]);
  }
  return true;
}

}  // namespace ply
