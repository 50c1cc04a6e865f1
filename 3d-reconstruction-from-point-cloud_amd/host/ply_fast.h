// ply_fast.h -- parallel ASCII PLY ingest: same grammar and same results as ply_io.h, which in
// turn accepts what the reference's tokenizer accepts (reference src/pointsTransfer.cpp:134-253 cloud, :266-455 mesh).
//
// The grammar is token based, not line based (9 / 11 / 4 whitespace-separated tokens per cloud vertex / mesh vertex /
// face; a record may span lines), so the file cannot simply be cut at newlines.  Instead:
//   1. the file is mmap'ed and the header parsed as in ply_io.h;
//   2. the body is cut into one byte range per thread, every cut moved forward to the end of the token it falls in;
//   3. pass 1 counts the tokens of every range, an exclusive scan gives each range the GLOBAL index of its first token;
//   4. pass 2 converts tokens: global index g -> record g / 9 (or 11, or 4 after the vertices), field g % 9, written
//      straight into the caller's record array -- no intermediate token list, no locks.
// Numbers: std::from_chars when it consumes the whole token (correctly rounded, like strtod), otherwise the strtod
// rules of ply_io.h on a NUL-terminated copy (leading '+', hex, inf/nan, trailing junk, non-numbers -> 0).
// Records are written into a RecordBuffer (malloc'ed, never value-initialised: zeroing 80 GB serially is the first
// thing that would dominate at 1e9 points).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "Point.h"
#include "ply_io.h"

namespace ply {

// uninitialised array of Point records (Point is trivially copyable; every field of every record is written by the parser)
class RecordBuffer {
 public:
  RecordBuffer() = default;
  RecordBuffer(const RecordBuffer&) = delete;
  RecordBuffer& operator=(const RecordBuffer&) = delete;
  ~RecordBuffer() { std::free(p_); }
  bool resize(size_t n) {
    std::free(p_);
    p_ = nullptr; n_ = 0;
    if (!n) return true;
    p_ = static_cast<Point*>(std::malloc(n * sizeof(Point)));
    if (!p_) return false;
    n_ = n;
    return true;
  }
  Point* data() { return p_; }
  const Point* data() const { return p_; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  Point& operator[](size_t i) { return p_[i]; }
  const Point& operator[](size_t i) const { return p_[i]; }

 private:
  Point* p_ = nullptr;
  size_t n_ = 0;
};

struct FastMesh {
  RecordBuffer vertices;
  std::vector<int> faces;   // 3 per face
  long vertex_count = 0, face_count = 0;
};

namespace detail {

class Mapping {   // read-only view of a whole file
 public:
  ~Mapping() { if (p_ && n_) munmap(const_cast<char*>(p_), n_); }
  bool open(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
    n_ = (size_t)st.st_size;
    if (n_) {
      void* m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { ::close(fd); n_ = 0; return false; }
      p_ = static_cast<const char*>(m);
      madvise(m, n_, MADV_SEQUENTIAL);
    }
    ::close(fd);
    return true;
  }
  const char* begin() const { return p_; }
  const char* end() const { return p_ + n_; }

 private:
  const char* p_ = nullptr;
  size_t n_ = 0;
};

// value of the token [t, e): the strtod prefix of it, 0 when it is not a number (ply_io.h next_number)
inline double token_value(const char* t, const char* e) {
  double v = 0.0;
  const auto r = std::from_chars(t, e, v);
  if (r.ec == std::errc() && r.ptr == e) return v;
  // rare: anything from_chars does not take whole.  strtod wants a terminator the mapping may not have.
  char tmp[128];
  std::string big;
  const size_t len = (size_t)(e - t);
  const char* z = tmp;
  if (len < sizeof(tmp)) { std::memcpy(tmp, t, len); tmp[len] = 0; }
  else { big.assign(t, e); z = big.c_str(); }
  char* q = nullptr;
  v = std::strtod(z, &q);
  return q == z ? 0.0 : v;
}

// threads > 0: at most that many; < 0: exactly -threads ranges whatever the size (tests); 0: automatic -- one per
// hardware thread, but no more than 64 and no more than one per 16 MB (measured on a 256-thread host, 1.2 GB file:
// 32 threads 5.6 GB/s, 64 5.1, 128 4.5, 256 3.4 -- beyond a few dozen, thread start-up and page faults cost more than they parse)
inline int resolve_threads(int threads, size_t bytes) {
  if (threads < 0) return -threads;
  size_t cap = bytes / (1u << 16) + 1;             // never a thread per few KB
  if (threads == 0) {
    threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    threads = std::min(threads, 64);
    cap = std::min(cap, bytes / (16u << 20) + 1);
  }
  return (int)std::min<size_t>((size_t)threads, cap);
}

// Cuts [b, e) into `parts` ranges on token boundaries: cuts[i] .. cuts[i+1].
inline std::vector<const char*> token_cuts(const char* b, const char* e, int parts) {
  std::vector<const char*> cuts((size_t)parts + 1);
  cuts[0] = b;
  cuts[(size_t)parts] = e;
  const size_t n = (size_t)(e - b);
  for (int i = 1; i < parts; ++i) {
    const char* c = b + n / (size_t)parts * (size_t)i;
    if (c > b && !is_ws(c[-1]))                    // inside a token (or right at its end): it belongs to the range before
      while (c < e && !is_ws(*c)) ++c;
    cuts[(size_t)i] = std::max(c, cuts[(size_t)i - 1]);
  }
  return cuts;
}

template <class F>
inline void for_each_token(const char* p, const char* e, F&& f) {
  while (true) {
    while (p < e && is_ws(*p)) ++p;
    if (p >= e) return;
    const char* t = p;
    while (p < e && !is_ws(*p)) ++p;
    f(t, p);
  }
}

template <class F>
inline void run_parallel(int parts, F&& f) {
  if (parts <= 1) { f(0); return; }
  std::vector<std::thread> th;
  th.reserve((size_t)parts - 1);
  for (int i = 1; i < parts; ++i) th.emplace_back([&f, i] { f(i); });
  f(0);
  for (auto& t : th) t.join();
}

// token counts per range -> first global token index per range (first[parts] = total)
inline std::vector<uint64_t> token_offsets(const std::vector<const char*>& cuts) {
  const int parts = (int)cuts.size() - 1;
  std::vector<uint64_t> first((size_t)parts + 1, 0);
  run_parallel(parts, [&](int i) {
    uint64_t c = 0;
    for_each_token(cuts[(size_t)i], cuts[(size_t)i + 1], [&](const char*, const char*) { ++c; });
    first[(size_t)i + 1] = c;
  });
  for (int i = 0; i < parts; ++i) first[(size_t)i + 1] += first[(size_t)i];
  return first;
}

inline void set_cloud_field(Point& q, unsigned field, double v) {
  switch (field) {
    case 0: q.ver[0] = v; q.U = 0.0; q.V = 0.0; break;   // the record's first token also clears what the cloud has not
    case 1: q.ver[1] = v; break;
    case 2: q.ver[2] = v; break;
    case 3: q.normal[0] = v; break;
    case 4: q.normal[1] = v; break;
    case 5: q.normal[2] = v; break;
    case 6: q.color[0] = (int)v; break;
    case 7: q.color[1] = (int)v; break;
    default: q.color[2] = (int)v; break;
  }
}
// mesh file order x y z nx ny nz u v r g b (reference :394)
inline void set_mesh_field(Point& q, unsigned field, double v) {
  switch (field) {
    case 0: q.ver[0] = v; break;
    case 1: q.ver[1] = v; break;
    case 2: q.ver[2] = v; break;
    case 3: q.normal[0] = v; break;
    case 4: q.normal[1] = v; break;
    case 5: q.normal[2] = v; break;
    case 6: q.U = v; break;
    case 7: q.V = v; break;
    case 8: q.color[0] = (int)v; break;
    case 9: q.color[1] = (int)v; break;
    default: q.color[2] = (int)v; break;
  }
}

}  // namespace detail

// Same contract as ply::read_cloud: false only when the file cannot be opened; `declared` = the header's count (or -1);
// points = the complete records present, at most `declared`.
inline bool read_cloud_fast(const std::string& path, RecordBuffer& points, long& declared, int threads = 0) {
  detail::Mapping map;
  if (!map.open(path)) return false;
  long faces;
  const char* body = parse_header(map.begin(), map.end(), declared, faces);
  points.resize(0);
  if (declared <= 0 || body >= map.end()) return true;
  const int parts = detail::resolve_threads(threads, (size_t)(map.end() - body));
  const auto cuts = detail::token_cuts(body, map.end(), parts);
  const auto first = detail::token_offsets(cuts);
  const uint64_t n = std::min<uint64_t>((uint64_t)declared, first[(size_t)parts] / 9);
  if (!points.resize((size_t)n)) return true;
  const uint64_t limit = n * 9;
  Point* out = points.data();
  detail::run_parallel(parts, [&](int i) {
    uint64_t g = first[(size_t)i];
    if (g >= limit) return;
    detail::for_each_token(cuts[(size_t)i], cuts[(size_t)i + 1], [&](const char* t, const char* e) {
      if (g < limit) detail::set_cloud_field(out[g / 9], (unsigned)(g % 9), detail::token_value(t, e));
      ++g;
    });
  });
  return true;
}

// Same contract as ply::read_mesh.
inline bool read_mesh_fast(const std::string& path, FastMesh& mesh, int threads = 0) {
  detail::Mapping map;
  if (!map.open(path)) return false;
  const char* body = parse_header(map.begin(), map.end(), mesh.vertex_count, mesh.face_count);
  mesh.vertices.resize(0);
  mesh.faces.clear();
  if (body >= map.end()) return true;
  const int parts = detail::resolve_threads(threads, (size_t)(map.end() - body));
  const auto cuts = detail::token_cuts(body, map.end(), parts);
  const auto first = detail::token_offsets(cuts);
  const uint64_t total = first[(size_t)parts];
  const uint64_t want_v = mesh.vertex_count > 0 ? (uint64_t)mesh.vertex_count : 0;
  const uint64_t nv = std::min<uint64_t>(want_v, total / 11);
  uint64_t nf = 0;   // (a truncated vertex list has eaten every token: no faces, as in ply_io.h)
  if (nv == want_v && mesh.face_count > 0) nf = std::min<uint64_t>((uint64_t)mesh.face_count, (total - nv * 11) / 4);
  if (!mesh.vertices.resize((size_t)nv)) return true;
  mesh.faces.resize((size_t)nf * 3);
  const uint64_t vlimit = nv * 11, flimit = vlimit + nf * 4;
  Point* out = mesh.vertices.data();
  int* fo = mesh.faces.data();
  detail::run_parallel(parts, [&](int i) {
    uint64_t g = first[(size_t)i];
    if (g >= flimit) return;
    detail::for_each_token(cuts[(size_t)i], cuts[(size_t)i + 1], [&](const char* t, const char* e) {
      if (g < vlimit) {
        detail::set_mesh_field(out[g / 11], (unsigned)(g % 11), detail::token_value(t, e));
      } else if (g < flimit) {
        const uint64_t f = g - vlimit;
        if (f % 4) fo[(f / 4) * 3 + (f % 4 - 1)] = (int)detail::token_value(t, e);   // `n i j k`: n ignored
      }
      ++g;
    });
  });
  return true;
}

}  // namespace ply
