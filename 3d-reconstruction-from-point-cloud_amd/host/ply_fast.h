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


// =====================================================================================================================
// Binary PLY and structure-of-arrays output (second half).
//
// The reference reads ASCII only (src/pointsTransfer.cpp:134-253, :266-455) and fills 80-byte Point records, of which the
// hot path uses 24 + 16 bytes.  Two additions, both supersets of the reference's behaviour:
//   * `format binary_little_endian 1.0` files: the header's property declarations give the record layout; the FIRST nine
//     (cloud) / eleven (mesh) scalar properties of the vertex element are taken in the reference's positional order
//     (x y z nx ny nz r g b  /  x y z nx ny nz u v r g b), whatever they are called, each converted from its declared type
//     exactly as the text reader converts a token (through double; colours truncated to int); further properties are
//     skipped.  Faces: a list property `count, indices...`; the first three indices are kept (the reference assumes triangles).
//   * clouds whose vertex element declares all nine fields under their usual names are read BY NAME (any order, other properties
//     skipped: cloud_layout below) by the planar readers; everything else -- and every mesh -- stays positional like the reference.
//   * clouds can be parsed straight into planar arrays x[] y[] z[] (double), rgb[][3] (bytes, clamped to 0..255 like the
//     device-side record split does) and nrm[][3] (float) -- 39 bytes per point instead of 80 -- in memory the caller
//     allocates (pinned, for the CLI), with a callback per finished range of records so that the upload of a range
//     overlaps the parsing of the others.
struct Header {
  long vertex_count = -1, face_count = -1;
  bool binary = false, big_endian = false;
  std::vector<int> vsize;          // byte size of every scalar property of the vertex element, in order
  std::vector<char> vkind;         // 'i' signed, 'u' unsigned, 'f' float, per property
  std::vector<std::string> vname;  // the properties' names, in order
  bool vertex_has_list = false;    // a list property inside the vertex element (positional reading only)
  int face_count_size = 1, face_index_size = 4;
  char face_index_kind = 'i';
  const char* body = nullptr;
};
namespace detail {
inline bool type_of(const std::string& t, int& size, char& kind) {
  if (t == "char" || t == "int8") { size = 1; kind = 'i'; }
  else if (t == "uchar" || t == "uint8") { size = 1; kind = 'u'; }
  else if (t == "short" || t == "int16") { size = 2; kind = 'i'; }
  else if (t == "ushort" || t == "uint16") { size = 2; kind = 'u'; }
  else if (t == "int" || t == "int32") { size = 4; kind = 'i'; }
  else if (t == "uint" || t == "uint32") { size = 4; kind = 'u'; }
  else if (t == "float" || t == "float32") { size = 4; kind = 'f'; }
  else if (t == "double" || t == "float64") { size = 8; kind = 'f'; }
  else return false;
  return true;
}
inline double load_scalar(const unsigned char* p, int size, char kind) {      // little-endian host (x86-64)
  switch (kind) {
    case 'f': if (size == 4) { float v; std::memcpy(&v, p, 4); return (double)v; } else { double v; std::memcpy(&v, p, 8); return v; }
    case 'u': if (size == 1) return (double)p[0]; if (size == 2) { uint16_t v; std::memcpy(&v, p, 2); return (double)v; } else { uint32_t v; std::memcpy(&v, p, 4); return (double)v; }
    default: if (size == 1) return (double)(int8_t)p[0]; if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return (double)v; } else { int32_t v; std::memcpy(&v, p, 4); return (double)v; }
  }
}
}  // namespace detail

// Header with the declarations a binary body needs.  For ASCII files it agrees with parse_header (same counts, same body).
inline Header parse_header_full(const char* begin, const char* end) {
  Header h;
  h.body = parse_header(begin, end, h.vertex_count, h.face_count);
  // second look at the header text, line by line, for `format` and the property declarations
  std::string element;
  const char* p = begin;
  while (p < h.body) {
    const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(h.body - p)));
    const char* le = nl ? nl : h.body;
    std::vector<std::string> tok;
    {
      const char* q = p;
      while (q < le) {
        while (q < le && is_ws(*q)) ++q;
        const char* t = q;
        while (q < le && !is_ws(*q)) ++q;
        if (q > t) tok.emplace_back(t, q);
      }
    }
    if (tok.size() >= 2 && tok[0] == "format") { h.binary = tok[1] != "ascii"; h.big_endian = tok[1] == "binary_big_endian"; }
    else if (tok.size() >= 2 && tok[0] == "element") element = tok[1];
    else if (tok.size() >= 3 && tok[0] == "property") {
      int sz; char kd;
      if (element == "vertex" && tok[1] != "list" && detail::type_of(tok[1], sz, kd)) { h.vsize.push_back(sz); h.vkind.push_back(kd); h.vname.push_back(tok[2]); }
      else if (element == "vertex" && tok[1] == "list") h.vertex_has_list = true;
      else if (element == "face" && tok[1] == "list" && tok.size() >= 5) {
        int cs, is; char ck, ik;
        if (detail::type_of(tok[2], cs, ck) && detail::type_of(tok[3], is, ik)) { h.face_count_size = cs; h.face_index_size = is; h.face_index_kind = ik; }
      }
    }
    p = nl ? nl + 1 : h.body;
  }
  if (h.binary && h.body < end && *h.body == '\r') ++h.body;       // "end_header\r\n"
  if (h.binary && h.body < end && *h.body == '\n') ++h.body;       // the body starts right after the header's last newline
  return h;
}

// planar cloud in caller-provided memory (see above): x, y, z: n doubles each; rgb: 3 n bytes; nrm: 3 n floats
struct CloudSoA {
  double *x = nullptr, *y = nullptr, *z = nullptr;
  uint8_t* rgb = nullptr;
  float* nrm = nullptr;
  size_t n = 0;
};
namespace detail {
inline uint8_t clamp_u8(double v) { const int c = (int)v; return (uint8_t)(c < 0 ? 0 : (c > 255 ? 255 : c)); }      // (int)v: as the AoS reader stores it
inline void set_cloud_soa(const CloudSoA& o, uint64_t rec, unsigned field, double v) {
  switch (field) {
    case 0: o.x[rec] = v; break;
    case 1: o.y[rec] = v; break;
    case 2: o.z[rec] = v; break;
    case 3: case 4: case 5: o.nrm[3 * rec + (field - 3)] = (float)v; break;
    default: o.rgb[3 * rec + (field - 6)] = clamp_u8(v); break;
  }
}
}  // namespace detail

// Which declared property feeds which of the nine cloud fields (x y z nx ny nz r g b).  The reference reads POSITIONALLY -- nine
// tokens per record, whatever the header says (src/pointsTransfer.cpp:204-250) -- and that stays the rule.  One extension, for files
// the reference would misread anyway: when the vertex element declares ALL nine fields under their usual names (x y z nx ny nz and
// red green blue, also r g b / diffuse_red ...), in any order and with any other scalar properties beside them (alpha, quality,
// scalar fields of scanner exports), the fields are taken BY NAME and the rest is skipped.  P = scalar properties (= tokens) per record.
struct CloudLayout {
  bool named = false;
  int P = 9;
  std::vector<int> fmap;        // per declared property: the field it feeds (0..8) or -1
};
inline CloudLayout cloud_layout(const Header& h) {
  CloudLayout L;
  static const char* const names[9][3] = {{"x", "", ""}, {"y", "", ""}, {"z", "", ""}, {"nx", "normal_x", ""}, {"ny", "normal_y", ""}, {"nz", "normal_z", ""},
                                          {"red", "r", "diffuse_red"}, {"green", "g", "diffuse_green"}, {"blue", "b", "diffuse_blue"}};
  if (h.vertex_has_list || h.vname.size() < 9) return L;
  std::vector<int> fmap(h.vname.size(), -1);
  int found = 0;
  for (int f = 0; f < 9; ++f) {
    int at = -1;
    for (size_t p = 0; p < h.vname.size() && at < 0; ++p)
      for (int a = 0; a < 3; ++a) if (names[f][a][0] && h.vname[p] == names[f][a] && fmap[p] < 0) { at = (int)p; break; }
    if (at >= 0) { fmap[(size_t)at] = f; ++found; }
  }
  if (found != 9) return L;
  bool positional_already = h.vname.size() == 9;
  for (int f = 0; f < 9 && positional_already; ++f) positional_already = fmap[(size_t)f] == f;
  if (positional_already) return L;                     // (the usual file: nothing to do differently)
  L.named = true; L.P = (int)h.vname.size(); L.fmap = fmap;
  return L;
}

// Records of the cloud file at `path` (ASCII or binary little-endian) into planar arrays.  `alloc(bytes)` provides the memory
// (five calls, once the record count is known; return nullptr to fail), `on_count(n)` announces that count before the first
// record is parsed (return false to abort), `range_done(first, count)` -- may do nothing -- is
// called from the parsing threads for every run of records that is complete while other ranges are still being parsed
// (records cut by a range boundary are reported after the join).  Returns false when the file cannot be opened or memory
// is refused; `declared` = the header's count (or -1); out.n = the complete records present, at most `declared`.
template <class Alloc, class OnCount, class RangeDone>
inline bool read_cloud_soa(const std::string& path, CloudSoA& out, long& declared, Alloc&& alloc, OnCount&& on_count, RangeDone&& range_done,
                           int threads = 0) {
  detail::Mapping map;
  if (!map.open(path)) return false;
  const Header h = parse_header_full(map.begin(), map.end());
  const CloudLayout L = cloud_layout(h);
  declared = h.vertex_count;
  out = CloudSoA();
  if (declared <= 0 || h.body >= map.end()) return true;
  if (h.binary && h.big_endian) return false;                        // (not produced by anything in this pipeline)
  auto allocate = [&](uint64_t n) -> bool {
    out.x = static_cast<double*>(alloc(n * 8)); out.y = static_cast<double*>(alloc(n * 8)); out.z = static_cast<double*>(alloc(n * 8));
    out.rgb = static_cast<uint8_t*>(alloc(n * 3)); out.nrm = static_cast<float*>(alloc(n * 12));
    out.n = (size_t)n;
    return out.x && out.y && out.z && out.rgb && out.nrm && on_count(n);
  };
  if (h.binary) {
    size_t stride = 0;
    for (int sz : h.vsize) stride += (size_t)sz;
    if (!stride) return true;
    const uint64_t n = std::min<uint64_t>((uint64_t)declared, (uint64_t)(map.end() - h.body) / stride);
    if (!n) return true;
    if (!allocate(n)) return false;
    const int nprop = (int)std::min<size_t>(h.vsize.size(), 9);
    const int parts = detail::resolve_threads(threads, (size_t)n * stride);
    detail::run_parallel(parts, [&](int i) {
      const uint64_t r0 = n * (uint64_t)i / (uint64_t)parts, r1 = n * (uint64_t)(i + 1) / (uint64_t)parts;
      const unsigned char* rec = reinterpret_cast<const unsigned char*>(h.body) + r0 * stride;
      for (uint64_t r = r0; r < r1; ++r, rec += stride) {
        const unsigned char* q = rec;
        if (L.named) {                                               // fields by name: every declared property is walked, the nine are kept
          for (int p = 0; p < L.P; ++p) {
            if (L.fmap[(size_t)p] >= 0) detail::set_cloud_soa(out, r, (unsigned)L.fmap[(size_t)p], detail::load_scalar(q, h.vsize[(size_t)p], h.vkind[(size_t)p]));
            q += h.vsize[(size_t)p];
          }
        } else {
          for (int f = 0; f < 9; ++f) {
            double v = 0.0;
            if (f < nprop) { v = detail::load_scalar(q, h.vsize[(size_t)f], h.vkind[(size_t)f]); q += h.vsize[(size_t)f]; }
            detail::set_cloud_soa(out, r, (unsigned)f, v);
          }
        }
      }
      if (r1 > r0) range_done(r0, r1 - r0);
    });
    return true;
  }
  const int parts = detail::resolve_threads(threads, (size_t)(map.end() - h.body));
  const auto cuts = detail::token_cuts(h.body, map.end(), parts);
  const auto first = detail::token_offsets(cuts);
  const uint64_t P = (uint64_t)L.P;                                  // tokens per record: 9 (the reference's rule), or the declared properties when read by name
  const uint64_t n = std::min<uint64_t>((uint64_t)declared, first[(size_t)parts] / P);
  if (!n) return true;
  if (!allocate(n)) return false;
  const uint64_t limit = n * P;
  detail::run_parallel(parts, [&](int i) {
    uint64_t g = first[(size_t)i];
    if (g >= limit) return;
    detail::for_each_token(cuts[(size_t)i], cuts[(size_t)i + 1], [&](const char* t, const char* e) {
      if (g < limit) {
        const int fld = L.named ? L.fmap[(size_t)(g % P)] : (int)(g % P);
        if (fld >= 0) detail::set_cloud_soa(out, g / P, (unsigned)fld, detail::token_value(t, e));
      }
      ++g;
    });
    // the records this range holds from first to last token
    const uint64_t r0 = (first[(size_t)i] + P - 1) / P, r1 = std::min<uint64_t>(first[(size_t)i + 1], limit) / P;
    if (r1 > r0) range_done(r0, r1 - r0);
  });
  uint64_t last = ~0ull;
  for (int i = 1; i < parts; ++i) {                                  // records cut by a range boundary: complete only now
    const uint64_t g = first[(size_t)i];                             // (several boundaries may cut the same record: reported once)
    if (g % P && g / P < n && g / P != last) { last = g / P; range_done(last, 1); }
  }
  return true;
}

// ONE PART of a cloud file for a job of `parts` cooperating readers (pointsTransfer --gpus N, host/sharded.h: rank r parses 1/N of
// the file instead of all of it).  Part `part` owns the records that BEGIN in its share of the body:
//   binary  records [n * part / parts, n * (part + 1) / parts) -- fixed stride, nothing to agree on;
//   ASCII   the body is cut into `parts` byte ranges on token boundaries (the same cuts for every reader: they depend on the file
//           only); a reader counts the tokens of its own range and `token_prefix(my_tokens, total_out)` -- the one exchange between the
//           readers, provided by the caller -- returns the number of tokens in the ranges before it and the file's total.  Record r
//           starts at token P r (P = 9, or the declared properties when the fields are read by name): the reader converts tokens [P r0, P r1) for the records that begin in its range, reading the last
//           record's tail out of the next range (fewer than P tokens) and leaving the head of a record begun earlier to its neighbour.
// out.n = records of this part, `first_record` their global index, `total_records` the file's complete records (what
// read_cloud_soa would deliver as out.n); record i of the part is global record first_record + i, field for field what
// read_cloud_soa stores there.  alloc as in read_cloud_soa (five calls).  Returns false when the file cannot be opened, memory is
// refused or token_prefix fails (returns false).
template <class Alloc, class Prefix>
inline bool read_cloud_soa_part(const std::string& path, int part, int parts, CloudSoA& out, uint64_t& first_record, uint64_t& total_records, long& declared,
                                Alloc&& alloc, Prefix&& token_prefix, int threads = 0) {
  detail::Mapping map;
  out = CloudSoA();
  first_record = total_records = 0;
  if (parts < 1 || part < 0 || part >= parts || !map.open(path)) return false;
  const Header h = parse_header_full(map.begin(), map.end());
  const CloudLayout L = cloud_layout(h);
  declared = h.vertex_count;
  auto allocate = [&](uint64_t n) -> bool {
    const uint64_t a = n ? n : 1;
    out.x = static_cast<double*>(alloc(a * 8)); out.y = static_cast<double*>(alloc(a * 8)); out.z = static_cast<double*>(alloc(a * 8));
    out.rgb = static_cast<uint8_t*>(alloc(a * 3)); out.nrm = static_cast<float*>(alloc(a * 12));
    out.n = (size_t)n;
    return out.x && out.y && out.z && out.rgb && out.nrm;
  };
  const bool empty = declared <= 0 || h.body >= map.end();
  if (h.binary) {
    if (h.big_endian) return false;
    size_t stride = 0;
    for (int sz : h.vsize) stride += (size_t)sz;
    const uint64_t n = (empty || !stride) ? 0 : std::min<uint64_t>((uint64_t)declared, (uint64_t)(map.end() - h.body) / stride);
    total_records = n;
    const uint64_t r0 = n * (uint64_t)part / (uint64_t)parts, r1 = n * (uint64_t)(part + 1) / (uint64_t)parts;
    first_record = r0;
    if (!allocate(r1 - r0)) return false;
    if (r1 == r0) return true;
    const int nprop = (int)std::min<size_t>(h.vsize.size(), 9);
    const int th = detail::resolve_threads(threads, (size_t)(r1 - r0) * stride);
    detail::run_parallel(th, [&](int i) {
      const uint64_t a = r0 + (r1 - r0) * (uint64_t)i / (uint64_t)th, b = r0 + (r1 - r0) * (uint64_t)(i + 1) / (uint64_t)th;
      const unsigned char* rec = reinterpret_cast<const unsigned char*>(h.body) + a * stride;
      for (uint64_t r = a; r < b; ++r, rec += stride) {
        const unsigned char* q = rec;
        if (L.named) {
          for (int p = 0; p < L.P; ++p) {
            if (L.fmap[(size_t)p] >= 0) detail::set_cloud_soa(out, r - r0, (unsigned)L.fmap[(size_t)p], detail::load_scalar(q, h.vsize[(size_t)p], h.vkind[(size_t)p]));
            q += h.vsize[(size_t)p];
          }
        } else {
          for (int f = 0; f < 9; ++f) {
            double v = 0.0;
            if (f < nprop) { v = detail::load_scalar(q, h.vsize[(size_t)f], h.vkind[(size_t)f]); q += h.vsize[(size_t)f]; }
            detail::set_cloud_soa(out, r - r0, (unsigned)f, v);
          }
        }
      }
    });
    return true;
  }
  // ASCII
  const char* body = empty ? map.end() : h.body;
  const auto rank_cuts = detail::token_cuts(body, map.end(), parts);                 // the same on every reader
  const char *mb = rank_cuts[(size_t)part], *me = rank_cuts[(size_t)part + 1];
  const int th = detail::resolve_threads(threads, (size_t)(me - mb));
  const auto cuts = detail::token_cuts(mb, me, th);
  const auto first = detail::token_offsets(cuts);                                     // pass 1 over this part only
  uint64_t total_tokens = 0, g0 = 0;
  if (!token_prefix(first[(size_t)th], g0, total_tokens)) return false;
  const uint64_t P = (uint64_t)L.P;                                                   // tokens per record (9, or the declared properties when read by name)
  const uint64_t n = empty ? 0 : std::min<uint64_t>((uint64_t)declared, total_tokens / P);
  total_records = n;
  const uint64_t g1 = g0 + first[(size_t)th];
  const uint64_t r0 = std::min<uint64_t>((g0 + P - 1) / P, n), r1 = std::min<uint64_t>((g1 + P - 1) / P, n);
  first_record = r0;
  if (!allocate(r1 - r0)) return false;
  if (r1 == r0) return true;
  const uint64_t lo = r0 * P, hi = r1 * P;                                            // the tokens this reader converts: [lo, hi)
  auto put = [&](uint64_t g, const char* t, const char* e) {
    const int fld = L.named ? L.fmap[(size_t)(g % P)] : (int)(g % P);
    if (fld >= 0) detail::set_cloud_soa(out, g / P - r0, (unsigned)fld, detail::token_value(t, e));
  };
  detail::run_parallel(th, [&](int i) {
    uint64_t g = g0 + first[(size_t)i];
    if (g >= hi) return;
    detail::for_each_token(cuts[(size_t)i], cuts[(size_t)i + 1], [&](const char* t, const char* e) {
      if (g >= lo && g < hi) put(g, t, e);
      ++g;
    });
  });
  if (g1 < hi) {                                                                      // the last record's tail lives in the next range(s)
    uint64_t g = g1;
    const char* p = me;
    while (g < hi) {
      while (p < map.end() && is_ws(*p)) ++p;
      if (p >= map.end()) break;                                                      // (cannot happen: n counts complete records only)
      const char* t = p;
      while (p < map.end() && !is_ws(*p)) ++p;
      put(g, t, p);
      ++g;
    }
  }
  return true;
}

// Binary little-endian meshes (ASCII ones go through read_mesh_fast): vertices positional like the text grammar, faces from
// the list property -- the first three indices of every face.  Same contract as read_mesh_fast.
inline bool read_mesh_any(const std::string& path, FastMesh& mesh, int threads = 0) {
  {
    detail::Mapping probe;
    if (!probe.open(path)) return false;
    const Header h = parse_header_full(probe.begin(), probe.end());
    if (!h.binary) return read_mesh_fast(path, mesh, threads);
    mesh.vertices.resize(0);
    mesh.faces.clear();
    mesh.vertex_count = h.vertex_count; mesh.face_count = h.face_count;
    if (h.big_endian || h.body >= probe.end() || h.vertex_count <= 0) return !h.big_endian;
    size_t stride = 0;
    for (int sz : h.vsize) stride += (size_t)sz;
    if (!stride) return true;
    const uint64_t avail = (uint64_t)(probe.end() - h.body);
    const uint64_t nv = std::min<uint64_t>((uint64_t)h.vertex_count, avail / stride);
    if (!mesh.vertices.resize((size_t)nv)) return true;
    const int nprop = (int)std::min<size_t>(h.vsize.size(), 11);
    const int parts = detail::resolve_threads(threads, (size_t)nv * stride);
    Point* out = mesh.vertices.data();
    detail::run_parallel(parts, [&](int i) {
      const uint64_t r0 = nv * (uint64_t)i / (uint64_t)parts, r1 = nv * (uint64_t)(i + 1) / (uint64_t)parts;
      const unsigned char* rec = reinterpret_cast<const unsigned char*>(h.body) + r0 * stride;
      for (uint64_t r = r0; r < r1; ++r, rec += stride) {
        const unsigned char* q = rec;
        out[r].normal[0] = out[r].normal[1] = out[r].normal[2] = 0.0;
        for (int f = 0; f < 11; ++f) {
          double v = 0.0;
          if (f < nprop) { v = detail::load_scalar(q, h.vsize[(size_t)f], h.vkind[(size_t)f]); q += h.vsize[(size_t)f]; }
          detail::set_mesh_field(out[r], (unsigned)f, v);
        }
      }
    });
    if (nv < (uint64_t)h.vertex_count || h.face_count <= 0) return true;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(h.body) + nv * stride;
    const unsigned char* e = reinterpret_cast<const unsigned char*>(probe.end());
    mesh.faces.reserve((size_t)h.face_count * 3);
    for (long f = 0; f < h.face_count; ++f) {                        // (variable-length lists: sequential)
      if (p + h.face_count_size > e) break;
      const uint64_t cnt = (uint64_t)detail::load_scalar(p, h.face_count_size, 'u');
      p += h.face_count_size;
      if ((uint64_t)(e - p) < cnt * (uint64_t)h.face_index_size) break;
      int v3[3] = {0, 0, 0};
      for (uint64_t c = 0; c < cnt && c < 3; ++c) v3[c] = (int)detail::load_scalar(p + c * (uint64_t)h.face_index_size, h.face_index_size, h.face_index_kind);
      p += cnt * (uint64_t)h.face_index_size;
      mesh.faces.push_back(v3[0]); mesh.faces.push_back(v3[1]); mesh.faces.push_back(v3[2]);
    }
  }
  return true;
}

}  // namespace ply
