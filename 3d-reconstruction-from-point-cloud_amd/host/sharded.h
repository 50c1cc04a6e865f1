// sharded.h -- `pointsTransfer cloud.ply mesh.ply --gpus N`: the source cloud cut into N spatial slabs, one PROCESS per GPU, the
// slab exchange over RCCL behind the C ABI (pt_comm_* / pt_query_exchange_blend).  The reference is a single process
// (src/pointsTransfer.cpp:109-626); this is the host side of SURVEY.md 8(e) in the reference's own language.
//
//   launcher  (the process the user started; never touches a GPU) makes a rendezvous directory, starts N fresh rank processes
//             (fork + exec of this binary with --rank r), waits for them, then starts one finalize process and waits for it.
//   rank r    opens its GPU, joins the RCCL communicator (rank 0 writes the 128-byte id into the rendezvous directory, the others
//             poll for it), parses ITS 1/N OF THE CLOUD FILE (ply::read_cloud_soa_part: a record range of a binary file, a byte
//             range of a text file with one exchange of token counts through the rendezvous directory) and publishes the parsed
//             piece there as a planar binary file; every rank maps all N pieces (shared page cache: the cloud is in host memory
//             once, not N times), keeps the points of slab r (equal-count quantiles of a sample along the longest axis -- every
//             rank derives the same bounds from the same pieces), builds its grid with global indices, uploads the attribute table
//             piece by piece, searches the mesh vertices homed in its slab, completes them through the exchange, and writes its
//             vertices' neighbour lists, blended attributes and the records of the cloud points those lists name to rank_r.bin.
//   finalize  merges the rank files, builds a small cloud of the REFERENCED points only (the bake needs nothing else), remaps the
//             neighbour lists into it, bakes and pads the texture on one GPU and writes texture.png / transfer.ply.
// stdout keeps the reference's lines and order: rank 0 prints the read / build / search lines, finalize the draw / output lines,
// the launcher the totals.
#pragma once
#include <signal.h>
#include <spawn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "Point.h"
#include "ply_fast.h"
#include "png_write.h"
#include "pt_api.h"

extern char** environ;

namespace sharded {

struct Options {
  std::string cloud, mesh, out_name = "transfer.ply", tex_name = "texture.png", rendezvous;
  int K = 20, device = 0, mode = PT_BLEND_MEAN, ply_threads = 0, resolution = 8192, pad = 25, gpus = 1, rank = -1;
  bool finalize = false;
};
using clk = std::chrono::steady_clock;
inline double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

struct RefPoint { uint32_t id; double x, y, z; uint8_t rgb[3]; };

// free bytes of the file system holding `dir` (0 when it cannot be told)
inline uint64_t free_bytes(const std::string& dir) {
  struct statvfs vs;
  if (statvfs(dir.c_str(), &vs) != 0) return 0;
  return (uint64_t)vs.f_bavail * (uint64_t)vs.f_frsize;
}

inline bool write_all(const std::string& path, const std::vector<std::pair<const void*, size_t>>& parts) {
  const std::string tmp = path + ".part";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  bool ok = true;
  for (auto& p : parts) ok = ok && (p.second == 0 || std::fwrite(p.first, 1, p.second, f) == p.second);
  ok = (std::fclose(f) == 0) && ok;
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;     // readers never see a half-written file
}

// ---- the cloud as N pieces in the rendezvous directory --------------------------------------------------------------------------
// piece_<r>.bin = {u64 first, u64 count} x[count] y[count] z[count] (f64) nrm[count][3] (f32) rgb[count][3] (u8): what rank r parsed.
inline bool wait_for_file(const std::string& path, double timeout_s) {
  const auto t0 = clk::now();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (since(t0) > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
  return true;
}
struct Piece {
  uint64_t first = 0, count = 0;
  const double *x = nullptr, *y = nullptr, *z = nullptr;
  const float* nrm = nullptr;
  const uint8_t* rgb = nullptr;
  void* map = nullptr;
  size_t bytes = 0;
};
struct Pieces {
  std::vector<Piece> p;
  uint64_t n = 0;
  ~Pieces() { for (Piece& q : p) if (q.map) munmap(q.map, q.bytes); }
  bool open(const std::string& dir, int parts, double timeout_s) {
    for (int r = 0; r < parts; ++r) {
      const std::string path = dir + "/piece_" + std::to_string(r) + ".bin";
      if (!wait_for_file(path, timeout_s)) return false;
      const int fd = ::open(path.c_str(), O_RDONLY);
      if (fd < 0) return false;
      struct stat st;
      if (fstat(fd, &st) != 0 || (size_t)st.st_size < 16) { ::close(fd); return false; }
      Piece q;
      q.bytes = (size_t)st.st_size;
      q.map = mmap(nullptr, q.bytes, PROT_READ, MAP_SHARED, fd, 0);
      ::close(fd);
      if (q.map == MAP_FAILED) { q.map = nullptr; return false; }
      const char* b = static_cast<const char*>(q.map);
      std::memcpy(&q.first, b, 8); std::memcpy(&q.count, b + 8, 8);
      if (q.bytes != 16 + q.count * 39) { munmap(q.map, q.bytes); return false; }
      q.x = reinterpret_cast<const double*>(b + 16); q.y = q.x + q.count; q.z = q.y + q.count;
      q.nrm = reinterpret_cast<const float*>(q.z + q.count);
      q.rgb = reinterpret_cast<const uint8_t*>(q.nrm + 3 * q.count);
      if (q.first != n) { munmap(q.map, q.bytes); return false; }          // the pieces tile the records in rank order
      n += q.count;
      p.push_back(q);
    }
    return true;
  }
  // piece holding global record g (g < n)
  const Piece& of(uint64_t g) const {
    size_t lo = 0, hi = p.size();
    while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (p[mid].first <= g) lo = mid; else hi = mid; }
    return p[lo];
  }
};

// slab bounds along `axis`: equal-count quantiles of every stride-th point (deterministic: every rank computes the same)
inline void slab_bounds(const Pieces& c, int world, int& axis, std::vector<double>& bounds) {
  const uint64_t stride = std::max<uint64_t>(1, c.n / 65536);
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  std::vector<double> sx, sy, sz;
  for (uint64_t i = 0; i < c.n; i += stride) {
    const Piece& q = c.of(i);
    const double v[3] = {q.x[i - q.first], q.y[i - q.first], q.z[i - q.first]};
    sx.push_back(v[0]); sy.push_back(v[1]); sz.push_back(v[2]);
    for (int a = 0; a < 3; ++a) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
  }
  // the longest axis (fewest targets near a cut) -- among axes within 10 % of each other the LAST one: the grid's macro blocks are laid out
  // x-fastest, and slabs that keep whole x / y rows of them search 8 - 10 % faster than slabs cut along x (tools/rehearse_slabs_c4.py)
  axis = 2;
  for (int a = 1; a >= 0; --a) if (mx[a] - mn[a] > 1.1 * (mx[axis] - mn[axis])) axis = a;
  std::vector<double>& s = axis == 0 ? sx : (axis == 1 ? sy : sz);
  std::sort(s.begin(), s.end());
  bounds.assign((size_t)world + 1, 0.0);
  bounds[0] = -std::numeric_limits<double>::infinity();
  bounds[(size_t)world] = std::numeric_limits<double>::infinity();
  for (int g = 1; g < world; ++g) bounds[(size_t)g] = s.empty() ? 0.0 : s[std::min(s.size() - 1, s.size() * (size_t)g / (size_t)world)];
  for (int g = 1; g <= world; ++g) if (bounds[(size_t)g] < bounds[(size_t)g - 1]) bounds[(size_t)g] = bounds[(size_t)g - 1];
}

inline int run_rank(const Options& o) {
  const int world = o.gpus, rank = o.rank;
  const auto t_start = clk::now();
  auto t_task = clk::now();
  pt_ctx* ctx = nullptr;
  int dev = o.device + rank;
  int rc = pt_ctx_create(&ctx, &dev, 1);                       // the process's first GPU call
  if (rc != PT_OK) { std::cerr << "pointsTransfer[rank " << rank << "]: no usable HIP device " << dev << " (pt_ctx_create returned " << rc << ")" << std::endl; return 1; }
  // error exits ABORT the communicator (pt_comm_abort: ncclCommAbort): a CommDestroy could wait for peers that wait for this rank
  auto die = [&](const char* what) { std::cerr << "pointsTransfer[rank " << rank << "]: " << what << ": " << pt_last_error(ctx) << std::endl; pt_comm_abort(ctx); pt_ctx_destroy(ctx); return 1; };
  {   // communicator: rank 0 creates the id, everybody reads it from the rendezvous directory
    unsigned char id[PT_COMM_ID_BYTES];
    const std::string idf = o.rendezvous + "/rccl_id";
    if (rank == 0) {
      if (pt_comm_unique_id(id) != PT_OK) return die("pt_comm_unique_id (is librccl loadable?)");
      if (!write_all(idf, {{id, sizeof id}})) return die("cannot write the rendezvous file");
    } else {
      bool got = false;
      for (int tries = 0; tries < 6000 && !got; ++tries) {       // up to 60 s
        if (FILE* f = std::fopen(idf.c_str(), "rb")) { got = std::fread(id, 1, sizeof id, f) == sizeof id; std::fclose(f); }
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
      if (!got) return die("timed out waiting for rank 0's RCCL id");
    }
    // (RCCL prints a version banner on stdout when a communicator comes up: stdout carries the reference's report lines and
    //  nothing else, so the banner is sent to stderr)
    std::cout.flush(); std::fflush(stdout);
    const int saved = dup(1);
    dup2(2, 1);
    const int crc = pt_comm_init(ctx, world, rank, id);
    std::fflush(stdout);
    dup2(saved, 1);
    close(saved);
    if (crc != PT_OK) return die("pt_comm_init");
  }
  pt_set_param(ctx, "k_hint", (double)o.K);
  // ---- cloud: this rank parses 1/world of the file and publishes the piece; the slab is then picked from all pieces --------------
  const double wait_s = 900.0;                   // (peers parse as long as this rank does -- a 1e9-point text cloud takes ~12 s per DESIGN 9; a peer that DIES ends the job through the launcher, this bound is for a launcher that was itself killed)
  long declared = 0;
  uint64_t my_first = 0, n_file = 0;
  {
    ply::CloudSoA part;
    std::vector<void*> mem;
    auto free_mem = [&]() { for (void* q : mem) std::free(q); mem.clear(); };
    auto prefix = [&](uint64_t mine, uint64_t& before, uint64_t& total) -> bool {     // text files: the readers' one exchange, token counts
      if (!write_all(o.rendezvous + "/tokens_" + std::to_string(rank), {{&mine, sizeof mine}})) return false;
      before = total = 0;
      for (int r = 0; r < world; ++r) {
        const std::string f = o.rendezvous + "/tokens_" + std::to_string(r);
        if (!wait_for_file(f, wait_s)) return false;
        uint64_t v = 0;
        FILE* fp = std::fopen(f.c_str(), "rb");
        const bool ok = fp && std::fread(&v, 1, sizeof v, fp) == sizeof v;
        if (fp) std::fclose(fp);
        if (!ok) return false;
        if (r < rank) before += v;
        total += v;
      }
      return true;
    };
    const bool opened = ply::read_cloud_soa_part(o.cloud, rank, world, part, my_first, n_file, declared,
                                                 [&](size_t b) { void* q = std::malloc(b ? b : 1); if (q) mem.push_back(q); return q; }, prefix, o.ply_threads);
    if (!opened) { if (rank == 0) std::cerr << "Cannot read or find point cloud file: " << o.cloud << std::endl; free_mem(); pt_comm_abort(ctx); pt_ctx_destroy(ctx); return 3; }
    if (n_file >= 0xFFFFFFF0ull) { free_mem(); pt_comm_abort(ctx); std::cerr << "pointsTransfer[rank " << rank << "]: " << n_file << " points: indices are 32-bit on this path (< 2^32 - 16 points)" << std::endl; pt_ctx_destroy(ctx); return 1; }
    const uint64_t hdr[2] = {my_first, (uint64_t)part.n};
    const bool wrote = write_all(o.rendezvous + "/piece_" + std::to_string(rank) + ".bin",
                                 {{hdr, sizeof hdr}, {part.x, part.n * 8}, {part.y, part.n * 8}, {part.z, part.n * 8}, {part.nrm, part.n * 12}, {part.rgb, part.n * 3}});
    std::cerr << "[pt_hip rank " << rank << "] parsed records [" << my_first << ", " << my_first + part.n << ") of " << n_file << " (1/" << world << " of the file) in "
              << since(t_task) << " s" << std::endl;
    free_mem();
    if (!wrote) {
      std::cerr << "pointsTransfer[rank " << rank << "]: writing " << ((16 + (uint64_t)part.n * 39) >> 20) << " MiB into " << o.rendezvous << " failed (" << std::strerror(errno) << "; "
                << (free_bytes(o.rendezvous) >> 20) << " MiB free there)" << std::endl;
      return die("cannot write this rank's piece of the cloud into the rendezvous directory");
    }
  }
  Pieces cloud;
  if (!cloud.open(o.rendezvous, world, wait_s) || cloud.n != n_file) return die("the pieces of the cloud in the rendezvous directory are incomplete");
  if (rank == 0) {
    std::cout << "PC Point count: " << declared << std::endl;
    std::cout << "Read point set in: " << since(t_task) << " seconds" << std::endl;
  }
  t_task = clk::now();
  int axis = 0;
  std::vector<double> bounds;
  slab_bounds(cloud, world, axis, bounds);
  const double lo = bounds[(size_t)rank], hi = bounds[(size_t)rank + 1];
  // The slab's points in the order of the file: their global indices ascend, so the records can carry POSITIONS in the slab
  // (pt_set_param "local_ids") and the attribute table holds this slab's points only -- 16 bytes x (points / ranks) on every GPU
  // instead of a copy of the whole table; what another slab contributes to a list travels with its record (round 4).
  std::vector<uint32_t> gidx;
  std::vector<double> sx, sy, sz;
  std::vector<uint8_t> srgb;
  std::vector<float> snrm;
  for (const Piece& q : cloud.p) {
    const double* ax = axis == 0 ? q.x : (axis == 1 ? q.y : q.z);
    for (uint64_t i = 0; i < q.count; ++i)
      if (ax[i] >= lo && ax[i] < hi) {
        gidx.push_back((uint32_t)(q.first + i)); sx.push_back(q.x[i]); sy.push_back(q.y[i]); sz.push_back(q.z[i]);
        srgb.insert(srgb.end(), q.rgb + 3 * i, q.rgb + 3 * i + 3); snrm.insert(snrm.end(), q.nrm + 3 * i, q.nrm + 3 * i + 3);
      }
  }
  const size_t ns = gidx.size();
  {
    std::vector<double> sxyz(std::max<size_t>(ns, 1) * 3);
    std::copy(sx.begin(), sx.end(), sxyz.begin()); std::copy(sy.begin(), sy.end(), sxyz.begin() + (long)ns); std::copy(sz.begin(), sz.end(), sxyz.begin() + 2 * (long)ns);
    std::vector<double>().swap(sx); std::vector<double>().swap(sy); std::vector<double>().swap(sz);
    pt_set_param(ctx, "local_ids", 1.0);
    if (pt_build_soa_indexed(ctx, sxyz.data(), PT_F64, gidx.data(), ns, 0) != PT_OK) return die("build failed");
  }
  if (pt_set_attributes_local(ctx, srgb.data(), snrm.data(), 0) != PT_OK) return die("attribute upload failed");
  std::vector<uint8_t>().swap(srgb); std::vector<float>().swap(snrm);
  if (rank == 0) std::cout << "Built Kd tree in: " << since(t_task) << " seconds" << std::endl;
  t_task = clk::now();
  // ---- mesh: the vertices homed in this slab ---------------------------------------------------------------------------------
  ply::FastMesh mesh;
  if (!ply::read_mesh_any(o.mesh, mesh, o.ply_threads)) { if (rank == 0) std::cerr << "Cannot read or find mesh file: " << o.mesh << std::endl; pt_comm_abort(ctx); pt_ctx_destroy(ctx); return 3; }
  if (rank == 0) {
    std::cout << "Mesh vertex count: " << mesh.vertex_count << std::endl;
    std::cout << "Mesh face count: " << mesh.face_count << std::endl;
    std::cout << "Read mesh faces: " << since(t_task) << " seconds" << std::endl;
  }
  t_task = clk::now();
  std::vector<uint32_t> home;
  for (size_t v = 0; v < mesh.vertices.size(); ++v) { const double c = mesh.vertices[v].ver[axis]; if (c >= lo && c < hi) home.push_back((uint32_t)v); }
  const size_t mh = home.size();
  std::vector<double> txyz(std::max<size_t>(mh, 1) * 3);
  for (size_t j = 0; j < mh; ++j) for (int a = 0; a < 3; ++a) txyz[(size_t)a * mh + j] = mesh.vertices[home[j]].ver[a];
  std::vector<uint32_t> idx(mh * (size_t)o.K);
  std::vector<double> d2(mh * (size_t)o.K);
  std::vector<float> rgb(mh * 3), nrm(mh * 3);
  pt_exchange_stats_t xs;
  if (pt_query_exchange_blend(ctx, txyz.data(), PT_F64, mh, o.K, axis, bounds.data(), o.mode, idx.data(), d2.data(), rgb.data(), nrm.data(), &xs) != PT_OK)
    return die("query / exchange failed");
  if (rank == 0) std::cout << "Neighbor search total time: " << since(t_task) << " seconds" << std::endl;
  // ---- hand-over to finalize: lists, blends, and the records of the cloud points the lists name ----------------------------
  std::vector<uint32_t> ref(idx);
  std::sort(ref.begin(), ref.end());
  ref.erase(std::unique(ref.begin(), ref.end()), ref.end());
  while (!ref.empty() && (ref.back() == PT_NOIDX || ref.back() >= cloud.n)) ref.pop_back();
  std::vector<RefPoint> pts(ref.size());
  for (size_t j = 0; j < ref.size(); ++j) {
    const uint32_t id = ref[j];
    const Piece& q = cloud.of(id);
    const uint64_t li = id - q.first;
    pts[j].id = id; pts[j].x = q.x[li]; pts[j].y = q.y[li]; pts[j].z = q.z[li];
    std::memcpy(pts[j].rgb, q.rgb + 3 * li, 3);
  }
  const uint64_t hdr[4] = {(uint64_t)mh, (uint64_t)o.K, (uint64_t)pts.size(), (uint64_t)cloud.n};
  const bool ok = write_all(o.rendezvous + "/rank_" + std::to_string(rank) + ".bin",
                            {{hdr, sizeof hdr}, {home.data(), mh * 4}, {idx.data(), idx.size() * 4}, {d2.data(), d2.size() * 8}, {rgb.data(), rgb.size() * 4},
                             {nrm.data(), nrm.size() * 4}, {pts.data(), pts.size() * sizeof(RefPoint)}});
  std::cerr << "[pt_hip rank " << rank << "] slab " << ns << " points, " << mh << " home vertices, " << xs.crossing << " requests out, " << xs.answered
            << " answered, exchange " << xs.ms << " ms, " << since(t_start) << " s in all" << std::endl;
  pt_comm_destroy(ctx);
  pt_ctx_destroy(ctx);
  return ok ? 0 : 1;
}

template <class WritePly>
inline int run_finalize(const Options& o, WritePly&& write_ply) {
  auto t_task = clk::now();
  ply::FastMesh mesh;
  if (!ply::read_mesh_any(o.mesh, mesh, o.ply_threads)) return 1;
  const size_t M = mesh.vertices.size(), K = (size_t)o.K;
  std::vector<uint32_t> idx(M * K, PT_NOIDX);
  std::vector<float> rgb(M * 3, 0.f), nrm(M * 3, 0.f);
  std::vector<RefPoint> pts;
  for (int r = 0; r < o.gpus; ++r) {
    std::ifstream f(o.rendezvous + "/rank_" + std::to_string(r) + ".bin", std::ios::binary);
    uint64_t hdr[4];
    if (!f.read(reinterpret_cast<char*>(hdr), sizeof hdr) || hdr[1] != K) { std::cerr << "pointsTransfer: rank " << r << " left no usable result" << std::endl; return 1; }
    const size_t mh = (size_t)hdr[0], np = (size_t)hdr[2];
    std::vector<uint32_t> home(mh), li(mh * K);
    std::vector<double> ld(mh * K);
    std::vector<float> lc(mh * 3), ln(mh * 3);
    const size_t base = pts.size();
    pts.resize(base + np);
    f.read(reinterpret_cast<char*>(home.data()), (std::streamsize)(mh * 4)); f.read(reinterpret_cast<char*>(li.data()), (std::streamsize)(li.size() * 4));
    f.read(reinterpret_cast<char*>(ld.data()), (std::streamsize)(ld.size() * 8)); f.read(reinterpret_cast<char*>(lc.data()), (std::streamsize)(lc.size() * 4));
    f.read(reinterpret_cast<char*>(ln.data()), (std::streamsize)(ln.size() * 4)); f.read(reinterpret_cast<char*>(pts.data() + base), (std::streamsize)(np * sizeof(RefPoint)));
    if (!f) { std::cerr << "pointsTransfer: rank " << r << "'s result file is truncated" << std::endl; return 1; }
    for (size_t j = 0; j < mh; ++j) {
      if (home[j] >= M) continue;
      std::memcpy(&idx[(size_t)home[j] * K], &li[j * K], K * 4);
      std::memcpy(&rgb[(size_t)home[j] * 3], &lc[j * 3], 12); std::memcpy(&nrm[(size_t)home[j] * 3], &ln[j * 3], 12);
    }
  }
  // the cloud of referenced points, ascending by original index: the bake's rules (union by original index, ascending) carry over
  std::sort(pts.begin(), pts.end(), [](const RefPoint& a, const RefPoint& b) { return a.id < b.id; });
  pts.erase(std::unique(pts.begin(), pts.end(), [](const RefPoint& a, const RefPoint& b) { return a.id == b.id; }), pts.end());
  const size_t np = pts.size();
  std::vector<double> cxyz(std::max<size_t>(np, 1) * 3);
  std::vector<uint8_t> crgb(std::max<size_t>(np, 1) * 3);
  std::vector<float> cnrm(std::max<size_t>(np, 1) * 3, 0.f);
  for (size_t j = 0; j < np; ++j) { cxyz[j] = pts[j].x; cxyz[np + j] = pts[j].y; cxyz[2 * np + j] = pts[j].z; std::memcpy(&crgb[3 * j], pts[j].rgb, 3); }
  std::vector<uint32_t> local(idx.size());
  for (size_t e = 0; e < idx.size(); ++e) {
    const auto it = std::lower_bound(pts.begin(), pts.end(), idx[e], [](const RefPoint& a, uint32_t id) { return a.id < id; });
    local[e] = (it != pts.end() && it->id == idx[e]) ? (uint32_t)(it - pts.begin()) : PT_NOIDX;
  }
  std::vector<uint8_t> texture;
  if (!o.tex_name.empty()) {
    pt_ctx* ctx = nullptr;
    int dev = o.device;
    int rc = pt_ctx_create(&ctx, &dev, 1);
    if (rc != PT_OK) { std::cerr << "pointsTransfer: no usable HIP device for the texture bake" << std::endl; return 1; }
    rc = pt_build_soa(ctx, cxyz.data(), PT_F64, crgb.data(), cnrm.data(), np, 0);
    texture.resize((size_t)o.resolution * (size_t)o.resolution * 4);
    if (rc == PT_OK)
      rc = pt_bake_texture(ctx, reinterpret_cast<const pt_point*>(mesh.vertices.data()), M, mesh.faces.data(), mesh.faces.size() / 3, local.data(), o.K, o.resolution,
                           o.pad, texture.data());
    if (rc != PT_OK) { std::cerr << "pointsTransfer: texture bake failed: " << pt_last_error(ctx) << std::endl; pt_ctx_destroy(ctx); return 1; }
    pt_ctx_destroy(ctx);
  }
  std::cout << "Draw triangles total time: " << since(t_task) << " seconds" << std::endl;
  t_task = clk::now();
  if (!o.tex_name.empty() && !png::write_bgra(o.tex_name, texture.data(), o.resolution, o.resolution)) { std::cerr << "pointsTransfer: cannot write " << o.tex_name << std::endl; return 1; }
  if (!o.out_name.empty()) write_ply(mesh, rgb, nrm);
  std::cout << "Output time: " << since(t_task) << " seconds" << std::endl;
  return 0;
}

// the launcher: fresh child processes only (this process never initialises HIP).  Children are reaped in the order they END; the
// first one that fails (non-zero exit, a signal) takes the others down -- its peers would otherwise wait for it inside an RCCL
// collective for ever, and the launcher with them: SIGTERM, a grace period, SIGKILL, all of them reaped before returning.
inline int spawn_and_wait(const std::vector<std::vector<std::string>>& cmds, double grace_s = 5.0) {
  std::vector<pid_t> pids;
  int worst = 0;
  for (const auto& cmd : cmds) {
    std::vector<char*> argv;
    for (const std::string& a : cmd) argv.push_back(const_cast<char*>(a.c_str()));
    argv.push_back(nullptr);
    pid_t pid = 0;
    if (posix_spawn(&pid, argv[0], nullptr, nullptr, argv.data(), environ) != 0) { std::cerr << "pointsTransfer: cannot start " << argv[0] << std::endl; worst = 1; break; }
    pids.push_back(pid);
  }
  size_t live = pids.size();
  auto reap = [&](pid_t pid, int st) {
    for (pid_t& q : pids) if (q == pid) { q = -1; --live; }
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if (rc != 0 && worst == 0) worst = rc;                              // the FIRST failure is the one reported (the rest are its victims)
  };
  while (live && worst == 0) {
    int st = 0;
    const pid_t pid = waitpid(-1, &st, 0);
    if (pid < 0) { if (errno == EINTR) continue; worst = 1; break; }
    reap(pid, st);
  }
  if (live) {                                                            // somebody failed: nobody else may be left waiting for it
    std::cerr << "pointsTransfer: a process of the job failed (exit " << worst << "): stopping the other " << live << std::endl;
    for (pid_t q : pids) if (q > 0) kill(q, SIGTERM);
    const auto t0 = clk::now();
    bool killed = false;
    while (live) {
      int st = 0;
      const pid_t pid = waitpid(-1, &st, WNOHANG);
      if (pid > 0) { const int keep = worst; reap(pid, st); worst = keep; continue; }
      if (pid < 0 && errno != EINTR) break;
      if (!killed && since(t0) > grace_s) { for (pid_t q : pids) if (q > 0) kill(q, SIGKILL); killed = true; }
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  return worst;
}

inline int run_launcher(const Options& o, const std::string& self, const std::vector<std::string>& passthrough) {
  // The rendezvous directory receives the parsed cloud (39 bytes per point: 39 GB at 1e9 points) -- under $TMPDIR when that is set
  // (a tmpfs /tmp holds it in RAM), else /tmp; `--rendezvous-root DIR` (o.rendezvous on the launcher) overrides both.
  const char* env = std::getenv("TMPDIR");
  const std::string root = !o.rendezvous.empty() ? o.rendezvous : std::string(env && *env ? env : "/tmp");
  std::string tmpl = root + "/pointsTransfer.XXXXXX";
  if (!mkdtemp(&tmpl[0])) { std::cerr << "pointsTransfer: cannot create a rendezvous directory under " << root << std::endl; return 1; }
  const std::string dir = tmpl;
  std::cerr << "[pt_hip launcher] rendezvous " << dir << std::endl;
  {   // room for the pieces?  The cloud file's size bounds the point count from above (a binary record is >= 27 bytes, a text record >= 18)
    struct stat st;
    if (stat(o.cloud.c_str(), &st) == 0) {
      const uint64_t need = (uint64_t)st.st_size / 18u * 39u / 2u, have = free_bytes(dir);      // (half the worst case: text records are ~60 bytes)
      if (have && need > have) {
        std::cerr << "pointsTransfer: the rendezvous directory " << dir << " has " << (have >> 20) << " MiB free, the parsed cloud may need " << (need >> 20)
                  << " MiB: set TMPDIR or --rendezvous-root to a larger file system" << std::endl;
        rmdir(dir.c_str());
        return 1;
      }
    }
  }
  auto cmd_for = [&](const std::vector<std::string>& extra) {
    std::vector<std::string> c = {self, o.cloud, o.mesh};
    c.insert(c.end(), passthrough.begin(), passthrough.end());
    c.insert(c.end(), {"--gpus", std::to_string(o.gpus), "--rendezvous", dir});
    c.insert(c.end(), extra.begin(), extra.end());
    return c;
  };
  std::vector<std::vector<std::string>> ranks;
  for (int r = 0; r < o.gpus; ++r) ranks.push_back(cmd_for({"--rank", std::to_string(r)}));
  int rc = spawn_and_wait(ranks);
  if (rc == 0) rc = spawn_and_wait({cmd_for({"--finalize"})});
  for (int r = 0; r < o.gpus; ++r)
    for (const char* f : {"/rank_", "/piece_", "/tokens_"}) { std::remove((dir + f + std::to_string(r) + (f[1] == 't' ? "" : ".bin")).c_str()); std::remove((dir + f + std::to_string(r) + (f[1] == 't' ? "" : ".bin") + ".part").c_str()); }
  std::remove((dir + "/rccl_id").c_str());
  rmdir(dir.c_str());
  return rc == 3 ? 0 : rc;          // 3: an input file could not be read -- reported, exit code 0 as the reference (:140, :272)
}

}  // namespace sharded
