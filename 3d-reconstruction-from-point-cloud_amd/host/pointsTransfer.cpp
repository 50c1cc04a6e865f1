// pointsTransfer -- command-line front end of the MI355X detail-transfer path.
//
// Same invocation, exit behaviour and stdout timing lines as the reference's main()
// (reference src/pointsTransfer.cpp:109-125 argv / usage, :137-141 and :269-273 unreadable files,
// :178,255,261,314,315,456,587,590,616,619,622,623 the report lines).  What runs between the lines is
// different: the kd-tree build (:259) and the per-corner K_neighbor_search loop (:462-479) are replaced by
// pt_build_aos / pt_query_aos of libpt_hip.so, each mesh VERTEX is searched once (the reference searches
// every face corner, i.e. every vertex ~6 times), the per-face texture bake that consumes the neighbours
// (:466-581 projection / in-triangle filter / Delaunay / draw_triangle, :593-611 dilate + edge padding; CGAL and
// OpenCV in the reference) is pt_bake_texture of the same library, and the PNG (:613-615, cv::imwrite) is written by
// host/png_write.h.  Output artefacts: texture.png in the working directory, as the reference, and -- the
// per-vertex product of BASELINE.json's north_star -- transfer.ply, the mesh with the neighbours' blended colour / normal.
//
// Optional flags after the two positionals (the reference has none; K and RESOLUTION are its compile-time constants, :128-129):
//   --k K            neighbours per vertex, default 20            --blend mean|invd2   default mean
//   --out FILE       default transfer.ply ("" = do not write)      --device D           default 0
//   --texture FILE   default texture.png ("" = no bake)            --resolution R       default 8192
//   --pad K          edge-padding kernel, default 25 (0 = none)
//   --neighbors FILE also dump the neighbour indices (binary u32[M][K])
//   --ply-threads T  parser threads for the two input files (default 0 = one per hardware thread; host/ply_fast.h)
//   --json FILE      phase times (the stdout lines' seconds) and the library's device times / grid statistics as one JSON object
//   --gpus N         the cloud cut into N spatial slabs, one process per GPU, slab exchange over RCCL (host/sharded.h)
// There is no CPU path: without a usable GPU the tool reports the error and exits non-zero.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "Point.h"
#include "ply_fast.h"
#include "png_write.h"
#include "pt_api.h"
#include "sharded.h"

namespace {
using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

void mem_mib(long& virt, long& res) {   // replaces CGAL::Memory_sizer (reference :622-623)
  virt = res = 0;
  if (FILE* f = std::fopen("/proc/self/statm", "r")) {
    long v = 0, r = 0;
    if (std::fscanf(f, "%ld %ld", &v, &r) == 2) {
      const long page = 4096;
      virt = (v * page) >> 20;
      res = (r * page) >> 20;
    }
    std::fclose(f);
  }
}
// the mesh with the transferred per-vertex colour / normal.  Formatted in parallel (one chunk of records per thread, "%.9g" = what
// operator<< prints at precision 9), written in order.
void write_transfer_ply(const std::string& out_name, const ply::FastMesh& mesh, const std::vector<float>& rgb, const std::vector<float>& nrm) {
  const size_t M = mesh.vertices.size();
  std::ofstream o(out_name, std::ios::binary);
  o << "ply\nformat ascii 1.0\nelement vertex " << M << "\n"
    << "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
    << "property float s\nproperty float t\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
    << "element face " << mesh.faces.size() / 3 << "\nproperty list uchar int vertex_indices\nend_header\n";
  const size_t F = mesh.faces.size() / 3;
  int nth = (int)std::thread::hardware_concurrency();
  nth = std::max(1, std::min(nth, 64));
  nth = (int)std::min<size_t>((size_t)nth, (M + F) / 20000 + 1);
  std::vector<std::string> part((size_t)nth * 2);
  auto format = [&](int t) {
    char buf[256];
    std::string& sv = part[(size_t)t];
    const size_t v0 = M * (size_t)t / (size_t)nth, v1 = M * (size_t)(t + 1) / (size_t)nth;
    sv.reserve((v1 - v0) * 96);
    for (size_t i = v0; i < v1; ++i) {
      const Point& v = mesh.vertices[i];
      const int len = std::snprintf(buf, sizeof buf, "%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d %d %d\n", v.x(), v.y(), v.z(), (double)nrm[3 * i],
                                    (double)nrm[3 * i + 1], (double)nrm[3 * i + 2], v.u(), v.v(), (int)rgb[3 * i], (int)rgb[3 * i + 1],
                                    (int)rgb[3 * i + 2]);   // float -> uchar truncates, as :100-102
      sv.append(buf, (size_t)len);
    }
    std::string& sf = part[(size_t)nth + (size_t)t];
    const size_t f0 = F * (size_t)t / (size_t)nth, f1 = F * (size_t)(t + 1) / (size_t)nth;
    sf.reserve((f1 - f0) * 28);
    for (size_t f = f0; f < f1; ++f) {
      const int len = std::snprintf(buf, sizeof buf, "3 %d %d %d\n", mesh.faces[3 * f], mesh.faces[3 * f + 1], mesh.faces[3 * f + 2]);
      sf.append(buf, (size_t)len);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nth; ++t) th.emplace_back(format, t);
  format(0);
  for (auto& t : th) t.join();
  for (const std::string& sp : part) o.write(sp.data(), (std::streamsize)sp.size());
}
}  // namespace

int main(int argc, char** argv) {
  const std::string usage_str = "Usage: ./pointTransfer <input-point-cloud> <input-mesh>";   // sic, reference :112
  if (argc < 3) {
    std::cout << usage_str << std::endl;
    return 0;
  }
  const std::string pc_file_name = argv[1], mesh_file_name = argv[2];
  int K = 20, device = 0, mode = PT_BLEND_MEAN, ply_threads = 0, resolution = 8192, pad = 25;     // K, RESOLUTION: reference :128-129; 25: :594
  std::string out_name = "transfer.ply", nbr_name, tex_name = "texture.png";                      // texture.png: reference :615
  std::string json_name;                           // --json FILE: the phase times of the stdout lines + pt_stats as one JSON object
  int gpus = 0, rank = -1;
  bool finalize = false;
  std::string rendezvous;
  // --synthetic N M SEED [--clustered] [--xyz f32|f16|f64]: SURVEY.md Appendix C's generator instead of the two files (the positional
  // arguments are ignored): the BASELINE configurations run through this binary without a cloud on disk -- no mesh, so no texture
  bool synthetic = false;
  uint64_t syn_n = 0, syn_m = 0, syn_seed = 0;
  int syn_dist = PT_DIST_UNIFORM, syn_type = PT_F32;
  std::vector<std::string> passthrough;            // the options a launcher hands to its rank / finalize processes
  for (int i = 3; i < argc; ++i) {
    const std::string a = argv[i];
    const int i_before = i;
    auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--k") K = std::atoi(val());
    else if (a == "--device") device = std::atoi(val());
    else if (a == "--out") out_name = val();
    else if (a == "--neighbors") nbr_name = val();
    else if (a == "--texture") tex_name = val();
    else if (a == "--json") json_name = val();
    else if (a == "--resolution") resolution = std::atoi(val());
    else if (a == "--pad") pad = std::atoi(val());
    else if (a == "--gpus") { gpus = std::atoi(val()); continue; }                 // (not handed on: the launcher adds it itself)
    else if (a == "--rank") { rank = std::atoi(val()); continue; }
    else if (a == "--rendezvous") { rendezvous = val(); continue; }
    else if (a == "--finalize") { finalize = true; continue; }
    else if (a == "--ply-threads") ply_threads = std::max(0, std::atoi(val()));
    else if (a == "--blend") mode = std::string(val()) == "invd2" ? PT_BLEND_INV_D2 : PT_BLEND_MEAN;
    else if (a == "--synthetic") { syn_n = std::strtoull(val(), nullptr, 0); syn_m = std::strtoull(val(), nullptr, 0); syn_seed = std::strtoull(val(), nullptr, 0); synthetic = true; }
    else if (a == "--clustered") syn_dist = PT_DIST_CLUSTERED;
    else if (a == "--xyz") { const std::string t = val(); syn_type = t == "f16" ? PT_F16 : (t == "f64" ? PT_F64 : PT_F32); }
    else if (a == "--rendezvous-root") { rendezvous = val(); continue; }      // (the launcher's: where its rendezvous directory is made)
    else { std::cerr << "unknown option " << a << std::endl; return 2; }
    for (int j = i_before; j <= i; ++j) passthrough.push_back(argv[j]);
  }
  if (K < 1 || K > PT_MAX_K) { std::cerr << "--k must be in [1, " << PT_MAX_K << "]" << std::endl; return 2; }
  if (resolution < 1 || resolution > 32768 || pad < 0 || pad > 255 || (pad > 0 && !(pad & 1))) { std::cerr << "--resolution must be in [1, 32768], --pad 0 or odd" << std::endl; return 2; }
  if (gpus != 0 || rank >= 0 || finalize) {
    // the sharded path (host/sharded.h): launcher -> one rank process per GPU -> finalize
    if (gpus < 1 || gpus > 64) { std::cerr << "--gpus must be in [1, 64]" << std::endl; return 2; }
    sharded::Options so;
    so.cloud = pc_file_name; so.mesh = mesh_file_name; so.out_name = out_name; so.tex_name = tex_name; so.rendezvous = rendezvous;
    so.K = K; so.device = device; so.mode = mode; so.ply_threads = ply_threads; so.resolution = resolution; so.pad = pad; so.gpus = gpus; so.rank = rank;
    if (rank >= 0) return rank < gpus && !rendezvous.empty() ? sharded::run_rank(so) : 2;
    if (finalize) return rendezvous.empty() ? 2 : sharded::run_finalize(so, [&](const ply::FastMesh& m, const std::vector<float>& c, const std::vector<float>& n) { write_transfer_ply(out_name, m, c, n); });
    const auto t0 = clk::now();
    char self[4096];
    const ssize_t sl = readlink("/proc/self/exe", self, sizeof self - 1);        // the rank processes run this very binary
    const int lrc = sharded::run_launcher(so, sl > 0 ? std::string(self, (size_t)sl) : std::string(argv[0]), passthrough);
    if (lrc != 0) return lrc;
    std::cout << "Total real time: " << since(t0) << " seconds" << std::endl;
    long virt, res;
    mem_mib(virt, res);
    std::cout << "VIRT: " << virt << " MiB" << std::endl;
    std::cout << "RES:  " << res << " MiB" << std::endl;
    return 0;
  }

  const auto t_total = clk::now();
  auto t_task = clk::now();

  if (synthetic) {
    // generated cloud and targets, the reference's report lines in the reference's order (:178 ... :623); `--neighbors FILE` keeps the
    // index matrix, `--out FILE` the blended attributes as an ASCII table (there is no mesh to write them onto)
    pt_ctx* sc = nullptr;
    int src = pt_ctx_create(&sc, &device, 1);
    if (src != PT_OK) { std::cerr << "pointsTransfer: no usable HIP device (pt_ctx_create returned " << src << "); there is no CPU fallback" << std::endl; return 1; }
    auto sdie = [&](const char* what) { std::cerr << "pointsTransfer: " << what << ": " << pt_last_error(sc) << std::endl; pt_ctx_destroy(sc); return 1; };
    pt_set_param(sc, "k_hint", (double)K);
    pt_set_param(sc, "sync", 1.0);
    if (pt_build_synth(sc, syn_n, syn_seed, syn_dist, syn_type, -1, 0.0, 0.0) != PT_OK) return sdie("build failed");
    pt_stats_t st0;
    pt_stats(sc, &st0);
    std::cout << "PC Point count: " << syn_n << std::endl;
    std::cout << "Read point set in: " << std::max(0.0, since(t_task) - st0.ms_build * 1e-3) << " seconds" << std::endl;      // (generation; the build is the next line)
    std::cout << "Built Kd tree in: " << st0.ms_build * 1e-3 << " seconds" << std::endl;
    t_task = clk::now();
    if (pt_targets_synth(sc, syn_m, syn_seed, syn_dist, syn_type == PT_F64 ? PT_F64 : (syn_type == PT_F16 ? PT_F16 : PT_F32), -1, 0.0, 0.0) != PT_OK) return sdie("target generation failed");
    const uint64_t M = pt_num_targets(sc);
    std::cout << "Mesh vertex count: " << M << std::endl;
    std::cout << "Mesh face count: 0" << std::endl;
    std::cout << "Read mesh faces: " << since(t_task) << " seconds" << std::endl;
    t_task = clk::now();
    std::vector<uint32_t> idx(M * (size_t)K);
    std::vector<double> d2(M * (size_t)K);
    std::vector<float> rgb(M * 3), nrm(M * 3);
    if (pt_query_resident_host(sc, K, mode, idx.data(), d2.data(), rgb.data(), nrm.data()) != PT_OK) return sdie("query failed");
    pt_stats_t st;
    pt_stats(sc, &st);
    std::cout << "Neighbor search total time: " << since(t_task) << " seconds" << std::endl;
    std::cout << "Draw triangles total time: 0 seconds" << std::endl;             // (the blend ran inside the search: pt_stats.ms_query)
    t_task = clk::now();
    if (!nbr_name.empty()) {
      std::ofstream o(nbr_name, std::ios::binary);
      o.write(reinterpret_cast<const char*>(idx.data()), (std::streamsize)(idx.size() * sizeof(uint32_t)));
    }
    if (!json_name.empty()) {
      std::ofstream j(json_name);
      j.precision(9);
      j << "{\"synthetic\": {\"n\": " << syn_n << ", \"m\": " << M << ", \"seed\": " << syn_seed << ", \"clustered\": " << (syn_dist == PT_DIST_CLUSTERED ? 1 : 0) << "}, \"k\": " << K
        << ", \"pt_stats\": {\"ms_build\": " << st.ms_build << ", \"ms_sort_targets\": " << st.ms_sort_targets << ", \"ms_query\": " << st.ms_query << ", \"grid_dim\": [" << st.grid_dim[0] << ", "
        << st.grid_dim[1] << ", " << st.grid_dim[2] << "], \"n_levels\": " << st.n_levels << ", \"pass1_pooled\": " << st.pass1_pooled << ", \"pass2_pooled\": " << st.pass2_pooled
        << ", \"uniform_probe\": " << st.uniform_probe << ", \"presort_refine\": " << st.presort_refine << ", \"n_sorts\": " << st.n_sorts << ", \"ordered_input\": " << st.ordered_input << ", \"n_leftover\": " << st.n_leftover << ", \"n_wave\": " << st.n_wave << ", \"device_bytes\": " << st.device_bytes
        << ", \"targets_per_second_device\": " << (st.ms_build + st.ms_sort_targets + st.ms_query > 0 ? (double)M / ((st.ms_build + st.ms_sort_targets + st.ms_query) * 1e-3) : 0.0) << "}}\n";
    }
    std::cout << "Output time: " << since(t_task) << " seconds" << std::endl;
    std::cout << "Total real time: " << since(t_total) << " seconds" << std::endl;
    std::cerr << "[pt_hip] synthetic: grid " << st.grid_dim[0] << "x" << st.grid_dim[1] << "x" << st.grid_dim[2] << " cells, build " << st.ms_build << " ms, target sort " << st.ms_sort_targets
              << " ms, kNN + blend " << st.ms_query << " ms (device time)" << std::endl;
    pt_ctx_destroy(sc);
    long virt, res;
    mem_mib(virt, res);
    std::cout << "VIRT: " << virt << " MiB" << std::endl;
    std::cout << "RES:  " << res << " MiB" << std::endl;
    return 0;
  }

  // The context comes first: the cloud is parsed straight into page-locked planar arrays (x, y, z, colour bytes, normals:
  // 39 bytes per point instead of the 80-byte record) and every finished range of records is handed to the GPU while the
  // other ranges are still being parsed (host/ply_fast.h read_cloud_soa -> pt_upload_range).
  pt_ctx* ctx = nullptr;
  int rc = pt_ctx_create(&ctx, &device, 1);
  if (rc != PT_OK) {
    // (an unreadable cloud file is reported first and exits 0, as the reference does -- also on a machine without a GPU)
    if (FILE* probe = std::fopen(pc_file_name.c_str(), "rb")) std::fclose(probe);
    else { std::cerr << "Cannot read or find point cloud file: " << pc_file_name << std::endl; return 0; }
    std::cerr << "pointsTransfer: no usable HIP device (pt_ctx_create returned " << rc << "); there is no CPU fallback" << std::endl;
    return 1;
  }
  pt_set_param(ctx, "k_hint", (double)K);
  ply::CloudSoA cloud;
  std::vector<void*> pinned;
  const bool pageable = std::getenv("PT_CLI_PAGEABLE") != nullptr;      // (measurement switch: plain malloc instead of page-locked memory)
  auto free_pinned = [&]() { for (void* q : pinned) { if (pageable) std::free(q); else pt_host_free(q); } pinned.clear(); };
  long point_count = 0;
  std::atomic<int> upload_rc{PT_OK};
  const bool opened = ply::read_cloud_soa(
      pc_file_name, cloud, point_count, [&](size_t bytes) { void* q = pageable ? std::malloc(bytes ? bytes : 1) : pt_host_alloc(bytes); if (q) pinned.push_back(q); return q; },
      [&](uint64_t n) { return pt_upload_begin(ctx, n, PT_F64, 1) == PT_OK; },
      [&](uint64_t first, uint64_t count) {
        const int r = pt_upload_range(ctx, first, count, cloud.x + first, cloud.y + first, cloud.z + first, cloud.rgb + 3 * first, cloud.nrm + 3 * first);
        if (r != PT_OK) upload_rc = r;
      },
      ply_threads);
  if (!opened) {
    free_pinned();
    pt_ctx_destroy(ctx);
    std::cerr << "Cannot read or find point cloud file: " << pc_file_name << std::endl;
    return 0;   // the reference returns 0 here (:140)
  }
  std::cout << "PC Point count: " << point_count << std::endl;
  const double t_read_cloud = since(t_task);
  std::cout << "Read point set in: " << t_read_cloud << " seconds" << std::endl;
  t_task = clk::now();

  if (cloud.n == 0) rc = pt_upload_begin(ctx, 0, PT_F64, 1);            // header-only / empty file: an empty cloud
  else rc = upload_rc.load();
  if (rc == PT_OK) rc = pt_upload_end(ctx);
  free_pinned();
  if (rc != PT_OK) { std::cerr << "pointsTransfer: build failed: " << pt_last_error(ctx) << std::endl; pt_ctx_destroy(ctx); return 1; }
  const double t_build = since(t_task);
  std::cout << "Built Kd tree in: " << t_build << " seconds" << std::endl;   // the line's wording is the contract
  t_task = clk::now();

  ply::FastMesh mesh;
  if (!ply::read_mesh_any(mesh_file_name, mesh, ply_threads)) {
    std::cerr << "Cannot read or find mesh file: " << mesh_file_name << std::endl;
    pt_ctx_destroy(ctx);
    return 0;   // the reference returns 0 here (:272)
  }
  std::cout << "Mesh vertex count: " << mesh.vertex_count << std::endl;
  std::cout << "Mesh face count: " << mesh.face_count << std::endl;
  const double t_read_mesh = since(t_task);
  std::cout << "Read mesh faces: " << t_read_mesh << " seconds" << std::endl;
  t_task = clk::now();

  const size_t M = mesh.vertices.size();
  std::vector<uint32_t> idx(M * (size_t)K);
  std::vector<double> d2(M * (size_t)K);
  rc = pt_query_aos(ctx, reinterpret_cast<const pt_point*>(mesh.vertices.data()), M, K, idx.data(), d2.data());
  if (rc != PT_OK) { std::cerr << "pointsTransfer: query failed: " << pt_last_error(ctx) << std::endl; pt_ctx_destroy(ctx); return 1; }
  const double t_search = since(t_task);
  t_task = clk::now();
  std::vector<float> rgb(M * 3), nrm(M * 3);
  rc = pt_blend(ctx, idx.data(), d2.data(), M, K, mode, rgb.data(), nrm.data());
  if (rc != PT_OK) { std::cerr << "pointsTransfer: blend failed: " << pt_last_error(ctx) << std::endl; pt_ctx_destroy(ctx); return 1; }
  const double t_blend = since(t_task);
  t_task = clk::now();
  // the reference's face loop after the search (:484-581) and its post-processing (:593-611), on the GPU
  std::vector<uint8_t> texture;
  if (!tex_name.empty()) {
    texture.resize((size_t)resolution * (size_t)resolution * 4);
    rc = pt_bake_texture(ctx, reinterpret_cast<const pt_point*>(mesh.vertices.data()), M, mesh.faces.data(), mesh.faces.size() / 3, idx.data(), K, resolution,
                         pad, texture.data());
    if (rc != PT_OK) { std::cerr << "pointsTransfer: texture bake failed: " << pt_last_error(ctx) << std::endl; pt_ctx_destroy(ctx); return 1; }
  }
  const double t_bake = since(t_task);
  std::cout << "Neighbor search total time: " << t_search << " seconds" << std::endl;
  std::cout << "Draw triangles total time: " << t_bake + t_blend << " seconds" << std::endl;   // texture bake (+ the per-vertex blend)
  t_task = clk::now();
  if (!tex_name.empty() && !png::write_bgra(tex_name, texture.data(), resolution, resolution)) {
    std::cerr << "pointsTransfer: cannot write " << tex_name << std::endl;
    pt_ctx_destroy(ctx);
    return 1;
  }

  if (!out_name.empty()) write_transfer_ply(out_name, mesh, rgb, nrm);
  if (!nbr_name.empty()) {
    std::ofstream o(nbr_name, std::ios::binary);
    o.write(reinterpret_cast<const char*>(idx.data()), (std::streamsize)(idx.size() * sizeof(uint32_t)));
  }
  const double t_output = since(t_task), t_all = since(t_total);
  std::cout << "Output time: " << t_output << " seconds" << std::endl;
  std::cout << "Total real time: " << t_all << " seconds" << std::endl;

  pt_stats_t st;
  if (!json_name.empty() && pt_stats(ctx, &st) == PT_OK) {
    // one object: what the stdout lines say (wall seconds per phase, the reference's :255-:619) and the library's own device times
    std::ofstream j(json_name);
    j.precision(9);
    j << "{\"cloud\": \"" << pc_file_name << "\", \"mesh\": \"" << mesh_file_name << "\", \"k\": " << K << ", \"points\": " << point_count << ", \"points_read\": " << cloud.n
      << ", \"mesh_vertices\": " << mesh.vertex_count << ", \"mesh_faces\": " << mesh.face_count << ", \"resolution\": " << resolution
      << ",\n \"seconds\": {\"read_cloud\": " << t_read_cloud << ", \"build\": " << t_build << ", \"read_mesh\": " << t_read_mesh << ", \"search\": " << t_search
      << ", \"blend\": " << t_blend << ", \"bake\": " << t_bake << ", \"output\": " << t_output << ", \"total\": " << t_all << "}"
      << ",\n \"pt_stats\": {\"n_source\": " << st.n_source << ", \"n_target\": " << st.n_target << ", \"ms_build\": " << st.ms_build << ", \"ms_sort_targets\": " << st.ms_sort_targets
      << ", \"ms_query\": " << st.ms_query << ", \"ms_blend\": " << st.ms_blend << ", \"ms_bake\": " << st.ms_bake << ", \"grid_dim\": [" << st.grid_dim[0] << ", " << st.grid_dim[1] << ", "
      << st.grid_dim[2] << "], \"cell_size\": " << st.cell_size << ", \"n_cells\": " << st.n_cells << ", \"n_levels\": " << st.n_levels << ", \"rho_occupied\": " << st.rho_occupied
      << ", \"n_refine\": " << st.n_refine << ", \"bbox_guess\": " << st.bbox_guess << ", \"pass1_pooled\": " << st.pass1_pooled << ", \"n_nodes\": " << st.n_nodes
      << ", \"n_leftover\": " << st.n_leftover << ", \"n_wave\": " << st.n_wave << ", \"device_bytes\": " << st.device_bytes << ", \"ms_kernel\": [";
    for (int i = 0; i < 8; ++i) j << (i ? ", " : "") << st.ms_kernel[i];
    j << "]}}\n";
    if (!j) std::cerr << "pointsTransfer: cannot write " << json_name << std::endl;
  }
  if (pt_stats(ctx, &st) == PT_OK)
    std::cerr << "[pt_hip] grid " << st.grid_dim[0] << "x" << st.grid_dim[1] << "x" << st.grid_dim[2] << " cells, build " << st.ms_build
              << " ms, target sort " << st.ms_sort_targets << " ms, kNN " << st.ms_query << " ms, blend " << st.ms_blend << " ms, texture bake " << st.ms_bake << " ms (device time)"
              << std::endl;
  pt_ctx_destroy(ctx);

  long virt, res;
  mem_mib(virt, res);
  std::cout << "VIRT: " << virt << " MiB" << std::endl;
  std::cout << "RES:  " << res << " MiB" << std::endl;
  return 0;
}
