// pt_api.hip -- the extern "C" boundary of libpt_hip.so (declared in include/pt_api.h) and the
// context that owns every HBM allocation of the detail-transfer path on one MI355X.
//
// Call sites this replaces in the reference (INTEGRATION.md has the patch):
//   pt_build_*   <->  Tree tree(points.begin(), points.end());           src/pointsTransfer.cpp:259
//   pt_query_*   <->  K_neighbor_search search(tree, v, K) + iteration    src/pointsTransfer.cpp:462-479
//   pt_stats     <->  the timer lines                                     src/pointsTransfer.cpp:261,587
// There is no CPU fallback anywhere in this file: without a usable HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: librccl is dlopen'ed by pt_comm_* (no link-time dependency)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_internal.h"

#include <hip/hip_fp16.h>

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct pt_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t sev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per-kernel marks of the source sort
  std::string err;
  double rho = 4.0;            // points per cell: 10^3-cell regions of ~4000 records let TWO tile workgroups share a CU
                               // (k-NN at C4: rho 8 -> 39.8 ms, 6 -> 32.4 ms, 4 with the small geometry -> 23.2 ms)
  int sync = 1;
  int adaptive = 1;            // refine the cell size when the occupied cells hold far more than rho points (non-uniform clouds)
  int tile = 1;                // 1: tile kernel + group kernel for leftovers (fp32, unbounded); 0: group kernel only
  int dup_runs = 1;            // leaves of refined cells that hold one position many times keep their PT_DUP_KEEP lowest indices in front ("dup_runs", 0: off -- a measurement switch)
  int tile_sparse = 2;         // the tile kernel over a LIST of the blocks that hold targets: 0 never, 1 always, 2 on clouds that leave most of their grid empty
  int tile_contrast = 0;       // clouds with strong density contrast: 0 every target gets a wave (round 2), 1 the tile kernel first (k <= 24), what it cannot settle gets a wave
  size_t dev_bytes = 0;

  // source cloud (slab-local when built from a slab)
  int src_type = -1;           // PT_F32 / PT_F64
  uint64_t n = 0, n_total = 0; // resident points; size of the attribute table
  DevBuf in_xyz, in_gidx, attr, rec, rec_tmp, cell_start;
  bool in_half = false;          // fp16 clouds: in_xyz holds the coordinates as fp16 (src_type says PT_F32: that is what they are sorted into)
  DevBuf xyz32;                  // ... and their fp32 image, made on demand for the few consumers of planar fp32 coordinates (PCA table, bake)
  bool xyz32_valid = false;
  DevBuf rec32;                // fp64 clouds: fp32 shadow of the sorted records (id = sorted position), the tile kernel's LDS image
  bool rec32_valid = false;
  float e_src = 0.f;           // fp64 clouds: largest rounding error of a source coordinate stored as fp32
  DevBuf posattr;              // fp32 clouds: {position, attributes} by original index for the PCA pass, built on first use
  bool has_gidx = false, has_attr = false, built = false, posattr_valid = false;
  // slabs (round 4): "local_ids" asked for AND the slab's gidx strictly ascending -> the records carry the point's POSITION in the slab's arrays
  // (same order as the global index), the attribute table may be the slab's own n records (attr_local), finished lists are translated through gidx
  bool want_local_ids = false, local_mode = false, attr_local = false;
  uint64_t synth_total = 0;    // a generated slab in local mode: the GENERATOR's point count (n_total is the slab's own then: its attribute table's length), which the clustered target generator needs
  uint64_t guess_min_points = 8u << 20;   // clouds at least this large lay their grid out from a sampled bounding box
  bool bbox_guess_ok = true;   // big clouds: lay the grid out from a sampled bounding box (cleared when a guess failed; reset by an upload)
  bool stream_bounds = true;   // pt_stream_query: later chunks are searched under the targets' current k-th distances and skipped when out of reach ("stream_bounds", a measurement switch)
  bool tile_bounds = false;    // run_query: bounds come with every target of the set (the tile kernel's bounded variant may take them)
  bool uniform_known = false;  // ... or a 1/64 sample taken before this cloud's first sort has answered the question (pt_grid.hip, uniform_probe_kernel)
  bool pool2_ok = true, pool2 = true, uniform_seen = false;   // pass 2 without its histogram: allowed ("pool2"), not failed yet on this cloud, and the last build of
                               // this resident cloud found it uniform (reset by an upload together with pool_ok)
  bool pool_ok = true;         // big clouds, two-level sorts: pass 1 without its histogram pass (cleared when a bin outgrew its sampled region; reset by an upload)
  int presort_refine = 1;                 // first builds of big non-uniform clouds refine the cell size from the sample, before the first sort ("presort_refine")
  uint64_t pool_min_points = 32u << 20;   // ... from this size up ("pool_min_points"; 0 switches the pooled pass 1 off)
  int n_cu = 256;              // compute units of the device: persistent workgroups of the pooled pass 1
  GridParams gp{};
  SortTables stb{};
  DevBuf stb_mem;

  // resident targets
  int tgt_type = -1;
  uint64_t m = 0;
  DevBuf t_xyz, t_gidx, trec, trec_tmp;
  DevBuf x_xyz;                // transient targets of pt_query_soa / _aos / _bounded_dev (resident targets stay untouched)
  bool t_has_gidx = false;
  // refinement of heavy cells (pt_refine.hip): sub-grids inside cells with more than refine_threshold points
  double refine_cpp = 2.0;              // ... and, in cells per point ("refine_cells_per_point")
  int refine_macros = PT_MAXBINS;       // finest grid the occupancy-driven refinement of h may ask for, in macro blocks (measured on the clustered
                                        // generator: beyond 1024 the extra sort pass costs more than the shorter scans save)
  // rebuilds of the SAME resident cloud (pt_rebuild) start from the cell size the last build ended with instead of searching for it again
  // (a refined grid costs one full sort per step of the search); the occupancy finalize reports checks the guess, a new cloud resets it
  double hint_h = 0.0, hint_rho_occ = 0.0;
  int hint_refines = 0, grid_hint = 1;
  bool grid_capped = false;             // the last choose_grid ran into the macro-block limit: no finer grid exists
  int wave_force = 0;                   // 1: the heavy / light split also on clouds without density contrast (tests, tuning)
  uint32_t wave_min = 1;                // targets with at least this many points in their 27 nearest cells get a wave each (0: never; 1: all of
                                        // them, and the group kernel is not run at all -- the default: 1B clustered / k=32 469 ms at 512, 448 at 1 before that shortcut)
  double refine_threshold = 8192.0;     // 0: never refine.  Measured on the clustered generator (tools/probe_wave.py) with the wave kernel taking the dense
                                        // neighbourhoods: a descent costs several dependent memory round trips, a scan of 64 records per step does
                                        // not, so only cells of many thousands of points are worth a sub-grid -- 1B / 50M / k = 32: 675 ms unrefined,
                                        // 585 / 565 / 553 / 554 at 2048 / 4096 / 8192 / 16384; 100M / 5M / k = 8: 20.3 / 22.1 / 21.9 / 21.2 / 21.1 ms
  DevBuf cell_node, nodes, near_node;   // near_node: one byte per cell, set for the 27 cells around every level-0 node
  uint32_t n_nodes = 0, refine_levels = 0;
  // slab exchange (pt_comm_* / pt_exchange_*)
  void* nccl_comm = nullptr;
  hipEvent_t xev[2] = {nullptr, nullptr};                     // exchange timing (run_query uses ev[0..2] itself)
  int world = 1, rank = 0;
  DevBuf x_bounds, x_counts, x_matrix, x_off, x_req, x_row, x_rreq, x_rxyz, x_rbound, x_ans_i, x_ans_d, x_back_i, x_back_d, x_flags, x_rows;
  DevBuf x_ans_a, x_back_a, x_rattr, l_idx;                 // sharded attributes: the candidates' records out and back, the merged rows' records; a list translated back to local ids
  std::vector<uint32_t> x_send, x_recv, x_soff, x_roff;      // per peer: packets to send / to answer, and their offsets
  uint32_t* h_matrix = nullptr;                               // pinned, world * world
  uint32_t* h_xoff = nullptr;                                 // pinned staging of the send offsets (xb_fill)
  hipEvent_t xoff_ev = nullptr;                               // recorded behind the copy that reads h_xoff
  bool comm_failed = false;                                   // an RCCL call failed: teardown aborts instead of waiting for peers
  // streamed upload (pt_upload_*)
  uint64_t up_n = 0;
  int up_type = -1, up_attr = 0;
  DevBuf up_rgb, up_nrm;
  SortTables ttb{};
  DevBuf ttb_mem;

  // scratch
  DevBuf bbox6, counter, q_idx, q_d2, b_rgb, b_nrm, aos_stage, misc, bounds, todo, retry, heavy, tlist;
  uint64_t* h_bbox = nullptr;   // pinned
  uint32_t* h_counter = nullptr;

  pt_stats_t st{};
};

namespace {

int fail(pt_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    c->err = buf;
  }
  return code;
}
#define HIPCHK(c, call)                                                                                     \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) return fail((c), PT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

int reserve(pt_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return PT_OK;
  if (b.p) { (void)hipFree(b.p); c->dev_bytes -= b.cap; b.p = nullptr; b.cap = 0; }
  if (bytes == 0) bytes = 16;
  bytes = (bytes + 255) & ~(size_t)255;
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) { b.p = nullptr; return fail(c, PT_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
  b.cap = bytes;
  c->dev_bytes += bytes;
  return PT_OK;
}
void release(pt_ctx* c, DevBuf& b) {
  if (b.p) { (void)hipFree(b.p); c->dev_bytes -= b.cap; }
  b.p = nullptr; b.cap = 0;
}
#define RES(c, buf, bytes)                       \
  do {                                           \
    int r_ = reserve((c), (buf), (bytes));       \
    if (r_ != PT_OK) return r_;                  \
  } while (0)

size_t tsize(int t) { return t == PT_F64 ? 8 : 4; }
size_t recsize(int t) { return t == PT_F64 ? sizeof(RecD) : sizeof(RecF); }

int finish(pt_ctx* c) {
  if (c->sync) HIPCHK(c, hipStreamSynchronize(c->stream));
  return PT_OK;
}

// carve the SortTables of one sort out of a single allocation
int make_tables(pt_ctx* c, DevBuf& mem, SortTables& tb, uint32_t nblocks, uint32_t npoints, size_t rec_size, uint64_t pool_records = 0, uint64_t pool2_records = 0) {
  const size_t small = PT_MAXBINS + 8;
  const size_t words = small * 4 /*counts1,start1,cursor1,tile_first2*/ + 16 /*tile_first1, seg_start1*/ +
                       ((size_t)nblocks + 8) * 3 + ((size_t)nblocks / 2048 + 16);
  const int gsh = pt_sort_group_shift(nblocks);                                            // > 0: three-level sort
  const size_t nmac = nblocks / PT_MACRO_BLOCKS, ngrp = (nmac + ((size_t)1 << gsh) - 1) >> gsh, nmacP = ngrp << gsh;
  const size_t nchunks = pt_sort_num_chunks(npoints, rec_size), nbins1 = ngrp + 1;
  const size_t chunk_words = nblocks > PT_MAXBINS ? (nchunks + 1) * nbins1 + (nchunks / 64 + 2) * nbins1 : 0;
  const size_t bid_points = std::max<uint64_t>(npoints, pool_records);             // (the pooled pass 1 writes ids for its slack and scratch area too)
  const size_t bid_words = nblocks > PT_MAXBINS && !gsh && !pool2_records ? (bid_points + 3) / 2 + 12 : 0;     // (a pooled pass 2 reads no block ids)     // u16 per point, two-level sorts only; 16-byte aligned, 16 bytes of slack (read 8 at a time)
  const size_t mac_words = gsh ? (nmacP + 8) * 4 : 0;
  const size_t rstart_words = pool2_records ? (size_t)nblocks + 8 : 0;
  RES(c, mem, (words + chunk_words + bid_words + mac_words + 16 + 2 * small + 8 + rstart_words) * sizeof(uint32_t));
  uint32_t* p = (uint32_t*)mem.p;
  tb.counts1 = p; p += small;
  tb.start1 = p; p += small;
  tb.cursor1 = p; p += small;
  tb.tile_first2 = p; p += small;
  tb.tile_first1 = p; p += 8;
  tb.seg_start1 = p; p += 8;
  tb.block_count = p; p += (size_t)nblocks + 8;
  tb.block_start = p; p += (size_t)nblocks + 8;
  tb.cursor2 = p; p += (size_t)nblocks + 8;
  tb.scan_tmp = p; p += (size_t)nblocks / 2048 + 16;
  tb.chunk_hist = tb.chunk_gsum = nullptr;                   // (chunked pass 1: grids of more than PT_MAXBINS blocks only -- as chunk_words counts them;
  if (chunk_words) { tb.chunk_hist = p; p += (nchunks + 1) * nbins1; tb.chunk_gsum = p; p += (nchunks / 64 + 2) * nbins1; }     //  carved unconditionally they ran past the allocation on one-level grids)
  if (bid_words) p += (4 - ((uintptr_t)p / sizeof(uint32_t)) % 4) % 4;                  // (at most 3 of bid_words' spare words)
  tb.bid = bid_words ? (uint16_t*)p : nullptr;
  p += bid_words ? bid_words - 4 : 0;
  tb.countsM = tb.startM = tb.cursorM = tb.tile_firstM = nullptr;
  if (gsh) { tb.countsM = p; p += nmacP + 8; tb.startM = p; p += nmacP + 8; tb.cursorM = p; p += nmacP + 8; tb.tile_firstM = p; p += nmacP + 8; }
  tb.pool_est = p; p += small; tb.pool_limit = p; p += small; tb.pool_flag = p; p += 8;
  tb.rstart = rstart_words ? p : nullptr; p += rstart_words;
  tb.pool2_records = pool2_records;
  tb.pool_records = pool_records; tb.pool_nwg = (uint32_t)c->n_cu;
  if ((size_t)(p - (uint32_t*)mem.p) * sizeof(uint32_t) > mem.cap) return fail(c, PT_ERR_STATE, "internal: sort tables carved past their allocation");
  tb.occupied = nullptr;
  tb.shadow32 = nullptr;
  tb.status = nullptr;
  tb.ev = nullptr;
  return PT_OK;
}

// choose the grid from the bounding box: cubic cells of side h with about rho points each
void choose_grid(pt_ctx* c, const double mn[3], const double mx[3], double force_h = 0.0) {
  GridParams& g = c->gp;
  double ext[3], maxext = 0.0;
  for (int a = 0; a < 3; ++a) { ext[a] = mx[a] - mn[a]; if (!(ext[a] >= 0)) ext[a] = 0; maxext = std::max(maxext, ext[a]); }
  double h = 1.0;
  if (maxext > 0 && c->n > 0) {
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) vol *= std::max(ext[a], maxext * 1e-6);
    h = std::cbrt(vol * c->rho / (double)c->n);
    if (!(h > 0) || !std::isfinite(h)) h = maxext;
    h = std::max(h, maxext / 60000.0);     // <= ~2^16 cells per axis
    if (force_h > 0) h = std::max(force_h, maxext / 60000.0);
  }
  // the refinement of h (force_h > 0) never asks for more than refine_macros macro blocks, nor for more than refine_cpp (2) cells per
  // point.  Measured with the wave kernel taking the dense cells: the clustered generator at 100M points is fastest at ~1 cell per
  // point (448^3: 19.0 ms per step; 19.8 at 2, 20.5 at 4 cells per point), a 50M-point surface (sphere shell, K = 20) at 3 - 4
  // (11.1 ms at 1.1, 9.5 at 2, 9.2 at 3 - 4); 1B clustered points want the 640^3 the macro limit allows
  const uint64_t cap = force_h > 0 ? std::max<uint64_t>(8, std::min<uint64_t>((uint64_t)c->refine_macros, (uint64_t)((double)c->n * c->refine_cpp / 262144.0))) : (uint64_t)PT_MAX_MACROS;
  auto lay = [&](double hh) -> uint64_t {                 // grid of cell side hh; returns its number of macro blocks
    const double inv_h = 1.0 / hh;
    uint64_t nmacro = 1;
    for (int a = 0; a < 3; ++a) {
      g.bbmin[a] = mn[a];
      const double cells = std::ceil(ext[a] * inv_h);
      g.dim[a] = (int)std::max(1.0, std::min(cells, 1.0e6));
      g.mdim[a] = (g.dim[a] + 63) / 64;
      nmacro *= (uint64_t)g.mdim[a];
    }
    g.inv_h = inv_h;
    g.h = hh;
    g.nblocks = (int)(std::min<uint64_t>(nmacro, PT_MAX_MACROS) * PT_MACRO_BLOCKS);
    return nmacro;
  };
  c->grid_capped = false;
  if (lay(h) > cap) {
    // too many macro blocks for the sort (and a cell table beyond 2^31 entries): the FINEST grid that fits -- the occupancy-driven
    // refinement would otherwise creep towards it in steps that cost a full sort each
    c->grid_capped = true;
    double lo = h, hi = h;                                 // lo does not fit, hi does
    do { hi *= 1.2599210498948732; } while (lay(hi) > cap);
    for (int it = 0; it < 12; ++it) { const double mid = std::sqrt(lo * hi); if (lay(mid) > cap) lo = mid; else hi = mid; }
    (void)lay(hi);
  }
}

template <class T, class Rec>
int run_source_sort(pt_ctx* c, uint64_t* bbox6_verify) {
  const T* x = (const T*)c->in_xyz.p;
  hipError_t e = hipSuccess;
  c->stb.status = &e;
  const Rec* r = pt_launch_grid_sort<T, Rec>(c->gp, x, x + c->n, x + 2 * c->n, (c->has_gidx && !c->local_mode) ? (const uint32_t*)c->in_gidx.p : nullptr, (uint32_t)c->n,
                                             (Rec*)c->rec.p, (Rec*)c->rec_tmp.p, (uint32_t*)c->cell_start.p, c->stb, true, c->stream, bbox6_verify);
  c->stb.status = nullptr;
  if (!r) return fail(c, PT_ERR_HIP, "grid build: a launch of the sort failed: %s", hipGetErrorString(e));
  return PT_OK;
}

// bounding box of the resident cloud: every point (sample_stride = 1) or every sample_stride-th one
int source_bbox(pt_ctx* c, uint32_t sample_stride, double (&mn)[3], double (&mx)[3]) {
  pt_launch_bbox_init((uint64_t*)c->bbox6.p, c->stream);
  if (c->in_half) {
    const __half* x = (const __half*)c->in_xyz.p;
    if (sample_stride > 1) pt_launch_bbox_sample<__half>(x, x + c->n, x + 2 * c->n, (uint32_t)c->n, sample_stride, (uint64_t*)c->bbox6.p, c->stream);
    else pt_launch_bbox<__half>(x, x + c->n, x + 2 * c->n, (uint32_t)c->n, (uint64_t*)c->bbox6.p, c->stream);
  } else if (c->src_type == PT_F32) {
    const float* x = (const float*)c->in_xyz.p;
    if (sample_stride > 1) pt_launch_bbox_sample<float>(x, x + c->n, x + 2 * c->n, (uint32_t)c->n, sample_stride, (uint64_t*)c->bbox6.p, c->stream);
    else pt_launch_bbox<float>(x, x + c->n, x + 2 * c->n, (uint32_t)c->n, (uint64_t*)c->bbox6.p, c->stream);
  } else {
    const double* x = (const double*)c->in_xyz.p;
    if (sample_stride > 1) pt_launch_bbox_sample<double>(x, x + c->n, x + 2 * c->n, (uint32_t)c->n, sample_stride, (uint64_t*)c->bbox6.p, c->stream);
    else pt_launch_bbox<double>(x, x + c->n, x + 2 * c->n, (uint32_t)c->n, (uint64_t*)c->bbox6.p, c->stream);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_bbox, c->bbox6.p, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int a = 0; a < 3; ++a) { mn[a] = pt_bbox_decode(c->h_bbox[a]); mx[a] = pt_bbox_decode(c->h_bbox[3 + a]); }
  for (int a = 0; a < 3; ++a)
    if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) return fail(c, PT_ERR_ARG, "source coordinates are not finite");
  return PT_OK;
}

// the resident cloud's planar coordinates as fp32 (fp32 clouds: the input buffer itself; fp16-resident clouds: widened once per cloud)
int source_xyz_f32(pt_ctx* c, const float** out) {
  if (!c->in_half) { *out = (const float*)c->in_xyz.p; return PT_OK; }
  if (!c->xyz32_valid) {
    RES(c, c->xyz32, std::max<uint64_t>(c->n, 1) * 3 * sizeof(float));
    pt_launch_half_to_float(c->in_xyz.p, (float*)c->xyz32.p, c->n * 3, c->stream);
    c->xyz32_valid = true;
  }
  *out = (const float*)c->xyz32.p;
  return PT_OK;
}

int rebuild(pt_ctx* c) {
  if (c->src_type != PT_F32 && c->src_type != PT_F64) return fail(c, PT_ERR_STATE, "no source cloud resident (call a pt_build_* first)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  // Big clouds: the grid is laid out from the bounding box of a SAMPLE (every 1024th point), widened by the cells the
  // 64-cell padding of the grid leaves free anyway (up to 2 % per side), and the first full pass over the coordinates
  // (pass 1's histogram) reduces the exact box along the way.  If the exact box fits the grid -- it does unless the cloud
  // has outliers the sample missed -- every point is inside the grid as the search requires and one 12-byte-per-point
  // pass has been saved; otherwise the build is redone from the exact box and the context stops guessing for this cloud.
  bool guessed = false;
  c->st.bbox_guess = 0;
  c->st.uniform_probe = 0;
  if (c->n) {
    guessed = c->bbox_guess_ok && c->n >= c->guess_min_points;
    { int r = source_bbox(c, guessed ? 1024u : 1u, mn, mx); if (r != PT_OK) return r; }
  }
  // Grid choice.  The first guess assumes the cloud fills its bounding box; finalize counts the non-empty cells, and when
  // those hold far more than rho points each (surfaces, clusters) the cell size is refined -- at most three times, and never
  // beyond what the dense cell table allows (choose_grid coarsens again if the macro-block limit is hit).
  double force_h = 0.0;
  bool hinted = false;
  if (c->grid_hint && c->adaptive && c->hint_h > 0.0) { force_h = c->hint_h; hinted = true; }
  uint32_t nblocks = 0;
  size_t ncells = 0;
  c->st.n_refine = 0;
  uint32_t max_cell = 0;            // points of the fullest cell of the final grid (adaptive builds)
  bool pool_failed = false;         // the pooled pass 1 overflowed a region and the build was redone with the exact pass 1
  bool pool2_failed = false;        // the same for the pooled pass 2
  int presort_iters = 0, nsorts = 0;   // refinements of the cell size taken from the sample, before any sort; sorts run by this build
  bool presort_unverified = false;     // ... and no sort has counted the occupied cells of the refined grid yet
  c->st.presort_refine = 0; c->st.n_sorts = 0; c->st.ordered_input = 0;
  for (int iter = 0;; ++iter) {
    choose_grid(c, mn, mx, force_h);
    int pad_cells[3] = {0, 0, 0};         // empty cells around the cloud on every side (a grid laid out from a sampled box)
    if (guessed && iter == 0) {           // widen the sampled box into the grid's own padding; no room on some axis: no guess
      int pad[3];
      bool room = c->gp.nblocks > PT_MAXBINS;     // (the one-level sort has no chunked pass 1 to verify in)
      for (int a = 0; a < 3; ++a) {
        pad[a] = std::min((c->gp.mdim[a] * 64 - c->gp.dim[a]) / 2, std::max(1, c->gp.dim[a] / 50));
        room = room && pad[a] >= 1;
      }
      if (room) {
        for (int a = 0; a < 3; ++a) { c->gp.bbmin[a] -= pad[a] * c->gp.h; c->gp.dim[a] += 2 * pad[a]; pad_cells[a] = pad[a]; }
      } else {
        guessed = false;
        { int r = source_bbox(c, 1u, mn, mx); if (r != PT_OK) return r; }
        choose_grid(c, mn, mx, force_h);
      }
    }
    nblocks = (uint32_t)c->gp.nblocks;
    ncells = (size_t)nblocks * PT_BLOCK_CELLS;
    RES(c, c->cell_start, (ncells + 1) * sizeof(uint32_t));
    // big clouds on the two-level sort: pass 1 takes its bin regions from a sample instead of a histogram pass of its own (pt_grid.hip,
    // scatter_pool_kernel); its output -- the records array -- then carries slack between the bins and a scratch area
    uint64_t pool_records = 0;
    if (c->pool_ok && c->pool_min_points && c->n >= c->pool_min_points && nblocks > PT_MAXBINS && !pt_sort_group_shift(nblocks))
      pool_records = pt_sort_pool_records((uint32_t)c->n, nblocks / PT_MACRO_BLOCKS, (uint32_t)c->n_cu, recsize(c->src_type));
    // a resident cloud whose last build found it uniform (no refinement of the cell size, occupied cells at rho) takes pass 2 without its
    // histogram too: block regions from the macro counts (pt_grid.hip, pool2_sizes_kernel); the pass-2 output then carries slack.
    // A cloud nothing is known about yet (its FIRST build) is asked through a 1/64 sample, one short read-back (round 4: a detail-transfer
    // run builds its index once -- reference src/pointsTransfer.cpp:259 -- so the first build is the one that counts)
    // The same sample also bounds the points per occupied cell from below (pt_grid.hip, uniform_check_kernel): when that bound already says
    // what the occupancy count after the sort would say -- far more than rho points per occupied cell: a surface, clusters -- the cell size
    // is refined BEFORE the first sort, by the rule the occupancy count is held to below, and the sample is asked again on the finer grid
    // (at most three times: a fraction of a millisecond each instead of a sort each; "presort_refine").  The sort's own count still decides.
    const bool presort_again = presort_iters > 0 && presort_iters < 3 && iter == 0 && nsorts == 0;
    if (((!c->uniform_known && force_h == 0.0) || presort_again) && iter == 0 && c->pool2_ok && c->pool2 && c->adaptive && c->pool_min_points && c->n >= c->pool_min_points &&
        nblocks > PT_MAXBINS && !pt_sort_group_shift(nblocks)) {
      int olo[3], ohi[3];
      for (int a = 0; a < 3; ++a) { olo[a] = pad_cells[a]; ohi[a] = c->gp.dim[a] - pad_cells[a]; }
      uint32_t* pflag = (uint32_t*)c->counter.p + 15;
      uint32_t* scratch = (uint32_t*)c->cell_start.p;                  // (written by finalize later; nblocks + nblocks / 512 words are a fraction of it)
      if (c->in_half) { const __half* x = (const __half*)c->in_xyz.p; pt_launch_uniform_probe<__half>(c->gp, x, x + c->n, x + 2 * c->n, (uint32_t)c->n, olo, ohi, scratch, pflag, c->stream); }
      else if (c->src_type == PT_F32) { const float* x = (const float*)c->in_xyz.p; pt_launch_uniform_probe<float>(c->gp, x, x + c->n, x + 2 * c->n, (uint32_t)c->n, olo, ohi, scratch, pflag, c->stream); }
      else { const double* x = (const double*)c->in_xyz.p; pt_launch_uniform_probe<double>(c->gp, x, x + c->n, x + 2 * c->n, (uint32_t)c->n, olo, ohi, scratch, pflag, c->stream); }
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 11, pflag, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->h_bbox + 6, scratch + pt_uniform_probe_acc_offset(nblocks), 32, hipMemcpyDeviceToHost, c->stream));     // (h_bbox: 10 pinned 64-bit words, 6 of them the box)
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (!presort_again) {
        const double chi2 = (double)c->h_bbox[6] / 1024.0, dof = (double)c->h_bbox[7];
        c->uniform_seen = c->h_counter[11] == 0 && dof > 0.0 && chi2 <= dof + 6.0 * std::sqrt(2.0 * dof) + 16.0;
        c->uniform_known = true;
        c->st.uniform_probe = c->uniform_seen ? 1 : -1;
      }
      // A cloud stored in SPATIAL order (scan lines, tiles, a previous sort) shows a sample of consecutive points a few crowded blocks and nothing
      // in between: neither the regions of the pooled passes nor the occupancy estimate below can be taken from it.  It gives itself away -- of 64
      // consecutive points most fall into ONE block -- and the build takes the exact passes and the sort's own count (one sort saved: the pooled
      // pass 1 would have overflowed and been redone).
      if (!presort_again) {
        const uint32_t stride = std::max<uint32_t>(16u, std::min<uint32_t>(256u, (uint32_t)c->n >> 22));
        const double sampled = (double)c->n / stride;
        c->st.ordered_input = (double)c->h_bbox[9] > 0.25 * sampled ? 1 : 0;
        if (c->st.ordered_input) { c->pool_ok = false; c->pool2_ok = false; c->uniform_seen = false; c->st.uniform_probe = -1; pool_records = 0; }
      }
      const double occ_ub = (double)c->h_bbox[8] / 16.0;
      const double rho_lb = occ_ub > 0.0 ? (double)c->n / occ_ub : 0.0;
      if (getenv("PT_DEBUG_PRESORT")) fprintf(stderr, "uniform probe: flag %u chi2 %.1f dof %.0f same-block %llu\n", c->h_counter[11], (double)c->h_bbox[6] / 1024.0, (double)c->h_bbox[7], (unsigned long long)c->h_bbox[9]);
      if (getenv("PT_DEBUG_PRESORT")) fprintf(stderr, "presort probe: grid %d %d %d h %.6g occ_ub %.0f rho_lb %.2f iters %d uniform %d\n", c->gp.dim[0], c->gp.dim[1], c->gp.dim[2], c->gp.h, occ_ub, rho_lb, presort_iters, (int)c->uniform_seen);
      if (c->presort_refine && !c->uniform_seen && !c->st.ordered_input && presort_iters < 3 && rho_lb > 1.5 * c->rho) {
        const double h_old = c->gp.h;
        const int d0 = c->gp.dim[0], d1 = c->gp.dim[1], d2 = c->gp.dim[2];
        const double h_new = h_old * std::pow(c->rho * 1.25 / rho_lb, 1.0 / 2.2);
        const GridParams keep = c->gp;
        choose_grid(c, mn, mx, h_new);
        const bool changed = c->gp.dim[0] != d0 || c->gp.dim[1] != d1 || c->gp.dim[2] != d2;
        const bool finer = c->gp.h < h_old * 0.95;
        c->gp = keep;
        if (changed && finer) {                    // as a refinement after a sort would: the same grid logic, restarted on the finer cell size
          force_h = h_new;
          ++presort_iters; ++c->st.n_refine;
          c->st.presort_refine = presort_iters;
          presort_unverified = true;
          --iter;
          continue;
        }
      }
      presort_iters = presort_iters ? 3 : 0;       // (asked for the last time)
    }
    uint64_t pool2_records = 0;
    if (c->pool2_ok && c->pool2 && c->uniform_seen && nblocks > PT_MAXBINS && !pt_sort_group_shift(nblocks) && c->n)
      pool2_records = pt_sort_pool2_records(pool_records ? c->n + (uint64_t)(nblocks / PT_MACRO_BLOCKS) * c->n_cu * 128u : c->n, nblocks, recsize(c->src_type));
    RES(c, c->rec, std::max<size_t>(std::max<uint64_t>(c->n, pool_records), 1) * recsize(c->src_type));
    RES(c, c->rec_tmp, std::max<size_t>(std::max<uint64_t>(c->n, pool2_records), 1) * recsize(c->src_type));
    { int r = make_tables(c, c->stb_mem, c->stb, nblocks, (uint32_t)c->n, recsize(c->src_type), pool_records, pool2_records); if (r != PT_OK) return r; }
    for (int a = 0; a < 3; ++a) { c->stb.occ_lo[a] = pad_cells[a]; c->stb.occ_hi[a] = c->gp.dim[a] - pad_cells[a]; }
    c->st.pass2_pooled = pool2_failed ? -1 : (pool2_records ? 1 : 0);
    c->st.pass1_pooled = pool_failed ? -1 : (pool_records ? 1 : 0);
    c->stb.ev = c->sev;
    c->rec32_valid = false;
    if (c->src_type == PT_F64 && c->tile && c->n) {           // the tile kernel's fp32 image of an fp64 cloud, written by finalize
      RES(c, c->rec32, c->n * sizeof(RecF));
      c->stb.shadow32 = (RecF*)c->rec32.p;
    }
    uint32_t* occ = (uint32_t*)c->counter.p + 8;
    c->stb.occupied = c->adaptive ? c->stb.block_count : nullptr;     // block_count is dead once block_start exists
    const bool verify = guessed && iter == 0;
    if (verify) pt_launch_bbox_init((uint64_t*)c->bbox6.p, c->stream);
    uint64_t* bv = verify ? (uint64_t*)c->bbox6.p : nullptr;
    ++nsorts; c->st.n_sorts = nsorts;
    { const int r = c->in_half ? run_source_sort<__half, RecF>(c, bv) : (c->src_type == PT_F32 ? run_source_sort<float, RecF>(c, bv) : run_source_sort<double, RecD>(c, bv)); if (r != PT_OK) return r; }
    c->h_counter[15] = 0;
    // ONE read-back per sort (round 4; there were up to three, each a drained pipeline -- a third of a 10 M-point rebuild): the pooled
    // passes' overflow flag, the verified bounding box and finalize's occupancy travel together; the occupancy sum is queued before it is
    // known whether a guess failed (then it sums whatever an aborted finalize left and is not looked at)
    const bool want_occ = c->adaptive && c->n;
    if (want_occ) {
      HIPCHK(c, hipMemsetAsync(occ, 0, 8, c->stream));
      pt_launch_sum_u32(c->stb.block_count, nblocks, occ, c->stream);
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 8, occ, 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (pool_records || pool2_records)     // did every bin stay inside the region its estimate gave it?
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 15, c->stb.pool_flag, 4, hipMemcpyDeviceToHost, c->stream));
    if (verify) HIPCHK(c, hipMemcpyAsync(c->h_bbox, c->bbox6.p, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (want_occ || pool_records || pool2_records || verify) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (verify) {
      bool inside = true, finite = true;
      for (int a = 0; a < 3; ++a) {
        mn[a] = pt_bbox_decode(c->h_bbox[a]); mx[a] = pt_bbox_decode(c->h_bbox[3 + a]);      // exact from here on
        finite = finite && std::isfinite(mn[a]) && std::isfinite(mx[a]);
        inside = inside && mn[a] >= c->gp.bbmin[a] && mx[a] <= c->gp.bbmin[a] + c->gp.dim[a] * c->gp.h;
      }
      if (!finite) return fail(c, PT_ERR_ARG, "source coordinates are not finite");
      c->st.bbox_guess = inside ? 1 : -1;
      if (!inside) {          // the sample missed part of the cloud: start over from the exact box, no more guessing for this cloud
        c->bbox_guess_ok = false;
        guessed = false;
        force_h = 0.0;
        if (c->st.uniform_probe) { c->uniform_known = false; c->uniform_seen = false; }      // (the sample was binned on the wrong grid: asked again on the exact one)
        presort_iters = 0; presort_unverified = false; c->st.n_refine = 0; c->st.presort_refine = 0;
        if (c->h_counter[15] & 1u) c->pool_ok = false;
        if (c->h_counter[15] & 2u) c->pool2_ok = false;
        --iter;
        continue;
      }
    }
    if (c->h_counter[15]) {                       // a bin outgrew its region (the sample missed a cluster; a macro block is not uniform inside):
      if (c->h_counter[15] & 1u) { c->pool_ok = false; pool_failed = true; }       // the same grid again with exact bins, and no more pooling of
      if (c->h_counter[15] & 2u) { c->pool2_ok = false; pool2_failed = true; }     // that pass for this cloud
      guessed = false;                            // (the box is exact by now, or was never guessed)
      --iter;
      continue;
    }
    c->st.rho_occupied = 0.0;
    if (!want_occ) break;
    const double occupied = std::max<double>(1.0, c->h_counter[8]);
    max_cell = c->h_counter[9];
    c->st.rho_occupied = (double)c->n / occupied;
    if (presort_unverified) {
      // The sample's bound holds for points in random order.  A cloud stored in SPATIAL order shows the sample (runs of consecutive points) a few
      // crowded blocks and nothing in between: the bound then asks for cells far too small.  The count says so -- and the grid is laid out
      // again from the box alone, this time with the sort's own count as the only judge (what every build did before round 4).
      presort_unverified = false;
      if (c->st.rho_occupied < 0.4 * c->rho) {
        c->st.presort_refine = -presort_iters;
        presort_iters = 3; force_h = 0.0; c->st.n_refine = 0; iter = -1;
        c->pool_ok = false; c->pool2_ok = false;          // (a sample that misjudged the occupancy misjudges the regions too)
        continue;
      }
    }
    if (hinted) {
      if (c->st.rho_occupied <= 1.25 * c->hint_rho_occ) { c->st.n_refine = c->hint_refines; break; }     // the grid the last build settled on still fits
      hinted = false; c->hint_h = 0.0; force_h = 0.0; c->st.n_refine = 0; iter = -1;                      // it does not (the resident cloud was changed under us): search again
      continue;
    }
    if (iter >= 3 || c->st.rho_occupied <= 1.5 * c->rho) break;
    const double h_old = c->gp.h;
    const int d0 = c->gp.dim[0], d1 = c->gp.dim[1], d2 = c->gp.dim[2];
    force_h = h_old * std::pow(c->rho * 1.25 / c->st.rho_occupied, 1.0 / 2.2);     // occupied cells grow ~ h^-2 .. h^-3
    GridParams probe = c->gp;
    choose_grid(c, mn, mx, force_h);
    const bool changed = c->gp.dim[0] != d0 || c->gp.dim[1] != d1 || c->gp.dim[2] != d2;
    const bool finer = c->gp.h < h_old * 0.95;
    c->gp = probe;
    if (!changed || !finer) break;                 // already at the resolution limit (or nothing left to split)
    ++c->st.n_refine;
  }
  if (c->st.n_refine > 0 && !hinted) { c->hint_h = c->gp.h; c->hint_rho_occ = c->st.rho_occupied; c->hint_refines = c->st.n_refine; }
  // ---- heavy cells get sub-grids (pt_refine.hip): only clouds whose fullest cell is over the threshold pay anything here ----
  c->n_nodes = 0; c->refine_levels = 0;
  if (c->refine_threshold >= 1.0 && c->adaptive && c->n && (double)max_cell > c->refine_threshold) {
    const uint32_t thr = (uint32_t)c->refine_threshold;
    // every node holds more than thr points of its level, and the levels nest: at most n / thr nodes per level
    const uint64_t cap64 = std::min<uint64_t>((uint64_t)PT_REFINE_DEPTH * (c->n / thr + 1) + 16, 0x7FFFFFF0ull / PT_NODE_WORDS * 8);
    const uint32_t cap = (uint32_t)std::min<uint64_t>(cap64, 64u << 20);
    RES(c, c->cell_node, (ncells + 1) * sizeof(uint32_t));
    RES(c, c->near_node, ncells + 16);
    HIPCHK(c, hipMemsetAsync(c->near_node.p, 0, ncells, c->stream));
    RES(c, c->nodes, (size_t)cap * PT_NODE_WORDS * sizeof(uint32_t));
    uint32_t* cnt = (uint32_t*)c->counter.p + 10;
    HIPCHK(c, hipMemsetAsync(cnt, 0, 4, c->stream));
    pt_launch_heavy_cells(c->gp, (const uint32_t*)c->cell_start.p, (uint32_t)ncells, thr, (uint32_t*)c->cell_node.p, cnt, cap, (uint32_t*)c->nodes.p, (uint8_t*)c->near_node.p, c->stream);
    uint32_t n0 = 0, n1 = 0;
    for (int level = 0; level < PT_REFINE_DEPTH; ++level) {
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 10, cnt, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));                       // one read-back per level: how many nodes the level has
      n1 = std::min(c->h_counter[10], cap);
      HIPCHK(c, hipMemcpyAsync(cnt, &n1, 4, hipMemcpyHostToDevice, c->stream));      // (a counter that ran past the capacity is put back)
      if (n1 == n0) break;
      if (c->src_type == PT_F32) pt_launch_refine_nodes<RecF>(c->gp, (RecF*)c->rec.p, (RecF*)c->rec_tmp.p, n0, n1, (uint32_t*)c->nodes.p, c->stream);
      else pt_launch_refine_nodes<RecD>(c->gp, (RecD*)c->rec.p, (RecD*)c->rec_tmp.p, n0, n1, (uint32_t*)c->nodes.p, c->stream);
      const bool last = level + 1 == PT_REFINE_DEPTH;
      pt_launch_heavy_subcells(n0, n1, last ? 0xFFFFFFFFu : thr, cnt, cap, (uint32_t*)c->nodes.p, c->stream);
      c->refine_levels = (uint32_t)level + 1;
      n0 = n1;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));                         // (n1 above is host memory the last copy reads)
    c->n_nodes = n1;
    // runs of identical points inside the nodes' leaves (pt_refine.hip, dedup_leaves_kernel): the PT_DUP_KEEP lowest indices to the front
    c->st.dup_leaves = 0;
    if (n1 && c->dup_runs) {
      static_assert(PT_MAX_K <= PT_DUP_KEEP, "a leaf's front must hold every point a query may return from it");
      uint32_t* dst = (uint32_t*)c->counter.p + 2;
      HIPCHK(c, hipMemsetAsync(dst, 0, 8, c->stream));
      if (c->src_type == PT_F32) pt_launch_dedup_leaves<RecF>((RecF*)c->rec.p, (RecF*)c->rec_tmp.p, 0, n1, (uint32_t*)c->nodes.p, dst, c->stream);
      else pt_launch_dedup_leaves<RecD>((RecD*)c->rec.p, (RecD*)c->rec_tmp.p, 0, n1, (uint32_t*)c->nodes.p, dst, c->stream);
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 2, dst, 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->st.dup_leaves = (int32_t)std::min<uint32_t>(c->h_counter[2], 0x7FFFFFFFu);
    }
    if (c->src_type == PT_F64 && c->stb.shadow32 && n1) pt_launch_reshadow((const RecD*)c->rec.p, (uint32_t)c->n, c->stb.shadow32, c->stream);
    HIPCHK(c, hipGetLastError());
  }
  c->st.n_nodes = c->n_nodes; c->st.refine_levels = (int32_t)c->refine_levels; c->st.max_cell_points = max_cell;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  HIPCHK(c, hipGetLastError());
  if (c->src_type == PT_F64 && c->stb.shadow32) {
    double amax = 0.0;                                          // largest |coordinate| a source point can have: the grid box's corners
    for (int a = 0; a < 3; ++a) amax = std::max(amax, std::max(std::fabs(c->gp.bbmin[a]), std::fabs(c->gp.bbmin[a] + c->gp.dim[a] * c->gp.h)));
    c->e_src = (float)(amax * 5.9604645e-8 * 1.000001);         // 2^-24 relative rounding, rounded up
    c->rec32_valid = std::isfinite(c->e_src) && amax < 1e30;  // (coordinates an fp32 cannot hold: the group kernel answers)
  }
  // what the next build of this resident cloud may assume: a cloud whose occupied cells hold about rho points without any refinement of
  // the cell size is uniform enough for block regions sized from the macro counts (verified again by that build's overflow flag)
  c->uniform_seen = c->adaptive && c->n && c->st.n_refine == 0 && !c->grid_capped && c->st.rho_occupied > 0.0 &&
                    c->st.rho_occupied <= 1.25 * c->rho / (1.0 - std::exp(-c->rho));       // (a uniform cloud's occupied cells hold rho / (1 - e^-rho))
  c->uniform_known = true;                                      // (what a finished build found outranks the sample)
  c->built = true;
  c->st.n_source = c->n;
  c->st.grid_dim[0] = c->gp.dim[0]; c->st.grid_dim[1] = c->gp.dim[1]; c->st.grid_dim[2] = c->gp.dim[2];
  c->st.cell_size = c->gp.h;
  c->st.n_cells = ncells;
  c->st.n_levels = nblocks <= PT_MAXBINS ? 1 : (pt_sort_group_shift(nblocks) ? 3 : 2);
  (void)ncells;
  const uint64_t s = tsize(c->src_type) * 3;
  c->st.bytes_alg_build = c->n * (2 * s + 4);
  if (c->sync) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->st.ms_build = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->sev[0]));
    c->st.ms_kernel[0] = ms;
    for (int i = 0; i < 5; ++i) {
      HIPCHK(c, hipEventElapsedTime(&ms, c->sev[i], c->sev[i + 1]));
      c->st.ms_kernel[1 + i] = ms;
    }
  }
  return PT_OK;
}

int check_n(pt_ctx* c, uint64_t n, const char* what) {
  if (n >= 0xFFFFFFF0ull) return fail(c, PT_ERR_ARG, "%s = %llu does not fit 32-bit indices", what, (unsigned long long)n);
  return PT_OK;
}

int copy_in(pt_ctx* c, void* dst, const void* src, size_t bytes, int on_device) {
  if (!bytes) return PT_OK;
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  if (!on_device) HIPCHK(c, hipStreamSynchronize(c->stream));   // pageable host memory: the caller may reuse it on return
  return PT_OK;
}

// Strong density contrast: the occupied cells hold clearly more points than those of a uniform cloud of the same density would
// (rho / (1 - exp(-rho)) per occupied cell) even after the refinement of h.  Such clouds skip the tile kernel (most of its regions
// overflow) for the wave kernel.  1.25: on the clustered generator at 30 M - 100 M points and k = 32 the occupied cells end at 1.4 rho
// (the refinement stops at ~1.1 cells per point) and the tile kernel still hands 60 - 80 % of the targets over -- 31.8 ms against
// 24.5 ms with every target on the wave kernel.
inline bool contrast_last(const pt_ctx* c) {
  return c->tile == 1 && c->adaptive && c->st.rho_occupied > 1.25 * c->rho / (1.0 - std::exp(-c->rho));
}

// optional second half of a query: blend the neighbours' attributes (fused into the tile kernel where that runs)
struct BlendReq { int mode; float* rgb_out; float* nrm_out; };

// sort the resident targets into cell order and run the k-NN kernel
int run_query(pt_ctx* c, const void* txyz, int ttype, uint64_t tm, int k, const double* bound2_dev, uint32_t* idx_dev, double* d2_dev,
              const BlendReq* br = nullptr) {
  if (!c->built) return fail(c, PT_ERR_STATE, "query before build");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k = %d out of range [1, %d]", k, PT_MAX_K);
  if (ttype != c->src_type) return fail(c, PT_ERR_UNSUPPORTED, "target xyz type %d differs from the source cloud's %d", ttype, c->src_type);
  if (!idx_dev && tm) return fail(c, PT_ERR_ARG, "idx output is null");
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t m = (uint32_t)tm;
  RES(c, c->trec, std::max<size_t>(m, 1) * recsize(ttype));
  RES(c, c->trec_tmp, std::max<size_t>(m, 1) * recsize(ttype));
  RES(c, c->todo, std::max<size_t>(m, 1) * sizeof(uint32_t));
  { int r = make_tables(c, c->ttb_mem, c->ttb, (uint32_t)c->gp.nblocks, m, recsize(ttype)); if (r != PT_OK) return r; }
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  // A cloud whose occupied cells stay far above rho even after refinement (blobs, strong density contrast) overflows
  // most tile regions: measured on the clustered generator the group kernel alone is ~8 % faster than tile kernel +
  // hand-over, so the automatic mode goes straight to it.
  const bool contrast = contrast_last(c);
  // (its fp32 pruning multiplies cell gaps by (float)(h*h): a cell side beyond ~1.8e19, or below ~1e-19, leaves fp32's range
  //  and would prune -- or keep -- everything; such clouds are answered by the group kernel, which prunes in fp64)
  const double h2d = c->gp.h * c->gp.h;
  const bool h2_ok = h2d <= 3.0e38 && h2d >= 1.2e-38;
  // radius-bounded queries run the group kernel (few targets, scattered: the slab exchange) -- unless the caller says the bounds come with
  // EVERY target of a block-sorted set (pt_stream_query from its second chunk on): then the tile kernel's bounded variant takes them
  // (round 4: fp64 clouds and k in 25..32 as well -- ADVICE r3: since every chunk of a streamed cloud brings bounds, those went to the
  //  8-lane group kernel in EVERY chunk, the first, unbounded one included)
  const bool tile_bounded = bound2_dev && c->tile_bounds && !br && k <= PT_TILE_MAX_K;
  const bool use_tile = c->tile && (!contrast || (c->tile_contrast && k <= 24)) && (!bound2_dev || tile_bounded) && m && k <= PT_TILE_MAX_K && h2_ok && (ttype == PT_F32 || c->rec32_valid);
  // The tile kernel over all blocks (+ the large geometry for the blocks the small one had to pass on); what it cannot
  // settle is on the todo list afterwards.  fp64 clouds: the LDS image is the fp32 shadow of the sorted records, the exact
  // 32-byte records are fetched for the few candidates that reach the ranking pass.
  std::function<void()> retry_launch;                   // the large-geometry launch over the blocks the small one passed on
  bool retry_pending = false;                           // ... its block count is still on its way to the host (see group_f32 / group_f64)
  const bool defer_retry = c->wave_min > 0 && c->wave_min <= 1 && c->n_nodes == 0 && (k > 16 || c->sync) && k <= 64;      // the leftovers get a wave each, sized by a read-back
  auto tile_launches = [&](const RecF* src32, const RecF* tgt32, const RecD* src64, const RecD* tgt64, uint32_t* todo_n) -> int {
    const double cells = (double)c->gp.dim[0] * c->gp.dim[1] * c->gp.dim[2];
    // regions (10^3 cells) that fit the small capacity with headroom run the two-workgroups-per-CU geometry
    const bool empty_grid = c->adaptive && (c->st.n_refine > 0 || c->st.rho_occupied > 1.25 * c->rho / (1.0 - std::exp(-c->rho)));      // surfaces, clusters: most cells are empty
    const bool fits_small = (double)c->n / cells * 1000.0 * 1.05 <= (double)(k <= 8 ? PT_TILE_CAP_SMALL_8 : PT_TILE_CAP_SMALL_16);
    // geometry: 1 = two 512-thread workgroups per CU (k <= 16), 4 = two 384-thread ones (k in 17..24), 0 = one large one.  A cloud that leaves most
    // of its grid empty tries the two-per-CU geometry first whatever the average density says: its blocks that are over budget get the large one
    const int tile_small = k <= 16 ? (c->tile == 2 || (c->tile == 1 && (fits_small || empty_grid)) ? 1 : 0)
                                   : (k <= 24 ? (c->tile == 2 || (c->tile == 1 && (fits_small || empty_grid)) ? 4 : 0) : 0);
    HIPCHK(c, hipMemsetAsync(todo_n, 0, 4, c->stream));
    const Attr* battr = br ? (const Attr*)c->attr.p : nullptr;
    const bool second_chance = tile_small != 0;               // two-per-CU geometries: over-budget blocks get the large one
    // clouds that leave most of their grid empty (surfaces, clusters: the occupied cells hold far more than rho points, or the cell size
    // was refined): one workgroup per block that HOLDS TARGETS instead of one per block of the grid -- one read-back for the list's length
    const uint32_t* blist = nullptr;
    uint32_t nlist = 0;
    if (c->tile_sparse == 1 || (c->tile_sparse == 2 && empty_grid)) {
      RES(c, c->tlist, (size_t)c->gp.nblocks * sizeof(uint32_t));
      uint32_t* lcnt = (uint32_t*)c->counter.p + 7;
      pt_launch_tblock_list(c->ttb.block_start, (uint32_t)c->gp.nblocks, (uint32_t*)c->tlist.p, lcnt, c->stream);
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 7, lcnt, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      blist = (const uint32_t*)c->tlist.p; nlist = c->h_counter[7];
      if (!nlist) return PT_OK;
    }
    uint32_t* retry_n = (uint32_t*)c->counter.p + 5;
    if (second_chance) {
      HIPCHK(c, hipMemsetAsync(retry_n, 0, 4, c->stream));
      RES(c, c->retry, (size_t)c->gp.nblocks * sizeof(uint32_t));
    }
    pt_launch_knn_tile(c->gp, src32, (const uint32_t*)c->cell_start.p, tgt32, c->ttb.block_start, k, idx_dev, d2_dev, (uint32_t*)c->todo.p, todo_n,
                       tile_small, battr, (uint32_t)c->n_total, br ? br->mode : 0, br ? br->rgb_out : nullptr, br ? br->nrm_out : nullptr, blist, nlist,
                       second_chance ? (uint32_t*)c->retry.p : nullptr, retry_n, src64, tgt64, c->e_src, c->stream, bound2_dev);
    if (second_chance) {
      retry_launch = [=]() {
        pt_launch_knn_tile(c->gp, src32, (const uint32_t*)c->cell_start.p, tgt32, c->ttb.block_start, k, idx_dev, d2_dev, (uint32_t*)c->todo.p, todo_n,
                           0, battr, (uint32_t)c->n_total, br ? br->mode : 0, br ? br->rgb_out : nullptr, br ? br->nrm_out : nullptr,
                           (const uint32_t*)c->retry.p, c->h_counter[5], nullptr, nullptr, src64, tgt64, c->e_src, c->stream, bound2_dev);
      };
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 5, retry_n, 4, hipMemcpyDeviceToHost, c->stream));
      // (round 4) when the leftover pass reads the todo list's length anyway -- one wave per leftover target -- the two counts share that
      // read-back: a 1 M-target query spent a third of its time in two drained pipelines.  Otherwise: one short read-back here; usually
      // 0 blocks and no launch
      if (defer_retry) { retry_pending = true; return PT_OK; }
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (c->h_counter[5]) retry_launch();
    }
    return PT_OK;
  };
  uint32_t* todo_n = (uint32_t*)c->counter.p + 4;
  // what the tile kernel does not take (or leaves over) goes to the 8-lanes-per-target kernel: the plain one, or -- when the build
  // refined heavy cells -- the one that descends into their sub-grids instead of scanning them end to end
  const bool hier = c->n_nodes > 0;
  // Clouds with strong density contrast: targets whose 27 nearest cells hold many points (wave_min or more) are listed by the group
  // kernel instead of being answered, and get ONE WAVE EACH afterwards (knn_wave_kernel); one read-back of the list length.
  // (also what the tile kernel leaves over: up to 12 % of the targets at k > 16 -- 100M / 10M uniform, k = 32: 24.4 -> 22.3 ms -- and 0.2 %
  //  at C4, 23.6 -> 23.1 ms; the list's length is read back to size that launch, so callers that asked for enqueue-only calls keep the
  //  group kernel at k <= 16)
  const bool wave = c->wave_min > 0 && (contrast || hier || c->st.n_refine > 0 || c->wave_force || (use_tile && (k > 16 || c->sync))) && m && k <= 64;
  // buffers of the split: marks (one byte per target), the two ordered lists (m words together), per-tile offsets and scan scratch
  const uint32_t mtiles = pt_mark_tiles(m);
  uint8_t* heavy = nullptr;
  uint32_t *hlist = nullptr, *hoff1 = nullptr, *hoff2 = nullptr, *hscan = nullptr, *hcnt = (uint32_t*)c->counter.p + 12;
  if (wave) {
    RES(c, c->heavy, (size_t)m * sizeof(uint32_t) + (((size_t)m + 15) & ~(size_t)15) + ((size_t)mtiles + 4) * 3 * sizeof(uint32_t) + 64);
    hlist = (uint32_t*)c->heavy.p;
    hoff1 = hlist + m; hoff2 = hoff1 + mtiles + 4; hscan = hoff2 + mtiles + 4;
    heavy = (uint8_t*)(hscan + mtiles + 4);
  }
  c->st.n_wave = 0;
  bool wave_blended = false;       // the wave kernel answered every target this call was about AND blended it: no blend pass afterwards
  const Attr* wattr = br ? (const Attr*)c->attr.p : nullptr;
  const uint32_t wnattr = (uint32_t)c->n_total;
  const int wmode = br ? br->mode : 0;
  float* wrgb = br ? br->rgb_out : nullptr;
  float* wnrm = br ? br->nrm_out : nullptr;
  // marks -> ordered lists; one read-back of the two lengths.  false: a HIP call failed (the caller's hipGetLastError reports it)
  auto wave_lists = [&]() -> bool {
    pt_launch_mark_count(heavy, m, hoff1, hoff2, hscan, c->stream);
    if (hipMemcpyAsync(c->h_counter + 12, hoff1 + mtiles, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->h_counter + 13, hoff2 + mtiles, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return false;
    c->st.n_wave = c->h_counter[12] + c->h_counter[13];
    if (!c->st.n_wave) return false;
    pt_launch_mark_write(heavy, m, hoff1, hoff2, hlist, hlist + c->h_counter[12], c->stream);
    return hipMemcpyAsync(hcnt, c->h_counter + 12, 8, hipMemcpyHostToDevice, c->stream) == hipSuccess;      // the kernels read their list length from the device
  };
  auto group_f32 = [&](const RecF* tg, const double* bnd, const uint32_t* list, const uint32_t* list_n) {
    if (wave && c->wave_min <= 1) {
      // every target gets a wave (measured on the clustered generator: the group kernel loses to it even on the sparse targets there,
      // and marking 50 M targets costs it 70 ms): no group kernel at all.  Without refined cells the list is the input list;
      // with them, a one-load-per-target pass marks who needs the descending variant.
      if (!hier) {
        uint32_t cnt = m;
        if (list) {                                           // a device-side list (what the tile kernel left over): its length sizes the launch
          if (hipMemcpyAsync(c->h_counter + 14, list_n, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return;
          if (retry_pending) {                                // the second-chance blocks' count came with it: their launch may add to the list
            retry_pending = false;
            if (c->h_counter[5]) {
              retry_launch();
              if (hipMemcpyAsync(c->h_counter + 14, list_n, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return;
            }
          }
          cnt = c->h_counter[14];
        }
        c->st.n_wave = cnt;
        pt_launch_knn_wave<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, nullptr, nullptr, 0xFFFFFFFFu, tg, cnt, k, bnd, idx_dev, d2_dev, list, list_n, c->stream,
                                 wattr, wnattr, wmode, wrgb, wnrm);
        wave_blended = true;
        return;
      }
      (void)hipMemsetAsync(heavy, 0, m, c->stream);
      pt_launch_mark_near<RecF>(c->gp, tg, list, list_n, m, (const uint8_t*)c->near_node.p, heavy, c->stream);
      if (wave_lists()) {
        pt_launch_knn_wave<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, nullptr, nullptr, 0xFFFFFFFFu, tg, c->h_counter[12], k, bnd, idx_dev, d2_dev,
                                 hlist, hcnt, c->stream, wattr, wnattr, wmode, wrgb, wnrm);
        pt_launch_knn_wave<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, (const uint32_t*)c->cell_node.p, (const uint32_t*)c->nodes.p, (uint32_t)c->refine_threshold,
                                 tg, c->h_counter[13], k, bnd, idx_dev, d2_dev, hlist + c->h_counter[12], hcnt + 1, c->stream, wattr, wnattr, wmode, wrgb, wnrm);
        wave_blended = true;                                // (every marked target is on one of the two lists)
      }
      return;
    }
    if (wave) (void)hipMemsetAsync(heavy, 0, m, c->stream);
    if (hier) pt_launch_knn_hier<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, (const uint32_t*)c->cell_node.p, (const uint32_t*)c->nodes.p, (uint32_t)c->refine_threshold, tg, m, k, bnd, idx_dev, d2_dev, list, list_n, c->stream, heavy, c->wave_min);
    else pt_launch_knn<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, tg, m, k, bnd, idx_dev, d2_dev, list, list_n, c->stream, heavy, c->wave_min);
    if (wave && wave_lists()) {                             // plain variant for the first list, descending variant for the second
      pt_launch_knn_wave<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, nullptr, nullptr, 0xFFFFFFFFu, tg, c->h_counter[12], k, bnd, idx_dev, d2_dev,
                               hlist, hcnt, c->stream);
      pt_launch_knn_wave<RecF>(c->gp, (const RecF*)c->rec.p, (const uint32_t*)c->cell_start.p, (const uint32_t*)c->cell_node.p, (const uint32_t*)c->nodes.p, (uint32_t)c->refine_threshold,
                               tg, c->h_counter[13], k, bnd, idx_dev, d2_dev, hlist + c->h_counter[12], hcnt + 1, c->stream);
    }
  };
  auto group_f64 = [&](const RecD* tg, const double* bnd, const uint32_t* list, const uint32_t* list_n) {
    if (wave && c->wave_min <= 1) {
      // every target gets a wave (measured on the clustered generator: the group kernel loses to it even on the sparse targets there,
      // and marking 50 M targets costs it 70 ms): no group kernel at all.  Without refined cells the list is the input list;
      // with them, a one-load-per-target pass marks who needs the descending variant.
      if (!hier) {
        uint32_t cnt = m;
        if (list) {                                           // a device-side list (what the tile kernel left over): its length sizes the launch
          if (hipMemcpyAsync(c->h_counter + 14, list_n, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return;
          if (retry_pending) {                                // the second-chance blocks' count came with it: their launch may add to the list
            retry_pending = false;
            if (c->h_counter[5]) {
              retry_launch();
              if (hipMemcpyAsync(c->h_counter + 14, list_n, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return;
            }
          }
          cnt = c->h_counter[14];
        }
        c->st.n_wave = cnt;
        pt_launch_knn_wave<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, nullptr, nullptr, 0xFFFFFFFFu, tg, cnt, k, bnd, idx_dev, d2_dev, list, list_n, c->stream,
                                 wattr, wnattr, wmode, wrgb, wnrm);
        wave_blended = true;
        return;
      }
      (void)hipMemsetAsync(heavy, 0, m, c->stream);
      pt_launch_mark_near<RecD>(c->gp, tg, list, list_n, m, (const uint8_t*)c->near_node.p, heavy, c->stream);
      if (wave_lists()) {
        pt_launch_knn_wave<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, nullptr, nullptr, 0xFFFFFFFFu, tg, c->h_counter[12], k, bnd, idx_dev, d2_dev,
                                 hlist, hcnt, c->stream, wattr, wnattr, wmode, wrgb, wnrm);
        pt_launch_knn_wave<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, (const uint32_t*)c->cell_node.p, (const uint32_t*)c->nodes.p, (uint32_t)c->refine_threshold,
                                 tg, c->h_counter[13], k, bnd, idx_dev, d2_dev, hlist + c->h_counter[12], hcnt + 1, c->stream, wattr, wnattr, wmode, wrgb, wnrm);
        wave_blended = true;                                // (every marked target is on one of the two lists)
      }
      return;
    }
    if (wave) (void)hipMemsetAsync(heavy, 0, m, c->stream);
    if (hier) pt_launch_knn_hier<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, (const uint32_t*)c->cell_node.p, (const uint32_t*)c->nodes.p, (uint32_t)c->refine_threshold, tg, m, k, bnd, idx_dev, d2_dev, list, list_n, c->stream, heavy, c->wave_min);
    else pt_launch_knn<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, tg, m, k, bnd, idx_dev, d2_dev, list, list_n, c->stream, heavy, c->wave_min);
    if (wave && wave_lists()) {
      pt_launch_knn_wave<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, nullptr, nullptr, 0xFFFFFFFFu, tg, c->h_counter[12], k, bnd, idx_dev, d2_dev,
                               hlist, hcnt, c->stream);
      pt_launch_knn_wave<RecD>(c->gp, (const RecD*)c->rec.p, (const uint32_t*)c->cell_start.p, (const uint32_t*)c->cell_node.p, (const uint32_t*)c->nodes.p, (uint32_t)c->refine_threshold,
                               tg, c->h_counter[13], k, bnd, idx_dev, d2_dev, hlist + c->h_counter[12], hcnt + 1, c->stream);
    }
  };
  if (ttype == PT_F32) {
    const float* x = (const float*)txyz;
    // targets only need to be grouped by block (tile kernel) -- the cell-level pass is skipped
    const RecF* tsorted = pt_launch_grid_sort<float, RecF>(c->gp, x, x + m, x + 2 * (size_t)m, nullptr, m, (RecF*)c->trec.p, (RecF*)c->trec_tmp.p, nullptr, c->ttb, false, c->stream);
    if (!tsorted) return fail(c, PT_ERR_HIP, "target binning: a launch of the sort failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    if (use_tile) {
      { int r = tile_launches((const RecF*)c->rec.p, tsorted, nullptr, nullptr, todo_n); if (r != PT_OK) return r; }
      group_f32(tsorted, bound2_dev, (const uint32_t*)c->todo.p, todo_n);       // (what the tile kernel left over keeps its bound, if it came with one)
      if (br && !wave_blended)   // the targets the tile kernel handed over get their blend from the lists the group kernel just wrote
        pt_launch_blend_list<RecF>((const uint32_t*)c->todo.p, todo_n, m, tsorted, idx_dev, d2_dev, k, br->mode, (const Attr*)c->attr.p,
                                   (uint32_t)c->n_total, br->rgb_out, br->nrm_out, c->stream);
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 4, todo_n, 4, hipMemcpyDeviceToHost, c->stream));
    } else {
      group_f32(tsorted, bound2_dev, nullptr, nullptr);
      if (br && !wave_blended) pt_launch_blend(idx_dev, d2_dev, m, k, br->mode, (const Attr*)c->attr.p, (uint32_t)c->n_total, br->rgb_out, br->nrm_out, c->stream);
    }
  } else {
    const double* x = (const double*)txyz;
    const RecD* tsorted = pt_launch_grid_sort<double, RecD>(c->gp, x, x + m, x + 2 * (size_t)m, nullptr, m, (RecD*)c->trec.p, (RecD*)c->trec_tmp.p, nullptr, c->ttb, false, c->stream);
    if (!tsorted) return fail(c, PT_ERR_HIP, "target binning: a launch of the sort failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    if (use_tile) {
      { int r = tile_launches((const RecF*)c->rec32.p, nullptr, (const RecD*)c->rec.p, tsorted, todo_n); if (r != PT_OK) return r; }
      group_f64(tsorted, bound2_dev, (const uint32_t*)c->todo.p, todo_n);       // (what the tile kernel left over keeps its bound, if it came with one)
      if (br && !wave_blended)
        pt_launch_blend_list<RecD>((const uint32_t*)c->todo.p, todo_n, m, tsorted, idx_dev, d2_dev, k, br->mode, (const Attr*)c->attr.p,
                                   (uint32_t)c->n_total, br->rgb_out, br->nrm_out, c->stream);
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 4, todo_n, 4, hipMemcpyDeviceToHost, c->stream));
    } else {
      group_f64(tsorted, bound2_dev, nullptr, nullptr);
      if (br && !wave_blended) pt_launch_blend(idx_dev, d2_dev, m, k, br->mode, (const Attr*)c->attr.p, (uint32_t)c->n_total, br->rgb_out, br->nrm_out, c->stream);
    }
  }
  if (c->local_mode && m) pt_launch_ids_to_global(idx_dev, (size_t)m * (size_t)k, (const uint32_t*)c->in_gidx.p, c->stream);      // positions in the slab -> global indices (every blend above used the positions)
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  HIPCHK(c, hipGetLastError());
  c->st.n_target = m;
  c->st.k = k;
  const uint64_t s = tsize(c->src_type) * 3;
  c->st.bytes_alg_query = c->n * s + (uint64_t)m * s + (uint64_t)m * k * 16 + (uint64_t)m * (4 * (uint64_t)k + 24);
  if (c->sync) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float a = 0, b = 0;
    HIPCHK(c, hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    c->st.ms_sort_targets = a;
    c->st.ms_query = b;
    c->st.ms_kernel[6] = a;
    c->st.ms_kernel[7] = b;
    c->st.n_leftover = use_tile ? c->h_counter[4] : 0;
  }
  return PT_OK;
}

// planar xyz of any supported type -> device buffer `dst` in the type the kernels run on (fp16 is widened to fp32, exactly)
int upload_xyz(pt_ctx* c, DevBuf& dst, const void* xyz, int& xyz_type, uint64_t n, int on_device) {
  if (xyz_type != PT_F32 && xyz_type != PT_F64 && xyz_type != PT_F16) return fail(c, PT_ERR_ARG, "unknown xyz_type %d", xyz_type);
  if (n && !xyz) return fail(c, PT_ERR_ARG, "xyz is null");
  if (xyz_type == PT_F16) {
    RES(c, dst, std::max<uint64_t>(n, 1) * 3 * sizeof(float));
    const void* src = xyz;
    if (!on_device) {
      RES(c, c->misc, std::max<uint64_t>(n, 1) * 3 * 2);
      { int r = copy_in(c, c->misc.p, xyz, n * 3 * 2, 0); if (r) return r; }
      src = c->misc.p;
    }
    pt_launch_half_to_float(src, (float*)dst.p, n * 3, c->stream);
    xyz_type = PT_F32;
    return PT_OK;
  }
  RES(c, dst, std::max<uint64_t>(n, 1) * 3 * tsize(xyz_type));
  return copy_in(c, dst.p, xyz, n * 3 * tsize(xyz_type), on_device);
}

// copy caller targets into the transient buffer (resident targets are not touched)
int load_transient(pt_ctx* c, const void* xyz, int& xyz_type, uint64_t m, int on_device) {
  { int r = check_n(c, m, "m"); if (r) return r; }
  { int r = upload_xyz(c, c->x_xyz, xyz, xyz_type, m, on_device); if (r) return r; }
  if (xyz_type == PT_F32 && c->src_type == PT_F64) {     // fp32 / fp16 targets against a double cloud: widen (exact)
    RES(c, c->misc, std::max<uint64_t>(m, 1) * 3 * sizeof(double));
    pt_launch_float_to_double((const float*)c->x_xyz.p, (double*)c->misc.p, m * 3, c->stream);
    std::swap(c->x_xyz, c->misc);
    xyz_type = PT_F64;
  }
  return PT_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

int pt_ctx_create(pt_ctx** out, const int* device_ids, int n_devices) {
  if (!out) return PT_ERR_ARG;
  *out = nullptr;
  if (n_devices != 1 && !(n_devices == 0 && !device_ids)) return PT_ERR_ARG;   // one process (context) per GPU
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PT_ERR_HIP;
  const int dev = device_ids ? device_ids[0] : 0;
  if (dev < 0 || dev >= count) return PT_ERR_ARG;
  if (hipSetDevice(dev) != hipSuccess) return PT_ERR_HIP;
  pt_ctx* c = new pt_ctx();
  c->device = dev;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PT_ERR_HIP; }
  c->stream = c->own_stream;
  { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) c->n_cu = cu; }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return PT_ERR_HIP; }
  for (auto& e : c->sev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return PT_ERR_HIP; }
  for (auto& e : c->xev)
    if (hipEventCreate(&e) != hipSuccess) { delete c; return PT_ERR_HIP; }
  if (hipHostMalloc((void**)&c->h_bbox, 10 * sizeof(uint64_t)) != hipSuccess || hipHostMalloc((void**)&c->h_counter, 64) != hipSuccess) {
    delete c;
    return PT_ERR_HIP;
  }
  if (reserve(c, c->bbox6, 64) != PT_OK || reserve(c, c->counter, 64) != PT_OK) { delete c; return PT_ERR_NOMEM; }
  *out = c;
  return PT_OK;
}

void pt_ctx_destroy(pt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  DevBuf* all[] = {&c->in_xyz, &c->in_gidx, &c->attr, &c->rec, &c->rec_tmp, &c->cell_start, &c->stb_mem, &c->t_xyz, &c->t_gidx, &c->trec,
                   &c->trec_tmp, &c->x_xyz, &c->ttb_mem, &c->bbox6, &c->counter, &c->q_idx, &c->q_d2, &c->b_rgb, &c->b_nrm, &c->aos_stage, &c->misc, &c->bounds, &c->todo, &c->posattr, &c->retry, &c->rec32, &c->up_rgb, &c->up_nrm, &c->x_bounds, &c->x_counts, &c->x_matrix, &c->x_off, &c->x_req, &c->x_row, &c->x_rreq,
                   &c->x_rxyz, &c->x_rbound, &c->x_ans_i, &c->x_ans_d, &c->x_back_i, &c->x_back_d, &c->x_flags, &c->x_rows, &c->cell_node, &c->nodes, &c->heavy, &c->near_node, &c->xyz32, &c->tlist, &c->x_ans_a, &c->x_back_a, &c->x_rattr, &c->l_idx};
  for (DevBuf* b : all) release(c, *b);
  if (c->h_bbox) (void)hipHostFree(c->h_bbox);
  if (c->h_counter) (void)hipHostFree(c->h_counter);
  if (c->h_matrix) (void)hipHostFree(c->h_matrix);
  if (c->h_xoff) (void)hipHostFree(c->h_xoff);
  if (c->xoff_ev) (void)hipEventDestroy(c->xoff_ev);
  (void)pt_comm_destroy(c);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->sev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->xev) if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int pt_set_stream(pt_ctx* c, void* hip_stream) {
  if (!c) return PT_ERR_ARG;
  c->stream = (hipStream_t)hip_stream;    // NULL is HIP's default stream (what torch uses unless told otherwise)
  return PT_OK;
}

int pt_set_param(pt_ctx* c, const char* name, double value) {
  if (!c || !name) return PT_ERR_ARG;
  c->hint_h = 0.0;                     // (whatever changes, the next build searches its cell size afresh)
  if (!strcmp(name, "rho")) { if (!(value >= 1e-4 && value <= 4096)) return fail(c, PT_ERR_ARG, "rho out of range"); c->rho = value; return PT_OK; }
  if (!strcmp(name, "k_hint")) {
    // cell density for the k the caller is going to ask for: ring 1 (3x3x3 cells) must usually contain the k nearest
    // (expected k-th distance ~0.8 cell sides), and a 10^3-cell region must fit the tile kernel's LDS budget
    const int k = (int)value;
    if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k_hint out of range");
    // (k in 25..32: 8.7 keeps the region inside LDS; ring 1 then fails for ~1 target in 8, which the group kernel finishes)
    c->rho = k <= 8 ? 4.0 : (k <= 16 ? 6.0 : (k <= 20 ? 7.0 : (k <= 24 ? 8.0 : (k <= PT_TILE_MAX_K ? 8.7 : 12.0))));   // (k = 20 at 100M / 10M: rho 6 7.2 ms, 7 7.0, 8 7.4)
    return PT_OK;
  }
  if (!strcmp(name, "sync")) { c->sync = value != 0; return PT_OK; }
  if (!strcmp(name, "adaptive")) { c->adaptive = value != 0; return PT_OK; }
  if (!strcmp(name, "tile")) { c->tile = (int)value; return PT_OK; }
  if (!strcmp(name, "tile_sparse")) { c->tile_sparse = (int)value; return PT_OK; }
  if (!strcmp(name, "dup_runs")) { c->dup_runs = value != 0; return PT_OK; }
  if (!strcmp(name, "local_ids")) { c->want_local_ids = value != 0; return PT_OK; }      // before pt_build_soa_indexed: see pt_set_attributes_local
  if (!strcmp(name, "tile_contrast")) { c->tile_contrast = (int)value; return PT_OK; }
  if (!strcmp(name, "grid_hint")) { c->grid_hint = value != 0; if (!c->grid_hint) c->hint_h = 0.0; return PT_OK; }
  if (!strcmp(name, "refine_cells_per_point")) { if (!(value > 0 && value <= 1e6)) return fail(c, PT_ERR_ARG, "refine_cells_per_point out of range"); c->refine_cpp = value; return PT_OK; }
  if (!strcmp(name, "refine_macros")) { if (!(value >= 1 && value <= PT_MAX_MACROS)) return fail(c, PT_ERR_ARG, "refine_macros out of range"); c->refine_macros = (int)value; return PT_OK; }
  if (!strcmp(name, "wave_force")) { c->wave_force = value != 0; return PT_OK; }
  if (!strcmp(name, "wave_min")) { if (!(value >= 0 && value <= 4e9)) return fail(c, PT_ERR_ARG, "wave_min out of range"); c->wave_min = (uint32_t)value; return PT_OK; }
  if (!strcmp(name, "refine_threshold")) { if (!(value >= 0 && value <= 1e9)) return fail(c, PT_ERR_ARG, "refine_threshold out of range"); c->refine_threshold = value; return PT_OK; }
  if (!strcmp(name, "pool2")) { c->pool2 = value != 0; c->pool2_ok = true; return PT_OK; }    // pass 2 without its histogram on clouds found uniform (1, default) or never (0)
  if (!strcmp(name, "stream_bounds")) { c->stream_bounds = value != 0; return PT_OK; }
  // "forget": the next build of the resident cloud decides everything a FIRST build decides (sampled bounding box, pooled passes, uniformity
  // sample, cell size) -- what bench.py times as its cold step, buffers already allocated
  if (!strcmp(name, "forget")) { if (value != 0) { c->bbox_guess_ok = true; c->pool_ok = true; c->pool2_ok = true; c->uniform_seen = false; c->uniform_known = false; c->hint_h = 0.0; } return PT_OK; }
  if (!strcmp(name, "presort_refine")) { c->presort_refine = value != 0; return PT_OK; }
  if (!strcmp(name, "pool_min_points")) { c->pool_min_points = value < 0 ? 0 : (uint64_t)value; c->pool_ok = true; return PT_OK; }   // pooled pass 1 from this size up (0: never)
  if (!strcmp(name, "guess_min_points")) { c->guess_min_points = value < 1 ? 1 : (uint64_t)value; return PT_OK; }   // sampled-bbox builds from this size up   // 0 group kernel only, 1 auto, 2 small tiles, 3 large tiles
  if (!strcmp(name, "own_stream")) { if (value != 0) c->stream = c->own_stream; return PT_OK; }
  return fail(c, PT_ERR_ARG, "unknown parameter '%s'", name);
}

const char* pt_last_error(pt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int pt_stats(pt_ctx* c, pt_stats_t* out) {
  if (!c || !out) return PT_ERR_ARG;
  c->st.device_bytes = c->dev_bytes;
  *out = c->st;
  return PT_OK;
}

int pt_synchronize(pt_ctx* c) {
  if (!c) return PT_ERR_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PT_OK;
}

uint64_t pt_num_source(pt_ctx* c) { return c ? c->n : 0; }
uint64_t pt_num_targets(pt_ctx* c) { return c ? c->m : 0; }

// ---- build ----------------------------------------------------------------------------------------
int pt_build_aos(pt_ctx* c, const pt_point* cloud, uint64_t n) {
  if (!c) return PT_ERR_ARG;
  if (n && !cloud) return fail(c, PT_ERR_ARG, "cloud is null");
  { int r = check_n(c, n, "n"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->aos_stage, std::max<uint64_t>(n, 1) * sizeof(pt_point));
  RES(c, c->in_xyz, std::max<uint64_t>(n, 1) * 3 * sizeof(double));
  RES(c, c->attr, std::max<uint64_t>(n, 1) * sizeof(Attr));
  { int r = copy_in(c, c->aos_stage.p, cloud, n * sizeof(pt_point), 0); if (r) return r; }
  double* x = (double*)c->in_xyz.p;
  pt_launch_aos_split(c->aos_stage.p, (uint32_t)n, x, x + n, x + 2 * n, (Attr*)c->attr.p, c->stream);
  c->src_type = PT_F64; c->n = n; c->n_total = n; c->has_gidx = false; c->local_mode = false; c->attr_local = false; c->has_attr = true; c->built = false; c->posattr_valid = false; c->bbox_guess_ok = true; c->pool_ok = true; c->pool2_ok = true; c->uniform_seen = false; c->uniform_known = false; c->hint_h = 0.0; c->in_half = false; c->xyz32_valid = false;
  return rebuild(c);
}

int pt_build_soa_indexed(pt_ctx* c, const void* xyz, int xyz_type, const uint32_t* gidx, uint64_t n, int on_device) {
  if (!c) return PT_ERR_ARG;
  { int r = check_n(c, n, "n"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  const bool keep_half = xyz_type == PT_F16;            // fp16 clouds stay fp16 in the resident input: 6 bytes per point for pass 1 to read
  if (keep_half) {
    if (n && !xyz) return fail(c, PT_ERR_ARG, "xyz is null");
    RES(c, c->in_xyz, std::max<uint64_t>(n, 1) * 3 * sizeof(__half));
    { int r = copy_in(c, c->in_xyz.p, xyz, n * 3 * sizeof(__half), on_device); if (r) return r; }
    xyz_type = PT_F32;
  } else {
    int r = upload_xyz(c, c->in_xyz, xyz, xyz_type, n, on_device); if (r) return r;
  }
  if (gidx) {
    RES(c, c->in_gidx, std::max<uint64_t>(n, 1) * sizeof(uint32_t));
    { int r = copy_in(c, c->in_gidx.p, gidx, n * sizeof(uint32_t), on_device); if (r) return r; }
  }
  c->src_type = xyz_type; c->n = n; c->has_gidx = gidx != nullptr; c->built = false;
  c->local_mode = false;
  if (gidx && c->want_local_ids) {
    uint32_t* flag = (uint32_t*)c->counter.p + 1;
    HIPCHK(c, hipMemsetAsync(flag, 0, 4, c->stream));
    pt_launch_ascending((const uint32_t*)c->in_gidx.p, (uint32_t)n, flag, c->stream);
    HIPCHK(c, hipMemcpyAsync(c->h_counter + 1, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_counter[1]) return fail(c, PT_ERR_ARG, "local_ids: the slab's global indices must be strictly ascending (positions then order like indices)");
    c->local_mode = true; c->synth_total = 0;
    c->has_attr = false; c->attr_local = false;
  }
  if (!gidx) { c->n_total = n; c->has_attr = false; c->attr_local = false; }
  c->posattr_valid = false; c->bbox_guess_ok = true; c->pool_ok = true; c->pool2_ok = true; c->uniform_seen = false; c->uniform_known = false; c->hint_h = 0.0; c->in_half = keep_half; c->xyz32_valid = false;
  return rebuild(c);
}

int pt_set_attributes_local(pt_ctx* c, const uint8_t* rgb, const float* nrm, int on_device) {
  if (!c) return PT_ERR_ARG;
  if (!c->local_mode) return fail(c, PT_ERR_STATE, "pt_set_attributes_local: build the slab with pt_set_param(\"local_ids\", 1) and ascending global indices first");
  const uint64_t n = c->n;
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->attr, std::max<uint64_t>(n, 1) * sizeof(Attr));
  const uint8_t* drgb = rgb;
  const float* dnrm = nrm;
  if (!on_device) {
    RES(c, c->misc, std::max<uint64_t>(n, 1) * 15);
    uint8_t* base = (uint8_t*)c->misc.p;
    if (nrm) { int r = copy_in(c, base, nrm, n * 12, 0); if (r) return r; dnrm = (const float*)base; }
    if (rgb) { int r = copy_in(c, base + n * 12, rgb, n * 3, 0); if (r) return r; drgb = base + n * 12; }
  }
  pt_launch_pack_attr(drgb, dnrm, (uint32_t)n, (Attr*)c->attr.p, c->stream);
  c->n_total = n;                      // the table's size: the slab's own points
  c->has_attr = true; c->attr_local = true; c->posattr_valid = false;
  return finish(c);
}

int pt_set_attributes(pt_ctx* c, const uint8_t* rgb, const float* nrm, uint64_t n_total, int on_device) {
  if (!c) return PT_ERR_ARG;
  if (c->local_mode) return fail(c, PT_ERR_STATE, "this slab keeps local ids: its attribute table is pt_set_attributes_local's (its own points' records)");
  { int r = check_n(c, n_total, "n_total"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->attr, std::max<uint64_t>(n_total, 1) * sizeof(Attr));
  const uint8_t* drgb = rgb;
  const float* dnrm = nrm;
  if (!on_device) {
    RES(c, c->misc, std::max<uint64_t>(n_total, 1) * 15);
    uint8_t* base = (uint8_t*)c->misc.p;
    if (nrm) { int r = copy_in(c, base, nrm, n_total * 12, 0); if (r) return r; dnrm = (const float*)base; }
    if (rgb) { int r = copy_in(c, base + n_total * 12, rgb, n_total * 3, 0); if (r) return r; drgb = base + n_total * 12; }
  }
  pt_launch_pack_attr(drgb, dnrm, (uint32_t)n_total, (Attr*)c->attr.p, c->stream);
  c->n_total = n_total;
  c->has_attr = true; c->posattr_valid = false;
  return finish(c);
}

int pt_set_attributes_range(pt_ctx* c, uint64_t first, uint64_t count, const uint8_t* rgb, const float* nrm, uint64_t n_total) {
  if (!c) return PT_ERR_ARG;
  if (c->local_mode) return fail(c, PT_ERR_STATE, "this slab keeps local ids: its attribute table is pt_set_attributes_local's (its own points' records)");
  { int r = check_n(c, n_total, "n_total"); if (r) return r; }
  if (first > n_total || count > n_total - first) return fail(c, PT_ERR_ARG, "pt_set_attributes_range: [first, first + count) outside the table of n_total records");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->has_attr || c->n_total != n_total || c->attr.cap < std::max<uint64_t>(n_total, 1) * sizeof(Attr)) {
    RES(c, c->attr, std::max<uint64_t>(n_total, 1) * sizeof(Attr));
    HIPCHK(c, hipMemsetAsync(c->attr.p, 0, std::max<uint64_t>(n_total, 1) * sizeof(Attr), c->stream));       // ranges never written read as black / zero normal
  }
  if (count) {
    RES(c, c->misc, count * 15);
    uint8_t* base = (uint8_t*)c->misc.p;
    const uint8_t* drgb = nullptr;
    const float* dnrm = nullptr;
    if (nrm) { int r = copy_in(c, base, nrm, count * 12, 0); if (r) return r; dnrm = (const float*)base; }
    if (rgb) { int r = copy_in(c, base + count * 12, rgb, count * 3, 0); if (r) return r; drgb = base + count * 12; }
    pt_launch_pack_attr(drgb, dnrm, (uint32_t)count, (Attr*)c->attr.p + first, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));              // the staging buffer is reused by the next range
  }
  c->n_total = n_total;
  c->has_attr = true; c->posattr_valid = false;
  return PT_OK;
}

int pt_build_soa(pt_ctx* c, const void* xyz, int xyz_type, const uint8_t* rgb, const float* nrm, uint64_t n, int on_device) {
  int r = pt_build_soa_indexed(c, xyz, xyz_type, nullptr, n, on_device);
  if (r != PT_OK) return r;
  if (rgb || nrm) return pt_set_attributes(c, rgb, nrm, n, on_device);
  return PT_OK;
}

int pt_build_synth(pt_ctx* c, uint64_t n_total, uint64_t seed, int dist, int xyz_type, int slab_axis, double slab_lo, double slab_hi) {
  if (!c) return PT_ERR_ARG;
  if (dist != PT_DIST_UNIFORM && dist != PT_DIST_CLUSTERED) return fail(c, PT_ERR_ARG, "unknown distribution %d", dist);
  if (xyz_type != PT_F32 && xyz_type != PT_F64 && xyz_type != PT_F16) return fail(c, PT_ERR_ARG, "unknown xyz_type %d", xyz_type);
  const int f16 = xyz_type == PT_F16;      // fp16 values: resident as fp16, sorted into fp32 records (widening is exact)
  if (f16) xyz_type = PT_F32;
  if (slab_axis > 2) return fail(c, PT_ERR_ARG, "slab_axis must be < 3");
  { int r = check_n(c, n_total, "n_total"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  uint64_t n = n_total;
  const bool slab = slab_axis >= 0;
  // "local_ids": the slab is generated in INDEX ORDER (per-workgroup counts, a scan, ranked writes: pt_attr.hip) so that its records can
  // carry positions, and its attribute table holds its own points only
  const bool ordered = slab && c->want_local_ids;
  uint32_t *wg_cnt = nullptr, *wg_off = nullptr;
  const uint32_t nwg = (uint32_t)((n_total + 255) / 256);
  if (ordered) {
    RES(c, c->misc, ((size_t)nwg + 8) * 2 * sizeof(uint32_t) + ((size_t)nwg / 2048 + 16) * sizeof(uint32_t));
    wg_cnt = (uint32_t*)c->misc.p; wg_off = wg_cnt + nwg + 8;
    uint32_t* scan_tmp = wg_off + nwg + 8;
    HIPCHK(c, hipMemsetAsync(wg_cnt + nwg, 0, 4, c->stream));
    pt_launch_synth_xyz<float>(seed, 0, (uint32_t)n_total, slab_axis, slab_lo, slab_hi, nullptr, nullptr, nullptr, nullptr, nullptr, 0, f16, dist, n_total, 0, c->stream, wg_cnt, nullptr);
    pt_launch_scan_u32(wg_cnt, wg_off, nwg + 1, scan_tmp, c->stream);
    HIPCHK(c, hipMemcpyAsync(c->h_counter, wg_off + nwg, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    n = *c->h_counter;
    RES(c, c->in_gidx, std::max<uint64_t>(n, 1) * sizeof(uint32_t));
  } else if (slab) {   // counting pass
    HIPCHK(c, hipMemsetAsync(c->counter.p, 0, 4, c->stream));
    pt_launch_synth_xyz<float>(seed, 0, (uint32_t)n_total, slab_axis, slab_lo, slab_hi, nullptr, nullptr, nullptr, nullptr, (uint32_t*)c->counter.p, 0, f16, dist, n_total, 0, c->stream);
    HIPCHK(c, hipMemcpyAsync(c->h_counter, c->counter.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    n = *c->h_counter;
    RES(c, c->in_gidx, std::max<uint64_t>(n, 1) * sizeof(uint32_t));
    HIPCHK(c, hipMemsetAsync(c->counter.p, 0, 4, c->stream));
  }
  const bool keep_half = f16 != 0;
  RES(c, c->in_xyz, std::max<uint64_t>(n, 1) * 3 * (keep_half ? sizeof(__half) : tsize(xyz_type)));
  uint32_t* g = slab ? (uint32_t*)c->in_gidx.p : nullptr;
  if (keep_half) { __half* x = (__half*)c->in_xyz.p; pt_launch_synth_xyz<__half>(seed, 0, (uint32_t)n_total, slab_axis, slab_lo, slab_hi, x, x + n, x + 2 * n, g, (uint32_t*)c->counter.p, (uint32_t)n, 1, dist, n_total, 0, c->stream, nullptr, ordered ? wg_off : nullptr); }
  else if (xyz_type == PT_F32) { float* x = (float*)c->in_xyz.p; pt_launch_synth_xyz<float>(seed, 0, (uint32_t)n_total, slab_axis, slab_lo, slab_hi, x, x + n, x + 2 * n, g, (uint32_t*)c->counter.p, (uint32_t)n, f16, dist, n_total, 0, c->stream, nullptr, ordered ? wg_off : nullptr); }
  else { double* x = (double*)c->in_xyz.p; pt_launch_synth_xyz<double>(seed, 0, (uint32_t)n_total, slab_axis, slab_lo, slab_hi, x, x + n, x + 2 * n, g, (uint32_t*)c->counter.p, (uint32_t)n, f16, dist, n_total, 0, c->stream, nullptr, ordered ? wg_off : nullptr); }
  if (ordered) {                                   // the slab's own records only: 16 n bytes instead of 16 n_total
    RES(c, c->attr, std::max<uint64_t>(n, 1) * sizeof(Attr));
    pt_launch_synth_attr(seed, (uint32_t)n, (Attr*)c->attr.p, c->stream, (const uint32_t*)c->in_gidx.p);
  } else {
    RES(c, c->attr, std::max<uint64_t>(n_total, 1) * sizeof(Attr));
    pt_launch_synth_attr(seed, (uint32_t)n_total, (Attr*)c->attr.p, c->stream);
  }
  c->src_type = xyz_type; c->n = n; c->n_total = ordered ? n : n_total; c->synth_total = n_total; c->has_gidx = slab; c->local_mode = ordered; c->attr_local = ordered; c->has_attr = true; c->built = false; c->posattr_valid = false; c->bbox_guess_ok = true; c->pool_ok = true; c->pool2_ok = true; c->uniform_seen = false; c->uniform_known = false; c->hint_h = 0.0; c->in_half = keep_half; c->xyz32_valid = false;
  return rebuild(c);
}

int pt_rebuild(pt_ctx* c) {
  if (!c) return PT_ERR_ARG;
  return rebuild(c);
}

// ---- query ----------------------------------------------------------------------------------------
int pt_targets_synth(pt_ctx* c, uint64_t m_total, uint64_t seed, int dist, int xyz_type, int slab_axis, double slab_lo, double slab_hi) {
  if (!c) return PT_ERR_ARG;
  if (dist != PT_DIST_UNIFORM && dist != PT_DIST_CLUSTERED) return fail(c, PT_ERR_ARG, "unknown distribution %d", dist);
  if (xyz_type != PT_F32 && xyz_type != PT_F64 && xyz_type != PT_F16) return fail(c, PT_ERR_ARG, "unknown xyz_type %d", xyz_type);
  const int f16 = xyz_type == PT_F16;
  if (f16) xyz_type = PT_F32;
  if (slab_axis > 2) return fail(c, PT_ERR_ARG, "slab_axis must be < 3");
  if (dist == PT_DIST_CLUSTERED && c->src_type < 0) return fail(c, PT_ERR_STATE, "clustered targets are a subsample of the sources: build the cloud first");
  { int r = check_n(c, m_total, "m_total"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  uint64_t m = m_total;
  const bool slab = slab_axis >= 0;
  if (slab) {
    HIPCHK(c, hipMemsetAsync(c->counter.p, 0, 4, c->stream));
    pt_launch_synth_xyz<float>(seed, 1, (uint32_t)m_total, slab_axis, slab_lo, slab_hi, nullptr, nullptr, nullptr, nullptr, (uint32_t*)c->counter.p, 0, f16, dist, (c->local_mode && c->synth_total) ? c->synth_total : c->n_total, m_total, c->stream);
    HIPCHK(c, hipMemcpyAsync(c->h_counter, c->counter.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    m = *c->h_counter;
    HIPCHK(c, hipMemsetAsync(c->counter.p, 0, 4, c->stream));
  }
  RES(c, c->t_gidx, std::max<uint64_t>(m, 1) * sizeof(uint32_t));
  RES(c, c->t_xyz, std::max<uint64_t>(m, 1) * 3 * tsize(xyz_type));
  uint32_t* g = (uint32_t*)c->t_gidx.p;
  if (xyz_type == PT_F32) { float* x = (float*)c->t_xyz.p; pt_launch_synth_xyz<float>(seed, 1, (uint32_t)m_total, slab_axis, slab_lo, slab_hi, x, x + m, x + 2 * m, g, (uint32_t*)c->counter.p, (uint32_t)m, f16, dist, (c->local_mode && c->synth_total) ? c->synth_total : c->n_total, m_total, c->stream); }
  else { double* x = (double*)c->t_xyz.p; pt_launch_synth_xyz<double>(seed, 1, (uint32_t)m_total, slab_axis, slab_lo, slab_hi, x, x + m, x + 2 * m, g, (uint32_t*)c->counter.p, (uint32_t)m, f16, dist, (c->local_mode && c->synth_total) ? c->synth_total : c->n_total, m_total, c->stream); }
  c->tgt_type = xyz_type; c->m = m; c->t_has_gidx = true;
  return finish(c);
}

int pt_targets_soa(pt_ctx* c, const void* xyz, int xyz_type, uint64_t m, int on_device) {
  if (!c) return PT_ERR_ARG;
  { int r = check_n(c, m, "m"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  { int r = upload_xyz(c, c->t_xyz, xyz, xyz_type, m, on_device); if (r) return r; }      // (fp16 is widened to fp32 here)
  c->tgt_type = xyz_type; c->m = m; c->t_has_gidx = false;
  return finish(c);
}

int pt_targets_aos(pt_ctx* c, const pt_point* targets, uint64_t m) {
  if (!c) return PT_ERR_ARG;
  if (m && !targets) return fail(c, PT_ERR_ARG, "targets is null");
  { int r = check_n(c, m, "m"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->aos_stage, std::max<uint64_t>(m, 1) * sizeof(pt_point));
  RES(c, c->t_xyz, std::max<uint64_t>(m, 1) * 3 * sizeof(double));
  { int r = copy_in(c, c->aos_stage.p, targets, m * sizeof(pt_point), 0); if (r) return r; }
  double* x = (double*)c->t_xyz.p;
  pt_launch_aos_split(c->aos_stage.p, (uint32_t)m, x, x + m, x + 2 * m, nullptr, c->stream);
  c->tgt_type = PT_F64; c->m = m; c->t_has_gidx = false;
  return finish(c);
}

int pt_query_resident(pt_ctx* c, int k, uint32_t* idx_dev, double* d2_dev_or_null) {
  if (!c) return PT_ERR_ARG;
  if (c->tgt_type < 0) return fail(c, PT_ERR_STATE, "no resident targets");
  return run_query(c, c->t_xyz.p, c->tgt_type, c->m, k, nullptr, idx_dev, d2_dev_or_null);
}

int pt_query_blend_resident(pt_ctx* c, int k, int mode, uint32_t* idx_dev, double* d2_dev_or_null, float* rgb_out_dev, float* nrm_out_dev) {
  if (!c) return PT_ERR_ARG;
  if (c->tgt_type < 0) return fail(c, PT_ERR_STATE, "no resident targets");
  if (!c->has_attr) return fail(c, PT_ERR_STATE, "no attribute table resident");
  if (mode != PT_BLEND_MEAN && mode != PT_BLEND_INV_D2) return fail(c, PT_ERR_ARG, "unknown blend mode %d", mode);
  if (mode == PT_BLEND_INV_D2 && !d2_dev_or_null) return fail(c, PT_ERR_ARG, "inverse-d2 blend needs d2");
  const BlendReq br{mode, rgb_out_dev, nrm_out_dev};
  c->st.ms_blend = 0.f;          // part of ms_query here
  return run_query(c, c->t_xyz.p, c->tgt_type, c->m, k, nullptr, idx_dev, d2_dev_or_null, &br);
}

int pt_query_resident_host(pt_ctx* c, int k, int blend_mode, uint32_t* idx_out, double* d2_out, float* rgb_out, float* nrm_out) {
  if (!c) return PT_ERR_ARG;
  if (c->tgt_type < 0) return fail(c, PT_ERR_STATE, "no resident targets");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k = %d out of range [1, %d]", k, PT_MAX_K);
  if (c->m && !idx_out) return fail(c, PT_ERR_ARG, "idx output is null");
  const bool blend = blend_mode >= 0 && (rgb_out || nrm_out);
  if (blend && blend_mode != PT_BLEND_MEAN && blend_mode != PT_BLEND_INV_D2) return fail(c, PT_ERR_ARG, "unknown blend mode %d", blend_mode);
  if (blend && !c->has_attr) return fail(c, PT_ERR_STATE, "no attribute table resident");
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t m = c->m;
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  RES(c, c->q_d2, std::max<uint64_t>(m, 1) * k * sizeof(double));
  if (blend) { RES(c, c->b_rgb, std::max<uint64_t>(m, 1) * 12); RES(c, c->b_nrm, std::max<uint64_t>(m, 1) * 12); }
  const BlendReq br{blend_mode, (float*)c->b_rgb.p, (float*)c->b_nrm.p};
  { int r = run_query(c, c->t_xyz.p, c->tgt_type, m, k, nullptr, (uint32_t*)c->q_idx.p, (double*)c->q_d2.p, blend ? &br : nullptr); if (r != PT_OK) return r; }
  if (m) {
    HIPCHK(c, hipMemcpyAsync(idx_out, c->q_idx.p, m * k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (d2_out) HIPCHK(c, hipMemcpyAsync(d2_out, c->q_d2.p, m * k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (blend && rgb_out) HIPCHK(c, hipMemcpyAsync(rgb_out, c->b_rgb.p, m * 12, hipMemcpyDeviceToHost, c->stream));
    if (blend && nrm_out) HIPCHK(c, hipMemcpyAsync(nrm_out, c->b_nrm.p, m * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return PT_OK;
}

int pt_resident_target_ids(pt_ctx* c, uint32_t* ids_dev) {
  if (!c || !ids_dev) return PT_ERR_ARG;
  if (c->tgt_type < 0) return fail(c, PT_ERR_STATE, "no resident targets");
  if (c->t_has_gidx) HIPCHK(c, hipMemcpyAsync(ids_dev, c->t_gidx.p, c->m * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  else pt_launch_iota(ids_dev, (uint32_t)c->m, c->stream);
  return finish(c);
}

int pt_resident_target_xyz(pt_ctx* c, void* xyz_dev) {
  if (!c || !xyz_dev) return PT_ERR_ARG;
  if (c->tgt_type < 0) return fail(c, PT_ERR_STATE, "no resident targets");
  HIPCHK(c, hipMemcpyAsync(xyz_dev, c->t_xyz.p, c->m * 3 * tsize(c->tgt_type), hipMemcpyDeviceToDevice, c->stream));
  return finish(c);
}

int pt_resident_source_xyz(pt_ctx* c, void* xyz_dev, int* xyz_type_out) {
  if (!c || !xyz_dev) return PT_ERR_ARG;
  if (c->src_type < 0) return fail(c, PT_ERR_STATE, "no source cloud resident");
  const size_t es = c->in_half ? sizeof(__half) : tsize(c->src_type);
  HIPCHK(c, hipMemcpyAsync(xyz_dev, c->in_xyz.p, c->n * 3 * es, hipMemcpyDeviceToDevice, c->stream));
  if (xyz_type_out) *xyz_type_out = c->in_half ? PT_F16 : c->src_type;
  return finish(c);
}

int pt_query_soa(pt_ctx* c, const void* xyz, int xyz_type, uint64_t m, int k, int on_device, uint32_t* idx, double* d2_or_null) {
  if (!c) return PT_ERR_ARG;
  if (!c->built) return fail(c, PT_ERR_STATE, "query before build");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k = %d out of range [1, %d]", k, PT_MAX_K);
  if (m && !idx) return fail(c, PT_ERR_ARG, "idx output is null");
  { int r = load_transient(c, xyz, xyz_type, m, on_device); if (r) return r; }
  if (on_device) return run_query(c, c->x_xyz.p, xyz_type, m, k, nullptr, idx, d2_or_null);
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  if (d2_or_null) RES(c, c->q_d2, std::max<uint64_t>(m, 1) * k * sizeof(double));
  { int r = run_query(c, c->x_xyz.p, xyz_type, m, k, nullptr, (uint32_t*)c->q_idx.p, d2_or_null ? (double*)c->q_d2.p : nullptr); if (r) return r; }
  if (m) {
    HIPCHK(c, hipMemcpyAsync(idx, c->q_idx.p, m * k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (d2_or_null) HIPCHK(c, hipMemcpyAsync(d2_or_null, c->q_d2.p, m * k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return PT_OK;
}

int pt_query_aos(pt_ctx* c, const pt_point* targets, uint64_t m, int k, uint32_t* idx, double* d2_or_null) {
  if (!c) return PT_ERR_ARG;
  if (!c->built) return fail(c, PT_ERR_STATE, "query before build");
  if (m && (!targets || !idx)) return fail(c, PT_ERR_ARG, "null argument");
  { int r = check_n(c, m, "m"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->aos_stage, std::max<uint64_t>(m, 1) * sizeof(pt_point));
  RES(c, c->x_xyz, std::max<uint64_t>(m, 1) * 3 * sizeof(double));
  { int r = copy_in(c, c->aos_stage.p, targets, m * sizeof(pt_point), 0); if (r) return r; }
  double* x = (double*)c->x_xyz.p;
  pt_launch_aos_split(c->aos_stage.p, (uint32_t)m, x, x + m, x + 2 * m, nullptr, c->stream);
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  if (d2_or_null) RES(c, c->q_d2, std::max<uint64_t>(m, 1) * k * sizeof(double));
  { int r = run_query(c, c->x_xyz.p, PT_F64, m, k, nullptr, (uint32_t*)c->q_idx.p, d2_or_null ? (double*)c->q_d2.p : nullptr); if (r) return r; }
  if (m) {
    HIPCHK(c, hipMemcpyAsync(idx, c->q_idx.p, m * k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (d2_or_null) HIPCHK(c, hipMemcpyAsync(d2_or_null, c->q_d2.p, m * k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return PT_OK;
}

int pt_query_bounded_dev(pt_ctx* c, const void* xyz_dev, int xyz_type, const double* bound2_dev, uint64_t m, int k, uint32_t* idx_dev,
                         double* d2_dev) {
  if (!c) return PT_ERR_ARG;
  if (!c->built) return fail(c, PT_ERR_STATE, "query before build");
  { int r = load_transient(c, xyz_dev, xyz_type, m, 1); if (r) return r; }
  return run_query(c, c->x_xyz.p, xyz_type, m, k, bound2_dev, idx_dev, d2_dev);
}

// ---- blend / PCA -----------------------------------------------------------------------------------
int pt_blend_dev(pt_ctx* c, const uint32_t* idx_dev, const double* d2_dev_or_null, uint64_t m, int k, int mode, float* rgb_out_dev,
                 float* nrm_out_dev) {
  if (!c) return PT_ERR_ARG;
  if (!c->has_attr) return fail(c, PT_ERR_STATE, "no attribute table resident");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k out of range");
  if (mode != PT_BLEND_MEAN && mode != PT_BLEND_INV_D2) return fail(c, PT_ERR_ARG, "unknown blend mode %d", mode);
  if (mode == PT_BLEND_INV_D2 && !d2_dev_or_null) return fail(c, PT_ERR_ARG, "inverse-d2 blend needs d2");
  if (m && !idx_dev) return fail(c, PT_ERR_ARG, "idx is null");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (c->attr_local) {
    // the table holds this slab's points by POSITION: the lists' global indices are looked up in the slab's ascending gidx; an entry that
    // names another slab's point (a row the exchange completed) cannot be blended from here -- pt_exchange_merge_* re-blends those rows itself
    RES(c, c->l_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
    pt_launch_ids_to_local(idx_dev, (size_t)m * k, (const uint32_t*)c->in_gidx.p, (uint32_t)c->n, (uint32_t*)c->l_idx.p, c->stream);
    idx_dev = (const uint32_t*)c->l_idx.p;
  }
  pt_launch_blend(idx_dev, d2_dev_or_null, (uint32_t)m, k, mode, (const Attr*)c->attr.p, (uint32_t)c->n_total, rgb_out_dev, nrm_out_dev, c->stream);
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  HIPCHK(c, hipGetLastError());
  if (c->sync) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->st.ms_blend = ms;
  }
  return PT_OK;
}

int pt_blend(pt_ctx* c, const uint32_t* idx, const double* d2_or_null, uint64_t m, int k, int mode, float* rgb_out, float* nrm_out) {
  if (!c) return PT_ERR_ARG;
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k out of range");
  if (m && !idx) return fail(c, PT_ERR_ARG, "idx is null");
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  RES(c, c->q_d2, std::max<uint64_t>(m, 1) * k * sizeof(double));
  RES(c, c->b_rgb, std::max<uint64_t>(m, 1) * 12);
  RES(c, c->b_nrm, std::max<uint64_t>(m, 1) * 12);
  { int r = copy_in(c, c->q_idx.p, idx, m * k * sizeof(uint32_t), 0); if (r) return r; }
  if (d2_or_null) { int r = copy_in(c, c->q_d2.p, d2_or_null, m * k * sizeof(double), 0); if (r) return r; }
  const int sync_save = c->sync;
  c->sync = 1;
  int r = pt_blend_dev(c, (const uint32_t*)c->q_idx.p, d2_or_null ? (const double*)c->q_d2.p : nullptr, m, k, mode, (float*)c->b_rgb.p, (float*)c->b_nrm.p);
  c->sync = sync_save;
  if (r != PT_OK) return r;
  if (m) {
    if (rgb_out) HIPCHK(c, hipMemcpy(rgb_out, c->b_rgb.p, m * 12, hipMemcpyDeviceToHost));
    if (nrm_out) HIPCHK(c, hipMemcpy(nrm_out, c->b_nrm.p, m * 12, hipMemcpyDeviceToHost));
  }
  return PT_OK;
}

int pt_blend_weighted_dev(pt_ctx* c, const uint32_t* idx_dev, const double* w_dev, uint64_t m, int k, float* rgb_out_dev, float* nrm_out_dev) {
  if (!c) return PT_ERR_ARG;
  if (!c->has_attr) return fail(c, PT_ERR_STATE, "no attribute table resident");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k out of range");
  if (m && (!idx_dev || !w_dev)) return fail(c, PT_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->attr_local) {                                       // (as pt_blend_dev: global indices -> positions in this slab's table)
    RES(c, c->l_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
    pt_launch_ids_to_local(idx_dev, (size_t)m * k, (const uint32_t*)c->in_gidx.p, (uint32_t)c->n, (uint32_t*)c->l_idx.p, c->stream);
    idx_dev = (const uint32_t*)c->l_idx.p;
  }
  pt_launch_blend_weighted(idx_dev, w_dev, (uint32_t)m, k, (const Attr*)c->attr.p, (uint32_t)c->n_total, rgb_out_dev, nrm_out_dev, c->stream);
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pt_blend_weighted(pt_ctx* c, const uint32_t* idx, const double* w, uint64_t m, int k, float* rgb_out, float* nrm_out) {
  if (!c) return PT_ERR_ARG;
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k out of range");
  if (m && (!idx || !w)) return fail(c, PT_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  RES(c, c->q_d2, std::max<uint64_t>(m, 1) * k * sizeof(double));
  RES(c, c->b_rgb, std::max<uint64_t>(m, 1) * 12);
  RES(c, c->b_nrm, std::max<uint64_t>(m, 1) * 12);
  { int r = copy_in(c, c->q_idx.p, idx, m * k * sizeof(uint32_t), 0); if (r) return r; }
  { int r = copy_in(c, c->q_d2.p, w, m * k * sizeof(double), 0); if (r) return r; }
  const int sync_save = c->sync;
  c->sync = 1;
  int r = pt_blend_weighted_dev(c, (const uint32_t*)c->q_idx.p, (const double*)c->q_d2.p, m, k, (float*)c->b_rgb.p, (float*)c->b_nrm.p);
  c->sync = sync_save;
  if (r != PT_OK) return r;
  if (m) {
    if (rgb_out) HIPCHK(c, hipMemcpy(rgb_out, c->b_rgb.p, m * 12, hipMemcpyDeviceToHost));
    if (nrm_out) HIPCHK(c, hipMemcpy(nrm_out, c->b_nrm.p, m * 12, hipMemcpyDeviceToHost));
  }
  return PT_OK;
}

int pt_pca_normals_dev(pt_ctx* c, const uint32_t* idx_dev, uint64_t m, int k, float* nrm_out_dev) {
  if (!c) return PT_ERR_ARG;
  if (c->src_type < 0) return fail(c, PT_ERR_STATE, "no source cloud resident");
  if (c->has_gidx) return fail(c, PT_ERR_UNSUPPORTED, "PCA normals need the whole cloud resident (not a slab)");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k out of range");
  if (m && (!idx_dev || !nrm_out_dev)) return fail(c, PT_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  const Attr* at = c->has_attr ? (const Attr*)c->attr.p : nullptr;
  if (c->src_type == PT_F32) {
    // one 32-byte gather per neighbour instead of four 4..16-byte ones: the table depends on the cloud only, so it is
    // built on the first call after an upload and reused by every later one (rebuilds of the grid do not touch it)
    const float* x = nullptr;
    if (!c->posattr_valid) { int r = source_xyz_f32(c, &x); if (r) return r; }
    if (!c->posattr_valid) {
      RES(c, c->posattr, std::max<uint64_t>(c->n, 1) * 32);
      pt_launch_pack_posattr(x, x + c->n, x + 2 * c->n, at, (uint32_t)c->n, c->posattr.p, c->stream);
      c->posattr_valid = true;
      HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    }
    pt_launch_pca_posattr(idx_dev, (uint32_t)m, k, c->posattr.p, (uint32_t)c->n, at ? 1 : 0, nrm_out_dev, c->stream);
  }
  else { const double* x = (const double*)c->in_xyz.p; pt_launch_pca<double>(idx_dev, (uint32_t)m, k, x, x + c->n, x + 2 * c->n, (uint32_t)c->n, at, nrm_out_dev, c->stream); }
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  HIPCHK(c, hipGetLastError());
  if (c->sync) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->st.ms_pca = ms;
  }
  return PT_OK;
}

int pt_pca_normals(pt_ctx* c, const uint32_t* idx, uint64_t m, int k, float* nrm_out) {
  if (!c) return PT_ERR_ARG;
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k out of range");
  if (m && (!idx || !nrm_out)) return fail(c, PT_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  RES(c, c->b_nrm, std::max<uint64_t>(m, 1) * 12);
  { int r = copy_in(c, c->q_idx.p, idx, m * k * sizeof(uint32_t), 0); if (r) return r; }
  const int sync_save = c->sync;
  c->sync = 1;
  int r = pt_pca_normals_dev(c, (const uint32_t*)c->q_idx.p, m, k, (float*)c->b_nrm.p);
  c->sync = sync_save;
  if (r != PT_OK) return r;
  if (m) HIPCHK(c, hipMemcpy(nrm_out, c->b_nrm.p, m * 12, hipMemcpyDeviceToHost));
  return PT_OK;
}

// ---- multi-GPU helpers -------------------------------------------------------------------------------
int pt_merge_candidates_dev(pt_ctx* c, const uint32_t* idx_lists_dev, const double* d2_lists_dev, int g, uint64_t m, int k,
                            uint32_t* idx_out_dev, double* d2_out_dev) {
  if (!c) return PT_ERR_ARG;
  if (g < 1 || g > 64 || k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "g or k out of range");
  if (m && (!idx_lists_dev || !d2_lists_dev || !idx_out_dev || !d2_out_dev)) return fail(c, PT_ERR_ARG, "null argument");
  { int r = check_n(c, m, "m"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  pt_launch_merge(idx_lists_dev, d2_lists_dev, g, (uint32_t)m, k, idx_out_dev, d2_out_dev, c->stream);
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pt_slab_need_dev(pt_ctx* c, const void* tgt_xyz_dev, int xyz_type, const double* d2_dev, uint64_t m, int k, int slab_axis,
                     const double* slab_bounds, int g, int my_slab, uint8_t* need_dev) {
  if (!c) return PT_ERR_ARG;
  if (xyz_type != PT_F32 && xyz_type != PT_F64) return fail(c, PT_ERR_UNSUPPORTED, "xyz_type %d not supported yet", xyz_type);
  if (g < 1 || g > 64 || k < 1 || k > PT_MAX_K || slab_axis < 0 || slab_axis > 2 || my_slab < 0 || my_slab >= g) return fail(c, PT_ERR_ARG, "argument out of range");
  if (m && (!tgt_xyz_dev || !d2_dev || !need_dev || !slab_bounds)) return fail(c, PT_ERR_ARG, "null argument");
  { int r = check_n(c, m, "m"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->bounds, 65 * sizeof(double));
  HIPCHK(c, hipMemcpyAsync(c->bounds.p, slab_bounds, (size_t)(g + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (xyz_type == PT_F32) { const float* x = (const float*)tgt_xyz_dev; pt_launch_slab_need<float>(x, x + m, x + 2 * m, d2_dev, (uint32_t)m, k, slab_axis, (const double*)c->bounds.p, g, my_slab, need_dev, c->stream); }
  else { const double* x = (const double*)tgt_xyz_dev; pt_launch_slab_need<double>(x, x + m, x + 2 * m, d2_dev, (uint32_t)m, k, slab_axis, (const double*)c->bounds.p, g, my_slab, need_dev, c->stream); }
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pt_pack_requests_dev(pt_ctx* c, const void* tgt_xyz_dev, int xyz_type, const double* d2_dev, uint64_t m, int k, int slab_axis,
                         const double* slab_bounds, int g, int my_slab, uint32_t* sel_out_dev, double* pkt_out_dev, uint32_t* count_out) {
  if (!c) return PT_ERR_ARG;
  if (xyz_type != PT_F32 && xyz_type != PT_F64) return fail(c, PT_ERR_UNSUPPORTED, "xyz_type %d not supported yet", xyz_type);
  if (g < 1 || g > 52 || k < 1 || k > PT_MAX_K || slab_axis < 0 || slab_axis > 2 || my_slab < 0 || my_slab >= g) return fail(c, PT_ERR_ARG, "argument out of range");
  if (!count_out || !slab_bounds || (m && (!tgt_xyz_dev || !d2_dev || !sel_out_dev || !pkt_out_dev))) return fail(c, PT_ERR_ARG, "null argument");
  { int r = check_n(c, m, "m"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->bounds, 65 * sizeof(double));
  uint32_t* cnt = (uint32_t*)c->counter.p + 6;
  HIPCHK(c, hipMemcpyAsync(c->bounds.p, slab_bounds, (size_t)(g + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(cnt, 0, 4, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));       // (slab_bounds is the caller's host memory)
  if (xyz_type == PT_F32) { const float* x = (const float*)tgt_xyz_dev; pt_launch_request_pack<float>(x, x + m, x + 2 * m, d2_dev, (uint32_t)m, k, slab_axis, (const double*)c->bounds.p, g, my_slab, cnt, sel_out_dev, pkt_out_dev, c->stream); }
  else { const double* x = (const double*)tgt_xyz_dev; pt_launch_request_pack<double>(x, x + m, x + 2 * m, d2_dev, (uint32_t)m, k, slab_axis, (const double*)c->bounds.p, g, my_slab, cnt, sel_out_dev, pkt_out_dev, c->stream); }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_counter + 6, cnt, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *count_out = c->h_counter[6];
  return PT_OK;
}

// ---- streamed upload -----------------------------------------------------------------------------------------------
void* pt_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void pt_host_free(void* p) { if (p) (void)hipHostFree(p); }

int pt_upload_begin(pt_ctx* c, uint64_t n, int xyz_type, int with_attributes) {
  if (!c) return PT_ERR_ARG;
  if (xyz_type != PT_F32 && xyz_type != PT_F64) return fail(c, PT_ERR_ARG, "pt_upload_begin: xyz_type must be PT_F32 or PT_F64");
  { int r = check_n(c, n, "n"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  RES(c, c->in_xyz, std::max<uint64_t>(n, 1) * 3 * tsize(xyz_type));
  if (with_attributes) {
    RES(c, c->up_rgb, std::max<uint64_t>(n, 1) * 3);
    RES(c, c->up_nrm, std::max<uint64_t>(n, 1) * 12);
    RES(c, c->attr, std::max<uint64_t>(n, 1) * sizeof(Attr));
  }
  c->up_n = n; c->up_type = xyz_type; c->up_attr = with_attributes ? 1 : 0;
  c->built = false; c->src_type = -1;            // nothing usable until pt_upload_end
  return PT_OK;
}

int pt_upload_range(pt_ctx* c, uint64_t first, uint64_t count, const void* x, const void* y, const void* z, const uint8_t* rgb, const float* nrm) {
  if (!c) return PT_ERR_ARG;
  if (c->up_type < 0) return PT_ERR_STATE;                                    // (no error text: several threads may be in here)
  if (first > c->up_n || count > c->up_n - first) return PT_ERR_ARG;
  if (count && (!x || !y || !z || (c->up_attr && (!rgb || !nrm)))) return PT_ERR_ARG;
  if (!count) return PT_OK;
  if (hipSetDevice(c->device) != hipSuccess) return PT_ERR_HIP;               // the calling thread's current device
  const size_t ts = tsize(c->up_type);
  char* base = (char*)c->in_xyz.p;
  const void* src[3] = {x, y, z};
  for (int a = 0; a < 3; ++a)
    if (hipMemcpyAsync(base + ((size_t)a * c->up_n + first) * ts, src[a], count * ts, hipMemcpyHostToDevice, c->stream) != hipSuccess) return PT_ERR_HIP;
  if (c->up_attr) {
    if (hipMemcpyAsync((char*)c->up_rgb.p + first * 3, rgb, count * 3, hipMemcpyHostToDevice, c->stream) != hipSuccess) return PT_ERR_HIP;
    if (hipMemcpyAsync((char*)c->up_nrm.p + first * 12, nrm, count * 12, hipMemcpyHostToDevice, c->stream) != hipSuccess) return PT_ERR_HIP;
  }
  return PT_OK;
}

int pt_upload_end(pt_ctx* c) {
  if (!c) return PT_ERR_ARG;
  if (c->up_type < 0) return fail(c, PT_ERR_STATE, "pt_upload_end without pt_upload_begin");
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t n = c->up_n;
  if (c->up_attr) pt_launch_pack_attr((const uint8_t*)c->up_rgb.p, (const float*)c->up_nrm.p, (uint32_t)n, (Attr*)c->attr.p, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));                                // the caller's buffers are free again
  c->src_type = c->up_type; c->n = n; c->n_total = n; c->has_gidx = false; c->local_mode = false; c->attr_local = false; c->has_attr = c->up_attr != 0; c->built = false;
  c->posattr_valid = false; c->bbox_guess_ok = true; c->pool_ok = true; c->pool2_ok = true; c->uniform_seen = false; c->uniform_known = false; c->hint_h = 0.0; c->in_half = false; c->xyz32_valid = false;
  c->up_type = -1;
  release(c, c->up_rgb); release(c, c->up_nrm);
  return rebuild(c);
}

// ---- out-of-core source --------------------------------------------------------------------------------------------
int pt_stream_query(pt_ctx* c, const void* xyz, int xyz_type, uint64_t n, uint64_t chunk_points, uint64_t first_id, int k, uint64_t* idx64_out,
                    double* d2_out) {
  if (!c) return PT_ERR_ARG;
  if (xyz_type != PT_F32 && xyz_type != PT_F64) return fail(c, PT_ERR_ARG, "pt_stream_query: xyz_type must be PT_F32 or PT_F64");
  if (c->tgt_type < 0) return fail(c, PT_ERR_STATE, "no resident targets (pt_targets_* first)");
  if (c->tgt_type != xyz_type) return fail(c, PT_ERR_UNSUPPORTED, "target xyz type %d differs from the cloud's %d", c->tgt_type, xyz_type);
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k = %d out of range [1, %d]", k, PT_MAX_K);
  if (chunk_points < 1) return fail(c, PT_ERR_ARG, "chunk_points must be positive");
  if ((n && !xyz) || (c->m && (!idx64_out || !d2_out))) return fail(c, PT_ERR_ARG, "null argument");
  chunk_points = std::min<uint64_t>(chunk_points, 0xFFFFFFF0ull - 1);
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t m = c->m;
  const size_t ts = tsize(xyz_type), lists = std::max<uint64_t>(m, 1) * (size_t)k;
  const uint64_t nchunks = n ? (n + chunk_points - 1) / chunk_points : 0;
  DevBuf best_i[2], best_d[2], ci, cd, stage[2], sbound, sfirst;
  uint64_t skipped = 0, revisited = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
  const int sync_save = c->sync;
  DevBuf keep_in = c->in_xyz;                       // the context's own input buffer: put back at the end (the chunks live in `stage`)
  const bool keep_half = c->in_half;
  auto cleanup = [&]() {
    c->in_xyz = keep_in;
    c->in_half = keep_half; c->xyz32_valid = false;
    c->sync = sync_save;
    c->tile_bounds = false;
    DevBuf* all[] = {&best_i[0], &best_i[1], &best_d[0], &best_d[1], &ci, &cd, &stage[0], &stage[1], &sbound, &sfirst};
    for (DevBuf* b : all) release(c, *b);
    for (auto& e : copied) if (e) (void)hipEventDestroy(e);
    for (auto& e : consumed) if (e) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
  };
  auto run = [&]() -> int {
    for (int b = 0; b < 2; ++b) { RES(c, best_i[b], lists * 8); RES(c, best_d[b], lists * 8); }
    RES(c, ci, lists * 4); RES(c, cd, lists * 8); RES(c, sbound, std::max<uint64_t>(m, 1) * 8);
    HIPCHK(c, hipMemsetAsync(best_i[0].p, 0xFF, lists * 8, c->stream));                          // ~0 = no neighbour yet
    HIPCHK(c, hipMemsetAsync(best_d[0].p, 0x7F, lists * 8, c->stream));    // (never compared: the merge reads a distance only beside a valid id)
    HIPCHK(c, hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      RES(c, stage[b], std::max<uint64_t>(std::min<uint64_t>(chunk_points, std::max<uint64_t>(n, 1)), 1) * 3 * ts);
      HIPCHK(c, hipEventCreateWithFlags(&copied[b], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
    }
    uint64_t held[2] = {~0ull, ~0ull};               // which chunk each stage buffer holds
    auto upload = [&](uint64_t ch) -> int {          // chunk ch -> stage[ch & 1], planar with the chunk's own length as the stride
      const int b = (int)(ch & 1);
      held[b] = ch;
      const uint64_t f0 = ch * chunk_points, cnt = std::min<uint64_t>(chunk_points, n - f0);
      HIPCHK(c, hipStreamWaitEvent(copy_stream, consumed[b], 0));      // (a never-recorded event does not block)
      for (int a = 0; a < 3; ++a)
        HIPCHK(c, hipMemcpyAsync((char*)stage[b].p + (size_t)a * cnt * ts, (const char*)xyz + ((size_t)a * n + f0) * ts, cnt * ts, hipMemcpyHostToDevice, copy_stream));
      HIPCHK(c, hipEventRecord(copied[b], copy_stream));
      return PT_OK;
    };
    c->sync = 1;
    int cur = 0;
    RES(c, sfirst, std::max<uint64_t>(m, 1) * 4);
    {   // first[t] = nchunks: not searched in any chunk yet
      std::vector<uint32_t> init((size_t)std::max<uint64_t>(m, 1), (uint32_t)nchunks);
      HIPCHK(c, hipMemcpyAsync(sfirst.p, init.data(), init.size() * 4, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    std::vector<double> boxes((size_t)nchunks * 6), margins((size_t)nchunks, 0.0);
    uint32_t* cnt_dev = (uint32_t*)c->counter.p + 6;
    // one (chunk, sweep): stage[b] holds chunk ch -- exact bounding box (forward sweep; remembered for the backward one), the bounds the
    // targets bring and how many of them reach the box at all, then build + search + merge unless nobody does
    auto adopt = [&](uint64_t ch, int b) {
      const uint64_t f0 = ch * chunk_points, cnt = std::min<uint64_t>(chunk_points, n - f0);
      c->in_xyz = stage[b];
      c->src_type = xyz_type; c->n = cnt; c->n_total = cnt; c->has_gidx = false; c->local_mode = false; c->attr_local = false; c->has_attr = false; c->built = false;
      c->posattr_valid = false; c->bbox_guess_ok = true; c->pool_ok = true; c->pool2_ok = true; c->uniform_seen = false; c->uniform_known = false; c->hint_h = 0.0; c->in_half = false; c->xyz32_valid = false;
    };
    auto bounds_and_reach = [&](uint64_t ch, int backward, uint32_t& reach) -> int {
      const double* mn = &boxes[(size_t)ch * 6];
      HIPCHK(c, hipMemsetAsync(cnt_dev, 0, 4, c->stream));
      if (xyz_type == PT_F32) pt_launch_stream_sweep<float>((const float*)c->t_xyz.p, (const unsigned long long*)best_i[cur].p, (const double*)best_d[cur].p, (uint32_t)m, k, (uint32_t)ch,
                                                            backward, (uint32_t*)sfirst.p, mn, mn + 3, margins[(size_t)ch], (double*)sbound.p, cnt_dev, c->stream);
      else pt_launch_stream_sweep<double>((const double*)c->t_xyz.p, (const unsigned long long*)best_i[cur].p, (const double*)best_d[cur].p, (uint32_t)m, k, (uint32_t)ch,
                                          backward, (uint32_t*)sfirst.p, mn, mn + 3, margins[(size_t)ch], (double*)sbound.p, cnt_dev, c->stream);
      HIPCHK(c, hipMemcpyAsync(c->h_counter + 6, cnt_dev, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      reach = c->h_counter[6];
      return PT_OK;
    };
    auto search_and_merge = [&](uint64_t ch, int b) -> int {
      const uint64_t f0 = ch * chunk_points;
      { int r = rebuild(c); if (r) return r; }
      HIPCHK(c, hipEventRecord(consumed[b], c->stream));               // the build no longer reads stage[b] (records hold the coordinates)
      c->tile_bounds = true;
      { int r = run_query(c, c->t_xyz.p, c->tgt_type, m, k, (const double*)sbound.p, (uint32_t*)ci.p, (double*)cd.p); c->tile_bounds = false; if (r) return r; }
      pt_launch_merge_stream((const unsigned long long*)best_i[cur].p, (const double*)best_d[cur].p, (const uint32_t*)ci.p, (const double*)cd.p,
                             (unsigned long long)(first_id + f0), (uint32_t)m, k, (unsigned long long*)best_i[cur ^ 1].p, (double*)best_d[cur ^ 1].p, c->stream);
      HIPCHK(c, hipGetLastError());
      cur ^= 1;
      return PT_OK;
    };
    // ---- forward sweep: every chunk travels once, the next one while this one is searched ----
    if (nchunks) { int r = upload(0); if (r) return r; }
    for (uint64_t ch = 0; ch < nchunks; ++ch) {
      const int b = (int)(ch & 1);
      if (ch + 1 < nchunks) { int r = upload(ch + 1); if (r) return r; }          // the next chunk travels while this one is searched
      HIPCHK(c, hipStreamWaitEvent(c->stream, copied[b], 0));
      adopt(ch, b);
      double mn[3], mx[3];
      { int r = source_bbox(c, 1u, mn, mx); if (r) return r; }                      // exact box: 12 bytes per point, against a build and a search
      for (int a = 0; a < 3; ++a) { boxes[(size_t)ch * 6 + a] = mn[a]; boxes[(size_t)ch * 6 + 3 + a] = mx[a]; }
      {   // "inside" is taken with a margin of two point spacings times the cube root of k: the chunk's own density says over what distance a
          // target just outside its box still finds its neighbours here (and an unbounded search from that far outside costs a ring or two)
        double ext[3], big = 0.0, vol = 1.0;
        for (int a = 0; a < 3; ++a) { ext[a] = mx[a] - mn[a]; big = std::max(big, ext[a]); }
        for (int a = 0; a < 3; ++a) vol *= std::max(ext[a], big * 1e-3);
        const uint64_t cntc = std::min<uint64_t>(chunk_points, n - ch * chunk_points);
        margins[(size_t)ch] = big > 0.0 ? 2.0 * std::cbrt(vol / (double)std::max<uint64_t>(cntc, 1)) * std::cbrt((double)k) : 0.0;
      }
      uint32_t reach = (uint32_t)m;
      if (m && c->stream_bounds) { int r = bounds_and_reach(ch, 0, reach); if (r) return r; }
      else if (m) {     // measurement switch: round 2's behaviour -- every target, unbounded, in every chunk
        std::vector<double> inf((size_t)m, std::numeric_limits<double>::infinity());
        HIPCHK(c, hipMemcpyAsync(sbound.p, inf.data(), (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
      }
      if (m && reach == 0) { HIPCHK(c, hipEventRecord(consumed[b], c->stream)); ++skipped; continue; }    // nobody can gain anything from this chunk now
      { int r = search_and_merge(ch, b); if (r) return r; }
    }
    // ---- backward sweep: the deferred (target, chunk) pairs -- targets that lay outside a chunk's box before they had a list.  None for a
    //      cloud whose chunks each cover the whole volume (every target is inside every box); for a cloud in spatial order the targets near
    //      the border of their own slab bring the earlier chunks back, one upload each, and only those ----
    if (m && c->stream_bounds) {
      for (uint64_t ch = nchunks; ch-- > 0;) {
        uint32_t reach = 0;
        { int r = bounds_and_reach(ch, 1, reach); if (r) return r; }
        if (!reach) continue;
        const int b = (int)(ch & 1);
        if (held[b] != ch) { int r = upload(ch); if (r) return r; }               // (the last two chunks of the forward sweep are still in the stage buffers)
        HIPCHK(c, hipStreamWaitEvent(c->stream, copied[b], 0));
        adopt(ch, b);
        { int r = search_and_merge(ch, b); if (r) return r; }
        ++revisited;
      }
    }
    if (m) {
      HIPCHK(c, hipMemcpyAsync(idx64_out, best_i[cur].p, (size_t)m * k * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(d2_out, best_d[cur].p, (size_t)m * k * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(copy_stream));
    if (m && !nchunks) for (size_t i = 0; i < (size_t)m * k; ++i) d2_out[i] = std::numeric_limits<double>::infinity();
    return PT_OK;
  };
  const int r = run();
  cleanup();
  c->st.stream_skipped = (int32_t)std::min<uint64_t>(skipped, 0x7FFFFFFF);
  c->st.stream_revisited = (int32_t)std::min<uint64_t>(revisited, 0x7FFFFFFF);
  // the chunks are gone with the stage buffers: no source cloud is resident any more (a later query needs a pt_build_* first)
  c->n = 0; c->n_total = 0; c->built = false; c->src_type = -1; c->has_attr = false;
  return r;
}

// ---- native slab exchange ----------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;            // optional: error paths must not wait for peers
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl() {          // loaded once, on first use (a function-local static: initialised exactly once, whatever thread comes first)
  static const RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (a.lib) {
      auto sym = [&](const char* n) { return dlsym(a.lib, n); };
      a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
      a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
      a.CommAbort = (decltype(a.CommAbort))sym("ncclCommAbort");
      a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
      a.Send = (decltype(a.Send))sym("ncclSend");
      a.Recv = (decltype(a.Recv))sym("ncclRecv");
      a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
      a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
      a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
      if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.Send || !a.Recv || !a.GroupStart || !a.GroupEnd) {
        dlclose(a.lib);
        a.lib = nullptr;
      }
    }
    return a;
  }();
  return api.lib ? const_cast<RcclApi*>(&api) : nullptr;
}
// An RCCL failure poisons the communicator: mark it, so that teardown ABORTS instead of a CommDestroy that would wait for peers
// which are themselves waiting for this rank.
int rccl_fail(pt_ctx* c, const char* what, ncclResult_t r) {
  c->comm_failed = true;
  return fail(c, PT_ERR_HIP, "%s failed: %s", what, rccl()->GetErrorString ? rccl()->GetErrorString(r) : "rccl error");
}
#define NCCLCHK(c, call)                                                                                                      \
  do {                                                                                                                        \
    ncclResult_t r_ = (call);                                                                                                 \
    if (r_ != ncclSuccess) return rccl_fail((c), #call, r_);                                                                  \
  } while (0)
// Inside ncclGroupStart .. ncclGroupEnd nothing may return: a group left open keeps every later RCCL call of this thread queued
// (CommDestroy included) and the peers wait for the matching receives forever.  Record the first error, always close the group.
#define NCCLGRP(first, what, call)                                                                                            \
  do {                                                                                                                        \
    if ((first) == ncclSuccess) { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (first) = r_; (what) = #call; } }        \
  } while (0)

struct XArgs { const void* xyz; int type; uint32_t m; int k, axis, g, me; uint32_t* idx; double* d2; };

// phase a: upload the slab bounds, count the crossing targets per destination slab into x_counts[g]
int xa_count(pt_ctx* c, const XArgs& A, const double* bounds) {
  RES(c, c->x_bounds, 65 * sizeof(double));
  RES(c, c->x_counts, 64 * sizeof(uint32_t));
  HIPCHK(c, hipMemcpyAsync(c->x_bounds.p, bounds, (size_t)(A.g + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->x_counts.p, 0, 64 * sizeof(uint32_t), c->stream));
  if (A.type == PT_F32) { const float* x = (const float*)A.xyz; pt_launch_xreq<float>(false, x, x + A.m, x + 2 * (size_t)A.m, A.d2, A.m, A.k, A.axis, (const double*)c->x_bounds.p, A.g, A.me, (uint32_t*)c->x_counts.p, nullptr, nullptr, nullptr, nullptr, c->stream); }
  else { const double* x = (const double*)A.xyz; pt_launch_xreq<double>(false, x, x + A.m, x + 2 * (size_t)A.m, A.d2, A.m, A.k, A.axis, (const double*)c->x_bounds.p, A.g, A.me, (uint32_t*)c->x_counts.p, nullptr, nullptr, nullptr, nullptr, c->stream); }
  HIPCHK(c, hipGetLastError());
  return PT_OK;
}
// phase b: with the count matrix on the host (matrix[r * g + s] = packets rank r sends to s), size the buffers and bucket the requests
int xb_fill(pt_ctx* c, const XArgs& A, const uint32_t* matrix) {
  const int g = A.g, me = A.me;
  c->x_send.assign((size_t)g, 0); c->x_recv.assign((size_t)g, 0); c->x_soff.assign((size_t)g + 1, 0); c->x_roff.assign((size_t)g + 1, 0);
  for (int p = 0; p < g; ++p) { c->x_send[(size_t)p] = matrix[(size_t)me * g + p]; c->x_recv[(size_t)p] = matrix[(size_t)p * g + me]; }
  for (int p = 0; p < g; ++p) { c->x_soff[(size_t)p + 1] = c->x_soff[(size_t)p] + c->x_send[(size_t)p]; c->x_roff[(size_t)p + 1] = c->x_roff[(size_t)p] + c->x_recv[(size_t)p]; }
  const size_t S = c->x_soff[(size_t)g], R = c->x_roff[(size_t)g];
  RES(c, c->x_off, 2 * 65 * sizeof(uint32_t));
  RES(c, c->x_req, std::max<size_t>(S, 1) * 32); RES(c, c->x_row, std::max<size_t>(S, 1) * 4);
  RES(c, c->x_rreq, std::max<size_t>(R, 1) * 32);
  RES(c, c->x_back_i, std::max<size_t>(S, 1) * (size_t)A.k * 4); RES(c, c->x_back_d, std::max<size_t>(S, 1) * (size_t)A.k * 8);
  RES(c, c->x_ans_i, std::max<size_t>(R, 1) * (size_t)A.k * 4); RES(c, c->x_ans_d, std::max<size_t>(R, 1) * (size_t)A.k * 8);
  if (c->attr_local) { RES(c, c->x_ans_a, std::max<size_t>(R, 1) * (size_t)A.k * sizeof(Attr)); RES(c, c->x_back_a, std::max<size_t>(S, 1) * (size_t)A.k * sizeof(Attr)); }
  uint32_t* off = (uint32_t*)c->x_off.p;
  uint32_t* cursor = off + 65;
  // the offsets travel from PINNED memory, so the copy needs no host wait; the event says when the staging words may be rewritten
  // (always long past: the count read-back of the next exchange sits in between -- the wait is for a caller who switched streams)
  if (!c->h_xoff) { HIPCHK(c, hipHostMalloc((void**)&c->h_xoff, 65 * sizeof(uint32_t))); HIPCHK(c, hipEventCreateWithFlags(&c->xoff_ev, hipEventDisableTiming)); }
  else HIPCHK(c, hipEventSynchronize(c->xoff_ev));
  memcpy(c->h_xoff, c->x_soff.data(), (size_t)(g + 1) * sizeof(uint32_t));
  HIPCHK(c, hipMemcpyAsync(off, c->h_xoff, (size_t)(g + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->xoff_ev, c->stream));
  HIPCHK(c, hipMemsetAsync(cursor, 0, 65 * sizeof(uint32_t), c->stream));
  if (S) {
    if (A.type == PT_F32) { const float* x = (const float*)A.xyz; pt_launch_xreq<float>(true, x, x + A.m, x + 2 * (size_t)A.m, A.d2, A.m, A.k, A.axis, (const double*)c->x_bounds.p, g, me, nullptr, off, cursor, (double*)c->x_req.p, (uint32_t*)c->x_row.p, c->stream); }
    else { const double* x = (const double*)A.xyz; pt_launch_xreq<double>(true, x, x + A.m, x + 2 * (size_t)A.m, A.d2, A.m, A.k, A.axis, (const double*)c->x_bounds.p, g, me, nullptr, off, cursor, (double*)c->x_req.p, (uint32_t*)c->x_row.p, c->stream); }
  }
  HIPCHK(c, hipGetLastError());
  return PT_OK;
}
// phase c: answer the requests received (x_rreq, x_roff[g] of them) with a radius-bounded search of this rank's slab
int xc_answer(pt_ctx* c, const XArgs& A) {
  const uint32_t R = c->x_roff[(size_t)A.g];
  if (!R) return PT_OK;
  RES(c, c->x_rxyz, (size_t)R * 3 * tsize(A.type)); RES(c, c->x_rbound, (size_t)R * 8);
  if (A.type == PT_F32) pt_launch_xunpack<float>((const double*)c->x_rreq.p, R, (float*)c->x_rxyz.p, (double*)c->x_rbound.p, c->stream);
  else pt_launch_xunpack<double>((const double*)c->x_rreq.p, R, (double*)c->x_rxyz.p, (double*)c->x_rbound.p, c->stream);
  { int r = run_query(c, c->x_rxyz.p, A.type, R, A.k, (const double*)c->x_rbound.p, (uint32_t*)c->x_ans_i.p, (double*)c->x_ans_d.p); if (r) return r; }
  // attribute table sharded with the slabs: the candidates' records ride along (they are this slab's points: found by position in its ascending gidx)
  if (c->attr_local && c->has_attr)
    pt_launch_xgather_attr((const uint32_t*)c->x_ans_i.p, (size_t)R * (size_t)A.k, (const uint32_t*)c->in_gidx.p, (uint32_t)c->n, (const Attr*)c->attr.p, (Attr*)c->x_ans_a.p, c->stream);
  return PT_OK;
}
// phase d: merge what came back (x_back_*, bucket by bucket) and redo the blend of the completed rows
int xd_merge(pt_ctx* c, const XArgs& A, int blend_mode, float* rgb, float* nrm) {
  const uint32_t S = c->x_soff[(size_t)A.g];
  if (!S) return PT_OK;
  const bool reblend = blend_mode >= 0 && (rgb || nrm) && c->has_attr;
  const bool sharded = reblend && c->attr_local;        // the table holds this slab's points only: the merge carries the candidates' records along
  uint8_t* flags = nullptr;
  if (reblend) {
    RES(c, c->x_flags, std::max<size_t>(A.m, 1)); RES(c, c->x_rows, (std::max<size_t>(A.m, 1) + 4) * 4);
    flags = (uint8_t*)c->x_flags.p;
    HIPCHK(c, hipMemsetAsync(flags, 0, A.m, c->stream));
  }
  if (sharded) RES(c, c->x_rattr, std::max<size_t>(A.m, 1) * (size_t)A.k * sizeof(Attr));      // (rows the exchange touches: written before they are read)
  for (int p = 0; p < A.g; ++p) {                     // one launch per bucket: a row may sit in two buckets (both neighbours)
    const uint32_t cnt = c->x_send[(size_t)p], o = c->x_soff[(size_t)p];
    if (sharded)
      pt_launch_xmerge_attr((const uint32_t*)c->x_row.p + o, cnt, (const uint32_t*)c->x_back_i.p + (size_t)o * A.k, (const double*)c->x_back_d.p + (size_t)o * A.k,
                            (const Attr*)c->x_back_a.p + (size_t)o * A.k, A.k, A.idx, A.d2, (Attr*)c->x_rattr.p, flags, (const uint32_t*)c->in_gidx.p, (uint32_t)c->n,
                            (const Attr*)c->attr.p, c->stream);
    else
      pt_launch_xmerge((const uint32_t*)c->x_row.p + o, cnt, (const uint32_t*)c->x_back_i.p + (size_t)o * A.k, (const double*)c->x_back_d.p + (size_t)o * A.k, A.k,
                       A.idx, A.d2, flags, c->stream);
  }
  if (reblend) {
    uint32_t* rows = (uint32_t*)c->x_rows.p;
    uint32_t* rows_n = rows + std::max<size_t>(A.m, 1);
    HIPCHK(c, hipMemsetAsync(rows_n, 0, 4, c->stream));
    pt_launch_xflag_rows(flags, A.m, rows, rows_n, c->stream);
    if (sharded) pt_launch_blend_rows_attr(rows, rows_n, std::min<uint32_t>(A.m, S), A.idx, A.d2, (const Attr*)c->x_rattr.p, A.k, blend_mode, rgb, nrm, c->stream);
    else pt_launch_blend_rows(rows, rows_n, std::min<uint32_t>(A.m, S), A.idx, A.d2, A.k, blend_mode, (const Attr*)c->attr.p, (uint32_t)c->n_total, rgb, nrm, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  return PT_OK;
}
int xcheck(pt_ctx* c, int xyz_type, uint64_t m, int k, int axis, int g, const double* bounds, const void* xyz, const uint32_t* idx, const double* d2) {
  if (xyz_type != PT_F32 && xyz_type != PT_F64) return fail(c, PT_ERR_UNSUPPORTED, "xyz_type %d not supported", xyz_type);
  if (!c->built) return fail(c, PT_ERR_STATE, "exchange before build");
  if (xyz_type != c->src_type) return fail(c, PT_ERR_UNSUPPORTED, "target xyz type %d differs from the source cloud's %d", xyz_type, c->src_type);
  if (g < 1 || g > 64 || k < 1 || k > PT_MAX_K || axis < 0 || axis > 2) return fail(c, PT_ERR_ARG, "argument out of range");
  if (!bounds || (m && (!xyz || !idx || !d2))) return fail(c, PT_ERR_ARG, "null argument");
  return check_n(c, m, "m");
}
}  // namespace

int pt_comm_unique_id(void* id_out) {
  if (!id_out || !rccl()) return PT_ERR_HIP;
  ncclUniqueId id;
  if (rccl()->GetUniqueId(&id) != ncclSuccess) return PT_ERR_HIP;
  static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "RCCL unique id size");
  memcpy(id_out, &id, sizeof id);
  return PT_OK;
}

int pt_comm_init(pt_ctx* c, int world, int rank, const void* id) {
  if (!c) return PT_ERR_ARG;
  if (world < 1 || world > 64 || rank < 0 || rank >= world || !id) return fail(c, PT_ERR_ARG, "pt_comm_init: world / rank / id out of range");
  if (c->nccl_comm) return fail(c, PT_ERR_STATE, "communicator already initialised");
  if (!rccl()) return fail(c, PT_ERR_HIP, "librccl could not be loaded");
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  NCCLCHK(c, rccl()->CommInitRank(&comm, world, uid, rank));
  c->nccl_comm = comm; c->world = world; c->rank = rank;
  if (!c->h_matrix) HIPCHK(c, hipHostMalloc((void**)&c->h_matrix, 64 * 64 * sizeof(uint32_t)));
  return PT_OK;
}

int pt_comm_destroy(pt_ctx* c) {
  if (!c) return PT_ERR_ARG;
  if (c->comm_failed) return pt_comm_abort(c);
  if (c->nccl_comm && rccl()) { (void)hipSetDevice(c->device); (void)rccl()->CommDestroy((ncclComm_t)c->nccl_comm); }
  c->nccl_comm = nullptr; c->world = 1; c->rank = 0;
  return PT_OK;
}

int pt_comm_abort(pt_ctx* c) {
  if (!c) return PT_ERR_ARG;
  if (c->nccl_comm && rccl()) {
    (void)hipSetDevice(c->device);
    if (rccl()->CommAbort) (void)rccl()->CommAbort((ncclComm_t)c->nccl_comm);
    else std::fprintf(stderr, "[pt_hip] rank %d: librccl has no ncclCommAbort: the communicator of a failed exchange is leaked rather than destroyed (a destroy could wait for peers that wait for this rank)\n", c->rank);
  }
  c->nccl_comm = nullptr; c->world = 1; c->rank = 0; c->comm_failed = false;
  return PT_OK;
}

int pt_exchange_merge_dev(pt_ctx* c, const void* tgt_xyz_dev, int xyz_type, uint64_t m, int k, int slab_axis, const double* slab_bounds, uint32_t* idx_dev,
                          double* d2_dev, int blend_mode, float* rgb_dev, float* nrm_dev, pt_exchange_stats_t* st) {
  if (!c) return PT_ERR_ARG;
  if (st) memset(st, 0, sizeof *st);
  const int g = c->world, me = c->rank;
  { int r = xcheck(c, xyz_type, m, k, slab_axis, g, slab_bounds, tgt_xyz_dev, idx_dev, d2_dev); if (r) return r; }
  if (g == 1) return PT_OK;
  if (!c->nccl_comm) return fail(c, PT_ERR_STATE, "pt_exchange_merge_dev before pt_comm_init");
  HIPCHK(c, hipSetDevice(c->device));
  ncclComm_t comm = (ncclComm_t)c->nccl_comm;
  const XArgs A{tgt_xyz_dev, xyz_type, (uint32_t)m, k, slab_axis, g, me, idx_dev, d2_dev};
  const int sync_save = c->sync;
  c->sync = 0;                                          // the bounded search only enqueues: no host wait between the phases
  auto body = [&]() -> int {
    HIPCHK(c, hipEventRecord(c->xev[0], c->stream));
    { int r = xa_count(c, A, slab_bounds); if (r) return r; }
    RES(c, c->x_matrix, 64 * 64 * sizeof(uint32_t));
    NCCLCHK(c, rccl()->AllGather(c->x_counts.p, c->x_matrix.p, 64, ncclUint32, comm, c->stream));      // rows of 64 counters, g of them
    HIPCHK(c, hipMemcpyAsync(c->h_matrix, c->x_matrix.p, (size_t)g * 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));         // the exchange's ONE host wait: every later size follows from the matrix
    std::vector<uint32_t> matrix((size_t)g * g);
    for (int r = 0; r < g; ++r) for (int s2 = 0; s2 < g; ++s2) matrix[(size_t)r * g + s2] = c->h_matrix[(size_t)r * 64 + s2];
    { int r = xb_fill(c, A, matrix.data()); if (r) return r; }
    ncclResult_t gerr = ncclSuccess;
    const char* gwhat = "";
    NCCLCHK(c, rccl()->GroupStart());
    for (int p = 0; p < g; ++p) {
      if (p == me) continue;
      if (c->x_send[(size_t)p]) NCCLGRP(gerr, gwhat, rccl()->Send((const double*)c->x_req.p + (size_t)c->x_soff[(size_t)p] * 4, (size_t)c->x_send[(size_t)p] * 4, ncclFloat64, p, comm, c->stream));
      if (c->x_recv[(size_t)p]) NCCLGRP(gerr, gwhat, rccl()->Recv((double*)c->x_rreq.p + (size_t)c->x_roff[(size_t)p] * 4, (size_t)c->x_recv[(size_t)p] * 4, ncclFloat64, p, comm, c->stream));
    }
    NCCLGRP(gerr, gwhat, rccl()->GroupEnd());
    if (gerr != ncclSuccess) { if (gwhat[0] && !strstr(gwhat, "GroupEnd")) (void)rccl()->GroupEnd(); return rccl_fail(c, gwhat, gerr); }
    { int r = xc_answer(c, A); if (r) return r; }
    NCCLCHK(c, rccl()->GroupStart());
    for (int p = 0; p < g; ++p) {
      if (p == me) continue;
      const size_t ro = (size_t)c->x_roff[(size_t)p] * k, rc = (size_t)c->x_recv[(size_t)p] * k, so = (size_t)c->x_soff[(size_t)p] * k, sc = (size_t)c->x_send[(size_t)p] * k;
      if (rc) { NCCLGRP(gerr, gwhat, rccl()->Send((const uint32_t*)c->x_ans_i.p + ro, rc, ncclUint32, p, comm, c->stream)); NCCLGRP(gerr, gwhat, rccl()->Send((const double*)c->x_ans_d.p + ro, rc, ncclFloat64, p, comm, c->stream)); }
      if (sc) { NCCLGRP(gerr, gwhat, rccl()->Recv((uint32_t*)c->x_back_i.p + so, sc, ncclUint32, p, comm, c->stream)); NCCLGRP(gerr, gwhat, rccl()->Recv((double*)c->x_back_d.p + so, sc, ncclFloat64, p, comm, c->stream)); }
      if (c->attr_local) {      // (every rank of a job runs the same mode: the candidates' 16-byte records as four words each)
        if (rc) NCCLGRP(gerr, gwhat, rccl()->Send((const uint32_t*)c->x_ans_a.p + ro * 4, rc * 4, ncclUint32, p, comm, c->stream));
        if (sc) NCCLGRP(gerr, gwhat, rccl()->Recv((uint32_t*)c->x_back_a.p + so * 4, sc * 4, ncclUint32, p, comm, c->stream));
      }
    }
    NCCLGRP(gerr, gwhat, rccl()->GroupEnd());
    if (gerr != ncclSuccess) { if (gwhat[0] && !strstr(gwhat, "GroupEnd")) (void)rccl()->GroupEnd(); return rccl_fail(c, gwhat, gerr); }
    { int r = xd_merge(c, A, blend_mode, rgb_dev, nrm_dev); if (r) return r; }
    HIPCHK(c, hipEventRecord(c->xev[1], c->stream));
    return PT_OK;
  };
  const int r = body();
  c->sync = sync_save;
  if (r != PT_OK) { c->comm_failed = true; return r; }      // whatever failed, the peers may be left waiting: teardown must abort, not wait
  if (st) {
    const uint64_t S = c->x_soff[(size_t)g], R = c->x_roff[(size_t)g];
    st->crossing = S; st->answered = R;
    const uint64_t cand = c->attr_local ? 28 : 12;        // (index + distance, + the attribute record when the table is sharded)
    st->bytes_sent = S * 32 + R * (uint64_t)k * cand; st->bytes_received = R * 32 + S * (uint64_t)k * cand;
  }
  if (c->sync || st) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st) { float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, c->xev[0], c->xev[1])); st->ms = ms; }
  }
  return PT_OK;
}

int pt_query_exchange_blend(pt_ctx* c, const void* tgt_xyz, int xyz_type, uint64_t m, int k, int slab_axis, const double* slab_bounds, int blend_mode,
                            uint32_t* idx_out, double* d2_out, float* rgb_out, float* nrm_out, pt_exchange_stats_t* st) {
  if (!c) return PT_ERR_ARG;
  if (st) memset(st, 0, sizeof *st);
  if (!c->built) return fail(c, PT_ERR_STATE, "query before build");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k = %d out of range [1, %d]", k, PT_MAX_K);
  if (m && (!tgt_xyz || !idx_out || !d2_out)) return fail(c, PT_ERR_ARG, "null argument");
  const bool blend = blend_mode >= 0 && (rgb_out || nrm_out);
  if (blend && blend_mode != PT_BLEND_MEAN && blend_mode != PT_BLEND_INV_D2) return fail(c, PT_ERR_ARG, "unknown blend mode %d", blend_mode);
  if (blend && !c->has_attr) return fail(c, PT_ERR_STATE, "no attribute table resident");
  { int r = pt_targets_soa(c, tgt_xyz, xyz_type, m, 0); if (r) return r; }
  if (c->tgt_type != c->src_type) return fail(c, PT_ERR_UNSUPPORTED, "target xyz type differs from the source cloud's");
  RES(c, c->q_idx, std::max<uint64_t>(m, 1) * k * sizeof(uint32_t));
  RES(c, c->q_d2, std::max<uint64_t>(m, 1) * k * sizeof(double));
  RES(c, c->b_rgb, std::max<uint64_t>(m, 1) * 12);
  RES(c, c->b_nrm, std::max<uint64_t>(m, 1) * 12);
  const BlendReq br{blend_mode, (float*)c->b_rgb.p, (float*)c->b_nrm.p};
  { int r = run_query(c, c->t_xyz.p, c->tgt_type, m, k, nullptr, (uint32_t*)c->q_idx.p, (double*)c->q_d2.p, blend ? &br : nullptr); if (r) return r; }
  if (c->world > 1) {
    int r = pt_exchange_merge_dev(c, c->t_xyz.p, c->tgt_type, m, k, slab_axis, slab_bounds, (uint32_t*)c->q_idx.p, (double*)c->q_d2.p, blend ? blend_mode : -1,
                                  blend ? (float*)c->b_rgb.p : nullptr, blend ? (float*)c->b_nrm.p : nullptr, st);
    if (r) return r;
  }
  if (m) {
    HIPCHK(c, hipMemcpyAsync(idx_out, c->q_idx.p, m * k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(d2_out, c->q_d2.p, m * k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (blend && rgb_out) HIPCHK(c, hipMemcpyAsync(rgb_out, c->b_rgb.p, m * 12, hipMemcpyDeviceToHost, c->stream));
    if (blend && nrm_out) HIPCHK(c, hipMemcpyAsync(nrm_out, c->b_nrm.p, m * 12, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PT_OK;
}

int pt_exchange_merge_local(pt_ctx* const* ctxs, int g, const void* const* tgt_xyz_dev, int xyz_type, const uint64_t* m, int k, int slab_axis,
                            const double* slab_bounds, uint32_t* const* idx_dev, double* const* d2_dev, int blend_mode, float* const* rgb_dev,
                            float* const* nrm_dev) {
  if (!ctxs || g < 1 || g > 64 || !tgt_xyz_dev || !m || !idx_dev || !d2_dev) return PT_ERR_ARG;
  for (int r = 0; r < g; ++r) if (!ctxs[r]) return PT_ERR_ARG;
  std::vector<XArgs> A((size_t)g);
  std::vector<int> sync_save((size_t)g);
  for (int r = 0; r < g; ++r) {
    pt_ctx* c = ctxs[r];
    { int e = xcheck(c, xyz_type, m[r], k, slab_axis, g, slab_bounds, tgt_xyz_dev[r], idx_dev[r], d2_dev[r]); if (e) return e; }
    A[(size_t)r] = XArgs{tgt_xyz_dev[r], xyz_type, (uint32_t)m[r], k, slab_axis, g, r, idx_dev[r], d2_dev[r]};
    sync_save[(size_t)r] = c->sync;
  }
  if (g == 1) return PT_OK;
  auto restore = [&]() { for (int r = 0; r < g; ++r) ctxs[r]->sync = sync_save[(size_t)r]; };
  auto all_sync = [&]() -> int { for (int r = 0; r < g; ++r) { pt_ctx* c = ctxs[r]; HIPCHK(c, hipStreamSynchronize(c->stream)); } return PT_OK; };
  auto body = [&]() -> int {
    std::vector<uint32_t> matrix((size_t)g * g, 0), row(64);
    for (int r = 0; r < g; ++r) { int e = xa_count(ctxs[r], A[(size_t)r], slab_bounds); if (e) return e; }
    for (int r = 0; r < g; ++r) {                                             // "all-gather" of the counters
      pt_ctx* c = ctxs[r];
      HIPCHK(c, hipMemcpyAsync(row.data(), c->x_counts.p, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      for (int s2 = 0; s2 < g; ++s2) matrix[(size_t)r * g + s2] = row[(size_t)s2];
    }
    for (int r = 0; r < g; ++r) { int e = xb_fill(ctxs[r], A[(size_t)r], matrix.data()); if (e) return e; }
    for (int r = 0; r < g; ++r)                                               // requests, owner to owner
      for (int p = 0; p < g; ++p) {
        pt_ctx *a = ctxs[r], *b = ctxs[p];
        if (p == r || !a->x_send[(size_t)p]) continue;
        HIPCHK(a, hipMemcpyAsync((double*)b->x_rreq.p + (size_t)b->x_roff[(size_t)r] * 4, (const double*)a->x_req.p + (size_t)a->x_soff[(size_t)p] * 4,
                                 (size_t)a->x_send[(size_t)p] * 32, hipMemcpyDeviceToDevice, a->stream));
      }
    { int e = all_sync(); if (e) return e; }
    for (int r = 0; r < g; ++r) { ctxs[r]->sync = 0; int e = xc_answer(ctxs[r], A[(size_t)r]); if (e) return e; }
    { int e = all_sync(); if (e) return e; }
    for (int r = 0; r < g; ++r)                                               // answers, back the same way
      for (int p = 0; p < g; ++p) {
        pt_ctx *a = ctxs[r], *b = ctxs[p];                                    // a answered b's requests
        if (p == r || !a->x_recv[(size_t)p]) continue;
        const size_t ro = (size_t)a->x_roff[(size_t)p] * k, cnt = (size_t)a->x_recv[(size_t)p] * k, so = (size_t)b->x_soff[(size_t)r] * k;
        HIPCHK(a, hipMemcpyAsync((uint32_t*)b->x_back_i.p + so, (const uint32_t*)a->x_ans_i.p + ro, cnt * 4, hipMemcpyDeviceToDevice, a->stream));
        HIPCHK(a, hipMemcpyAsync((double*)b->x_back_d.p + so, (const double*)a->x_ans_d.p + ro, cnt * 8, hipMemcpyDeviceToDevice, a->stream));
        if (a->attr_local && b->attr_local) HIPCHK(a, hipMemcpyAsync((Attr*)b->x_back_a.p + so, (const Attr*)a->x_ans_a.p + ro, cnt * sizeof(Attr), hipMemcpyDeviceToDevice, a->stream));
      }
    { int e = all_sync(); if (e) return e; }
    for (int r = 0; r < g; ++r) { int e = xd_merge(ctxs[r], A[(size_t)r], blend_mode, rgb_dev ? rgb_dev[r] : nullptr, nrm_dev ? nrm_dev[r] : nullptr); if (e) return e; }
    return all_sync();
  };
  const int e = body();
  restore();
  return e;
}

// ---- texture bake (pt_bake.hip) ------------------------------------------------------------------------------------
int pt_bake_texture(pt_ctx* c, const pt_point* mesh_vertices, uint64_t nv, const int32_t* faces, uint64_t nf, const uint32_t* nbr_idx, int k,
                    int resolution, int pad_ksize, uint8_t* bgra_out) {
  if (!c) return PT_ERR_ARG;
  if (c->src_type != PT_F32 && c->src_type != PT_F64) return fail(c, PT_ERR_STATE, "no source cloud resident (call a pt_build_* first)");
  if (c->has_gidx) return fail(c, PT_ERR_UNSUPPORTED, "the texture bake needs the whole cloud resident (not a slab)");
  if (!c->has_attr) return fail(c, PT_ERR_STATE, "no attribute table resident (the bake reads the source colours)");
  if (k < 1 || k > PT_MAX_K) return fail(c, PT_ERR_ARG, "k = %d out of range [1, %d]", k, PT_MAX_K);
  if (resolution < 1 || resolution > 32768) return fail(c, PT_ERR_ARG, "resolution out of range [1, 32768]");
  if (pad_ksize < 0 || (pad_ksize > 0 && !(pad_ksize & 1)) || pad_ksize > 255) return fail(c, PT_ERR_ARG, "pad_ksize must be 0 or an odd number <= 255");
  if (!bgra_out || (nv && !mesh_vertices) || (nf && (!faces || !nbr_idx))) return fail(c, PT_ERR_ARG, "null argument");
  if (nf >= (1ull << 24)) return fail(c, PT_ERR_ARG, "nf = %llu: the pixel key holds 24 bits of face index", (unsigned long long)nf);
  { int r = check_n(c, nv, "nv"); if (r) return r; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t npix = (size_t)resolution * (size_t)resolution;
  DevBuf keys, tex, tmp, out, dv, df, dn;
  auto cleanup = [&]() { DevBuf* all[] = {&keys, &tex, &tmp, &out, &dv, &df, &dn}; for (DevBuf* b : all) release(c, *b); };
  auto run = [&]() -> int {
    RES(c, keys, npix * 8); RES(c, tex, npix * 4);
    RES(c, dv, std::max<uint64_t>(nv, 1) * sizeof(pt_point)); RES(c, df, std::max<uint64_t>(nf, 1) * 12); RES(c, dn, std::max<uint64_t>(nv, 1) * (size_t)k * 4);
    HIPCHK(c, hipMemsetAsync(keys.p, 0, npix * 8, c->stream));
    { int r = copy_in(c, dv.p, mesh_vertices, nv * sizeof(pt_point), 0); if (r) return r; }
    { int r = copy_in(c, df.p, faces, nf * 12, 0); if (r) return r; }
    { int r = copy_in(c, dn.p, nbr_idx, nv * (size_t)k * 4, 0); if (r) return r; }
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (c->src_type == PT_F32) {
      const float* x = nullptr;
      { int r = source_xyz_f32(c, &x); if (r) return r; }
      pt_launch_bake_faces<float>(x, x + c->n, x + 2 * c->n, (const Attr*)c->attr.p, (uint32_t)c->n, dv.p, (uint32_t)nv, (const int32_t*)df.p, (uint32_t)nf,
                                  (const uint32_t*)dn.p, k, resolution, (unsigned long long*)keys.p, c->stream);
    } else {
      const double* x = (const double*)c->in_xyz.p;
      pt_launch_bake_faces<double>(x, x + c->n, x + 2 * c->n, (const Attr*)c->attr.p, (uint32_t)c->n, dv.p, (uint32_t)nv, (const int32_t*)df.p, (uint32_t)nf,
                                   (const uint32_t*)dn.p, k, resolution, (unsigned long long*)keys.p, c->stream);
    }
    pt_launch_bake_resolve((const unsigned long long*)keys.p, (uint32_t*)tex.p, npix, c->stream);
    const void* result = tex.p;
    if (pad_ksize > 0) {
      RES(c, tmp, npix * 4); RES(c, out, npix * 4);
      pt_launch_dilate_pad((const uint32_t*)tex.p, (uint32_t*)tmp.p, (uint32_t*)out.p, resolution, pad_ksize, c->stream);
      result = out.p;
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(bgra_out, result, npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->st.ms_bake = ms;
    return PT_OK;
  };
  const int r = run();
  cleanup();
  return r;
}

int pt_texture_pad(pt_ctx* c, const uint8_t* bgra_in, int resolution, int ksize, uint8_t* bgra_out) {
  if (!c) return PT_ERR_ARG;
  if (resolution < 1 || resolution > 32768) return fail(c, PT_ERR_ARG, "resolution out of range [1, 32768]");
  if (ksize < 1 || !(ksize & 1) || ksize > 255) return fail(c, PT_ERR_ARG, "ksize must be an odd number in [1, 255]");
  if (!bgra_in || !bgra_out) return fail(c, PT_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t npix = (size_t)resolution * (size_t)resolution;
  DevBuf tex, tmp, out;
  auto run = [&]() -> int {
    RES(c, tex, npix * 4); RES(c, tmp, npix * 4); RES(c, out, npix * 4);
    { int r = copy_in(c, tex.p, bgra_in, npix * 4, 0); if (r) return r; }
    pt_launch_dilate_pad((const uint32_t*)tex.p, (uint32_t*)tmp.p, (uint32_t*)out.p, resolution, ksize, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(bgra_out, out.p, npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
  };
  const int r = run();
  release(c, tex); release(c, tmp); release(c, out);
  return r;
}

}  // extern "C"
