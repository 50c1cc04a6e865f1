// pt_grid.hip -- on-device spatial grid build for gfx950 (MI355X).
//
// Replaces the reference's kd-tree construction `Tree tree(points.begin(), points.end())`
// (reference src/pointsTransfer.cpp:259; CGAL Kd_tree, lazily built on the first query) with a
// hand-written MSD radix partition of the points by grid-cell key (pt_common.h):
//
//   pass 1   partition by macro block (<= 1024 bins)          read planar xyz, write 16/32-B records
//   pass 2   partition every macro segment by Morton block id  (512 bins)   records -> records
//   finalize one workgroup per 8x8x8-cell block: LDS counting sort by local cell, emits cell_start
//
// The key is a pure function of xyz, so no key array is ever stored or moved: every pass recomputes the
// digit it needs from the coordinates.  Order inside a cell is irrelevant (the query imposes the total
// order (d2, original index) itself), so no pass needs to be stable: ranks inside a tile come from LDS
// atomics and bin space is reserved with one global atomic per (tile, non-empty bin).  Records are regrouped
// in LDS before they are written, so every bin's share of a tile leaves the CU as one contiguous run.
//
// All of this is HBM-bound byte shuffling: no MFMA, wave64 everywhere, 256-thread workgroups.
#include "pt_internal.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstring>

namespace {

constexpr int WG = 256;

template <class T> struct RecOf;
template <> struct RecOf<float> { using type = RecF; };
template <> struct RecOf<double> { using type = RecD; };
template <> struct RecOf<__half> { using type = RecF; };          // fp16 clouds stay fp16 in the resident input (6 B / point read by pass 1 and
                                                                  // its histogram instead of 12); the records they are sorted into are fp32 (exact)
template <class T> __device__ inline T pt_widen(T v) { return v; }
__device__ inline float pt_widen(__half v) { return __half2float(v); }

template <class T>
struct PlanarLoader {
  using Rec = typename RecOf<T>::type;
  const T *x, *y, *z;
  const uint32_t* gidx;
  __device__ Rec load(uint32_t i) const {
    Rec r;
    r.x = pt_widen(x[i]); r.y = pt_widen(y[i]); r.z = pt_widen(z[i]);
    r.id = gidx ? gidx[i] : i;
    return r;
  }
  // the same with the question "is there an index array" answered by the caller, once per tile: inside load() the optional fourth
  // load lands in one scratch register per record, and the wait that protects it waits for the whole record (ISA: s_waitcnt
  // vmcnt(0) after every record's loads, whether or not there is an index array)
  __device__ bool has_ids() const { return gidx != nullptr; }
  template <bool IDS>
  __device__ Rec load_t(uint32_t i) const {
    Rec r;
    r.x = pt_widen(x[i]); r.y = pt_widen(y[i]); r.z = pt_widen(z[i]);
    if constexpr (IDS) r.id = gidx[i]; else r.id = i;
    return r;
  }
};
template <class R>
struct RecLoader {
  using Rec = R;
  const R* p;
  __device__ Rec load(uint32_t i) const { return p[i]; }
  __device__ bool has_ids() const { return false; }
  template <bool IDS>
  __device__ Rec load_t(uint32_t i) const { return p[i]; }
};

struct BinSpec {
  int mode;    // 0: local bin = blk >> shift, global bin = local      (pass 1)
               // 1: local bin = blk & mask,   global bin = blk         (pass 2; seg = blk >> shift)
               // 2: local bin = macro & mask, global bin = macro       (middle pass of a three-level sort; macro = blk >> 9, seg = macro >> shift)
  int shift;
  int nbins;   // local bins (<= PT_MAXBINS)
};
__device__ inline uint32_t local_bin(const BinSpec& b, uint32_t blk) {
  return b.mode == 0 ? (blk >> b.shift) : ((b.mode == 1 ? blk : blk >> 9) & ((1u << b.shift) - 1u));
}
__device__ inline uint32_t global_bin(const BinSpec& b, uint32_t seg, uint32_t local) {
  return b.mode == 0 ? local : ((seg << b.shift) + local);
}
template <class Rec>
__device__ inline uint32_t block_of_rec(const GridParams& gp, const Rec& r) {
  int cx, cy, cz;
  pt_cell_of(gp, r, cx, cy, cz);
  return pt_block_id(gp.mdim, cx, cy, cz);
}

// tile -> (segment, [s,e)).  tile_first is the exclusive scan of tiles per segment (nseg+1 entries).
// seg_end: null when the segments are dense (segment i ends where i + 1 starts); the pooled pass 1 leaves slack between its bins and
// says where each one ends.
__device__ inline bool tile_range(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ tile_first, int nseg,
                                  uint32_t tile, uint32_t tile_pts, uint32_t& seg, uint32_t& s, uint32_t& e,
                                  const uint32_t* __restrict__ seg_end = nullptr) {
  if (tile >= tile_first[nseg]) return false;
  int lo = 0, hi = nseg;               // invariant: tile_first[lo] <= tile < tile_first[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_first[mid] <= tile) lo = mid; else hi = mid;
  }
  seg = (uint32_t)lo;
  s = seg_start[lo] + (tile - tile_first[lo]) * tile_pts;
  const uint32_t send = seg_end ? seg_end[lo] : seg_start[lo + 1];
  e = (send - s > tile_pts) ? s + tile_pts : send;
  return true;
}

// ---- bounding box ------------------------------------------------------------------------------
__device__ inline uint64_t enc_f64(double d) {
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// Reduce every thread's (mn, mx) over the workgroup and fold the result into out6 with ONE set of six 64-bit atomics:
// thousands of waves updating six addresses serialise (measured: 0.6 ms for a 10 M-point cloud, 1 ms for a sample).
__device__ inline void wg_bbox_commit(double (&mn)[3], double (&mx)[3], uint64_t* out6) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fmin(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmax(mx[a], __shfl_xor(mx[a], o)); }
  }
  __shared__ double wbox[WG / 64][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { wbox[threadIdx.x >> 6][a] = mn[a]; wbox[threadIdx.x >> 6][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = wbox[0][threadIdx.x];
    for (int w = 1; w < WG / 64; ++w) v = threadIdx.x < 3 ? fmin(v, wbox[w][threadIdx.x]) : fmax(v, wbox[w][threadIdx.x]);
    if (threadIdx.x < 3) { if (v != INFINITY) atomicMin((unsigned long long*)&out6[threadIdx.x], (unsigned long long)enc_f64(v)); }
    else if (v != -INFINITY) atomicMax((unsigned long long*)&out6[threadIdx.x], (unsigned long long)enc_f64(v));
  }
}
template <class T>
__global__ __launch_bounds__(WG) void bbox_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, uint32_t n,
                                                  uint64_t* out6) {
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
    const double v[3] = {(double)pt_widen(x[i]), (double)pt_widen(y[i]), (double)pt_widen(z[i])};
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = (v[a] != v[a]) ? -INFINITY : fmin(mn[a], v[a]); mx[a] = fmax(mx[a], v[a]); }   // NaN: fmin / fmax would drop it
  }
  wg_bbox_commit(mn, mx, out6);
}
// bounding box of a SAMPLE of the cloud (a guess of its extent): one run of WG consecutive points out of every
// `stride` runs -- consecutive so that the loads coalesce (one point every 4 KB cost a TLB miss each: 1 ms for 1e6 points)
template <class T>
__global__ __launch_bounds__(WG) void bbox_sample_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, uint32_t n,
                                                         uint32_t stride, uint64_t* out6) {
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  const uint64_t span = (uint64_t)stride * WG;
  for (uint64_t i = (uint64_t)blockIdx.x * span + threadIdx.x; i < n; i += (uint64_t)gridDim.x * span) {
    const double v[3] = {(double)pt_widen(x[i]), (double)pt_widen(y[i]), (double)pt_widen(z[i])};
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = (v[a] != v[a]) ? -INFINITY : fmin(mn[a], v[a]); mx[a] = fmax(mx[a], v[a]); }   // NaN: fmin / fmax would drop it
  }
  wg_bbox_commit(mn, mx, out6);
}
__global__ void bbox_init_kernel(uint64_t* out6) {
  if (threadIdx.x < 3) out6[threadIdx.x] = ~0ull;
  else if (threadIdx.x < 6) out6[threadIdx.x] = 0ull;
}

// ---- small table kernels -----------------------------------------------------------------------
__global__ void single_segment_kernel(uint32_t n, uint32_t tile_pts, uint32_t* seg_start, uint32_t* tile_first) {
  if (threadIdx.x == 0) { seg_start[0] = 0; seg_start[1] = n; tile_first[0] = 0; tile_first[1] = (n + tile_pts - 1) / tile_pts; }
}
// pass-1 histogram -> segment starts, scatter cursors and the tile table of pass 2 (nseg <= 1024)
__global__ __launch_bounds__(WG) void seg_setup_kernel(const uint32_t* __restrict__ counts, int nseg, uint32_t n, uint32_t tile_pts,
                                                       uint32_t* start, uint32_t* cursor, uint32_t* tile_first) {
  __shared__ uint32_t wsum[4];
  uint32_t c[4], t[4], sc = 0, st = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = threadIdx.x * 4 + i;
    c[i] = b < nseg ? counts[b] : 0u;
    t[i] = (c[i] + tile_pts - 1) / tile_pts;
    sc += c[i]; st += t[i];
  }
  uint32_t tot;
  uint32_t ec = block_excl_scan(sc, wsum, tot);
  __syncthreads();
  uint32_t et = block_excl_scan(st, wsum, tot);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = threadIdx.x * 4 + i;
    if (b < nseg) { start[b] = ec; cursor[b] = ec; tile_first[b] = et; }
    ec += c[i]; et += t[i];
  }
  if (threadIdx.x == 0) { start[nseg] = n; tile_first[nseg] = tot; }
}

// the same for any number of segments (three-level sorts: the macro blocks), one workgroup walking them 1024 at a time
__global__ __launch_bounds__(WG) void seg_setup_big_kernel(const uint32_t* __restrict__ counts, int nseg, uint32_t n, uint32_t tile_pts,
                                                           uint32_t* start, uint32_t* cursor, uint32_t* tile_first) {
  __shared__ uint32_t wsum[4];
  uint32_t basec = 0, baset = 0;
  for (int b0 = 0; b0 < nseg; b0 += WG * 4) {
    uint32_t c[4], t[4], sc = 0, st = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = b0 + threadIdx.x * 4 + i;
      c[i] = b < nseg ? counts[b] : 0u;
      t[i] = (c[i] + tile_pts - 1) / tile_pts;
      sc += c[i]; st += t[i];
    }
    uint32_t totc, tott;
    uint32_t ec = basec + block_excl_scan(sc, wsum, totc);
    __syncthreads();
    uint32_t et = baset + block_excl_scan(st, wsum, tott);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = b0 + threadIdx.x * 4 + i;
      if (b < nseg) { start[b] = ec; cursor[b] = ec; tile_first[b] = et; }
      ec += c[i]; et += t[i];
    }
    basec += totc; baset += tott;
  }
  if (threadIdx.x == 0) { start[nseg] = n; tile_first[nseg] = baset; }
}

// ---- generic exclusive scan of u32 ---------------------------------------------------------------
constexpr int SCAN_ITEMS = 8, SCAN_TILE = WG * SCAN_ITEMS;
__global__ __launch_bounds__(WG) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n, uint32_t* sums) {
  __shared__ uint32_t wsum[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
  uint32_t tot;
  block_excl_scan(s, wsum, tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(WG) void scan_sums_kernel(uint32_t* sums, uint32_t nt) {
  __shared__ uint32_t wsum[4];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nt; base += WG * 4) {
    uint32_t c[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint32_t j = base + threadIdx.x * 4 + i; c[i] = j < nt ? sums[j] : 0u; s += c[i]; }
    uint32_t tot;
    uint32_t e = block_excl_scan(s, wsum, tot) + carry;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint32_t j = base + threadIdx.x * 4 + i; if (j < nt) sums[j] = e; e += c[i]; }
    carry += tot;
    __syncthreads();
  }
}
__global__ __launch_bounds__(WG) void scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t* out, uint32_t n,
                                                        const uint32_t* __restrict__ sums) {
  __shared__ uint32_t wsum[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t c[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { c[i] = base + i < n ? in[base + i] : 0u; s += c[i]; }
  uint32_t tot;
  uint32_t e = block_excl_scan(s, wsum, tot) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) out[base + i] = e; e += c[i]; }
}

// ---- partition pass: histogram -------------------------------------------------------------------
template <class Loader, int ITEMS>
__global__ __launch_bounds__(WG) void hist_kernel(Loader in, GridParams gp, BinSpec bs, const uint32_t* __restrict__ seg_start,
                                                  const uint32_t* __restrict__ tile_first, int nseg, uint32_t* counts, int tiles_per_wg) {
  __shared__ uint32_t hist[PT_MAXBINS];
  constexpr uint32_t TILE = WG * ITEMS;
  int cur_seg = -1;
  for (int tt = 0; tt < tiles_per_wg; ++tt) {
    uint32_t seg, s, e;
    if (!tile_range(seg_start, tile_first, nseg, blockIdx.x * tiles_per_wg + tt, TILE, seg, s, e)) break;
    if ((int)seg != cur_seg) {
      __syncthreads();
      if (cur_seg >= 0)
        for (int b = threadIdx.x; b < bs.nbins; b += WG) { const uint32_t c = hist[b]; if (c) atomicAdd(&counts[global_bin(bs, cur_seg, b)], c); }
      __syncthreads();
      for (int b = threadIdx.x; b < bs.nbins; b += WG) hist[b] = 0;
      __syncthreads();
      cur_seg = (int)seg;
    }
    constexpr int HBK = ITEMS < 8 ? ITEMS : 8;             // eight records' loads in flight, then their bins (see scatter_kernel)
#pragma unroll
    for (int j0 = 0; j0 < ITEMS; j0 += HBK) {
      decltype(in.load(0)) r[HBK];
#pragma unroll
      for (int j = 0; j < HBK; ++j) {
        const uint32_t i = s + (j0 + j) * WG + threadIdx.x;
        if (i < e) r[j] = in.load(i);
      }
#pragma unroll
      for (int j = 0; j < HBK; ++j) {
        const uint32_t i = s + (j0 + j) * WG + threadIdx.x;
        if (i < e) atomicAdd(&hist[local_bin(bs, block_of_rec(gp, r[j]))], 1u);
      }
    }
  }
  __syncthreads();
  if (cur_seg >= 0)
    for (int b = threadIdx.x; b < bs.nbins; b += WG) { const uint32_t c = hist[b]; if (c) atomicAdd(&counts[global_bin(bs, cur_seg, b)], c); }
}

// ---- partition pass: scatter (one tile per workgroup) ----------------------------------------------
template <class Loader, int ITEMS, int SW>
__global__ __launch_bounds__(SW) void scatter_kernel(Loader in, typename Loader::Rec* __restrict__ out, GridParams gp, BinSpec bs,
                                                     const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ tile_first, int nseg,
                                                     uint32_t* cursor, const uint32_t* __restrict__ seg_end = nullptr,
                                                     const uint32_t* __restrict__ limit = nullptr, uint32_t* flag = nullptr, uint32_t scratch_base = 0,
                                                     const uint32_t* abort_flag = nullptr) {
  using Rec = typename Loader::Rec;
  constexpr uint32_t TILE = SW * ITEMS;
  constexpr int BPT = PT_MAXBINS / SW;            // bins per thread in the scan
  // A pooled pass 1 whose sampled regions overflowed (flag bit 1) leaves reserved-but-unwritten stretches inside its bins: whatever is
  // there -- an earlier build's records, uninitialised memory -- must not be partitioned as if it were this build's (counts and
  // placement would disagree, writes could leave their blocks).  Everything downstream of a failed guess returns at once; the host, which
  // reads the flag in the read-back it makes anyway, redoes the build exactly.
  if (abort_flag && (*abort_flag & 1u)) return;
  // seg_end != null: the input is what the pooled pass 1 wrote -- segments with slack between them and, inside them, the SENTINEL
  // records (id = PT_NOIDX_U) that pad the blocks a workgroup left partly filled; those are dropped here.
  const bool skip = seg_end != nullptr;
  __shared__ uint32_t binA[PT_MAXBINS];           // counts, then local start of each bin in `stage`
  __shared__ uint32_t binB[PT_MAXBINS];           // global start of the bin's run minus its local start
  __shared__ uint32_t wsum[SW / 64];
  __shared__ Rec stage[TILE];

  uint32_t seg, s, e;
  if (!tile_range(seg_start, tile_first, nseg, blockIdx.x, TILE, seg, s, e, seg_end) || e <= s) return;      // (workgroup-uniform; tiles are never empty)
  for (int b = threadIdx.x; b < PT_MAXBINS; b += SW) binA[b] = 0;
  __syncthreads();

  Rec r[ITEMS];
  uint32_t lb[ITEMS], rank[ITEMS];
  // every load of the tile is issued before the first record is binned (in one loop, the 16-byte record loads came out with a full
  // wait after each: ITEMS memory latencies in a row per tile)
  if (in.has_ids()) {                                        // (workgroup-uniform; see PlanarLoader::load_t and scatter_chunk_kernel)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) r[j] = in.template load_t<true>(min(s + j * SW + threadIdx.x, e - 1u));
  } else {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) r[j] = in.template load_t<false>(min(s + j * SW + threadIdx.x, e - 1u));
  }
  bool live[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const uint32_t i = s + j * SW + threadIdx.x;
    live[j] = i < e && !(skip && r[j].id == PT_NOIDX_U);
    if (live[j]) {
      lb[j] = local_bin(bs, block_of_rec(gp, r[j]));
      rank[j] = atomicAdd(&binA[lb[j]], 1u);
    }
  }
  __syncthreads();
  uint32_t cnt;                                   // records of the tile that are staged (all of them unless sentinels were dropped)
  {
    uint32_t c[BPT], sum = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) { c[i] = binA[threadIdx.x * BPT + i]; sum += c[i]; }
    uint32_t ex = block_excl_scan_n<SW / 64>(sum, wsum, cnt);
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int b = threadIdx.x * BPT + i;
      binA[b] = ex;
      if (c[i]) {
        const uint32_t gb = global_bin(bs, seg, b);
        const uint32_t base = atomicAdd(&cursor[gb], c[i]);                         // one reservation per (tile, bin)
        // limit != null (pooled pass 2): the bins are REGIONS sized from an estimate (limit[gb + 1] = where bin gb's region ends); a
        // share that does not fit goes to the scratch area and raises the flag -- the host redoes the build with exact bin sizes
        const uint32_t lim = limit ? min(limit[gb + 1], scratch_base) : 0xFFFFFFFFu;      // (regions beyond the allocation count as full)
        if (limit && (base > lim || c[i] > lim - base)) { atomicOr(flag, 2u); binB[b] = scratch_base - ex; }
        else binB[b] = base - ex;
      }
      ex += c[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j)
    if (live[j]) stage[binA[lb[j]] + rank[j]] = r[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const uint32_t slot = j * SW + threadIdx.x;
    if (slot < cnt) {
      const Rec v = stage[slot];
      out[binB[local_bin(bs, block_of_rec(gp, v))] + slot] = v;     // consecutive slots of a bin -> consecutive addresses
    }
  }
}

// ---- pass 1 without atomics: fixed chunks of tiles, per-chunk histograms, column scan, chunk-local cursors ----
// (with only <= 1024 bins, a global cursor per bin would be hit by every tile of the cloud: at 1e9 points that
//  serialises ~5e5 reservations per address.  Chunks make the placement deterministic and contention-free.)
// (bbox6, optional: the exact bounding box of everything read, for a build whose grid was laid out from a SAMPLE of the
//  cloud -- the first full pass over the coordinates verifies the guess instead of a pass of its own preceding it)
template <class Loader, int ITEMS>
__global__ __launch_bounds__(WG) void hist_chunk_kernel(Loader in, GridParams gp, BinSpec bs, uint32_t n, int chunk_tiles,
                                                        uint32_t* __restrict__ chunk_hist, uint64_t* bbox6) {
  __shared__ uint32_t hist[PT_MAXBINS];
  for (int b = threadIdx.x; b < bs.nbins; b += WG) hist[b] = 0;
  __syncthreads();
  const uint64_t span = (uint64_t)chunk_tiles * (WG * ITEMS);
  const uint64_t base = (uint64_t)blockIdx.x * span;
  const uint32_t end = (uint32_t)min((uint64_t)n, base + span);
  using CT = decltype(in.load(0).x);                 // min / max in the cloud's own type: exact, and cheap for fp32
  CT tmn[3] = {(CT)INFINITY, (CT)INFINITY, (CT)INFINITY}, tmx[3] = {(CT)-INFINITY, (CT)-INFINITY, (CT)-INFINITY};
  bool nan_seen = false;
  auto take = [&](const decltype(in.load(0))& r) {
    atomicAdd(&hist[local_bin(bs, block_of_rec(gp, r))], 1u);
    if (bbox6) {
      tmn[0] = r.x < tmn[0] ? r.x : tmn[0]; tmx[0] = r.x > tmx[0] ? r.x : tmx[0];
      tmn[1] = r.y < tmn[1] ? r.y : tmn[1]; tmx[1] = r.y > tmx[1] ? r.y : tmx[1];
      tmn[2] = r.z < tmn[2] ? r.z : tmn[2]; tmx[2] = r.z > tmx[2] ? r.z : tmx[2];
      nan_seen |= (r.x != r.x) | (r.y != r.y) | (r.z != r.z);          // the comparisons above ignore a NaN
    }
  };
  // HB records (their 3 HB loads) in flight per thread; one at a time the pass ran at 4.3 TB/s, a plain reduction of the same bytes at 5
  constexpr int HB = 4;
  uint32_t i = (uint32_t)base + threadIdx.x;
  for (; (uint64_t)i + (HB - 1) * WG < end; i += HB * WG) {
    decltype(in.load(0)) r[HB];
#pragma unroll
    for (int j = 0; j < HB; ++j) r[j] = in.load(i + j * WG);
#pragma unroll
    for (int j = 0; j < HB; ++j) take(r[j]);
  }
  for (; i < end; i += WG) take(in.load(i));
  if (bbox6) {
    if (nan_seen) tmn[0] = (CT)-INFINITY;                               // "not finite" for the host's check of the verified box
    double mn[3] = {(double)tmn[0], (double)tmn[1], (double)tmn[2]}, mx[3] = {(double)tmx[0], (double)tmx[1], (double)tmx[2]};
    wg_bbox_commit(mn, mx, bbox6);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bs.nbins; b += WG) chunk_hist[(size_t)blockIdx.x * bs.nbins + b] = hist[b];
}
constexpr int COL_GROUP = 64;   // chunk rows per column-scan group
// The three column kernels run on a (row groups) x (bins / 64) grid, one bin per thread, with the loads of eight rows in flight:
// as one workgroup walking four bins per thread row by row, the scan alone was 108 us of dependent loads at 1B points.
constexpr int COL_WG = 64;
__global__ __launch_bounds__(COL_WG) void colsum_kernel(const uint32_t* __restrict__ mat, int nrows, int nbins, uint32_t* __restrict__ gsum) {
  const int r0 = blockIdx.x * COL_GROUP, r1 = min(nrows, r0 + COL_GROUP);
  const int b = blockIdx.y * COL_WG + threadIdx.x;
  if (b >= nbins) return;
  uint32_t sacc = 0;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = mat[(size_t)(r + j) * nbins + b];
#pragma unroll
    for (int j = 0; j < 8; ++j) sacc += t[j];
  }
  for (; r < r1; ++r) sacc += mat[(size_t)r * nbins + b];
  gsum[(size_t)blockIdx.x * nbins + b] = sacc;
}
__global__ __launch_bounds__(COL_WG) void colscan_kernel(uint32_t* __restrict__ gsum, int ngroups, int nbins, uint32_t* __restrict__ totals) {
  const int b = blockIdx.x * COL_WG + threadIdx.x;
  if (b >= nbins) return;
  uint32_t run = 0;
  int g = 0;
  for (; g + 8 <= ngroups; g += 8) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = gsum[(size_t)(g + j) * nbins + b];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gsum[(size_t)(g + j) * nbins + b] = run; run += t[j]; }
  }
  for (; g < ngroups; ++g) { const uint32_t t = gsum[(size_t)g * nbins + b]; gsum[(size_t)g * nbins + b] = run; run += t; }
  totals[b] = run;
}
__global__ __launch_bounds__(COL_WG) void colapply_kernel(uint32_t* __restrict__ mat, int nrows, int nbins, const uint32_t* __restrict__ gsum,
                                                          const uint32_t* __restrict__ bin_start) {
  const int r0 = blockIdx.x * COL_GROUP, r1 = min(nrows, r0 + COL_GROUP);
  const int b = blockIdx.y * COL_WG + threadIdx.x;
  if (b >= nbins) return;
  uint32_t run = bin_start[b] + gsum[(size_t)blockIdx.x * nbins + b];
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = mat[(size_t)(r + j) * nbins + b];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mat[(size_t)(r + j) * nbins + b] = run; run += t[j]; }
  }
  for (; r < r1; ++r) { const uint32_t t = mat[(size_t)r * nbins + b]; mat[(size_t)r * nbins + b] = run; run += t; }
}
template <class Loader, int ITEMS, int SW>
__global__ __launch_bounds__(SW) void scatter_chunk_kernel(Loader in, typename Loader::Rec* __restrict__ out, GridParams gp, BinSpec bs, uint32_t n,
                                                           int chunk_tiles, const uint32_t* __restrict__ chunk_base, uint16_t* __restrict__ bid) {
  using Rec = typename Loader::Rec;
  constexpr uint32_t TILE = SW * ITEMS;
  constexpr int BPT = PT_MAXBINS / SW;
  __shared__ uint32_t cursor[PT_MAXBINS];         // next free global slot of every bin, for this chunk
  __shared__ uint32_t binA[PT_MAXBINS];
  __shared__ uint32_t binB[PT_MAXBINS];
  __shared__ uint32_t wsum[SW / 64];
  __shared__ Rec stage[TILE];
  for (int b = threadIdx.x; b < PT_MAXBINS; b += SW) cursor[b] = b < bs.nbins ? chunk_base[(size_t)blockIdx.x * bs.nbins + b] : 0u;
  for (int t = 0; t < chunk_tiles; ++t) {
    const uint64_t s64 = ((uint64_t)blockIdx.x * chunk_tiles + t) * TILE;
    if (s64 >= n) break;
    const uint32_t s = (uint32_t)s64, e = (uint32_t)min((uint64_t)n, s64 + TILE);
    for (int b = threadIdx.x; b < PT_MAXBINS; b += SW) binA[b] = 0;
    __syncthreads();
    Rec r[ITEMS];
    uint32_t lb[ITEMS], rank[ITEMS];
    // all of the tile's loads first (see scatter_kernel: in one loop with the binning, each record's three loads were waited for
    // before the next record's were issued -- eight memory latencies in a row per tile)
    // (unconditional loads, the index clamped into the tile: a load behind `if (i < e)` is copied into its array slot at the join,
    //  and that copy waits for it on the spot)
    if (in.has_ids()) {                                      // (workgroup-uniform)
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) r[j] = in.template load_t<true>(min(s + j * SW + threadIdx.x, e - 1u));
    } else {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) r[j] = in.template load_t<false>(min(s + j * SW + threadIdx.x, e - 1u));
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t i = s + j * SW + threadIdx.x;
      if (i < e) {
        lb[j] = local_bin(bs, block_of_rec(gp, r[j]));
        rank[j] = atomicAdd(&binA[lb[j]], 1u);
      }
    }
    __syncthreads();
    {
      uint32_t c[BPT], sum = 0;
#pragma unroll
      for (int i = 0; i < BPT; ++i) { c[i] = binA[threadIdx.x * BPT + i]; sum += c[i]; }
      uint32_t tot;
      uint32_t ex = block_excl_scan_n<SW / 64>(sum, wsum, tot);
#pragma unroll
      for (int i = 0; i < BPT; ++i) {
        const int b = threadIdx.x * BPT + i;
        binA[b] = ex;
        binB[b] = cursor[b] - ex;
        cursor[b] += c[i];               // one thread owns each bin: no race
        ex += c[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t i = s + j * SW + threadIdx.x;
      if (i < e) stage[binA[lb[j]] + rank[j]] = r[j];
    }
    __syncthreads();
    const uint32_t cnt = e - s;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t slot = j * SW + threadIdx.x;
      if (slot < cnt) {
        const Rec v = stage[slot];
        const uint32_t blk = block_of_rec(gp, v);
        const uint32_t pos = binB[local_bin(bs, blk)] + slot;
        out[pos] = v;
        if (bid) bid[pos] = (uint16_t)(blk & (PT_MACRO_BLOCKS - 1));   // 2 bytes here save pass 2's histogram a 16-byte read
      }
    }
    __syncthreads();
  }
}

// ---- pass 1 WITHOUT a histogram pass (round 3): bin regions sized from a SAMPLE, space taken in blocks ----------------------------
// The chunked pass 1 above reads the cloud twice: once for the per-chunk histograms that make its placement exact, once to move the
// records (12 + 12 bytes per point read).  This form reads it once.  A sample of the cloud (one run of 256 points in 64) estimates
// every bin's population; each bin gets a region of that size plus slack (8 sigma of the estimate + room for the padding below), laid
// out back to back; PERSISTENT workgroups (one per CU) walk their share of the tiles and take space inside a bin's region in BLOCKS of
// POOL_B records with one global atomic per block -- n / POOL_B reservations per address instead of one per tile -- filling a block
// over as many tiles as it takes.  What a workgroup leaves unfilled in its last block of every bin at the end is padded with SENTINEL
// records (id = PT_NOIDX_U, block id 0xFFFF: < 2 % of the records at 1e9 points), which pass 2 drops.  A bin that outgrows its region
// (the sample missed a cluster) raises a flag: the records go to a scratch area, the host sees the flag in the read-back it makes
// anyway and redoes the build with the exact histogram -- the same guess-and-verify shape as the sampled bounding box.
constexpr uint32_t POOL_B = 128;                 // records per reservation block (2 KB of fp32 records)
constexpr uint32_t POOL_SAMPLE_RUN = 256;        // consecutive points per sampled run (coalesced)
struct PoolTables {
  uint32_t* est;        // [PT_MAXBINS]  sample counts per bin
  uint32_t* start;      // [PT_MAXBINS+1] region start of every bin (multiple of POOL_B)
  uint32_t* cursor;     // [PT_MAXBINS]  next unreserved record of every bin's region; after pass 1: where the bin ends
  uint32_t* limit;      // [PT_MAXBINS]  end of every bin's region
  uint32_t* flag;       // [4] {overflow, sampled points, records incl. sentinels, -}
};
template <class Loader>
__global__ __launch_bounds__(WG) void pool_sample_kernel(Loader in, GridParams gp, BinSpec bs, uint32_t n, uint32_t stride, PoolTables pt) {
  __shared__ uint32_t hist[PT_MAXBINS];
  for (int b = threadIdx.x; b < bs.nbins; b += WG) hist[b] = 0;
  __syncthreads();
  const uint64_t span = (uint64_t)stride * POOL_SAMPLE_RUN;
  uint32_t mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * span + threadIdx.x; i < n; i += (uint64_t)gridDim.x * span) {
    atomicAdd(&hist[local_bin(bs, block_of_rec(gp, in.load((uint32_t)i)))], 1u);
    ++mine;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bs.nbins; b += WG) { const uint32_t c = hist[b]; if (c) atomicAdd(&pt.est[b], c); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&pt.flag[1], mine);
}
// regions from the sample: cap = n / ns * (s + 8 sqrt(s) + 32) + padding room, rounded up to whole blocks
__global__ __launch_bounds__(WG) void pool_setup_kernel(PoolTables pt, int nbins, uint32_t n, uint32_t nwg, uint32_t pool_records /* capacity before the scratch area */) {
  __shared__ uint32_t wsum[4];
  const double f = pt.flag[1] ? (double)n / (double)pt.flag[1] : 1.0;
  uint32_t cap[4], sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = threadIdx.x * 4 + i;
    cap[i] = 0;
    if (b < nbins) {
      const double sct = (double)pt.est[b];
      const double want = f * (sct + 8.0 * sqrt(sct) + 32.0) + (double)nwg * (double)POOL_B;
      const double capd = fmin(want, 4.0e9);
      cap[i] = (uint32_t)(((uint64_t)capd + POOL_B) / POOL_B * POOL_B);
    }
    sum += cap[i] / POOL_B;                       // (in blocks: the total may pass 2^32 records before it is refused)
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(sum, wsum, tot);
  const bool fits = (uint64_t)tot * POOL_B <= (uint64_t)pool_records;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = threadIdx.x * 4 + i;
    if (b < nbins) {
      // (regions that do not fit the allocation: every bin gets an empty region -- all of pass 1 lands in the scratch area, flagged)
      const uint32_t st = fits ? ex * POOL_B : 0u;
      pt.start[b] = st; pt.cursor[b] = st; pt.limit[b] = fits ? st + cap[i] : 0u;
    }
    ex += cap[i] / POOL_B;
  }
  if (threadIdx.x == 0) { pt.start[nbins] = fits ? tot * POOL_B : 0u; if (!fits) pt.flag[0] = 1u; }
}
template <class Loader, int ITEMS, int SW>
__global__ __launch_bounds__(SW) void scatter_pool_kernel(Loader in, typename Loader::Rec* __restrict__ out, GridParams gp, BinSpec bs, uint32_t n,
                                                          PoolTables pt, uint32_t scratch_base, uint16_t* __restrict__ bid, uint64_t* bbox6) {
  using Rec = typename Loader::Rec;
  constexpr uint32_t TILE = SW * ITEMS;
  static_assert(PT_MAXBINS == SW, "one bin per thread in the reservation step");
  static_assert(TILE % POOL_B == 0, "the scratch area holds one tile");
  __shared__ uint32_t lpos[PT_MAXBINS];           // next free record of the block this workgroup is filling in every bin (0: none yet)
  __shared__ uint32_t binA[PT_MAXBINS];           // counts, then local start of each bin in `stage`
  __shared__ uint32_t thrS[PT_MAXBINS];           // stage slot at which the bin's share leaves the old block for the new one(s)
  __shared__ uint32_t adjA[PT_MAXBINS];           // global position - stage slot, old block
  __shared__ uint32_t adjB[PT_MAXBINS];           //                             , new block(s)
  __shared__ uint32_t wsum[SW / 64];
  __shared__ Rec stage[TILE];
  __shared__ double wbox[SW / 64][6];
  lpos[threadIdx.x] = 0;
  const uint32_t ntiles = (uint32_t)(((uint64_t)n + TILE - 1) / TILE);
  const uint32_t t0 = (uint32_t)((uint64_t)ntiles * blockIdx.x / gridDim.x), t1 = (uint32_t)((uint64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
  using CT = decltype(in.load(0).x);
  CT tmn[3] = {(CT)INFINITY, (CT)INFINITY, (CT)INFINITY}, tmx[3] = {(CT)-INFINITY, (CT)-INFINITY, (CT)-INFINITY};
  bool nan_seen = false;
  for (uint32_t t = t0; t < t1; ++t) {
    const uint64_t s64 = (uint64_t)t * TILE;
    const uint32_t s = (uint32_t)s64, e = (uint32_t)min((uint64_t)n, s64 + TILE);
    binA[threadIdx.x] = 0;
    __syncthreads();
    Rec r[ITEMS];
    uint32_t lb[ITEMS], rank[ITEMS];
    if (in.has_ids()) {                                      // (workgroup-uniform; all of the tile's loads first: see scatter_chunk_kernel)
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) r[j] = in.template load_t<true>(min(s + j * SW + threadIdx.x, e - 1u));
    } else {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) r[j] = in.template load_t<false>(min(s + j * SW + threadIdx.x, e - 1u));
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t i = s + j * SW + threadIdx.x;
      if (i < e) {
        lb[j] = local_bin(bs, block_of_rec(gp, r[j]));
        rank[j] = atomicAdd(&binA[lb[j]], 1u);
        if (bbox6) {                                           // the exact bounding box of everything read (the grid was laid out from a sample)
          tmn[0] = r[j].x < tmn[0] ? r[j].x : tmn[0]; tmx[0] = r[j].x > tmx[0] ? r[j].x : tmx[0];
          tmn[1] = r[j].y < tmn[1] ? r[j].y : tmn[1]; tmx[1] = r[j].y > tmx[1] ? r[j].y : tmx[1];
          tmn[2] = r[j].z < tmn[2] ? r[j].z : tmn[2]; tmx[2] = r[j].z > tmx[2] ? r[j].z : tmx[2];
          nan_seen |= (r[j].x != r[j].x) | (r[j].y != r[j].y) | (r[j].z != r[j].z);
        }
      }
    }
    __syncthreads();
    {
      const int b = threadIdx.x;
      const uint32_t c = binA[b];
      uint32_t tot;
      const uint32_t ex = block_excl_scan_n<SW / 64>(c, wsum, tot);
      binA[b] = ex;
      if (c) {
        const uint32_t pos = lpos[b];
        const uint32_t room = (POOL_B - (pos & (POOL_B - 1u))) & (POOL_B - 1u);      // left in the block being filled (regions start on block boundaries)
        adjA[b] = pos - ex;
        if (c <= room) { thrS[b] = ex + c; adjB[b] = 0u; lpos[b] = pos + c; }
        else {
          const uint32_t need = c - room, take = (need + POOL_B - 1u) / POOL_B * POOL_B;
          const uint32_t base = atomicAdd(&pt.cursor[b], take);
          thrS[b] = ex + room;
          if (base > pt.limit[b] || take > pt.limit[b] - base) {      // the region is full: this share goes to the scratch area, the build is redone
            atomicOr(&pt.flag[0], 1u);
            adjB[b] = scratch_base - thrS[b];                          // (slot - thrS < TILE: inside the scratch area)
            lpos[b] = pos + room;
          } else {
            adjB[b] = base - thrS[b];
            lpos[b] = base + need;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t i = s + j * SW + threadIdx.x;
      if (i < e) stage[binA[lb[j]] + rank[j]] = r[j];
    }
    __syncthreads();
    const uint32_t cnt = e - s;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint32_t slot = j * SW + threadIdx.x;
      if (slot < cnt) {
        const Rec v = stage[slot];
        const uint32_t blk = block_of_rec(gp, v);
        const uint32_t lbn = local_bin(bs, blk);
        const uint32_t pos = (slot < thrS[lbn] ? adjA[lbn] : adjB[lbn]) + slot;
        out[pos] = v;
        if (bid) bid[pos] = (uint16_t)(blk & (PT_MACRO_BLOCKS - 1));
      }
    }
    __syncthreads();
  }
  // pad the blocks left partly filled: pass 2 reads every bin's region from its start to its cursor and drops these
  {
    const uint32_t pos = lpos[threadIdx.x];
    const uint32_t room = (POOL_B - (pos & (POOL_B - 1u))) & (POOL_B - 1u);
    if (room && pos + room <= pt.limit[threadIdx.x]) {
      Rec sv;
      sv.x = (decltype(sv.x))gp.bbmin[0]; sv.y = (decltype(sv.y))gp.bbmin[1]; sv.z = (decltype(sv.z))gp.bbmin[2]; sv.id = PT_NOIDX_U;
      for (uint32_t q = 0; q < room; ++q) { out[pos + q] = sv; if (bid) bid[pos + q] = (uint16_t)0xFFFFu; }
    }
  }
  if (bbox6) {
    if (nan_seen) tmn[0] = (CT)-INFINITY;
    double mn[3] = {(double)tmn[0], (double)tmn[1], (double)tmn[2]}, mx[3] = {(double)tmx[0], (double)tmx[1], (double)tmx[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mn[a] = fmin(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmax(mx[a], __shfl_xor(mx[a], o)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { wbox[threadIdx.x >> 6][a] = mn[a]; wbox[threadIdx.x >> 6][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      double v = wbox[0][threadIdx.x];
      for (int w = 1; w < SW / 64; ++w) v = threadIdx.x < 3 ? fmin(v, wbox[w][threadIdx.x]) : fmax(v, wbox[w][threadIdx.x]);
      if (threadIdx.x < 3) { if (v != INFINITY) atomicMin((unsigned long long*)&bbox6[threadIdx.x], (unsigned long long)enc_f64(v)); }
      else if (v != -INFINITY) atomicMax((unsigned long long*)&bbox6[threadIdx.x], (unsigned long long)enc_f64(v));
    }
  }
}
// after pass 1: every bin ends at its cursor (clamped to its region); tiles of pass 2 per bin, scanned
__global__ __launch_bounds__(WG) void pool_finish_kernel(PoolTables pt, int nbins, uint32_t tile_pts, uint32_t* counts, uint32_t* tile_first) {
  __shared__ uint32_t wsum[4];
  uint32_t t[4], st = 0, sc = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = threadIdx.x * 4 + i;
    t[i] = 0;
    if (b < nbins) {
      const uint32_t end = min(pt.cursor[b], pt.limit[b]);
      pt.cursor[b] = end;
      const uint32_t c = end - pt.start[b];
      counts[b] = c;
      t[i] = (c + tile_pts - 1) / tile_pts;
      sc += c;
    }
    st += t[i];
  }
  uint32_t tot, totc;
  uint32_t et = block_excl_scan(st, wsum, tot);
  __syncthreads();
  (void)block_excl_scan(sc, wsum, totc);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = threadIdx.x * 4 + i;
    if (b < nbins) tile_first[b] = et;
    et += t[i];
  }
  if (threadIdx.x == 0) { tile_first[nbins] = tot; pt.flag[2] = totc; }
}

// ---- pass 2 WITHOUT its histogram (round 3; clouds the previous build of the same resident cloud found uniform) -------------------
// Pass 2's scatter already takes its space with one atomic per (tile, block); what the histogram + scan before it provided were the
// blocks' exact STARTS.  Regions sized from an estimate do as well: a block's expected population is its macro segment's count times
// the share of the macro's cells (inside the grid, outside the empty padding of a sampled bounding box) that are the block's, plus 6 sigma
// and a constant (a 2048-point block: + 15 %); the scatter counts as it goes,
// the counts are scanned AFTER it, and finalize reads every block from the start of its region and writes it to its exact place.
// Gone: the 2-byte block ids pass 1 wrote beside the records (16-byte pieces: 5.7 GB of write traffic for 2 GB of ids) and the pass that
// read them.  A block that outgrows its region (a cloud that is not uniform inside its macro blocks after all) raises flag bit 2.
struct OccBox { int lo[3], hi[3]; };
__global__ __launch_bounds__(WG) void pool2_sizes_kernel(GridParams gp, OccBox ob, const uint32_t* __restrict__ macro_count, uint32_t nblocks, uint32_t* __restrict__ rsize) {
  const uint32_t b = blockIdx.x * WG + threadIdx.x;
  if (b > nblocks) return;
  if (b == nblocks) { rsize[b] = 0; return; }
  const uint32_t macro = b >> 9, m9 = b & 511u;
  const int mx = (int)(macro % (uint32_t)gp.mdim[0]), my = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]), mz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1]));
  const int bx = mx * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u));
  const int by = my * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u));
  const int bz = mz * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u));
  // cells of [lo, lo + len) the cloud is expected to occupy: inside the grid, and not in the empty padding a sampled bounding box was given
  auto inside = [](int lo, int len, int olo, int ohi) { return max(0, min(lo + len, ohi) - max(lo, olo)); };
  const double cov_b = (double)inside(bx * 8, 8, ob.lo[0], ob.hi[0]) * inside(by * 8, 8, ob.lo[1], ob.hi[1]) * inside(bz * 8, 8, ob.lo[2], ob.hi[2]);
  const double cov_m = (double)inside(mx * 64, 64, ob.lo[0], ob.hi[0]) * inside(my * 64, 64, ob.lo[1], ob.hi[1]) * inside(mz * 64, 64, ob.lo[2], ob.hi[2]);
  const double e = cov_m > 0.0 ? (double)macro_count[macro] * cov_b / cov_m : 0.0;
  rsize[b] = (uint32_t)(e + 6.0 * sqrt(e) + 48.0);          // (48 also for blocks expected empty: outliers the sampled box missed land there)
}
__global__ __launch_bounds__(WG) void pool2_check_kernel(const uint32_t* __restrict__ rstart, uint32_t nblocks, uint32_t capacity, uint32_t* flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && rstart[nblocks] > capacity) atomicOr(flag, 2u);      // the regions do not fit the allocation
}
__global__ __launch_bounds__(WG) void pool2_counts_kernel(const uint32_t* __restrict__ rstart, const uint32_t* __restrict__ cursor, uint32_t nblocks,
                                                          uint32_t* __restrict__ count) {
  const uint32_t b = blockIdx.x * WG + threadIdx.x;
  if (b > nblocks) return;
  count[b] = b < nblocks ? min(cursor[b], rstart[b + 1]) - rstart[b] : 0u;
}

// Is the cloud uniform enough for those regions?  Round 3 only knew from a PREVIOUS build of the same resident cloud (finalize's occupancy);
// round 4 asks a sample (about 4 M points) BEFORE the first sort: the sample's count per 8^3-cell block against what its macro block's sample count and
// the block's share of the macro's occupied cells predict.  A uniform cloud keeps every block within 6 sigma of that (a 2048-point block
// shows 32 +- 5.7 sample points); surfaces, clusters and anything with structure below the macro scale do not, by orders of magnitude.
// The answer only CHOOSES the pass 2: the pooled one still verifies itself (flag bit 2) and is redone exactly if a block outgrows its region.
template <class Loader>
__global__ __launch_bounds__(WG) void uniform_probe_kernel(Loader in, GridParams gp, uint32_t n, uint32_t stride, uint32_t* __restrict__ blk_cnt,
                                                           uint32_t* __restrict__ cell_bits, unsigned long long* __restrict__ acc) {
  const uint64_t span = (uint64_t)stride * POOL_SAMPLE_RUN;
  uint32_t same = 0;         // lanes whose point fell into the block of the wave's first point: consecutive points of a cloud in RANDOM order almost never do
  for (uint64_t i = (uint64_t)blockIdx.x * span + threadIdx.x; i < n; i += (uint64_t)gridDim.x * span) {
    int cx, cy, cz;
    pt_cell_of(gp, in.load((uint32_t)i), cx, cy, cz);
    const uint32_t blk = pt_block_id(gp.mdim, cx, cy, cz);
    same += (uint32_t)__popcll(__ballot(blk == (uint32_t)__builtin_amdgcn_readfirstlane((int)blk))) - 1u;
    atomicAdd(&blk_cnt[blk], 1u);                       // (n / 64 atomics over nblocks addresses: ~30 per address)
    const uint32_t cib = (uint32_t)((cx & 7) | ((cy & 7) << 3) | ((cz & 7) << 6));
    atomicOr(&cell_bits[(size_t)blk * 16 + (cib >> 5)], 1u << (cib & 31u));      // which of the block's 512 cells the sample has seen
  }
  if ((threadIdx.x & 63) == 0 && same) atomicAdd(&acc[3], (unsigned long long)same);   // (every lane of a wave counted the same ballots)
}
// the macro blocks' sample counts = sums of their 512 blocks' (one atomic per sample point on <= 1024 addresses cost 4 ms at 1e9 points)
// A block is INTERIOR when all of its cells lie inside the occupied box shrunk by one cell: the cloud's outermost cell layers are filled to
// whatever fraction its extent leaves of them, and a block that owns such a layer (or only a sliver of the box) misses its expected count by
// tens of percent -- at 167 sample points per block (64 M points, rho 6) the 2500 face blocks alone tripled the chi-square sum of a uniform
// cloud.  The chi-square is taken over interior blocks against the mean of their macro block's interior blocks.
__device__ inline bool uniform_interior(const GridParams& gp, const OccBox& ob, uint32_t b) {
  const uint32_t macro = b >> 9, m9 = b & 511u;
  const int mx = (int)(macro % (uint32_t)gp.mdim[0]), my = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]), mz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1]));
  const int c0[3] = {(mx * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u))) * 8, (my * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u))) * 8,
                     (mz * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u))) * 8};
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) in = in && c0[a] >= ob.lo[a] + 1 && c0[a] + 8 <= ob.hi[a] - 1;
  return in;
}
__global__ __launch_bounds__(WG) void uniform_macro_kernel(GridParams gp, OccBox ob, const uint32_t* __restrict__ blk_cnt, uint32_t* __restrict__ macro_cnt,
                                                           uint32_t* __restrict__ macro_int, uint32_t* __restrict__ macro_nint) {
  __shared__ uint32_t wsum[4];
  const uint32_t b0 = blockIdx.x * PT_MACRO_BLOCKS + threadIdx.x, b1 = b0 + WG;
  const uint32_t v0 = blk_cnt[b0], v1 = blk_cnt[b1];
  const bool i0 = uniform_interior(gp, ob, b0), i1 = uniform_interior(gp, ob, b1);
  uint32_t tot, tin, nin;
  (void)block_excl_scan(v0 + v1, wsum, tot);
  __syncthreads();
  (void)block_excl_scan((i0 ? v0 : 0u) + (i1 ? v1 : 0u), wsum, tin);
  __syncthreads();
  (void)block_excl_scan((uint32_t)i0 + (uint32_t)i1, wsum, nin);
  if (threadIdx.x == 0) { macro_cnt[blockIdx.x] = tot; macro_int[blockIdx.x] = tin; macro_nint[blockIdx.x] = nin; }
}
// Two tests per block against e = (its macro block's sample count) x (its share of the macro's occupied cells): a MAXIMUM test -- more
// than e + 6 sqrt(e) + 8 sample points is a clump -- and the block's term (s - e)^2 / e of a CHI-SQUARE sum over all blocks with e >= 4,
// which is what sees smooth density gradients (a +-15 % drift across a macro block, enough to overflow the regions' 6-sigma slack, moves
// a 32-point sample count by less than one sigma per block but the sum by tens of sigmas).  acc[0] += term * 1024 (fixed point), acc[1] += 1.
// Third sum (acc[2], x 16): an estimate of how many CELLS the cloud occupies.  A block's s sample points were seen in u distinct cells: points
// spread evenly over a of the block's cells show a (1 - exp(-s / a)) of them, which gives a (u = s, every sample point in a cell of its own,
// says nothing: a = all of the block's cells, the cautious end); the m = stride x s points the block really holds then occupy
// a (1 - exp(-m / a)) cells.  n over the sum estimates the points per occupied cell the sort would count, which is what the choice of the cell
// size goes by (pt_api.hip, rebuild) -- from below wherever the sample is thin, so that a refinement it suggests is one the count would ask for.
__global__ __launch_bounds__(WG) void uniform_check_kernel(GridParams gp, OccBox ob, const uint32_t* __restrict__ macro_cnt, const uint32_t* __restrict__ macro_int,
                                                           const uint32_t* __restrict__ macro_nint, const uint32_t* __restrict__ blk_cnt,
                                                           const uint32_t* __restrict__ cell_bits, uint32_t nblocks, uint32_t stride, uint32_t* flag, unsigned long long* acc) {
  const uint32_t b = blockIdx.x * WG + threadIdx.x;
  float term = 0.f, occ = 0.f;
  uint32_t used = 0;
  if (b < nblocks) {
  const uint32_t macro = b >> 9, m9 = b & 511u;
  const int mx = (int)(macro % (uint32_t)gp.mdim[0]), my = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]), mz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1]));
  const int bx = mx * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u));
  const int by = my * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u));
  const int bz = mz * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u));
  auto inside = [](int lo, int len, int olo, int ohi) { return max(0, min(lo + len, ohi) - max(lo, olo)); };
  const double cov_b = (double)inside(bx * 8, 8, ob.lo[0], ob.hi[0]) * inside(by * 8, 8, ob.lo[1], ob.hi[1]) * inside(bz * 8, 8, ob.lo[2], ob.hi[2]);
  const double cov_m = (double)inside(mx * 64, 64, ob.lo[0], ob.hi[0]) * inside(my * 64, 64, ob.lo[1], ob.hi[1]) * inside(mz * 64, 64, ob.lo[2], ob.hi[2]);
  const double e = cov_m > 0.0 ? (double)macro_cnt[macro] * cov_b / cov_m : 0.0;
  const double sb = (double)blk_cnt[b];
  if (sb > e + 6.0 * sqrt(e) + 8.0) atomicOr(flag, 1u);
  if (uniform_interior(gp, ob, b)) {
    const double ei = (double)macro_int[macro] / (double)max(macro_nint[macro], 1u);      // the mean of the macro block's interior blocks
    if (ei >= 4.0) { term = (float)((sb - ei) * (sb - ei) / ei); used = 1; }
  }
  if (sb > 0.0 && cov_b > 0.0) {
    uint32_t u = 0;
    const uint4* bits = reinterpret_cast<const uint4*>(cell_bits + (size_t)b * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint4 w = bits[i]; u += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w); }
    float a = (float)cov_b;
    const float fs = (float)sb, fu = (float)u;
    if (fu < fs && a * (1.f - __expf(-fs / a)) > fu) {      // fewer distinct cells than an even spread over the whole block would show: solve for a
      float lo = fu, hi = a;
      for (int it = 0; it < 14; ++it) { const float mid = 0.5f * (lo + hi); if (mid * (1.f - __expf(-fs / mid)) < fu) lo = mid; else hi = mid; }
      a = hi;
    }
    occ = a * (1.f - __expf(-(float)stride * fs / a));
  }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { term += __shfl_xor(term, o); used += __shfl_xor(used, o); occ += __shfl_xor(occ, o); }
  if ((threadIdx.x & 63) == 0) {
    if (used) {
      atomicAdd(&acc[0], (unsigned long long)(term * 1024.0f + 0.5f));
      atomicAdd(&acc[1], (unsigned long long)used);
    }
    if (occ > 0.f) atomicAdd(&acc[2], (unsigned long long)(occ * 16.0f + 0.5f));
  }
}

// pass-2 histogram from the block ids pass 1 left beside the records: same tiling and flush as hist_kernel
template <int ITEMS>
__global__ __launch_bounds__(WG) void hist_bid_kernel(const uint16_t* __restrict__ bid, int nbins, const uint32_t* __restrict__ seg_start,
                                                      const uint32_t* __restrict__ tile_first, int nseg, uint32_t* counts, int tiles_per_wg,
                                                      const uint32_t* __restrict__ seg_end = nullptr, const uint32_t* abort_flag = nullptr) {
  __shared__ uint32_t hist[PT_MACRO_BLOCKS];
  if (abort_flag && (*abort_flag & 1u)) return;      // (pooled pass 1 overflowed: see scatter_kernel)
  constexpr uint32_t TILE = WG * ITEMS;
  int cur_seg = -1;
  for (int tt = 0; tt < tiles_per_wg; ++tt) {
    uint32_t seg, s, e;
    if (!tile_range(seg_start, tile_first, nseg, blockIdx.x * tiles_per_wg + tt, TILE, seg, s, e, seg_end)) break;
    if ((int)seg != cur_seg) {
      __syncthreads();
      if (cur_seg >= 0)
        for (int b = threadIdx.x; b < nbins; b += WG) { const uint32_t c = hist[b]; if (c) atomicAdd(&counts[(uint32_t)cur_seg * PT_MACRO_BLOCKS + b], c); }
      __syncthreads();
      for (int b = threadIdx.x; b < nbins; b += WG) hist[b] = 0;
      __syncthreads();
      cur_seg = (int)seg;
    }
    // eight ids per 16-byte load (the table is 16-byte aligned and padded: pt_api.hip make_tables), from the aligned group the tile
    // starts in; the ids outside [s, e) are masked.  Two bytes per lane and load -- the first form of this -- moved 128 bytes per
    // wave-instruction.
    static_assert(TILE % 8 == 0 && (TILE + 8) <= 3 * WG * 8, "three trips cover a tile and its misalignment");
    const uint32_t a = s & ~7u;
    uint4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {                          // (the three loads first and unconditional: a group beyond the tile reads the tile's first)
      const uint32_t g = a + ((uint32_t)j * WG + threadIdx.x) * 8u;
      v[j] = *reinterpret_cast<const uint4*>(bid + (g < e ? g : a));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const uint32_t g = a + ((uint32_t)j * WG + threadIdx.x) * 8u;
      if (g < e) {
        const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t i = g + (uint32_t)q, id = (w[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
          if (i >= s && i < e && id < (uint32_t)PT_MACRO_BLOCKS) atomicAdd(&hist[id], 1u);      // (0xFFFF: a sentinel of the pooled pass 1)
        }
      }
    }
  }
  __syncthreads();
  if (cur_seg >= 0)
    for (int b = threadIdx.x; b < nbins; b += WG) { const uint32_t c = hist[b]; if (c) atomicAdd(&counts[(uint32_t)cur_seg * PT_MACRO_BLOCKS + b], c); }
}

// ---- finalize: counting sort of one 8x8x8-cell block by local cell, in LDS -------------------------
constexpr int FWG = 512;        // finalize workgroup: one thread per local cell in the scan
constexpr int FITEMS = 12;      // records a thread keeps in registers: blocks up to 6144 points are read once
__device__ inline void put_shadow(RecF* shadow, uint32_t pos, double x, double y, double z) {
  RecF v;
  v.x = (float)x; v.y = (float)y; v.z = (float)z; v.id = pos;      // rounded coordinates + where the exact record lives
  shadow[pos] = v;
}
template <class Rec>
__global__ __launch_bounds__(FWG) void finalize_kernel(const Rec* __restrict__ in, Rec* __restrict__ out, GridParams gp,
                                                       const uint32_t* __restrict__ block_start, uint32_t* cell_start, uint32_t* occupied,
                                                       RecF* __restrict__ shadow, const uint32_t* __restrict__ in_start = nullptr,
                                                       const uint32_t* abort_flag = nullptr) {
  constexpr int FSTAGE = 64 * 1024 / (int)sizeof(Rec);        // records of a block that fit the 64-KB output stage
  if (abort_flag && *abort_flag) return;      // a pooled pass overflowed (either bit): this build's tables are void, the host redoes it (see scatter_kernel)
  static_assert(FSTAGE <= FWG * FITEMS, "staged blocks are register-resident blocks");
  __shared__ uint32_t cnt[PT_BLOCK_CELLS];
  __shared__ uint32_t wsum[FWG / 64];
  __shared__ Rec stage[FSTAGE];
  const uint32_t b = blockIdx.x;
  const uint32_t s = block_start[b], e = block_start[b + 1];
  // (pooled pass 2: the block's records sit at the start of its REGION in `in`, not at its final place)
  if (in_start) in += (ptrdiff_t)in_start[b] - (ptrdiff_t)s;
  if (s == e) {   // empty block: only the table
    if (occupied && threadIdx.x == 0) occupied[b] = 0;
    if (cell_start) {
      cell_start[b * PT_BLOCK_CELLS + threadIdx.x] = s;
      if (b == gridDim.x - 1 && threadIdx.x == 0) cell_start[(b + 1) * PT_BLOCK_CELLS] = e;
    }
    return;
  }
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const bool in_regs = (e - s) <= (uint32_t)(FWG * FITEMS);
  Rec r[FITEMS];
  uint32_t lc[FITEMS];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < FITEMS; ++j) r[j] = in[min(s + j * FWG + threadIdx.x, e - 1u)];     // (all loads first, unconditional: see scatter_chunk_kernel)
#pragma unroll
    for (int j = 0; j < FITEMS; ++j) {
      const uint32_t i = s + j * FWG + threadIdx.x;
      if (i < e) {
        int cx, cy, cz;
        pt_cell_of(gp, r[j], cx, cy, cz);
        lc[j] = pt_local_cell(cx, cy, cz);
        atomicAdd(&cnt[lc[j]], 1u);
      }
    }
  } else {
    for (uint32_t i = s + threadIdx.x; i < e; i += FWG) {
      int cx, cy, cz;
      pt_cell_of(gp, in[i], cx, cy, cz);
      atomicAdd(&cnt[pt_local_cell(cx, cy, cz)], 1u);
    }
  }
  __syncthreads();
  // exclusive scan of the 512 cell counts, one per thread
  const uint32_t c0 = cnt[threadIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // non-empty cells of this block (the host derives the points per OCCUPIED cell from their sum).  One plain store per
  // block into a table: a single shared atomic counter would serialise ~5e5 workgroups on one address.
  const uint32_t nzw = (uint32_t)__popcll(__ballot(c0 > 0));
  uint32_t cmax = c0;                          // most populated cell of the wave: what tells the host that cells need refining
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, o));
  const uint32_t incl = wave_incl_scan(c0);
  __shared__ uint32_t wnz[FWG / 64];
  __shared__ uint32_t wmx[FWG / 64];
  if (lane == 63) wsum[w] = incl;
  if (lane == 0) { wnz[w] = nzw; wmx[w] = cmax; }
  __syncthreads();
  uint32_t off = 0;
#pragma unroll
  for (int i = 0; i < FWG / 64; ++i) if (i < w) off += wsum[i];
  if (occupied && threadIdx.x == 0) {
    uint32_t nz = 0, mx = 0;
#pragma unroll
    for (int i = 0; i < FWG / 64; ++i) { nz += wnz[i]; mx = max(mx, wmx[i]); }
    occupied[b] = nz | (min(mx, 0x3FFFFFu) << 10);       // non-empty cells (<= 512) | points of the fullest cell, saturated
  }
  const uint32_t ex = off + incl - c0;
  cnt[threadIdx.x] = ex;                      // cursor, relative to s
  if (cell_start) {
    cell_start[b * PT_BLOCK_CELLS + threadIdx.x] = s + ex;
    if (b == gridDim.x - 1 && threadIdx.x == 0) cell_start[(b + 1) * PT_BLOCK_CELLS] = e;
  }
  __syncthreads();
  if (e - s <= (uint32_t)FSTAGE) {
    // usual case: the sorted block is assembled in LDS and leaves as full lines (a direct store would be 64 separate
    // 16-byte requests per wave instruction)
#pragma unroll
    for (int j = 0; j < FITEMS; ++j) {
      const uint32_t i = s + j * FWG + threadIdx.x;
      if (i < e) stage[atomicAdd(&cnt[lc[j]], 1u)] = r[j];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < e - s; i += FWG) {
      const Rec v = stage[i];
      out[s + i] = v;
      if (shadow) put_shadow(shadow, s + i, (double)v.x, (double)v.y, (double)v.z);
    }
  } else if (in_regs) {
#pragma unroll
    for (int j = 0; j < FITEMS; ++j) {
      const uint32_t i = s + j * FWG + threadIdx.x;
      if (i < e) {
        const uint32_t pos = s + atomicAdd(&cnt[lc[j]], 1u);
        out[pos] = r[j];
        if (shadow) put_shadow(shadow, pos, (double)r[j].x, (double)r[j].y, (double)r[j].z);
      }
    }
  } else {
    for (uint32_t i = s + threadIdx.x; i < e; i += FWG) {   // oversized block: second read (mostly L2)
      const Rec v = in[i];
      int cx, cy, cz;
      pt_cell_of(gp, v, cx, cy, cz);
      const uint32_t pos = s + atomicAdd(&cnt[pt_local_cell(cx, cy, cz)], 1u);
      out[pos] = v;
      if (shadow) put_shadow(shadow, pos, (double)v.x, (double)v.y, (double)v.z);
    }
  }
}

// finalize's per-block words {non-empty cells | fullest cell << 10}: out[0] += sum of the former, out[1] = max of the latter
__global__ __launch_bounds__(WG) void sum_u32_kernel(const uint32_t* __restrict__ v, uint32_t n, uint32_t* out) {
  uint32_t acc = 0, mx = 0;
  for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) { const uint32_t w = v[i]; acc += w & 0x3FFu; mx = max(mx, w >> 10); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
  if ((threadIdx.x & 63) == 0) { if (acc) atomicAdd(out, acc); if (mx) atomicMax(out + 1, mx); }
}

template <class Rec> constexpr int items_for() { return sizeof(Rec) == 16 ? 8 : 4; }

}  // namespace

// =================================================================================================
int pt_sort_tile_points(size_t rec_size) { return rec_size == 16 ? 4096 : 2048; }   // 64 KB of records per scatter tile
// tiles per pass-1 chunk: enough chunks to fill the chip, few enough that the chunk-histogram table stays small
int pt_sort_chunk_tiles(uint32_t n, size_t rec_size) {
  const uint32_t tile = (uint32_t)pt_sort_tile_points(rec_size);
  const uint32_t ntiles = (n + tile - 1) / tile;
  return (int)std::max<uint32_t>(1, std::min<uint32_t>(32, ntiles / 4096));
}
// The pooled pass 1's capacity.  Regions: cap_b <= f (s_b + 8 sqrt(s_b) + 32) + (nwg + 1) B with f = n / ns and sum(s_b) = ns, and
// sum(sqrt(s_b)) <= sqrt(nbins ns) whatever the distribution; plus 2 % and the scratch area of one tile.
uint64_t pt_sort_pool_records(uint32_t n, uint32_t nbins, uint32_t nwg, size_t rec_size) {
  if (!n || !nbins) return 0;
  const double f = 64.0, ns = std::max(1.0, (double)n / f);
  const double regions = (double)n + f * (8.0 * std::sqrt((double)nbins * ns) + 32.0 * nbins) + (double)nbins * ((double)nwg + 1.0) * POOL_B;
  const uint64_t tile1 = (uint64_t)1024 * (rec_size == 16 ? 8 : 4);
  const uint64_t total = (uint64_t)(regions * 1.02) + 4096 + tile1;
  return total < 0xFFFFFF00ull ? total : 0;                    // (positions are 32-bit)
}
// The pooled pass 2's capacity (records of the pass-2 output): regions e_b + 6 sqrt(e_b) + 48 over all blocks, with sum(e_b) = n_in (what
// pass 1 left, sentinels included) and sum(sqrt(e_b)) <= sqrt(nblocks n_in); plus 1 % and the scratch area of one tile.  0 = too large.
uint64_t pt_sort_pool2_records(uint64_t n_in, uint32_t nblocks, size_t rec_size) {
  if (!n_in || !nblocks) return 0;
  const double regions = (double)n_in + 6.0 * std::sqrt((double)nblocks * (double)n_in) + 49.0 * (double)nblocks;
  const uint64_t total = (uint64_t)(regions * 1.01) + 4096 + (uint64_t)pt_sort_tile_points(rec_size);
  return total < 0xFFFFFF00ull ? total : 0;
}
uint32_t pt_sort_num_chunks(uint32_t n, size_t rec_size) {
  const uint32_t tile = (uint32_t)pt_sort_tile_points(rec_size);
  const uint32_t ntiles = (n + tile - 1) / tile, ct = (uint32_t)pt_sort_chunk_tiles(n, rec_size);
  return (ntiles + ct - 1) / ct;
}

void pt_launch_bbox_init(uint64_t* out6, hipStream_t s) { hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, s, out6); }
template <class T>
void pt_launch_bbox(const T* x, const T* y, const T* z, uint32_t n, uint64_t* out6, hipStream_t s) {
  if (!n) return;
  const uint32_t g = (uint32_t)std::min<uint64_t>(((uint64_t)n + WG - 1) / WG, 1024);
  hipLaunchKernelGGL(bbox_kernel<T>, dim3(g), dim3(WG), 0, s, x, y, z, n, out6);
}
template <class T>
void pt_launch_bbox_sample(const T* x, const T* y, const T* z, uint32_t n, uint32_t stride, uint64_t* out6, hipStream_t s) {
  if (!n) return;
  const uint64_t span = (uint64_t)stride * WG;                      // one run of WG points per `span` points
  const uint32_t runs = (uint32_t)((n + span - 1) / span);
  hipLaunchKernelGGL(bbox_sample_kernel<T>, dim3(std::min<uint32_t>(runs, 256u)), dim3(WG), 0, s, x, y, z, n, stride, out6);
}
template void pt_launch_bbox_sample<__half>(const __half*, const __half*, const __half*, uint32_t, uint32_t, uint64_t*, hipStream_t);
template void pt_launch_bbox<__half>(const __half*, const __half*, const __half*, uint32_t, uint64_t*, hipStream_t);
template void pt_launch_bbox_sample<float>(const float*, const float*, const float*, uint32_t, uint32_t, uint64_t*, hipStream_t);
template void pt_launch_bbox_sample<double>(const double*, const double*, const double*, uint32_t, uint32_t, uint64_t*, hipStream_t);
template void pt_launch_bbox<float>(const float*, const float*, const float*, uint32_t, uint64_t*, hipStream_t);
template void pt_launch_bbox<double>(const double*, const double*, const double*, uint32_t, uint64_t*, hipStream_t);
double pt_bbox_decode(uint64_t enc) {
  const uint64_t b = (enc >> 63) ? (enc & 0x7FFFFFFFFFFFFFFFull) : ~enc;
  double d;
  memcpy(&d, &b, 8);
  return d;
}

void pt_launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tmp, hipStream_t s) {
  if (!n) return;
  const uint32_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nt), dim3(WG), 0, s, in, n, tmp);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(WG), 0, s, tmp, nt);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nt), dim3(WG), 0, s, in, out, n, tmp);
}

template <class T, class Rec>
const Rec* pt_launch_grid_sort(const GridParams& gp, const T* x, const T* y, const T* z, const uint32_t* gidx, uint32_t n, Rec* out_final,
                               Rec* tmp, uint32_t* cell_start, const SortTables& tb, bool do_finalize, hipStream_t s, uint64_t* bbox6_verify) {
  constexpr int SW = 512;                              // scatter workgroup: 8 waves, 64 KB of staged records, 2 per CU
  constexpr int ITEMS_S = items_for<Rec>();            // records per scatter thread (8 / 4)
  constexpr int ITEMS = ITEMS_S * (SW / WG);           // records per histogram thread (same tile, 256 threads)
  constexpr uint32_t TILE = SW * ITEMS_S;
  const uint32_t nblocks = (uint32_t)gp.nblocks;
  const uint32_t nmacro = nblocks / PT_MACRO_BLOCKS;
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  PlanarLoader<T> pl{x, y, z, gidx};
  // every HIP call of the sort is checked: the first failure is kept, reported through tb.status and turns the result into nullptr
  hipError_t first = hipSuccess;
  auto ck = [&](hipError_t e) { if (first == hipSuccess && e != hipSuccess) first = e; };
  auto done = [&](const Rec* r) -> const Rec* { ck(hipGetLastError()); if (tb.status) *tb.status = first; return first == hipSuccess ? r : nullptr; };
  auto mark = [&](int i) { if (tb.ev) ck(hipEventRecord(tb.ev[i], s)); ck(hipGetLastError()); };     // (also collects launch errors of the phase just queued)
  mark(0);
  ck(hipMemsetAsync(tb.block_count, 0, sizeof(uint32_t) * ((size_t)nblocks + 1), s));
  hipLaunchKernelGGL(single_segment_kernel, dim3(1), dim3(64), 0, s, n, TILE, tb.seg_start1, tb.tile_first1);

  const Rec* blocked = nullptr;   // records partitioned by block id
  if (nblocks <= PT_MAXBINS) {
    // one level: bins = blocks.  planar -> tmp (by block) -> out_final (by cell)
    static_assert(PT_MAXBINS == 1024, "one-level bin mask assumes 10 bits");
    const BinSpec bs{1, 10, (int)nblocks};
    const int tpw = 4;
    if (n) {
      hipLaunchKernelGGL((hist_kernel<PlanarLoader<T>, ITEMS>), dim3((ntiles + tpw - 1) / tpw), dim3(WG), 0, s, pl, gp, bs,
                         tb.seg_start1, tb.tile_first1, 1, tb.block_count, tpw);
    }
    pt_launch_scan_u32(tb.block_count, tb.block_start, nblocks + 1, tb.scan_tmp, s);
    ck(hipMemcpyAsync(tb.cursor2, tb.block_start, sizeof(uint32_t) * nblocks, hipMemcpyDeviceToDevice, s));
    mark(1); mark(2); mark(3);
    if (n)
      hipLaunchKernelGGL((scatter_kernel<PlanarLoader<T>, ITEMS_S, SW>), dim3(ntiles), dim3(SW), 0, s, pl, tmp, gp, bs, tb.seg_start1,
                         tb.tile_first1, 1, tb.cursor2);
    mark(4);
    blocked = tmp;
    if (do_finalize) hipLaunchKernelGGL(finalize_kernel<Rec>, dim3(nblocks), dim3(FWG), 0, s, blocked, out_final, gp, tb.block_start, cell_start, tb.occupied, tb.shadow32);
    mark(5);
    return done(do_finalize ? out_final : tmp);
  }
  const BinSpec b2{1, 9, PT_MACRO_BLOCKS};
  if (const int gsh = pt_sort_group_shift(nblocks)) {
    // three levels (more than PT_MAXBINS macro blocks: fine grids over clouds with strong density contrast):
    // planar -> tmp (by GROUP of 2^gsh macro blocks) -> out_final (by macro block) -> tmp (by block) -> out_final (by cell).
    // The two later passes take their histograms from the records themselves (no 2-byte ids beside them).
    const uint32_t ngrp = (nmacro + (1u << gsh) - 1u) >> gsh, nmacP = ngrp << gsh;
    const BinSpec bg{0, 9 + gsh, (int)ngrp}, bm{2, gsh, 1 << gsh};
    const int chunk_tiles = pt_sort_chunk_tiles(n, sizeof(Rec));
    const uint32_t nchunks = n ? (ntiles + chunk_tiles - 1) / chunk_tiles : 0;
    const uint32_t ngroups = (nchunks + COL_GROUP - 1) / COL_GROUP;
    const int tpw = 4;
    ck(hipMemsetAsync(tb.counts1, 0, sizeof(uint32_t) * (PT_MAXBINS + 1), s));
    ck(hipMemsetAsync(tb.countsM, 0, sizeof(uint32_t) * ((size_t)nmacP + 1), s));
    if (n) {
      hipLaunchKernelGGL((hist_chunk_kernel<PlanarLoader<T>, ITEMS>), dim3(nchunks), dim3(WG), 0, s, pl, gp, bg, n, chunk_tiles, tb.chunk_hist, bbox6_verify);
      hipLaunchKernelGGL(colsum_kernel, dim3(ngroups, (ngrp + COL_WG - 1) / COL_WG), dim3(COL_WG), 0, s, tb.chunk_hist, (int)nchunks, (int)ngrp, tb.chunk_gsum);
      hipLaunchKernelGGL(colscan_kernel, dim3((ngrp + COL_WG - 1) / COL_WG), dim3(COL_WG), 0, s, tb.chunk_gsum, (int)ngroups, (int)ngrp, tb.counts1);
    }
    hipLaunchKernelGGL(seg_setup_kernel, dim3(1), dim3(WG), 0, s, tb.counts1, (int)ngrp, n, TILE, tb.start1, tb.cursor1, tb.tile_first2);
    mark(1);
    if (n) {
      hipLaunchKernelGGL(colapply_kernel, dim3(ngroups, (ngrp + COL_WG - 1) / COL_WG), dim3(COL_WG), 0, s, tb.chunk_hist, (int)nchunks, (int)ngrp, tb.chunk_gsum, tb.start1);
      if (chunk_tiles % 2 == 0)
        hipLaunchKernelGGL((scatter_chunk_kernel<PlanarLoader<T>, ITEMS_S, 2 * SW>), dim3(nchunks), dim3(2 * SW), 0, s, pl, tmp, gp, bg, n, chunk_tiles / 2, tb.chunk_hist,
                           (uint16_t*)nullptr);
      else
        hipLaunchKernelGGL((scatter_chunk_kernel<PlanarLoader<T>, ITEMS_S, SW>), dim3(nchunks), dim3(SW), 0, s, pl, tmp, gp, bg, n, chunk_tiles, tb.chunk_hist,
                           (uint16_t*)nullptr);
    }
    mark(2);
    RecLoader<Rec> rg{tmp};
    const uint32_t ntilesG = ntiles + ngrp;
    if (n) hipLaunchKernelGGL((hist_kernel<RecLoader<Rec>, ITEMS>), dim3((ntilesG + tpw - 1) / tpw), dim3(WG), 0, s, rg, gp, bm, tb.start1, tb.tile_first2, (int)ngrp, tb.countsM, tpw);
    hipLaunchKernelGGL(seg_setup_big_kernel, dim3(1), dim3(WG), 0, s, tb.countsM, (int)nmacP, n, TILE, tb.startM, tb.cursorM, tb.tile_firstM);
    if (n) hipLaunchKernelGGL((scatter_kernel<RecLoader<Rec>, ITEMS_S, SW>), dim3(ntilesG), dim3(SW), 0, s, rg, out_final, gp, bm, tb.start1, tb.tile_first2, (int)ngrp, tb.cursorM);
    RecLoader<Rec> rm{out_final};
    const uint32_t ntilesM = ntiles + nmacP;
    if (n) hipLaunchKernelGGL((hist_kernel<RecLoader<Rec>, ITEMS>), dim3((ntilesM + tpw - 1) / tpw), dim3(WG), 0, s, rm, gp, b2, tb.startM, tb.tile_firstM, (int)nmacP, tb.block_count, tpw);
    pt_launch_scan_u32(tb.block_count, tb.block_start, nblocks + 1, tb.scan_tmp, s);
    ck(hipMemcpyAsync(tb.cursor2, tb.block_start, sizeof(uint32_t) * nblocks, hipMemcpyDeviceToDevice, s));
    mark(3);
    if (n) hipLaunchKernelGGL((scatter_kernel<RecLoader<Rec>, ITEMS_S, SW>), dim3(ntilesM), dim3(SW), 0, s, rm, tmp, gp, b2, tb.startM, tb.tile_firstM, (int)nmacP, tb.cursor2);
    mark(4);
    if (do_finalize) hipLaunchKernelGGL(finalize_kernel<Rec>, dim3(nblocks), dim3(FWG), 0, s, (const Rec*)tmp, out_final, gp, tb.block_start, cell_start, tb.occupied, tb.shadow32);
    mark(5);
    return done(do_finalize ? out_final : tmp);
  }
  // two levels: planar -> out_final (by macro block) -> tmp (by block) -> out_final (by cell)
  const BinSpec b1{0, 9, (int)nmacro};
  // pass 2 + finalize without the pass-2 histogram (pool2_sizes_kernel): regions from the macro counts, counts scanned after the scatter
  auto pooled_pass2 = [&](uint32_t ntiles2, const uint32_t* seg_end) {
    RecLoader<Rec> rl{out_final};
    const uint32_t scratch2 = (uint32_t)(tb.pool2_records - TILE);
    OccBox ob;
    for (int a = 0; a < 3; ++a) { ob.lo[a] = tb.occ_lo[a]; ob.hi[a] = tb.occ_hi[a]; }
    hipLaunchKernelGGL(pool2_sizes_kernel, dim3((nblocks + 1 + WG - 1) / WG), dim3(WG), 0, s, gp, ob, tb.counts1, nblocks, tb.block_count);
    pt_launch_scan_u32(tb.block_count, tb.rstart, nblocks + 1, tb.scan_tmp, s);
    hipLaunchKernelGGL(pool2_check_kernel, dim3(1), dim3(WG), 0, s, tb.rstart, nblocks, scratch2, tb.pool_flag);
    ck(hipMemcpyAsync(tb.cursor2, tb.rstart, sizeof(uint32_t) * nblocks, hipMemcpyDeviceToDevice, s));
    mark(3);
    hipLaunchKernelGGL((scatter_kernel<RecLoader<Rec>, ITEMS_S, SW>), dim3(ntiles2), dim3(SW), 0, s, rl, tmp, gp, b2, tb.start1, tb.tile_first2,
                       (int)nmacro, tb.cursor2, seg_end, tb.rstart, tb.pool_flag, scratch2, (const uint32_t*)tb.pool_flag);
    mark(4);
    hipLaunchKernelGGL(pool2_counts_kernel, dim3((nblocks + 1 + WG - 1) / WG), dim3(WG), 0, s, tb.rstart, tb.cursor2, nblocks, tb.block_count);
    pt_launch_scan_u32(tb.block_count, tb.block_start, nblocks + 1, tb.scan_tmp, s);
    if (do_finalize) hipLaunchKernelGGL(finalize_kernel<Rec>, dim3(nblocks), dim3(FWG), 0, s, (const Rec*)tmp, out_final, gp, tb.block_start, cell_start, tb.occupied, tb.shadow32, tb.rstart, (const uint32_t*)tb.pool_flag);
    mark(5);
  };
  const bool pool2 = tb.pool2_records && n && do_finalize;
  if (tb.pool_records && n) {
    // pass 1 without its histogram pass: regions from a sample, blocks, sentinels (scatter_pool_kernel); pass 2 reads every bin from its
    // region's start to where pass 1 stopped (seg_end) and drops the sentinels
    constexpr uint32_t TILE1 = 1024 * ITEMS_S;
    const PoolTables pt{tb.pool_est, tb.start1, tb.cursor1, tb.pool_limit, tb.pool_flag};
    ck(hipMemsetAsync(tb.pool_est, 0, sizeof(uint32_t) * PT_MAXBINS, s));
    ck(hipMemsetAsync(tb.pool_flag, 0, sizeof(uint32_t) * 4, s));
    const uint32_t stride = 64;                                       // one run of 256 points in 64: what pt_sort_pool_records assumes
    const uint64_t runs = ((uint64_t)n + POOL_SAMPLE_RUN - 1) / POOL_SAMPLE_RUN;
    const uint32_t gs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((runs + stride - 1) / stride, 1), 1024);
    hipLaunchKernelGGL((pool_sample_kernel<PlanarLoader<T>>), dim3(gs), dim3(WG), 0, s, pl, gp, b1, n, stride, pt);
    const uint32_t scratch = (uint32_t)(tb.pool_records - TILE1);
    hipLaunchKernelGGL(pool_setup_kernel, dim3(1), dim3(WG), 0, s, pt, (int)nmacro, n, tb.pool_nwg, scratch);
    mark(1);
    const uint32_t nt1 = (uint32_t)(((uint64_t)n + TILE1 - 1) / TILE1);
    hipLaunchKernelGGL((scatter_pool_kernel<PlanarLoader<T>, ITEMS_S, 1024>), dim3(std::min(tb.pool_nwg, nt1)), dim3(1024), 0, s, pl, out_final, gp, b1, n, pt,
                       scratch, pool2 ? (uint16_t*)nullptr : tb.bid, bbox6_verify);
    hipLaunchKernelGGL(pool_finish_kernel, dim3(1), dim3(WG), 0, s, pt, (int)nmacro, TILE, tb.counts1, tb.tile_first2);
    mark(2);
    RecLoader<Rec> rl{out_final};
    const uint32_t ntiles2 = (uint32_t)(((uint64_t)n + (uint64_t)nmacro * tb.pool_nwg * POOL_B) / TILE) + nmacro + 1;
    if (pool2) { pooled_pass2(ntiles2, tb.cursor1); return done(out_final); }
    const int tpw = 4;
    hipLaunchKernelGGL((hist_bid_kernel<ITEMS>), dim3((ntiles2 + tpw - 1) / tpw), dim3(WG), 0, s, tb.bid, (int)PT_MACRO_BLOCKS, tb.start1,
                       tb.tile_first2, (int)nmacro, tb.block_count, tpw, tb.cursor1, (const uint32_t*)tb.pool_flag);
    pt_launch_scan_u32(tb.block_count, tb.block_start, nblocks + 1, tb.scan_tmp, s);
    ck(hipMemcpyAsync(tb.cursor2, tb.block_start, sizeof(uint32_t) * nblocks, hipMemcpyDeviceToDevice, s));
    mark(3);
    hipLaunchKernelGGL((scatter_kernel<RecLoader<Rec>, ITEMS_S, SW>), dim3(ntiles2), dim3(SW), 0, s, rl, tmp, gp, b2, tb.start1, tb.tile_first2,
                       (int)nmacro, tb.cursor2, tb.cursor1, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (const uint32_t*)tb.pool_flag);
    mark(4);
    if (do_finalize) hipLaunchKernelGGL(finalize_kernel<Rec>, dim3(nblocks), dim3(FWG), 0, s, (const Rec*)tmp, out_final, gp, tb.block_start, cell_start, tb.occupied, tb.shadow32,
                                        (const uint32_t*)nullptr, (const uint32_t*)tb.pool_flag);
    mark(5);
    return done(do_finalize ? out_final : tmp);
  }
  const int chunk_tiles = pt_sort_chunk_tiles(n, sizeof(Rec));
  const uint32_t nchunks = n ? (ntiles + chunk_tiles - 1) / chunk_tiles : 0;
  const uint32_t ngroups = (nchunks + COL_GROUP - 1) / COL_GROUP;
  ck(hipMemsetAsync(tb.counts1, 0, sizeof(uint32_t) * (PT_MAXBINS + 1), s));
  if (n) {
    hipLaunchKernelGGL((hist_chunk_kernel<PlanarLoader<T>, ITEMS>), dim3(nchunks), dim3(WG), 0, s, pl, gp, b1, n, chunk_tiles, tb.chunk_hist,
                       bbox6_verify);
    hipLaunchKernelGGL(colsum_kernel, dim3(ngroups, (nmacro + COL_WG - 1) / COL_WG), dim3(COL_WG), 0, s, tb.chunk_hist, (int)nchunks, (int)nmacro, tb.chunk_gsum);
    hipLaunchKernelGGL(colscan_kernel, dim3((nmacro + COL_WG - 1) / COL_WG), dim3(COL_WG), 0, s, tb.chunk_gsum, (int)ngroups, (int)nmacro, tb.counts1);
  }
  hipLaunchKernelGGL(seg_setup_kernel, dim3(1), dim3(WG), 0, s, tb.counts1, (int)nmacro, n, TILE, tb.start1, tb.cursor1, tb.tile_first2);
  mark(1);
  if (n) {
    hipLaunchKernelGGL(colapply_kernel, dim3(ngroups, (nmacro + COL_WG - 1) / COL_WG), dim3(COL_WG), 0, s, tb.chunk_hist, (int)nchunks, (int)nmacro, tb.chunk_gsum, tb.start1);
    if (chunk_tiles % 2 == 0)     // big clouds: 1024-thread workgroups, tiles twice as long -> twice the bytes per bin and tile
      hipLaunchKernelGGL((scatter_chunk_kernel<PlanarLoader<T>, ITEMS_S, 2 * SW>), dim3(nchunks), dim3(2 * SW), 0, s, pl, out_final, gp, b1, n,
                         chunk_tiles / 2, tb.chunk_hist, pool2 ? (uint16_t*)nullptr : tb.bid);
    else
      hipLaunchKernelGGL((scatter_chunk_kernel<PlanarLoader<T>, ITEMS_S, SW>), dim3(nchunks), dim3(SW), 0, s, pl, out_final, gp, b1, n, chunk_tiles,
                         tb.chunk_hist, pool2 ? (uint16_t*)nullptr : tb.bid);
  }
  mark(2);
  RecLoader<Rec> rl{out_final};
  const uint32_t ntiles2 = ntiles + nmacro;   // upper bound: every segment adds at most one partial tile
  if (pool2) {
    ck(hipMemsetAsync(tb.pool_flag, 0, sizeof(uint32_t) * 4, s));
    pooled_pass2(ntiles2, nullptr);
    return done(out_final);
  }
  if (n) {
    const int tpw = 4;
    hipLaunchKernelGGL((hist_bid_kernel<ITEMS>), dim3((ntiles2 + tpw - 1) / tpw), dim3(WG), 0, s, tb.bid, (int)PT_MACRO_BLOCKS, tb.start1,
                       tb.tile_first2, (int)nmacro, tb.block_count, tpw);
  }
  pt_launch_scan_u32(tb.block_count, tb.block_start, nblocks + 1, tb.scan_tmp, s);
  ck(hipMemcpyAsync(tb.cursor2, tb.block_start, sizeof(uint32_t) * nblocks, hipMemcpyDeviceToDevice, s));
  mark(3);
  if (n)
    hipLaunchKernelGGL((scatter_kernel<RecLoader<Rec>, ITEMS_S, SW>), dim3(ntiles2), dim3(SW), 0, s, rl, tmp, gp, b2, tb.start1, tb.tile_first2,
                       (int)nmacro, tb.cursor2);
  mark(4);
  if (do_finalize) hipLaunchKernelGGL(finalize_kernel<Rec>, dim3(nblocks), dim3(FWG), 0, s, (const Rec*)tmp, out_final, gp, tb.block_start, cell_start, tb.occupied, tb.shadow32);
  mark(5);
  return done(do_finalize ? out_final : tmp);
}
template const RecF* pt_launch_grid_sort<float, RecF>(const GridParams&, const float*, const float*, const float*, const uint32_t*, uint32_t, RecF*,
                                                      RecF*, uint32_t*, const SortTables&, bool, hipStream_t, uint64_t*);
template const RecF* pt_launch_grid_sort<__half, RecF>(const GridParams&, const __half*, const __half*, const __half*, const uint32_t*, uint32_t, RecF*,
                                                      RecF*, uint32_t*, const SortTables&, bool, hipStream_t, uint64_t*);
template const RecD* pt_launch_grid_sort<double, RecD>(const GridParams&, const double*, const double*, const double*, const uint32_t*, uint32_t,
                                                       RecD*, RecD*, uint32_t*, const SortTables&, bool, hipStream_t, uint64_t*);
template <class T>
void pt_launch_uniform_probe(const GridParams& gp, const T* x, const T* y, const T* z, uint32_t n, const int occ_lo[3], const int occ_hi[3],
                             uint32_t* scratch, uint32_t* flag, hipStream_t s) {
  const uint32_t nblocks = (uint32_t)gp.nblocks, nmacro = nblocks / PT_MACRO_BLOCKS;
  uint32_t* blk_cnt = scratch;
  uint32_t* macro_cnt = scratch + nblocks;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(scratch + pt_uniform_probe_acc_offset(nblocks));
  uint32_t* cell_bits = scratch + pt_uniform_probe_acc_offset(nblocks) + 8;      // 16 words per block, 16-byte aligned (the offset is a multiple of four words, the scratch an allocation of its own)
  uint32_t* macro_int = cell_bits + (size_t)nblocks * 16;                         // per macro block: the sample points of its interior blocks, and how many those are
  uint32_t* macro_nint = macro_int + nmacro;
  (void)hipMemsetAsync(scratch, 0, sizeof(uint32_t) * ((size_t)pt_uniform_probe_acc_offset(nblocks) + 8 + (size_t)nblocks * 16 + 2 * (size_t)nmacro), s);
  (void)hipMemsetAsync(flag, 0, sizeof(uint32_t), s);
  if (!n) return;
  PlanarLoader<T> pl{x, y, z, nullptr};
  // about 4 M sample points whatever the cloud's size (one run of 256 consecutive points in `stride` runs): the scattered atomics of a
  // 1/64 sample of 1e9 points took 1.4 ms, and eight sample points per block still put a +-15 % density drift tens of sigmas out
  const uint32_t stride = std::max<uint32_t>(16u, std::min<uint32_t>(256u, n >> 22));
  const uint64_t runs = ((uint64_t)n + POOL_SAMPLE_RUN - 1) / POOL_SAMPLE_RUN;
  const uint32_t gs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((runs + stride - 1) / stride, 1), 4096);
  hipLaunchKernelGGL((uniform_probe_kernel<PlanarLoader<T>>), dim3(gs), dim3(WG), 0, s, pl, gp, n, stride, blk_cnt, cell_bits, acc);
  static_assert(PT_MACRO_BLOCKS == 2 * WG, "two blocks per thread in the macro sums");
  OccBox ob;
  for (int a = 0; a < 3; ++a) { ob.lo[a] = occ_lo[a]; ob.hi[a] = occ_hi[a]; }
  hipLaunchKernelGGL(uniform_macro_kernel, dim3(nmacro), dim3(WG), 0, s, gp, ob, blk_cnt, macro_cnt, macro_int, macro_nint);
  hipLaunchKernelGGL(uniform_check_kernel, dim3((nblocks + WG - 1) / WG), dim3(WG), 0, s, gp, ob, macro_cnt, macro_int, macro_nint, blk_cnt, cell_bits, nblocks, stride, flag, acc);
}
template void pt_launch_uniform_probe<float>(const GridParams&, const float*, const float*, const float*, uint32_t, const int*, const int*, uint32_t*, uint32_t*, hipStream_t);
template void pt_launch_uniform_probe<__half>(const GridParams&, const __half*, const __half*, const __half*, uint32_t, const int*, const int*, uint32_t*, uint32_t*, hipStream_t);
template void pt_launch_uniform_probe<double>(const GridParams&, const double*, const double*, const double*, uint32_t, const int*, const int*, uint32_t*, uint32_t*, hipStream_t);

void pt_launch_sum_u32(const uint32_t* v, uint32_t n, uint32_t* out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(sum_u32_kernel, dim3(std::min<uint32_t>((n + WG - 1) / WG, 64)), dim3(WG), 0, s, v, n, out);
}
