// pt_query.hip -- exact k-nearest-neighbour search on the cell grid, for gfx950 (MI355X).
//
// Replaces the reference's query loop
//     K_neighbor_search search(tree, vertices[...], K);  for (it = search.begin(); ...)
// (reference src/pointsTransfer.cpp:462-479; CGAL Orthogonal_k_neighbor_search, eps = 0, results ascending)
// with one launch over all targets.  Metric: reference src/Distance.h:6-11, evaluated in fp64 as
// (dx*dx + dy*dy) + dz*dz with contraction off -- the same 3 mul + 2 add the reference's flags produce.
// Cell pruning is Distance::min_distance_to_rectangle (reference src/Distance.h:27-57) applied to cell
// boxes; ring termination is the same bound applied to the faces of the box already scanned.
//
// Mapping onto the wave: 8 lanes per target, 8 targets per wave64, 32 per 256-thread workgroup.
//   - a target's candidate cells are x-runs of cells (contiguous in the sorted records), so the 8 lanes of a
//     group read 8 consecutive 16-B records = one 128-B line per step;
//   - the running top-k lives in registers, distributed over the group's lanes (lane L holds ranks
//     [L*KPL, (L+1)*KPL), KPL = ceil(k/8)), ordered by the total order (d2, original index);
//   - a candidate is offered with one group ballot; an insertion is a one-position shift across lanes.
// Control flow is uniform inside a group (all 8 lanes take every branch together), so cross-lane operations
// never see an inactive partner; different groups of a wave diverge freely.
#include "pt_internal.h"

namespace {

constexpr int WG = 256;
constexpr int GL = 8;   // lanes per target

__device__ inline bool key_lt(double ad, uint32_t ai, double bd, uint32_t bi) { return ad < bd || (ad == bd && ai < bi); }

template <class Rec>
__device__ inline double dist2(const double (&q)[3], const Rec& r) {
#pragma clang fp contract(off)
  const double dx = q[0] - (double)r.x;
  const double dy = q[1] - (double)r.y;
  const double dz = q[2] - (double)r.z;
  return (dx * dx + dy * dy) + dz * dz;     // reference src/Distance.h:10, left to right, unfused
}

template <class Rec, int KPL>
struct GroupSearch {
  const GridParams& gp;
  const Rec* __restrict__ src;
  const uint32_t* __restrict__ cs;
  double q[3], u[3];
  int c[3];
  double ld[KPL];          // this lane's slice of the sorted top list
  uint32_t li[KPL];
  double lim_d, bnd_d;     // acceptance limit = min(k-th entry, caller's bound); both with index NOIDX when unset
  uint32_t lim_i;
  double h2;
  int L, gshift, hl, hr;

  __device__ GroupSearch(const GridParams& g, const Rec* s, const uint32_t* cell_start) : gp(g), src(s), cs(cell_start) {}

  // distance (cell units, >= 0) from the target to the cell interval [lo, hi] along axis a, minus the slack
  __device__ double gap(int a, int lo, int hi) const {
    const double g = fmax((double)lo - u[a], u[a] - (double)(hi + 1)) - PT_CELL_EPS;
    return fmax(g, 0.0);
  }

  __device__ void insert(double xd, uint32_t xi) {
    bool cj[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) cj[j] = key_lt(xd, xi, ld[j], li[j]);
    // the lane below hands over its last entry if the new key sorts before it
    const double pd = __shfl_up(ld[KPL - 1], 1, GL);
    const uint32_t pi = __shfl_up(li[KPL - 1], 1, GL);
    int pc = __shfl_up((int)cj[KPL - 1], 1, GL);
    if (L == 0) pc = 0;
#pragma unroll
    for (int j = KPL - 1; j >= 1; --j) {
      if (cj[j - 1]) { ld[j] = ld[j - 1]; li[j] = li[j - 1]; }
      else if (cj[j]) { ld[j] = xd; li[j] = xi; }
    }
    if (pc) { ld[0] = pd; li[0] = pi; }
    else if (cj[0]) { ld[0] = xd; li[0] = xi; }
    // new acceptance limit: the entry of rank k-1, unless the caller's bound is tighter
    double kd = ld[0];
    uint32_t ki = li[0];
#pragma unroll
    for (int j = 1; j < KPL; ++j) if (hr == j) { kd = ld[j]; ki = li[j]; }
    kd = __shfl(kd, hl, GL);
    ki = __shfl(ki, hl, GL);
    if (key_lt(kd, ki, bnd_d, PT_NOIDX_U)) { lim_d = kd; lim_i = ki; }
    else { lim_d = bnd_d; lim_i = PT_NOIDX_U; }
  }

  // offer the records [s, e) of the sorted cloud
  __device__ void scan_range(uint32_t s, uint32_t e) {
    for (uint32_t base = s; base < e; base += GL) {
      const uint32_t p = base + L;
      double d = INFINITY;
      uint32_t id = PT_NOIDX_U;
      if (p < e) {
        const Rec r = src[p];
        d = dist2(q, r);
        id = r.id;
      }
      const bool pass = key_lt(d, id, lim_d, lim_i);
      uint32_t mask = (uint32_t)(__ballot(pass) >> gshift) & 0xFFu;
      while (mask) {
        const int t = __ffs(mask) - 1;
        mask &= mask - 1;
        const double xd = __shfl(d, t, GL);
        const uint32_t xi = __shfl(id, t, GL);
        if (key_lt(xd, xi, lim_d, lim_i)) insert(xd, xi);   // re-test: the limit may have tightened this step
      }
    }
  }

  // cells [xa, xb] x {y} x {z} (inside the grid); prunes by the box lower bound, then walks the run block by block
  __device__ void scan_row(int xa, int xb, int y, int z) {
    const double gy = gap(1, y, y), gz = gap(2, z, z);
    const double s2 = gy * gy + gz * gz;
    if (s2 * h2 > lim_d) return;
    while (xa < xb) { const double g = gap(0, xa, xa); if ((g * g + s2) * h2 > lim_d) ++xa; else break; }
    while (xb > xa) { const double g = gap(0, xb, xb); if ((g * g + s2) * h2 > lim_d) --xb; else break; }
    { const double g = gap(0, xa, xb); if ((g * g + s2) * h2 > lim_d) return; }
    for (int bx = xa >> 3; bx <= (xb >> 3); ++bx) {
      const int pa = max(xa, bx << 3), pb = min(xb, (bx << 3) + 7);
      const uint32_t key = (pt_block_id(gp.mdim, pa, y, z) << 9) + pt_local_cell(pa, y, z);
      const uint32_t s = cs[key], e = cs[key + (uint32_t)(pb - pa) + 1u];
      scan_range(s, e);
    }
  }
};

template <class Rec, int KPL>
__global__ __launch_bounds__(WG) void knn_kernel(GridParams gp, const Rec* __restrict__ src, const uint32_t* __restrict__ cell_start,
                                                 const Rec* __restrict__ tgt, uint32_t m, int k, const double* __restrict__ bound2,
                                                 uint32_t* __restrict__ out_idx, double* __restrict__ out_d2) {
  const uint32_t gid = (blockIdx.x * WG + threadIdx.x) / GL;
  if (gid >= m) return;                       // whole groups leave together
  GroupSearch<Rec, KPL> S(gp, src, cell_start);
  S.L = threadIdx.x & (GL - 1);
  S.gshift = (threadIdx.x & 63) & ~(GL - 1);
  S.hl = (k - 1) / KPL;
  S.hr = (k - 1) % KPL;
  S.h2 = gp.h * gp.h;
  const Rec T = tgt[gid];
  S.q[0] = (double)T.x; S.q[1] = (double)T.y; S.q[2] = (double)T.z;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    S.u[a] = (S.q[a] - gp.bbmin[a]) * gp.inv_h;
    S.c[a] = (int)fmin(fmax(S.u[a], 0.0), (double)(gp.dim[a] - 1));
  }
#pragma unroll
  for (int j = 0; j < KPL; ++j) { S.ld[j] = INFINITY; S.li[j] = PT_NOIDX_U; }
  S.bnd_d = bound2 ? bound2[T.id] : INFINITY;
  S.lim_d = S.bnd_d;
  S.lim_i = PT_NOIDX_U;

  const int c0 = S.c[0], c1 = S.c[1], c2 = S.c[2];
  for (int r = 1;; ++r) {
    const int x0 = max(c0 - r, 0), x1 = min(c0 + r, gp.dim[0] - 1);
    const int y0 = max(c1 - r, 0), y1 = min(c1 + r, gp.dim[1] - 1);
    const int z0 = max(c2 - r, 0), z1 = min(c2 + r, gp.dim[2] - 1);
    if (r == 1) {
      // the 3x3x3 box, centre row first so that the limit tightens before the outer rows are tested
      // (dy,dz)+1 packed 2 bits each: dy = 0,-1,1,0,0,-1,1,-1,1 ; dz = 0,0,0,-1,1,-1,-1,1,1
      constexpr uint32_t OY = 139617u, OZ = 164373u;
#pragma unroll 1
      for (int i = 0; i < 9; ++i) {
        const int y = c1 + (int)((OY >> (2 * i)) & 3u) - 1, z = c2 + (int)((OZ >> (2 * i)) & 3u) - 1;
        if (y >= y0 && y <= y1 && z >= z0 && z <= z1) S.scan_row(x0, x1, y, z);
      }
    } else {
      // the shell box(r) \ box(r-1)
      for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
          const bool shell = (z == c2 - r) || (z == c2 + r) || (y == c1 - r) || (y == c1 + r);
          if (shell) S.scan_row(x0, x1, y, z);
          else {
            if (c0 - r >= 0) S.scan_row(c0 - r, c0 - r, y, z);
            if (c0 + r <= gp.dim[0] - 1) S.scan_row(c0 + r, c0 + r, y, z);
          }
        }
    }
    // stop when the box covers the grid, or when nothing outside it can beat the limit:
    // every unscanned point lies beyond one of the box faces that still has cells behind it
    bool covered = true;
    double dout = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int lo = S.c[a] - r, hi = S.c[a] + r;
      if (lo > 0) { covered = false; dout = fmin(dout, S.u[a] - (double)lo); }
      if (hi < gp.dim[a] - 1) { covered = false; dout = fmin(dout, (double)(hi + 1) - S.u[a]); }
    }
    if (covered) break;
    dout = fmax(dout - PT_CELL_EPS, 0.0);
    if (dout * dout * S.h2 > S.lim_d) break;
  }

  const size_t row = (size_t)T.id * (size_t)k;
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int e = S.L * KPL + j;
    if (e < k) {
      out_idx[row + e] = S.li[j];
      if (out_d2) out_d2[row + e] = S.ld[j];
    }
  }
}

// ---- G-way merge of candidate lists under (d2, idx): one thread per (target, list slot) --------------
__global__ __launch_bounds__(WG) void merge_kernel(const uint32_t* __restrict__ idx_lists, const double* __restrict__ d2_lists, int g,
                                                   uint32_t m, int k, uint32_t* __restrict__ idx_out, double* __restrict__ d2_out) {
  const int per = g * k;
  const uint64_t gt = (uint64_t)blockIdx.x * WG + threadIdx.x;
  const uint32_t t = (uint32_t)(gt / (uint64_t)per);
  if (t >= m) return;
  const int slot = (int)(gt % (uint64_t)per);
  const int mg = slot / k, mj = slot % k;
  const size_t mo = ((size_t)mg * m + t) * (size_t)k + (size_t)mj;
  const uint32_t myi = idx_lists[mo];
  const double myd = d2_lists[mo];
  int rank = 0, cnt = 0;
  for (int r = 0; r < g; ++r)
    for (int j = 0; j < k; ++j) {
      const size_t o = ((size_t)r * m + t) * (size_t)k + (size_t)j;
      const uint32_t oi = idx_lists[o];
      if (oi == PT_NOIDX_U) continue;
      ++cnt;
      const double od = d2_lists[o];
      if (key_lt(od, oi, myd, myi) || (od == myd && oi == myi && (r * k + j) < slot)) ++rank;
    }
  if (myi != PT_NOIDX_U && rank < k) { idx_out[(size_t)t * k + rank] = myi; d2_out[(size_t)t * k + rank] = myd; }
  if (slot < k && slot >= cnt) { idx_out[(size_t)t * k + slot] = PT_NOIDX_U; d2_out[(size_t)t * k + slot] = INFINITY; }
}

// ---- which other slabs can still hold one of a target's k nearest (reference Distance.h:27-57 on slab boxes)
template <class T>
__global__ __launch_bounds__(WG) void slab_need_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                                                       const double* __restrict__ d2, uint32_t m, int k, int axis,
                                                       const double* __restrict__ bounds, int g, int my_slab, uint8_t* __restrict__ need) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  const double c = (double)(axis == 0 ? x[t] : (axis == 1 ? y[t] : z[t]));
  const double kth = d2[(size_t)t * k + (k - 1)];
  for (int s = 0; s < g; ++s) {
    uint8_t v = 0;
    if (s != my_slab) {
      const double lo = bounds[s], hi = bounds[s + 1];
      const double gapd = c < lo ? lo - c : (c >= hi ? c - hi : 0.0);
      v = (gapd * gapd * (1.0 - 1e-12) <= kth) ? 1 : 0;     // `<=`: an equal-distance lower-index point would win the tie
    }
    need[(size_t)s * m + t] = v;
  }
}

}  // namespace

template <class Rec>
void pt_launch_knn(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const Rec* tgt, uint32_t m, int k, const double* bound2,
                   uint32_t* out_idx, double* out_d2, hipStream_t s) {
  if (!m) return;
  const uint32_t nwg = (uint32_t)(((uint64_t)m * GL + WG - 1) / WG);
  if (k <= 8)
    hipLaunchKernelGGL((knn_kernel<Rec, 1>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2);
  else if (k <= 16)
    hipLaunchKernelGGL((knn_kernel<Rec, 2>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2);
  else
    hipLaunchKernelGGL((knn_kernel<Rec, 4>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2);
}
template void pt_launch_knn<RecF>(const GridParams&, const RecF*, const uint32_t*, const RecF*, uint32_t, int, const double*, uint32_t*, double*,
                                  hipStream_t);
template void pt_launch_knn<RecD>(const GridParams&, const RecD*, const uint32_t*, const RecD*, uint32_t, int, const double*, uint32_t*, double*,
                                  hipStream_t);

void pt_launch_merge(const uint32_t* idx_lists, const double* d2_lists, int g, uint32_t m, int k, uint32_t* idx_out, double* d2_out,
                     hipStream_t s) {
  if (!m) return;
  const uint64_t threads = (uint64_t)m * (uint64_t)(g * k);
  hipLaunchKernelGGL(merge_kernel, dim3((uint32_t)((threads + WG - 1) / WG)), dim3(WG), 0, s, idx_lists, d2_lists, g, m, k, idx_out, d2_out);
}

template <class T>
void pt_launch_slab_need(const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g,
                         int my_slab, uint8_t* need, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(slab_need_kernel<T>, dim3((m + WG - 1) / WG), dim3(WG), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, my_slab, need);
}
template void pt_launch_slab_need<float>(const float*, const float*, const float*, const double*, uint32_t, int, int, const double*, int, int,
                                         uint8_t*, hipStream_t);
template void pt_launch_slab_need<double>(const double*, const double*, const double*, const double*, uint32_t, int, int, const double*, int,
                                          int, uint8_t*, hipStream_t);
