// pt_exchange.hip -- device side of the multi-GPU slab exchange ("v2": pruned, owner-to-owner).
//
// The reference is a single process (src/pointsTransfer.cpp:462-479 searches one tree); with the cloud cut into G slabs along
// one axis, a target homed in slab `me` needs slab s only if dist2(target, slab s) <= its current k-th squared distance --
// Distance::min_distance_to_rectangle (reference src/Distance.h:27-57) applied to the slab boxes, `<=` because an equal-distance
// point with a lower index would win the tie.  These kernels
//   * count, then bucket by destination slab, the request packets {x, y, z, current k-th d2} of the targets that cross;
//   * unpack received requests into the planar coordinates + bounds the radius-bounded search takes;
//   * merge the k candidates another slab returned for a row into that row's list under the total order (d2, index).
// The transport between the phases (RCCL all-gather of the count matrix, grouped ncclSend / ncclRecv of requests and answers --
// or plain device copies when G logical slabs share one process) lives in pt_api.hip.
#include "pt_internal.h"

namespace {

constexpr int XW = 256;

__device__ inline bool slab_needed(double c, double kth, double lo, double hi) {
  const double gapd = c < lo ? lo - c : (c >= hi ? c - hi : 0.0);
  return gapd * gapd * (1.0 - 1e-12) <= kth;          // same test as slab_need_kernel / request_pack_kernel (pt_query.hip)
}

template <class T, bool FILL>
__global__ __launch_bounds__(XW) void xreq_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const double* __restrict__ d2,
                                                  uint32_t m, int k, int axis, const double* __restrict__ bounds, int g, int me,
                                                  uint32_t* __restrict__ counts, const uint32_t* __restrict__ off, uint32_t* __restrict__ cursor,
                                                  double* __restrict__ req, uint32_t* __restrict__ req_row) {
  // One GLOBAL atomic per workgroup and destination, not per target (round 4; measured on config 4 as eight logical slabs: the 70 k crossing
  // targets of a rank -- one in ninety, spread evenly over the array, so a wave rarely holds two -- took their slots one by one from eight
  // counters: 0.92 ms for each of the two passes where the bytes they read cost 0.1).  A workgroup walks a contiguous piece of the targets,
  // counts per destination in LDS, reserves its ranges with one atomic each, and (FILL) walks the piece again handing out slots from LDS.
  __shared__ uint32_t cnt[64], base[64];
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t per = ((m + gridDim.x - 1) / gridDim.x + XW - 1) / XW * XW, t0 = blockIdx.x * per, t1 = min(m, t0 + per);
  for (uint32_t t = t0 + threadIdx.x; t < t1; t += XW) {
    const double c = (double)(axis == 0 ? x[t] : (axis == 1 ? y[t] : z[t]));
    const double kth = d2[(size_t)t * k + (k - 1)];
    for (int s = 0; s < g; ++s)
      if (s != me && slab_needed(c, kth, bounds[s], bounds[s + 1])) atomicAdd(&cnt[s], 1u);
  }
  __syncthreads();
  if constexpr (!FILL) {
    if (threadIdx.x < (uint32_t)g && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]);
  } else {
    if (threadIdx.x < (uint32_t)g) { base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0u; cnt[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += XW) {
      const double c = (double)(axis == 0 ? x[t] : (axis == 1 ? y[t] : z[t]));
      const double kth = d2[(size_t)t * k + (k - 1)];
      for (int s = 0; s < g; ++s) {
        if (s != me && slab_needed(c, kth, bounds[s], bounds[s + 1])) {
          const uint32_t pos = off[s] + base[s] + atomicAdd(&cnt[s], 1u);
          double* o = req + (size_t)pos * 4;
          o[0] = (double)x[t]; o[1] = (double)y[t]; o[2] = (double)z[t]; o[3] = kth;
          req_row[pos] = t;
        }
      }
    }
  }
}
template <class T>
__global__ __launch_bounds__(XW) void xunpack_kernel(const double* __restrict__ rreq, uint32_t r, T* __restrict__ xyz, double* __restrict__ bound) {
  const uint32_t i = blockIdx.x * XW + threadIdx.x;
  if (i >= r) return;
  const double* q = rreq + (size_t)i * 4;
  xyz[i] = (T)q[0]; xyz[(size_t)r + i] = (T)q[1]; xyz[2 * (size_t)r + i] = (T)q[2];      // exact: they were widened from this type
  bound[i] = q[3];
}

// rows[e]: the local row request e of this bucket came from; (bi, bd)[e][k]: what the other slab found within that row's bound.
// Different slabs hold different points, so the two lists share no index; both are ascending under (d2, index).
__global__ __launch_bounds__(XW) void xmerge_kernel(const uint32_t* __restrict__ rows, uint32_t cnt, const uint32_t* __restrict__ bi,
                                                    const double* __restrict__ bd, int k, uint32_t* __restrict__ idx, double* __restrict__ d2,
                                                    uint8_t* __restrict__ flags) {
  const uint32_t e = blockIdx.x * XW + threadIdx.x;
  if (e >= cnt) return;
  const size_t row = (size_t)rows[e] * (size_t)k, src = (size_t)e * (size_t)k;
  uint32_t oi[PT_TILE_MAX_K];
  double od[PT_TILE_MAX_K];
  int a = 0, b = 0;
  for (int o = 0; o < k; ++o) {
    const uint32_t ia = a < k ? idx[row + a] : PT_NOIDX_U, ib = b < k ? bi[src + b] : PT_NOIDX_U;
    const double da = ia != PT_NOIDX_U ? d2[row + a] : INFINITY, db = ib != PT_NOIDX_U ? bd[src + b] : INFINITY;
    const bool take_a = ia != PT_NOIDX_U && (ib == PT_NOIDX_U || da < db || (da == db && ia < ib));
    if (take_a) { oi[o] = ia; od[o] = da; ++a; }
    else if (ib != PT_NOIDX_U) { oi[o] = ib; od[o] = db; ++b; }
    else { oi[o] = PT_NOIDX_U; od[o] = INFINITY; }
  }
  for (int o = 0; o < k; ++o) { idx[row + o] = oi[o]; d2[row + o] = od[o]; }
  if (flags) flags[rows[e]] = 1;
}

// ---- slabs that keep LOCAL ids in their records and their own points' attribute records only (round 4) --------------------------------
// A slab built from strictly ascending global indices sorts the POSITION of a point in the slab's arrays into its records instead of
// the global index: the order (d2, position) is the order (d2, index), the fused blend gathers from a table of the slab's own n records
// (16 n bytes per GPU instead of 16 N), and the finished lists are translated through gidx.  Whoever meets a GLOBAL index afterwards --
// a candidate another slab answered, a list handed back to pt_blend -- finds its position by binary search in the ascending gidx.
__device__ inline uint32_t local_of(const uint32_t* __restrict__ gidx, uint32_t n, uint32_t g) {       // position of global index g in this slab, PT_NOIDX_U if it is not here
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (gidx[mid] < g) lo = mid + 1; else hi = mid; }
  return (lo < n && gidx[lo] == g) ? lo : PT_NOIDX_U;
}
__global__ __launch_bounds__(XW) void ids_to_global_kernel(uint32_t* __restrict__ idx, size_t count, const uint32_t* __restrict__ gidx) {
  const size_t i = (size_t)blockIdx.x * XW + threadIdx.x;
  if (i < count) { const uint32_t v = idx[i]; if (v != PT_NOIDX_U) idx[i] = gidx[v]; }
}
__global__ __launch_bounds__(XW) void ids_to_local_kernel(const uint32_t* __restrict__ in, size_t count, const uint32_t* __restrict__ gidx, uint32_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * XW + threadIdx.x;
  if (i < count) { const uint32_t v = in[i]; out[i] = v == PT_NOIDX_U ? PT_NOIDX_U : local_of(gidx, n, v); }
}
// the attribute records of the candidates a slab answers with (global indices of ITS points; an empty slot gets a zero record)
__global__ __launch_bounds__(XW) void xgather_attr_kernel(const uint32_t* __restrict__ ids, size_t count, const uint32_t* __restrict__ gidx, uint32_t n,
                                                          const Attr* __restrict__ attr, Attr* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * XW + threadIdx.x;
  if (i >= count) return;
  Attr a; a.rgba = 0; a.nx = a.ny = a.nz = 0.f;
  const uint32_t v = ids[i];
  if (v != PT_NOIDX_U) { const uint32_t l = local_of(gidx, n, v); if (l != PT_NOIDX_U) a = attr[l]; }
  out[i] = a;
}
// xmerge_kernel with the candidates' attribute records carried along: rattr[row][k] holds the records of the row's list in list order,
// filled from the slab's own table the first time a row is touched (flags[row] == 0: its list still names home points only)
__global__ __launch_bounds__(XW) void xmerge_attr_kernel(const uint32_t* __restrict__ rows, uint32_t cnt, const uint32_t* __restrict__ bi, const double* __restrict__ bd,
                                                         const Attr* __restrict__ ba, int k, uint32_t* __restrict__ idx, double* __restrict__ d2, Attr* __restrict__ rattr,
                                                         uint8_t* __restrict__ flags, const uint32_t* __restrict__ gidx, uint32_t n, const Attr* __restrict__ attr) {
  const uint32_t e = blockIdx.x * XW + threadIdx.x;
  if (e >= cnt) return;
  const uint32_t r = rows[e];
  const size_t row = (size_t)r * (size_t)k, src = (size_t)e * (size_t)k;
  if (!flags[r]) {
    for (int j = 0; j < k; ++j) {
      Attr a; a.rgba = 0; a.nx = a.ny = a.nz = 0.f;
      const uint32_t v = idx[row + j];
      if (v != PT_NOIDX_U) { const uint32_t l = local_of(gidx, n, v); if (l != PT_NOIDX_U) a = attr[l]; }
      rattr[row + j] = a;
    }
  }
  uint32_t oi[PT_TILE_MAX_K];
  double od[PT_TILE_MAX_K];
  Attr oa[PT_TILE_MAX_K];
  int a = 0, b = 0;
  for (int o = 0; o < k; ++o) {
    const uint32_t ia = a < k ? idx[row + a] : PT_NOIDX_U, ib = b < k ? bi[src + b] : PT_NOIDX_U;
    const double da = ia != PT_NOIDX_U ? d2[row + a] : INFINITY, db = ib != PT_NOIDX_U ? bd[src + b] : INFINITY;
    const bool take_a = ia != PT_NOIDX_U && (ib == PT_NOIDX_U || da < db || (da == db && ia < ib));
    if (take_a) { oi[o] = ia; od[o] = da; oa[o] = rattr[row + a]; ++a; }
    else if (ib != PT_NOIDX_U) { oi[o] = ib; od[o] = db; oa[o] = ba[src + b]; ++b; }
    else { oi[o] = PT_NOIDX_U; od[o] = INFINITY; oa[o].rgba = 0; oa[o].nx = oa[o].ny = oa[o].nz = 0.f; }
  }
  for (int o = 0; o < k; ++o) { idx[row + o] = oi[o]; d2[row + o] = od[o]; rattr[row + o] = oa[o]; }
  flags[r] = 1;
}
// the blend of pt_attr.hip's blend_one for the listed rows, from the records the merge carried along
__global__ __launch_bounds__(XW) void blend_rows_attr_kernel(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ rows_n, const uint32_t* __restrict__ idx,
                                                             const double* __restrict__ d2, const Attr* __restrict__ rattr, int k, int mode, float* __restrict__ rgb_out,
                                                             float* __restrict__ nrm_out) {
  const uint32_t i = blockIdx.x * XW + threadIdx.x;
  if (i >= *rows_n) return;
  const uint32_t t = rows[i];
  double wsum = 0.0, c[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
  for (int j = 0; j < k; ++j) {
    if (idx[(size_t)t * k + j] == PT_NOIDX_U) continue;
    const Attr a = rattr[(size_t)t * k + j];
    const double w = (mode == 1) ? 1.0 / (d2[(size_t)t * k + j] + 1e-12) : 1.0;
    wsum += w;
    c[0] += w * (double)(a.rgba & 0xFFu); c[1] += w * (double)((a.rgba >> 8) & 0xFFu); c[2] += w * (double)((a.rgba >> 16) & 0xFFu);
    nn[0] += w * (double)a.nx; nn[1] += w * (double)a.ny; nn[2] += w * (double)a.nz;
  }
  if (wsum > 0.0) {
    const double iw = 1.0 / wsum;
    for (int q = 0; q < 3; ++q) { c[q] *= iw; nn[q] *= iw; }
    const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    if (len >= 1e-12) { nn[0] /= len; nn[1] /= len; nn[2] /= len; }
  }
  if (rgb_out) { rgb_out[3 * (size_t)t] = (float)c[0]; rgb_out[3 * (size_t)t + 1] = (float)c[1]; rgb_out[3 * (size_t)t + 2] = (float)c[2]; }
  if (nrm_out) { nrm_out[3 * (size_t)t] = (float)nn[0]; nrm_out[3 * (size_t)t + 1] = (float)nn[1]; nrm_out[3 * (size_t)t + 2] = (float)nn[2]; }
}
// is gidx strictly ascending?  *flag |= 1 when it is not
__global__ __launch_bounds__(XW) void ascending_kernel(const uint32_t* __restrict__ gidx, uint32_t n, uint32_t* flag) {
  const uint32_t i = blockIdx.x * XW + threadIdx.x;
  if (i + 1 < n && !(gidx[i] < gidx[i + 1])) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(XW) void xflag_rows_kernel(const uint8_t* __restrict__ flags, uint32_t m, uint32_t* __restrict__ rows, uint32_t* __restrict__ count) {
  // (as xreq_kernel: a workgroup counts the flagged rows of its piece in LDS, takes its range with ONE atomic and hands the slots out from LDS;
  //  70 k rows of a rank one by one from one counter took 0.73 ms)
  __shared__ uint32_t cnt, base;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const uint32_t per = ((m + gridDim.x - 1) / gridDim.x + XW - 1) / XW * XW, t0 = blockIdx.x * per, t1 = min(m, t0 + per);
  for (uint32_t t = t0 + threadIdx.x; t < t1; t += XW) if (flags[t]) atomicAdd(&cnt, 1u);
  __syncthreads();
  if (threadIdx.x == 0) { base = cnt ? atomicAdd(count, cnt) : 0u; cnt = 0; }
  __syncthreads();
  for (uint32_t t = t0 + threadIdx.x; t < t1; t += XW) if (flags[t]) rows[base + atomicAdd(&cnt, 1u)] = t;
}

}  // namespace

static inline dim3 xgrid(uint32_t n) { return dim3((n + XW - 1) / XW); }

template <class T>
void pt_launch_xreq(bool fill, const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g, int me,
                    uint32_t* counts, const uint32_t* off, uint32_t* cursor, double* req, uint32_t* req_row, hipStream_t s) {
  if (!m) return;
  const dim3 grid(std::min<uint32_t>((m + XW - 1) / XW, 2048u));       // a piece of the targets per workgroup (xreq_kernel)
  if (fill) hipLaunchKernelGGL((xreq_kernel<T, true>), grid, dim3(XW), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, me, counts, off, cursor, req, req_row);
  else hipLaunchKernelGGL((xreq_kernel<T, false>), grid, dim3(XW), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, me, counts, off, cursor, req, req_row);
}
template void pt_launch_xreq<float>(bool, const float*, const float*, const float*, const double*, uint32_t, int, int, const double*, int, int, uint32_t*,
                                    const uint32_t*, uint32_t*, double*, uint32_t*, hipStream_t);
template void pt_launch_xreq<double>(bool, const double*, const double*, const double*, const double*, uint32_t, int, int, const double*, int, int, uint32_t*,
                                     const uint32_t*, uint32_t*, double*, uint32_t*, hipStream_t);
template <class T>
void pt_launch_xunpack(const double* rreq, uint32_t r, T* xyz, double* bound, hipStream_t s) {
  if (!r) return;
  hipLaunchKernelGGL(xunpack_kernel<T>, xgrid(r), dim3(XW), 0, s, rreq, r, xyz, bound);
}
template void pt_launch_xunpack<float>(const double*, uint32_t, float*, double*, hipStream_t);
template void pt_launch_xunpack<double>(const double*, uint32_t, double*, double*, hipStream_t);
void pt_launch_xmerge(const uint32_t* rows, uint32_t cnt, const uint32_t* bi, const double* bd, int k, uint32_t* idx, double* d2, uint8_t* flags, hipStream_t s) {
  if (!cnt) return;
  hipLaunchKernelGGL(xmerge_kernel, xgrid(cnt), dim3(XW), 0, s, rows, cnt, bi, bd, k, idx, d2, flags);
}
void pt_launch_xflag_rows(const uint8_t* flags, uint32_t m, uint32_t* rows, uint32_t* count, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(xflag_rows_kernel, dim3(std::min<uint32_t>((m + XW - 1) / XW, 2048u)), dim3(XW), 0, s, flags, m, rows, count);
}

static inline dim3 xgrid64(size_t n) { return dim3((unsigned)((n + XW - 1) / XW)); }
void pt_launch_ids_to_global(uint32_t* idx, size_t count, const uint32_t* gidx, hipStream_t s) {
  if (count) hipLaunchKernelGGL(ids_to_global_kernel, xgrid64(count), dim3(XW), 0, s, idx, count, gidx);
}
void pt_launch_ids_to_local(const uint32_t* in, size_t count, const uint32_t* gidx, uint32_t n, uint32_t* out, hipStream_t s) {
  if (count) hipLaunchKernelGGL(ids_to_local_kernel, xgrid64(count), dim3(XW), 0, s, in, count, gidx, n, out);
}
void pt_launch_xgather_attr(const uint32_t* ids, size_t count, const uint32_t* gidx, uint32_t n, const Attr* attr, Attr* out, hipStream_t s) {
  if (count) hipLaunchKernelGGL(xgather_attr_kernel, xgrid64(count), dim3(XW), 0, s, ids, count, gidx, n, attr, out);
}
void pt_launch_xmerge_attr(const uint32_t* rows, uint32_t cnt, const uint32_t* bi, const double* bd, const Attr* ba, int k, uint32_t* idx, double* d2, Attr* rattr,
                           uint8_t* flags, const uint32_t* gidx, uint32_t n, const Attr* attr, hipStream_t s) {
  if (cnt) hipLaunchKernelGGL(xmerge_attr_kernel, xgrid(cnt), dim3(XW), 0, s, rows, cnt, bi, bd, ba, k, idx, d2, rattr, flags, gidx, n, attr);
}
void pt_launch_blend_rows_attr(const uint32_t* rows, const uint32_t* rows_n, uint32_t m_max, const uint32_t* idx, const double* d2, const Attr* rattr, int k, int mode,
                               float* rgb_out, float* nrm_out, hipStream_t s) {
  if (m_max) hipLaunchKernelGGL(blend_rows_attr_kernel, xgrid(m_max), dim3(XW), 0, s, rows, rows_n, idx, d2, rattr, k, mode, rgb_out, nrm_out);
}
void pt_launch_ascending(const uint32_t* gidx, uint32_t n, uint32_t* flag, hipStream_t s) {
  if (n > 1) hipLaunchKernelGGL(ascending_kernel, xgrid(n), dim3(XW), 0, s, gidx, n, flag);
}
