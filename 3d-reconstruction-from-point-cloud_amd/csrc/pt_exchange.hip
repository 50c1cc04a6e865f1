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
  const uint32_t t = blockIdx.x * XW + threadIdx.x;
  if (t >= m) return;
  const double c = (double)(axis == 0 ? x[t] : (axis == 1 ? y[t] : z[t]));
  const double kth = d2[(size_t)t * k + (k - 1)];
  for (int s = 0; s < g; ++s) {
    if (s == me || !slab_needed(c, kth, bounds[s], bounds[s + 1])) continue;
    if constexpr (!FILL) {
      atomicAdd(&counts[s], 1u);
    } else {
      const uint32_t pos = off[s] + atomicAdd(&cursor[s], 1u);
      double* o = req + (size_t)pos * 4;
      o[0] = (double)x[t]; o[1] = (double)y[t]; o[2] = (double)z[t]; o[3] = kth;
      req_row[pos] = t;
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

__global__ __launch_bounds__(XW) void xflag_rows_kernel(const uint8_t* __restrict__ flags, uint32_t m, uint32_t* __restrict__ rows, uint32_t* __restrict__ count) {
  const uint32_t t = blockIdx.x * XW + threadIdx.x;
  if (t < m && flags[t]) rows[atomicAdd(count, 1u)] = t;
}

}  // namespace

static inline dim3 xgrid(uint32_t n) { return dim3((n + XW - 1) / XW); }

template <class T>
void pt_launch_xreq(bool fill, const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g, int me,
                    uint32_t* counts, const uint32_t* off, uint32_t* cursor, double* req, uint32_t* req_row, hipStream_t s) {
  if (!m) return;
  if (fill) hipLaunchKernelGGL((xreq_kernel<T, true>), xgrid(m), dim3(XW), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, me, counts, off, cursor, req, req_row);
  else hipLaunchKernelGGL((xreq_kernel<T, false>), xgrid(m), dim3(XW), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, me, counts, off, cursor, req, req_row);
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
  hipLaunchKernelGGL(xflag_rows_kernel, xgrid(m), dim3(XW), 0, s, flags, m, rows, count);
}
