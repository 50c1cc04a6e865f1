// pt_attr.hip -- everything around the search that touches attributes or raw input, for gfx950 (MI355X):
//   * the index-addressable synthetic generator of SURVEY.md Appendix C (bit-identical to the CPU checker under oracle/
//     for xyz and colour: integer hashing + one exact int->float conversion);
//   * the AoS -> planar split of the reference's 80-byte Point records (reference src/Point.h:1-6);
//   * attribute gather + blend.  The reference's only blend arithmetic is the barycentric colour mix of its
//     rasteriser (reference src/pointsTransfer.cpp:95-97: weights * colours summed as double products);
//     the per-vertex k-neighbour blend keeps that shape (out = sum_j w_j * a_j) with build-defined weights;
//   * PCA normals from the neighbours (BASELINE config 3; no reference counterpart).
#include "pt_internal.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int WG = 256;

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t stream_key(uint64_t seed, uint64_t stream) { return splitmix64(seed ^ (stream << 56)); }
__device__ inline float u24(uint64_t h) { return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f); }

// clustered distribution (BASELINE config 5): the same arithmetic, op for op, as the CPU checker's generator
__device__ inline float gauss4(uint64_t h) {
  const uint64_t a = splitmix64(h), b = splitmix64(a), c = splitmix64(b), d = splitmix64(c);
  return ((u24(a) + u24(b)) + (u24(c) + u24(d)) - 2.0f) * 1.7320508f;
}
__device__ inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 0.99999994f ? 0.99999994f : v); }
__device__ inline void clustered_source(uint64_t seed, uint64_t i, float (&out)[3]) {
  const uint64_t k0 = stream_key(seed, 0), k4 = stream_key(seed, 4);
  const uint64_t sel = splitmix64(k4 + 4ull * i);
  const uint32_t t = (uint32_t)(sel % 100u);
  const float ua = u24(splitmix64(k0 + 4ull * i + 0)), ub = u24(splitmix64(k0 + 4ull * i + 1));
  const uint64_t hc = splitmix64(k0 + 4ull * i + 2);
  if (t < 70u) {
    const uint64_t p = (sel >> 8) % 64u, kp = stream_key(seed, 5);
    const float g = gauss4(hc);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float o = u24(splitmix64(kp + 4ull * p + (uint64_t)c));
      const float e1 = (2.0f * u24(splitmix64(kp + 4ull * (64u + p) + (uint64_t)c)) - 1.0f) * 0.3f;
      const float e2 = (2.0f * u24(splitmix64(kp + 4ull * (128u + p) + (uint64_t)c)) - 1.0f) * 0.3f;
      const float nn = 2.0f * u24(splitmix64(kp + 4ull * (192u + p) + (uint64_t)c)) - 1.0f;
      out[c] = clamp01(((o + ua * e1) + ub * e2) + (1e-4f * g) * nn);
    }
  } else if (t < 95u) {
    const uint64_t q = (sel >> 8) % 256u, kq = stream_key(seed, 6);
    const float sigma = 1e-3f + 1.9e-2f * u24(splitmix64(kq + 4ull * q + 3));
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = clamp01(u24(splitmix64(kq + 4ull * q + (uint64_t)c)) + sigma * gauss4(splitmix64(hc + (uint64_t)c)));
  } else {
    out[0] = ua; out[1] = ub; out[2] = u24(hc);
  }
}

template <class T>
__global__ __launch_bounds__(WG) void synth_xyz_kernel(uint64_t key, uint32_t n_total, int axis, double lo, double hi, T* __restrict__ x,
                                                       T* __restrict__ y, T* __restrict__ z, uint32_t* __restrict__ gidx, uint32_t* counter,
                                                       uint32_t capacity, int round_f16, int dist, uint64_t seed, int stream,
                                                       uint64_t src_total, uint64_t tgt_total, uint32_t* __restrict__ wg_cnt, const uint32_t* __restrict__ wg_off) {
#pragma clang fp contract(off)
  // ORDERED slabs (round 4; wg_cnt / wg_off): the points of a slab keep the order of their global indices -- every workgroup of 256
  // consecutive indices counts its members (first launch: wg_cnt), the counts are scanned, and the second launch (wg_off) writes each
  // member at its workgroup's offset + its rank inside the workgroup.  Ascending indices are what lets a slab carry positions in its
  // records (pt_exchange.hip).  Without either the members are appended through one atomic counter, in any order.
  __shared__ uint32_t wsum[4];
  const bool ordered = wg_cnt != nullptr || wg_off != nullptr;
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= n_total && !ordered) return;
  const bool live = i < n_total;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (live) {
  if (dist == 0) {
    px = u24(splitmix64(key + 4ull * i + 0));
    py = u24(splitmix64(key + 4ull * i + 1));
    pz = u24(splitmix64(key + 4ull * i + 2));
  } else {
    float p[3];
    if (stream == 0) clustered_source(seed, i, p);
    else {
      const uint64_t step = (tgt_total && src_total / tgt_total) ? src_total / tgt_total : 1;
      clustered_source(seed, ((uint64_t)i * step) % (src_total ? src_total : 1), p);
#pragma unroll
      for (int c = 0; c < 3; ++c) p[c] = clamp01(p[c] + 5e-4f * gauss4(splitmix64(splitmix64(key + 4ull * i + (uint64_t)c))));
    }
    px = p[0]; py = p[1]; pz = p[2];
  }
  if (round_f16) {   // BASELINE config 5: xyz = half_rn(fp32 value), widened back exactly
    px = __half2float(__float2half_rn(px)); py = __half2float(__float2half_rn(py)); pz = __half2float(__float2half_rn(pz));
  }
  }
  uint32_t pos = i;
  if (axis >= 0) {
    const double c = (double)(axis == 0 ? px : (axis == 1 ? py : pz));
    const bool in = live && c >= lo && c < hi;
    if (ordered) {
      uint32_t tot;
      const uint32_t rank = block_excl_scan(in ? 1u : 0u, wsum, tot);
      if (wg_cnt) { if (threadIdx.x == 0) wg_cnt[blockIdx.x] = tot; return; }
      if (!in) return;
      pos = wg_off[blockIdx.x] + rank;
      if (!x || pos >= capacity) return;
    } else {
      if (!in) return;
      pos = atomicAdd(counter, 1u);      // hipcc folds this into one atomic per wave
      if (!x || pos >= capacity) return;   // counting pass, or overflow (host checks the counter)
    }
  }
  x[pos] = (T)px; y[pos] = (T)py; z[pos] = (T)pz;
  if (gidx) gidx[pos] = i;
}

// gidx != null: attr[j] is the record of point gidx[j] (a slab's own table, in the slab's order); else attr[i] of point i
__global__ __launch_bounds__(WG) void synth_attr_kernel(uint64_t key_rgb, uint64_t key_nrm, uint32_t n_total, Attr* __restrict__ attr, const uint32_t* __restrict__ gidx) {
#pragma clang fp contract(off)
  const uint32_t j = blockIdx.x * WG + threadIdx.x;
  if (j >= n_total) return;
  const uint64_t i = gidx ? gidx[j] : j;
  Attr a;
  a.rgba = (uint32_t)(splitmix64(key_rgb + 4ull * i) & 0xFFFFFFu);
  float nx = 2.0f * u24(splitmix64(key_nrm + 4ull * i + 0)) - 1.0f;
  float ny = 2.0f * u24(splitmix64(key_nrm + 4ull * i + 1)) - 1.0f;
  float nz = 2.0f * u24(splitmix64(key_nrm + 4ull * i + 2)) - 1.0f;
  const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
  if (len < 1e-12f) { nx = 0.f; ny = 0.f; nz = 1.f; }
  else { nx = nx / len; ny = ny / len; nz = nz / len; }
  a.nx = nx; a.ny = ny; a.nz = nz;
  attr[j] = a;
}

// reference Point: ver f64x3 @0, normal f64x3 @24, color i32x3 @48, U @64, V @72 (80 B)
__global__ __launch_bounds__(WG) void aos_split_kernel(const unsigned char* __restrict__ aos, uint32_t n, double* __restrict__ x,
                                                       double* __restrict__ y, double* __restrict__ z, Attr* __restrict__ attr) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  const double* d = (const double*)(aos + (size_t)i * 80);
  const int* col = (const int*)(aos + (size_t)i * 80 + 48);
  x[i] = d[0]; y[i] = d[1]; z[i] = d[2];
  if (attr) {
    Attr a;
    const uint32_t r = (uint32_t)min(max(col[0], 0), 255), g = (uint32_t)min(max(col[1], 0), 255), b = (uint32_t)min(max(col[2], 0), 255);
    a.rgba = r | (g << 8) | (b << 16);
    a.nx = (float)d[3]; a.ny = (float)d[4]; a.nz = (float)d[5];
    attr[i] = a;
  }
}

__global__ __launch_bounds__(WG) void pack_attr_kernel(const uint8_t* __restrict__ rgb, const float* __restrict__ nrm, uint32_t n,
                                                       Attr* __restrict__ attr) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  Attr a;
  a.rgba = rgb ? ((uint32_t)rgb[3 * (size_t)i] | ((uint32_t)rgb[3 * (size_t)i + 1] << 8) | ((uint32_t)rgb[3 * (size_t)i + 2] << 16)) : 0u;
  a.nx = nrm ? nrm[3 * (size_t)i] : 0.f;
  a.ny = nrm ? nrm[3 * (size_t)i + 1] : 0.f;
  a.nz = nrm ? nrm[3 * (size_t)i + 2] : 0.f;
  attr[i] = a;
}

// gather the k neighbours' 16-B attribute records of target row t and blend them
__device__ inline void blend_one(const uint32_t* __restrict__ idx, const double* __restrict__ d2, uint32_t t, int k, int mode,
                                 const Attr* __restrict__ attr, uint32_t n_attr, float* __restrict__ rgb_out, float* __restrict__ nrm_out) {
  double wsum = 0.0, c[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
  // four neighbours' records in flight per thread (one by one, a thread waited k random-access latencies in a row); the sums keep
  // their order.  A missing neighbour reads record 0 (cached) instead of branching around its load.
  constexpr int GB = 4;
  for (int j0 = 0; j0 < k && n_attr; j0 += GB) {
    uint32_t id[GB];
    Attr a[GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) id[q] = j0 + q < k ? idx[(size_t)t * k + j0 + q] : PT_NOIDX_U;
#pragma unroll
    for (int q = 0; q < GB; ++q) a[q] = pt_gather_attr(attr, (id[q] != PT_NOIDX_U && id[q] < n_attr) ? id[q] : 0u);
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      if (id[q] == PT_NOIDX_U || id[q] >= n_attr) continue;
      const double w = (mode == 1) ? 1.0 / (d2[(size_t)t * k + j0 + q] + 1e-12) : 1.0;
      wsum += w;
      c[0] += w * (double)(a[q].rgba & 0xFFu); c[1] += w * (double)((a[q].rgba >> 8) & 0xFFu); c[2] += w * (double)((a[q].rgba >> 16) & 0xFFu);
      nn[0] += w * (double)a[q].nx; nn[1] += w * (double)a[q].ny; nn[2] += w * (double)a[q].nz;
    }
  }
  if (wsum > 0.0) {
    const double iw = 1.0 / wsum;
#pragma unroll
    for (int a = 0; a < 3; ++a) { c[a] *= iw; nn[a] *= iw; }
    const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    if (len >= 1e-12) { nn[0] /= len; nn[1] /= len; nn[2] /= len; }
  }
  if (rgb_out) { rgb_out[3 * (size_t)t] = (float)c[0]; rgb_out[3 * (size_t)t + 1] = (float)c[1]; rgb_out[3 * (size_t)t + 2] = (float)c[2]; }
  if (nrm_out) { nrm_out[3 * (size_t)t] = (float)nn[0]; nrm_out[3 * (size_t)t + 1] = (float)nn[1]; nrm_out[3 * (size_t)t + 2] = (float)nn[2]; }
}

// one thread per target
__global__ __launch_bounds__(WG) void blend_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ d2, uint32_t m, int k, int mode,
                                                   const Attr* __restrict__ attr, uint32_t n_attr, float* __restrict__ rgb_out,
                                                   float* __restrict__ nrm_out) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  blend_one(idx, d2, t, k, mode, attr, n_attr, rgb_out, nrm_out);
}

// the reference's own mix (src/pointsTransfer.cpp:95-97: double weight x int colour, summed left to right in double, stored
// to a float) for k terms with caller-given weights; no normalisation
__global__ __launch_bounds__(WG) void blend_weighted_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ w, uint32_t m, int k,
                                                            const Attr* __restrict__ attr, uint32_t n_attr, float* __restrict__ rgb_out,
                                                            float* __restrict__ nrm_out) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  double c[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
  bool first = true;
  for (int j = 0; j < k; ++j) {
    const uint32_t id = idx[(size_t)t * k + j];
    if (id == PT_NOIDX_U || id >= n_attr) continue;
    const double wj = w[(size_t)t * k + j];
    const Attr a = pt_gather_attr(attr, id);
    const double pc[3] = {wj * (double)(a.rgba & 0xFFu), wj * (double)((a.rgba >> 8) & 0xFFu), wj * (double)((a.rgba >> 16) & 0xFFu)};
    const double pn[3] = {wj * (double)a.nx, wj * (double)a.ny, wj * (double)a.nz};
#pragma unroll
    for (int q = 0; q < 3; ++q) { c[q] = first ? pc[q] : c[q] + pc[q]; nn[q] = first ? pn[q] : nn[q] + pn[q]; }
    first = false;
  }
  if (rgb_out) { rgb_out[3 * (size_t)t] = (float)c[0]; rgb_out[3 * (size_t)t + 1] = (float)c[1]; rgb_out[3 * (size_t)t + 2] = (float)c[2]; }
  if (nrm_out) { nrm_out[3 * (size_t)t] = (float)nn[0]; nrm_out[3 * (size_t)t + 1] = (float)nn[1]; nrm_out[3 * (size_t)t + 2] = (float)nn[2]; }
}

// the same for the targets on a list of positions in the sorted target array (what the tile kernel left to the group kernel)
template <class Rec>
__global__ __launch_bounds__(WG) void blend_list_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n,
                                                        const Rec* __restrict__ tgt, const uint32_t* __restrict__ idx, const double* __restrict__ d2,
                                                        int k, int mode, const Attr* __restrict__ attr, uint32_t n_attr, float* __restrict__ rgb_out,
                                                        float* __restrict__ nrm_out) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= *list_n) return;
  blend_one(idx, d2, tgt[list[i]].id, k, mode, attr, n_attr, rgb_out, nrm_out);
}

// the same for a list of ROW ids (the rows a slab exchange completed with foreign candidates)
__global__ __launch_bounds__(WG) void blend_rows_kernel(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ rows_n, const uint32_t* __restrict__ idx,
                                                        const double* __restrict__ d2, int k, int mode, const Attr* __restrict__ attr, uint32_t n_attr,
                                                        float* __restrict__ rgb_out, float* __restrict__ nrm_out) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= *rows_n) return;
  blend_one(idx, d2, rows[i], k, mode, attr, n_attr, rgb_out, nrm_out);
}

// cyclic Jacobi on a symmetric 3x3 (fp64), same sweep order as the oracle
__device__ inline void jacobi3(double (&a)[3][3], double (&v)[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off <= 1e-20 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;   // converged far below fp64 resolution
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(a[p][q]) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double arp = a[r][p], arq = a[r][q]; a[r][p] = cs * arp - sn * arq; a[r][q] = sn * arp + cs * arq; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double apr = a[p][r], aqr = a[q][r]; a[p][r] = cs * apr - sn * aqr; a[q][r] = sn * apr + cs * aqr; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double vrp = v[r][p], vrq = v[r][q]; v[r][p] = cs * vrp - sn * vrq; v[r][q] = sn * vrp + cs * vrq; }
      }
  }
}

// position + attributes of one source point by ORIGINAL index, 32 bytes: what the PCA pass gathers per neighbour (one
// sector instead of four: x, y, z and the attribute record live in four different arrays)
struct PosAttr { float x, y, z; uint32_t rgba; float nx, ny, nz; uint32_t pad; };
static_assert(sizeof(PosAttr) == 32, "PosAttr is two 16-byte words");

__global__ __launch_bounds__(WG) void pack_posattr_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                          const Attr* __restrict__ attr, uint32_t n, PosAttr* __restrict__ out) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  PosAttr r;
  r.x = x[i]; r.y = y[i]; r.z = z[i]; r.rgba = 0; r.nx = r.ny = r.nz = 0.f; r.pad = 0;
  if (attr) { const Attr a = attr[i]; r.rgba = a.rgba; r.nx = a.nx; r.ny = a.ny; r.nz = a.nz; }
  out[i] = r;
}

// PCA normal of one target from its neighbours: `fetch(id, p, nrm)` yields the neighbour's position (fp64) and normal
template <class Fetch>
__device__ inline void pca_one(const uint32_t* __restrict__ idx, uint32_t t, int k, uint32_t n, bool has_attr, float* __restrict__ nrm_out, Fetch fetch) {
  // one gather pass: moments about the first neighbour (a shift keeps Sum(dd^T) - Sum(d)Sum(d)^T/n free of cancellation)
  double mn[3] = {0, 0, 0}, sd[3] = {0, 0, 0}, o[3] = {0, 0, 0};
  double cv[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  int ke = 0;
  // four neighbours' records in flight per thread, consumed in order (as blend_one; a missing neighbour fetches record 0)
  constexpr int GB = 4;
  for (int j0 = 0; j0 < k && n; j0 += GB) {
    uint32_t id[GB];
    double pq[GB][3];
    float aq[GB][3];
#pragma unroll
    for (int q = 0; q < GB; ++q) id[q] = j0 + q < k ? idx[(size_t)t * k + j0 + q] : PT_NOIDX_U;
#pragma unroll
    for (int q = 0; q < GB; ++q) fetch((id[q] != PT_NOIDX_U && id[q] < n) ? id[q] : 0u, pq[q], aq[q]);
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      if (id[q] == PT_NOIDX_U || id[q] >= n) continue;
      const double* p = pq[q];
      if (ke == 0) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
      const double d[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        sd[a] += d[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) cv[a][b] += d[a] * d[b];
      }
      ++ke;
      if (has_attr) { mn[0] += (double)aq[q][0]; mn[1] += (double)aq[q][1]; mn[2] += (double)aq[q][2]; }
    }
  }
  float* o3 = nrm_out + 3 * (size_t)t;
  if (ke < 3) { o3[0] = 0.f; o3[1] = 0.f; o3[2] = 1.f; return; }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) cv[a][b] -= sd[a] * sd[b] / (double)ke;
  double v[3][3];
  jacobi3(cv, v);
  const double e0 = cv[0][0], e1 = cv[1][1], e2 = cv[2][2];
  double nn[3];
  if (e0 <= e1 && e0 <= e2) { nn[0] = v[0][0]; nn[1] = v[1][0]; nn[2] = v[2][0]; }
  else if (e1 <= e2) { nn[0] = v[0][1]; nn[1] = v[1][1]; nn[2] = v[2][1]; }
  else { nn[0] = v[0][2]; nn[1] = v[1][2]; nn[2] = v[2][2]; }
  const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
  const double ref = has_attr ? (nn[0] * mn[0] + nn[1] * mn[1] + nn[2] * mn[2]) : nn[2];
  const double sgn = (ref < 0 ? -1.0 : 1.0) / len;
  o3[0] = (float)(nn[0] * sgn); o3[1] = (float)(nn[1] * sgn); o3[2] = (float)(nn[2] * sgn);
}

template <class T>
__global__ __launch_bounds__(WG) void pca_kernel(const uint32_t* __restrict__ idx, uint32_t m, int k, const T* __restrict__ x,
                                                 const T* __restrict__ y, const T* __restrict__ z, uint32_t n, const Attr* __restrict__ attr,
                                                 float* __restrict__ nrm_out) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  pca_one(idx, t, k, n, attr != nullptr, nrm_out, [&](uint32_t id, double (&p)[3], float (&an)[3]) {
    p[0] = (double)x[id]; p[1] = (double)y[id]; p[2] = (double)z[id];
    if (attr) { const Attr at = attr[id]; an[0] = at.nx; an[1] = at.ny; an[2] = at.nz; }
  });
}

__global__ __launch_bounds__(WG) void pca_pa_kernel(const uint32_t* __restrict__ idx, uint32_t m, int k, const PosAttr* __restrict__ pa, uint32_t n,
                                                    int has_attr, float* __restrict__ nrm_out) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  pca_one(idx, t, k, n, has_attr != 0, nrm_out, [&](uint32_t id, double (&p)[3], float (&an)[3]) {
    const float4 a = reinterpret_cast<const float4*>(pa)[2 * (size_t)id], b = reinterpret_cast<const float4*>(pa)[2 * (size_t)id + 1];
    p[0] = (double)a.x; p[1] = (double)a.y; p[2] = (double)a.z;
    an[0] = b.x; an[1] = b.y; an[2] = b.z;
  });
}

// fp16 planar xyz -> fp32 planar xyz (exact widening): fp16 clouds run through the fp32 path unchanged
__global__ __launch_bounds__(WG) void half_to_float_kernel(const __half* __restrict__ in, float* __restrict__ out, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i < count) out[i] = __half2float(in[i]);
}

__global__ __launch_bounds__(WG) void float_to_double_kernel(const float* __restrict__ in, double* __restrict__ out, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i < count) out[i] = (double)in[i];
}

__global__ __launch_bounds__(WG) void iota_kernel(uint32_t* p, uint32_t n) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i < n) p[i] = i;
}

inline dim3 grid_for(uint32_t n) { return dim3((n + WG - 1) / WG); }

}  // namespace

template <class T>
void pt_launch_synth_xyz(uint64_t seed, uint64_t stream, uint32_t n_total, int axis, double lo, double hi, T* x, T* y, T* z, uint32_t* gidx,
                         uint32_t* counter, uint32_t capacity, int round_f16, int dist, uint64_t src_total, uint64_t tgt_total, hipStream_t s,
                         uint32_t* wg_cnt, const uint32_t* wg_off) {
  if (!n_total) return;
  hipLaunchKernelGGL(synth_xyz_kernel<T>, grid_for(n_total), dim3(WG), 0, s, stream_key(seed, stream), n_total, axis, lo, hi, x, y, z, gidx,
                     counter, capacity, round_f16, dist, seed, (int)stream, src_total, tgt_total, wg_cnt, wg_off);
}
template void pt_launch_synth_xyz<float>(uint64_t, uint64_t, uint32_t, int, double, double, float*, float*, float*, uint32_t*, uint32_t*, uint32_t,
                                         int, int, uint64_t, uint64_t, hipStream_t, uint32_t*, const uint32_t*);
template void pt_launch_synth_xyz<__half>(uint64_t, uint64_t, uint32_t, int, double, double, __half*, __half*, __half*, uint32_t*, uint32_t*, uint32_t,
                                          int, int, uint64_t, uint64_t, hipStream_t, uint32_t*, const uint32_t*);
template void pt_launch_synth_xyz<double>(uint64_t, uint64_t, uint32_t, int, double, double, double*, double*, double*, uint32_t*, uint32_t*,
                                          uint32_t, int, int, uint64_t, uint64_t, hipStream_t, uint32_t*, const uint32_t*);

void pt_launch_synth_attr(uint64_t seed, uint32_t n_total, Attr* attr, hipStream_t s, const uint32_t* gidx) {
  if (!n_total) return;
  hipLaunchKernelGGL(synth_attr_kernel, grid_for(n_total), dim3(WG), 0, s, stream_key(seed, 2), stream_key(seed, 3), n_total, attr, gidx);
}
void pt_launch_aos_split(const void* aos, uint32_t n, double* x, double* y, double* z, Attr* attr, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(aos_split_kernel, grid_for(n), dim3(WG), 0, s, (const unsigned char*)aos, n, x, y, z, attr);
}
void pt_launch_pack_attr(const uint8_t* rgb, const float* nrm, uint32_t n, Attr* attr, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(pack_attr_kernel, grid_for(n), dim3(WG), 0, s, rgb, nrm, n, attr);
}
void pt_launch_blend_rows(const uint32_t* rows, const uint32_t* rows_n, uint32_t m_max, const uint32_t* idx, const double* d2, int k, int mode,
                          const Attr* attr, uint32_t n_attr, float* rgb_out, float* nrm_out, hipStream_t s) {
  if (!m_max) return;
  hipLaunchKernelGGL(blend_rows_kernel, grid_for(m_max), dim3(WG), 0, s, rows, rows_n, idx, d2, k, mode, attr, n_attr, rgb_out, nrm_out);
}
void pt_launch_blend(const uint32_t* idx, const double* d2, uint32_t m, int k, int mode, const Attr* attr, uint32_t n_attr, float* rgb_out,
                     float* nrm_out, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(blend_kernel, grid_for(m), dim3(WG), 0, s, idx, d2, m, k, mode, attr, n_attr, rgb_out, nrm_out);
}
void pt_launch_blend_weighted(const uint32_t* idx, const double* w, uint32_t m, int k, const Attr* attr, uint32_t n_attr, float* rgb_out,
                              float* nrm_out, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(blend_weighted_kernel, grid_for(m), dim3(WG), 0, s, idx, w, m, k, attr, n_attr, rgb_out, nrm_out);
}
template <class Rec>
void pt_launch_blend_list(const uint32_t* list, const uint32_t* list_n, uint32_t m_max, const Rec* tgt, const uint32_t* idx, const double* d2, int k,
                          int mode, const Attr* attr, uint32_t n_attr, float* rgb_out, float* nrm_out, hipStream_t s) {
  if (!m_max) return;
  hipLaunchKernelGGL(blend_list_kernel<Rec>, grid_for(m_max), dim3(WG), 0, s, list, list_n, tgt, idx, d2, k, mode, attr, n_attr, rgb_out, nrm_out);
}
template void pt_launch_blend_list<RecF>(const uint32_t*, const uint32_t*, uint32_t, const RecF*, const uint32_t*, const double*, int, int, const Attr*,
                                         uint32_t, float*, float*, hipStream_t);
template void pt_launch_blend_list<RecD>(const uint32_t*, const uint32_t*, uint32_t, const RecD*, const uint32_t*, const double*, int, int, const Attr*,
                                         uint32_t, float*, float*, hipStream_t);
template <class T>
void pt_launch_pca(const uint32_t* idx, uint32_t m, int k, const T* x, const T* y, const T* z, uint32_t n, const Attr* attr, float* nrm_out,
                   hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(pca_kernel<T>, grid_for(m), dim3(WG), 0, s, idx, m, k, x, y, z, n, attr, nrm_out);
}
template void pt_launch_pca<float>(const uint32_t*, uint32_t, int, const float*, const float*, const float*, uint32_t, const Attr*, float*,
                                   hipStream_t);
template void pt_launch_pca<double>(const uint32_t*, uint32_t, int, const double*, const double*, const double*, uint32_t, const Attr*, float*,
                                    hipStream_t);
void pt_launch_pack_posattr(const float* x, const float* y, const float* z, const Attr* attr, uint32_t n, void* out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(pack_posattr_kernel, grid_for(n), dim3(WG), 0, s, x, y, z, attr, n, (PosAttr*)out);
}
void pt_launch_pca_posattr(const uint32_t* idx, uint32_t m, int k, const void* posattr, uint32_t n, int has_attr, float* nrm_out, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(pca_pa_kernel, grid_for(m), dim3(WG), 0, s, idx, m, k, (const PosAttr*)posattr, n, has_attr, nrm_out);
}
void pt_launch_iota(uint32_t* p, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(WG), 0, s, p, n);
}
void pt_launch_half_to_float(const void* in_half, float* out, uint64_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(half_to_float_kernel, dim3((uint32_t)((count + WG - 1) / WG)), dim3(WG), 0, s, (const __half*)in_half, out, count);
}
void pt_launch_float_to_double(const float* in, double* out, uint64_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(float_to_double_kernel, dim3((uint32_t)((count + WG - 1) / WG)), dim3(WG), 0, s, in, out, count);
}
