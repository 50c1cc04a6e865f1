// pt_bake.hip -- per-face detail transfer into the UV atlas ("texture bake") and edge padding, for gfx950 (MI355X).
//
// Replaces the body of the reference's face loop after the neighbour search and its rasteriser and post-processing:
//   reference src/pointsTransfer.cpp:466-479  union of the three corners' neighbour lists
//                                    :484-537  plane through the face, projection, barycentric in-triangle filter
//                                    :539-581  per-face Delaunay triangulation of corners + interior points, UV of the interior
//                                              points by barycentric interpolation of the corner UVs
//                                    :66-107   draw_triangle: barycentric colour mix per pixel, BGRA bytes at (resolution - j, i)
//                                    :593-611  25x25 dilate, ~alpha mask, add  (edge padding)
// The reference leans on CGAL (Plane_3::to_2d, Triangle_coordinates_2, Delaunay_triangulation_2) and OpenCV (dilate) for these;
// neither is in this image, so the exact results are BUILD-DEFINED where the libraries (or undefined behaviour) decide them --
// the definition is written out in DESIGN.md ("Texture bake") and restated on the CPU by the test oracle; this file follows it
// operation for operation in fp64 with contraction off, so that the atlas bytes are identical:
//   * neighbour union de-duplicated by ORIGINAL INDEX, ascending;   * orthonormal plane frame at corner 0 (e1 along corner 0 -> 1);
//   * Delaunay = every non-degenerate triple whose circumcircle holds no other point strictly inside, predicates evaluated on the
//     index-sorted tuple (consistent signs), enumerated in lexicographic (i, j, k) order;
//   * a pixel covered by several triangles keeps the LAST one in (face, triangle) order -- the single-threaded reference's result:
//     here every covered pixel does one 64-bit atomicMax on a key {face * 256 + triangle + 1, BGRA}; a second pass keeps the BGRA.
//
// One wave64 per face (four per workgroup).  Nothing here is bandwidth-critical: the atlas is 64 M pixels, a face a few dozen.
#include "pt_internal.h"

#include <algorithm>

namespace {

constexpr int BK_WG = 256, BK_WAVES = BK_WG / 64;
constexpr int BK_MAXNBR = 96;             // 3 corners x k <= 32
constexpr int BK_MAXPTS = 3 + BK_MAXNBR;
constexpr int BK_MAXTRI = 255;            // triangle number within a face must fit 8 bits of the pixel key

struct BakeWave {
  uint32_t raw[BK_MAXNBR + 32];           // the three neighbour lists as read, padded to 128
  uint32_t sorted[BK_MAXNBR];             // distinct indices, ascending
  double cx[BK_MAXNBR], cy[BK_MAXNBR];    // 2-D image of candidate e (sorted order)
  double cu[BK_MAXNBR], cv[BK_MAXNBR];    // its interpolated UV
  uint8_t cin[BK_MAXNBR + 32];            // inside the face?
  double px[BK_MAXPTS], py[BK_MAXPTS], pu[BK_MAXPTS], pv[BK_MAXPTS];   // kept points: corners 0..2, then interior points
  uint32_t pc[BK_MAXPTS];                 // their colour, r | g << 8 | b << 16
  uint32_t tri[BK_MAXTRI];                // accepted triangles, i | j << 8 | k << 16, in enumeration order
};

__device__ inline double cross2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
__device__ inline bool finite_d(double v) { return v == v && v - v == 0.0; }
// barycentric coordinates of X in (v0, v1, v2), A = cross(v1 - v0, v2 - v0)
__device__ inline void bary2(double Xx, double Xy, double x0, double y0, double x1, double y1, double x2, double y2, double A, double (&b)[3]) {
  b[0] = cross2(x1 - Xx, y1 - Xy, x2 - Xx, y2 - Xy) / A;
  b[1] = cross2(x2 - Xx, y2 - Xy, x0 - Xx, y0 - Xy) / A;
  b[2] = (1.0 - b[0]) - b[1];
}
__device__ inline double incircle_sorted(const double* px, const double* py, int w, int x, int y, int z) {
  const double adx = px[w] - px[z], ady = py[w] - py[z], bdx = px[x] - px[z], bdy = py[x] - py[z], cdx = px[y] - px[z], cdy = py[y] - py[z];
  const double al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
  return (al * (bdx * cdy - bdy * cdx) - bl * (adx * cdy - ady * cdx)) + cl * (adx * bdy - ady * bdx);
}
__device__ inline bool in_circumcircle(const double* px, const double* py, int i, int j, int k, int l, int os) {
  double d;
  int par;
  if (l > k) { d = incircle_sorted(px, py, i, j, k, l); par = 1; }
  else if (l > j) { d = incircle_sorted(px, py, i, j, l, k); par = -1; }
  else if (l > i) { d = incircle_sorted(px, py, i, l, j, k); par = 1; }
  else { d = incircle_sorted(px, py, l, i, j, k); par = -1; }
  return (double)(os * par) * d > 0.0;
}

// reference draw_triangle (:66-107), the wave's 64 lanes striding over the pixels of the bounding box that land inside the texture
__device__ inline void draw_triangle(const double (&U)[3], const double (&V)[3], const uint32_t (&col)[3], int R, unsigned long long seq,
                                     unsigned long long* __restrict__ keys, int lane) {
  const double px = U[0] * R, py = V[0] * R, qx = U[1] * R, qy = V[1] * R, rx = U[2] * R, ry = V[2] * R;
  if (!(finite_d(px) && finite_d(py) && finite_d(qx) && finite_d(qy) && finite_d(rx) && finite_d(ry))) return;
  const double A = cross2(qx - px, qy - py, rx - px, ry - py);
  if (!(A != 0.0) || !finite_d(A)) return;
  const double xmin = fmin(px, fmin(qx, rx)), xmax = fmax(px, fmax(qx, rx));
  const double ymin = fmin(py, fmin(qy, ry)), ymax = fmax(py, fmax(qy, ry));
  const int i0 = (int)fmax(floor(xmin), 0.0), i1 = (int)fmin(floor(xmax), (double)(R - 1));      // column i in [0, R)
  const int j0 = (int)fmax(floor(ymin), 1.0), j1 = (int)fmin(floor(ymax), (double)R);            // row R - j in [0, R)
  if (i1 < i0 || j1 < j0) return;
  const unsigned long long ni = (unsigned long long)(i1 - i0 + 1), total = ni * (unsigned long long)(j1 - j0 + 1);
  for (unsigned long long pix = (unsigned long long)lane; pix < total; pix += 64ull) {
    const int i = i0 + (int)(pix % ni), j = j0 + (int)(pix / ni);
    const int x = i >= R ? R - 1 : i, y = j >= R ? R - 1 : j;
    double b[3];
    bary2((double)x, (double)y, px, py, qx, qy, rx, ry, A, b);
    if (b[0] >= 0 && b[1] >= 0 && b[2] >= 0) {
      uint32_t bgra = 0xFF000000u;
#pragma unroll
      for (int c = 0; c < 3; ++c) {                        // :95-97: double products summed, stored to a float, truncated to a byte
        const double c0 = (double)((col[0] >> (8 * c)) & 0xFFu), c1 = (double)((col[1] >> (8 * c)) & 0xFFu), c2 = (double)((col[2] >> (8 * c)) & 0xFFu);
        const float f = (float)((b[0] * c0 + b[1] * c1) + b[2] * c2);
        const float g = f < 0.f ? 0.f : (f > 255.f ? 255.f : f);
        bgra |= (uint32_t)g << (8 * (2 - c));              // byte 0 = B, 1 = G, 2 = R
      }
      atomicMax(&keys[(size_t)(R - j) * (size_t)R + (size_t)i], (seq << 32) | (unsigned long long)bgra);
    }
  }
}

template <class T>
__global__ __launch_bounds__(BK_WG) void bake_faces_kernel(const T* __restrict__ sx, const T* __restrict__ sy, const T* __restrict__ sz,
                                                           const Attr* __restrict__ attr, uint32_t n, const unsigned char* __restrict__ verts /* 80-B records */,
                                                           uint32_t nv, const int32_t* __restrict__ faces, uint32_t nf,
                                                           const uint32_t* __restrict__ nbr, int k, int R, unsigned long long* __restrict__ keys) {
  __shared__ BakeWave sh[BK_WAVES];
  const int lane = threadIdx.x & 63;
  const uint32_t f = blockIdx.x * BK_WAVES + (threadIdx.x >> 6);
  if (f >= nf) return;                                     // whole waves leave; there is no workgroup barrier below
  BakeWave& W = sh[threadIdx.x >> 6];
  const int32_t f0 = faces[3 * (size_t)f], f1 = faces[3 * (size_t)f + 1], f2 = faces[3 * (size_t)f + 2];
  if (f0 < 0 || f1 < 0 || f2 < 0 || (uint32_t)f0 >= nv || (uint32_t)f1 >= nv || (uint32_t)f2 >= nv) return;      // malformed face
  const int32_t fv[3] = {f0, f1, f2};
  double c3[3][3], cu[3], cv[3];
  uint32_t ccol[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double* d = reinterpret_cast<const double*>(verts + (size_t)fv[c] * 80);
    const int* ci = reinterpret_cast<const int*>(verts + (size_t)fv[c] * 80 + 48);
    c3[c][0] = d[0]; c3[c][1] = d[1]; c3[c][2] = d[2];
    cu[c] = d[8]; cv[c] = d[9];
    ccol[c] = (uint32_t)min(max(ci[0], 0), 255) | ((uint32_t)min(max(ci[1], 0), 255) << 8) | ((uint32_t)min(max(ci[2], 0), 255) << 16);
  }
  // ---- union of the three neighbour lists, by original index, ascending -----------------------------------------------
  const int nraw = 3 * k;
  for (int e = lane; e < BK_MAXNBR + 32; e += 64) {
    uint32_t id = PT_NOIDX_U;
    if (e < nraw) { id = nbr[(size_t)fv[e / k] * (size_t)k + (size_t)(e % k)]; if (id >= n) id = PT_NOIDX_U; }
    W.raw[e] = id;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int nid = 0;
  {
    uint32_t val[2], pos[2];
    bool first[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = lane + 64 * h;
      val[h] = W.raw[e];
      first[h] = val[h] != PT_NOIDX_U;
      for (int q = 0; q < nraw && q < e; ++q) if (W.raw[q] == val[h]) first[h] = false;     // an equal index earlier in the lists
      W.cin[e] = first[h] ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pos[h] = 0;
      for (int q = 0; q < nraw; ++q) if (W.cin[q] && W.raw[q] < val[h]) ++pos[h];           // rank among the distinct values
    }
    nid = (int)__popcll(__ballot(first[0])) + (int)__popcll(__ballot(first[1]));
#pragma unroll
    for (int h = 0; h < 2; ++h) if (first[h]) W.sorted[pos[h]] = val[h];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- plane frame: origin corner 0, e1 along corner 0 -> 1, e2 = n x e1 (every lane computes the same values) ---------
  const double ax = c3[1][0] - c3[0][0], ay = c3[1][1] - c3[0][1], az = c3[1][2] - c3[0][2];
  const double bx = c3[2][0] - c3[0][0], by = c3[2][1] - c3[0][1], bz = c3[2][2] - c3[0][2];
  const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
  const double la = sqrt((ax * ax + ay * ay) + az * az);
  const double e1x = ax / la, e1y = ay / la, e1z = az / la;
  const double tx = ny * e1z - nz * e1y, ty = nz * e1x - nx * e1z, tz = nx * e1y - ny * e1x;
  const double lt = sqrt((tx * tx + ty * ty) + tz * tz);
  const double e2x = tx / lt, e2y = ty / lt, e2z = tz / lt;
  const bool frame_ok = la > 0.0 && lt > 0.0 && finite_d(la) && finite_d(lt);
  const double P0x = 0.0, P0y = 0.0;
  const double P1x = (ax * e1x + ay * e1y) + az * e1z, P1y = (ax * e2x + ay * e2y) + az * e2z;
  const double P2x = (bx * e1x + by * e1y) + bz * e1z, P2y = (bx * e2x + by * e2y) + bz * e2z;
  const double A = frame_ok ? cross2(P1x - P0x, P1y - P0y, P2x - P0x, P2y - P0y) : 0.0;
  if (lane < 3) {
    W.px[lane] = lane == 0 ? P0x : (lane == 1 ? P1x : P2x);
    W.py[lane] = lane == 0 ? P0y : (lane == 1 ? P1y : P2y);
    W.pu[lane] = cu[lane]; W.pv[lane] = cv[lane]; W.pc[lane] = ccol[lane];
  }
  int np = 3;
  if (frame_ok && A != 0.0 && finite_d(A)) {
    // ---- project the candidates, keep what is inside the face (:505-537) ----------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = lane + 64 * h;
      bool inside = false;
      if (e < nid) {
        const size_t id = W.sorted[e];
        const double dx = (double)sx[id] - c3[0][0], dy = (double)sy[id] - c3[0][1], dz = (double)sz[id] - c3[0][2];
        const double Xx = (dx * e1x + dy * e1y) + dz * e1z, Xy = (dx * e2x + dy * e2y) + dz * e2z;
        double b[3];
        bary2(Xx, Xy, P0x, P0y, P1x, P1y, P2x, P2y, A, b);
        inside = b[0] >= 0 && b[1] >= 0 && b[2] >= 0;
        W.cx[e] = Xx; W.cy[e] = Xy;
        W.cu[e] = (b[0] * cu[0] + b[1] * cu[1]) + b[2] * cu[2];             // :571-572
        W.cv[e] = (b[0] * cv[0] + b[1] * cv[1]) + b[2] * cv[2];
      }
      if (e < BK_MAXNBR + 32) W.cin[e] = inside ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // a candidate whose 2-D image equals a corner's or an EARLIER inside candidate's is dropped (the earliest of equals stays)
    bool keep[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = lane + 64 * h;
      keep[h] = e < nid && W.cin[e];
      if (keep[h]) {
        const double Xx = W.cx[e], Xy = W.cy[e];
        if ((Xx == P0x && Xy == P0y) || (Xx == P1x && Xy == P1y) || (Xx == P2x && Xy == P2y)) keep[h] = false;
        for (int q = 0; q < e; ++q) if (W.cin[q] && W.cx[q] == Xx && W.cy[q] == Xy) keep[h] = false;
      }
    }
    const unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (keep[h]) {
        const int e = lane + 64 * h;
        const int slot = 3 + (h ? (int)__popcll(m0) + (int)__popcll(m1 & below) : (int)__popcll(m0 & below));
        W.px[slot] = W.cx[e]; W.py[slot] = W.cy[e]; W.pu[slot] = W.cu[e]; W.pv[slot] = W.cv[e];
        W.pc[slot] = attr[W.sorted[e]].rgba & 0xFFFFFFu;
      }
    }
    np = 3 + (int)__popcll(m0) + (int)__popcll(m1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const unsigned long long seq0 = (unsigned long long)f * 256ull + 1ull;
  if (np == 3) {                                            // no interior points: the face itself (:540-544)
    draw_triangle(cu, cv, ccol, R, seq0, keys, lane);
    return;
  }
  // ---- Delaunay by exhaustion: triples in lexicographic order, lanes over k (:546-581 with the build's definition) --------
  int ntri = 0;
  for (int i = 0; i < np - 2; ++i)
    for (int j = i + 1; j < np - 1; ++j)
      for (int kb = j + 1; kb < np; kb += 64) {
        const int kk = kb + lane;
        bool ok = false;
        if (kk < np) {
          const double o = cross2(W.px[j] - W.px[i], W.py[j] - W.py[i], W.px[kk] - W.px[i], W.py[kk] - W.py[i]);
          if (o != 0.0) {
            const int os = o > 0.0 ? 1 : -1;
            ok = true;
            for (int l = 0; l < np && ok; ++l)
              if (l != i && l != j && l != kk && in_circumcircle(W.px, W.py, i, j, kk, l, os)) ok = false;
          }
        }
        const unsigned long long m = __ballot(ok);
        if (ok) {
          const int slot = ntri + (int)__popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
          if (slot < BK_MAXTRI) W.tri[slot] = (uint32_t)i | ((uint32_t)j << 8) | ((uint32_t)kk << 16);
        }
        ntri += (int)__popcll(m);
      }
  if (ntri > BK_MAXTRI) ntri = BK_MAXTRI;                   // (2 np - 5 <= 193 triangles unless many points are co-circular)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int t = 0; t < ntri; ++t) {
    const uint32_t tr = W.tri[t];
    const int a = (int)(tr & 0xFFu), b = (int)((tr >> 8) & 0xFFu), c = (int)((tr >> 16) & 0xFFu);
    const double U[3] = {W.pu[a], W.pu[b], W.pu[c]}, V[3] = {W.pv[a], W.pv[b], W.pv[c]};
    const uint32_t col[3] = {W.pc[a], W.pc[b], W.pc[c]};
    draw_triangle(U, V, col, R, seq0 + (unsigned long long)t, keys, lane);
  }
}

__global__ __launch_bounds__(BK_WG) void bake_resolve_kernel(const unsigned long long* __restrict__ keys, uint32_t* __restrict__ bgra, size_t npix) {
  const size_t i = (size_t)blockIdx.x * BK_WG + threadIdx.x;
  if (i < npix) bgra[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull);           // untouched pixels hold key 0: transparent black
}

__device__ inline uint32_t bytemax(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) { const uint32_t x = (a >> (8 * c)) & 0xFFu, y = (b >> (8 * c)) & 0xFFu; r |= (x > y ? x : y) << (8 * c); }
  return r;
}
// separable ksize x ksize maximum per channel (what lies outside the image does not count), reference :594-597
__global__ __launch_bounds__(BK_WG) void dilate_rows_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int R, int h) {
  const size_t i = (size_t)blockIdx.x * BK_WG + threadIdx.x;
  if (i >= (size_t)R * R) return;
  const int x = (int)(i % (size_t)R);
  const size_t row = i - (size_t)x;
  uint32_t m = 0;
  for (int d = -h; d <= h; ++d) { const int xx = x + d; if (xx >= 0 && xx < R) m = bytemax(m, in[row + (size_t)xx]); }
  out[i] = m;
}
// vertical maximum of the row maxima, then edges = dilated & ~alpha (all four channels), padded = texture + edges, saturating (:598-611)
__global__ __launch_bounds__(BK_WG) void dilate_cols_pad_kernel(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ tex, uint32_t* __restrict__ out,
                                                                int R, int h) {
  const size_t i = (size_t)blockIdx.x * BK_WG + threadIdx.x;
  if (i >= (size_t)R * R) return;
  const int x = (int)(i % (size_t)R), y = (int)(i / (size_t)R);
  uint32_t m = 0;
  for (int d = -h; d <= h; ++d) { const int yy = y + d; if (yy >= 0 && yy < R) m = bytemax(m, rows[(size_t)yy * R + x]); }
  const uint32_t t = tex[i];
  const uint32_t mask = (~(t >> 24)) & 0xFFu;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t s = ((t >> (8 * c)) & 0xFFu) + (((m >> (8 * c)) & 0xFFu) & mask);
    r |= (s > 255u ? 255u : s) << (8 * c);
  }
  out[i] = r;
}

}  // namespace

template <class T>
void pt_launch_bake_faces(const T* sx, const T* sy, const T* sz, const Attr* attr, uint32_t n, const void* verts_aos, uint32_t nv, const int32_t* faces,
                          uint32_t nf, const uint32_t* nbr, int k, int R, unsigned long long* keys, hipStream_t s) {
  if (!nf) return;
  hipLaunchKernelGGL(bake_faces_kernel<T>, dim3((nf + BK_WAVES - 1) / BK_WAVES), dim3(BK_WG), 0, s, sx, sy, sz, attr, n, (const unsigned char*)verts_aos, nv,
                     faces, nf, nbr, k, R, keys);
}
template void pt_launch_bake_faces<float>(const float*, const float*, const float*, const Attr*, uint32_t, const void*, uint32_t, const int32_t*, uint32_t,
                                          const uint32_t*, int, int, unsigned long long*, hipStream_t);
template void pt_launch_bake_faces<double>(const double*, const double*, const double*, const Attr*, uint32_t, const void*, uint32_t, const int32_t*, uint32_t,
                                           const uint32_t*, int, int, unsigned long long*, hipStream_t);
void pt_launch_bake_resolve(const unsigned long long* keys, uint32_t* bgra, size_t npix, hipStream_t s) {
  if (!npix) return;
  hipLaunchKernelGGL(bake_resolve_kernel, dim3((uint32_t)((npix + BK_WG - 1) / BK_WG)), dim3(BK_WG), 0, s, keys, bgra, npix);
}
void pt_launch_dilate_pad(const uint32_t* tex, uint32_t* tmp, uint32_t* out, int R, int ksize, hipStream_t s) {
  const size_t npix = (size_t)R * R;
  if (!npix) return;
  const uint32_t g = (uint32_t)((npix + BK_WG - 1) / BK_WG);
  hipLaunchKernelGGL(dilate_rows_kernel, dim3(g), dim3(BK_WG), 0, s, tex, tmp, R, ksize / 2);
  hipLaunchKernelGGL(dilate_cols_pad_kernel, dim3(g), dim3(BK_WG), 0, s, tmp, tex, out, R, ksize / 2);
}
