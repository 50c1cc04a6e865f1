// Point.h -- the 80-byte point record of the detail-transfer path, layout-compatible
// with the reference's `struct Point` (reference src/Point.h:1-6: ver f64x3 @0,
// normal f64x3 @24, color i32x3 @48, U f64 @64, V f64 @72; pinned by
// oracle/_ref/point_layout.json which is produced from the reference's own header).
//
// This is a build-authored header: same field names, same member-function names and
// same semantics (so code written against the reference's Point keeps compiling), laid
// out so that `Point*` can be handed to the C ABI (include/pt_api.h) as `pt_point*`.
#ifndef PT_POINT_H
#define PT_POINT_H

#include <cstddef>
#include <cstdio>
#include <string>
#include <type_traits>

#include "pt_api.h"

struct Point {
  double ver[3];      // position
  double normal[3];   // parsed by the reference, never read by it afterwards (src/pointsTransfer.cpp:221-231)
  int color[3];       // r, g, b as ints
  double U, V;        // texture coordinates (mesh vertices only)

  // Construction follows reference src/Point.h:8-23: the default and xyz-only
  // constructors initialise `ver` alone.
  Point() : ver{0.0, 0.0, 0.0} {}
  Point(double px, double py, double pz) : ver{px, py, pz} {}
  Point(double px, double py, double pz, double nx_, double ny_, double nz_, int r_, int g_, int b_)
      : ver{px, py, pz}, normal{nx_, ny_, nz_}, color{r_, g_, b_} {}
  Point(double px, double py, double pz, double nx_, double ny_, double nz_, int r_, int g_, int b_, double u_,
        double v_)
      : ver{px, py, pz}, normal{nx_, ny_, nz_}, color{r_, g_, b_}, U(u_), V(v_) {}

  // Accessors, const and mutable, as in reference src/Point.h:25-53.
#define PT_POINT_ACCESSOR(type, name, expr)  \
  type name() const { return expr; }         \
  type& name() { return expr; }
  PT_POINT_ACCESSOR(double, x, ver[0])
  PT_POINT_ACCESSOR(double, y, ver[1])
  PT_POINT_ACCESSOR(double, z, ver[2])
  PT_POINT_ACCESSOR(double, nx, normal[0])
  PT_POINT_ACCESSOR(double, ny, normal[1])
  PT_POINT_ACCESSOR(double, nz, normal[2])
  PT_POINT_ACCESSOR(int, r, color[0])
  PT_POINT_ACCESSOR(int, g, color[1])
  PT_POINT_ACCESSOR(int, b, color[2])
  PT_POINT_ACCESSOR(double, u, U)
  PT_POINT_ACCESSOR(double, v, V)
#undef PT_POINT_ACCESSOR

  // Text dumps (reference src/Point.h:55-74; unused by its main()): std::to_string formatting,
  // space separated, no trailing space.
  std::string detail() const { return join(false); }
  std::string detailWithUV() const { return join(true); }

  // Equality is positional only (reference src/Point.h:76-81): two records with the same
  // xyz are "the same point" whatever their attributes -- this is what makes exact
  // duplicates exact distance ties in the search.
  bool operator==(const Point& o) const { return ver[0] == o.ver[0] && ver[1] == o.ver[1] && ver[2] == o.ver[2]; }
  bool operator!=(const Point& o) const { return !(*this == o); }

 private:
  std::string join(bool with_uv) const {
    std::string s;
    for (double c : ver) s += std::to_string(c) + ' ';
    for (double c : normal) s += std::to_string(c) + ' ';
    s += std::to_string(color[0]) + ' ' + std::to_string(color[1]) + ' ' + std::to_string(color[2]);
    if (with_uv) s += ' ' + std::to_string(U) + ' ' + std::to_string(V);
    return s;
  }
};

// Coordinate range adaptor (reference src/Point.h:85-92): "the coordinates of a Point are the three
// contiguous doubles at ver".  CGAL's Search_traits calls exactly these two operators.
struct Construct_coord_iterator {
  typedef const double* result_type;
  result_type operator()(const Point& p) const { return p.ver; }
  result_type operator()(const Point& p, int) const { return p.ver + 3; }
};

// reference src/Point.h:94-102 (`point_set_comparator`) is deliberately NOT reproduced: it is not a
// strict weak ordering and belongs to the per-face union of the texture
// bake, outside this path.

static_assert(sizeof(Point) == 80 && alignof(Point) == 8, "Point must keep the reference's 80-byte layout");
static_assert(offsetof(Point, ver) == 0 && offsetof(Point, normal) == 24 && offsetof(Point, color) == 48 &&
                  offsetof(Point, U) == 64 && offsetof(Point, V) == 72,
              "Point field offsets must match reference src/Point.h:2-6");
static_assert(std::is_trivially_copyable<Point>::value && std::is_standard_layout<Point>::value,
              "Point crosses the C ABI by memcpy");
static_assert(sizeof(pt_point) == sizeof(Point) && offsetof(pt_point, color) == offsetof(Point, color) &&
                  offsetof(pt_point, U) == offsetof(Point, U) && offsetof(pt_point, V) == offsetof(Point, V),
              "pt_point (C ABI) and Point must be layout-identical");

#endif  // PT_POINT_H
