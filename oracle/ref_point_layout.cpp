// Probe compiled against the REFERENCE's own src/Point.h where it lies under
// /root/reference (never copied into this repo).  Point.h needs nothing but
// <string>, which the reference's includer provides (pointsTransfer.cpp:31-35),
// so this is a real build of the reference's record type -- no stand-ins.
// Distance.h is NOT built: it needs CGAL::Dimension_tag / CGAL::Kd_tree_rectangle,
// CGAL is absent from the image, and writing stand-ins for it is not allowed.
//
// Output: one JSON line with the layout facts the C-ABI (include/pt_api.h,
// include/Point.h) must reproduce.  TEST INFRASTRUCTURE ONLY.
#include <cstddef>
#include <cstdio>
#include <string>
#include <type_traits>
#include "Point.h"   // -I/root/reference/src

int main() {
  Point a(1.0, 2.0, 3.0, 0.0, 0.0, 1.0, 10, 20, 30, 0.25, 0.75);
  Point b(1.0, 2.0, 3.0);                 // same xyz, different everything else
  Point c;                                // default ctor zeroes ver only (Point.h:8)
  Construct_coord_iterator it;
  std::printf(
      "{\"sizeof\": %zu, \"alignof\": %zu, \"off_ver\": %zu, \"off_normal\": %zu, \"off_color\": %zu, "
      "\"off_U\": %zu, \"off_V\": %zu, \"trivially_copyable\": %d, \"standard_layout\": %d, "
      "\"eq_xyz_only\": %d, \"default_ver_zero\": %d, \"coord_begin_is_ver\": %d, \"coord_len\": %td}\n",
      sizeof(Point), alignof(Point), offsetof(Point, ver), offsetof(Point, normal), offsetof(Point, color),
      offsetof(Point, U), offsetof(Point, V), (int)std::is_trivially_copyable<Point>::value,
      (int)std::is_standard_layout<Point>::value, (int)(a == b), (int)(c.x() == 0 && c.y() == 0 && c.z() == 0),
      (int)(it(a) == a.ver), it(a, 0) - it(a));
  return 0;
}
