// Distance.h -- the squared-Euclidean metric and box bounds of the detail-transfer path,
// with the member signatures CGAL's GeneralDistance concept expects (reference
// src/Distance.h:1-101), so the functor still plugs into
// CGAL::Orthogonal_k_neighbor_search<Traits, Distance> (reference src/pointsTransfer.cpp:39)
// when CGAL is available, and works stand-alone (host-side slab pruning, tests) when not.
//
// Build-authored.  Semantics per reference line:
//   transformed_distance(p,q)            :6-11   (dx*dx + dy*dy) + dz*dz, no FMA
//   min_distance_to_rectangle(p,b)       :13-25  the reference's y-term multiplies (h-max)*(h-min)
//                                                (typo at :20); this header returns the CORRECT
//                                                value -- deliberate, documented deviation; the
//                                                overload is dead code on the search path.
//   min_distance_to_rectangle(p,b,dists) :27-57  squared distance to the box + per-axis offsets
//   max_distance_to_rectangle(p,b[,d])   :60-90
//   new_distance(dist,old,new,dim)       :92-95  dist + new^2 - old^2
//   transformed_distance(d)              :97     d*d
//   inverse_of_transformed_distance(d)   :99     sqrt(d)
// Any rectangle type with min_coord(int)/max_coord(int) is accepted (CGAL::Kd_tree_rectangle has
// them); pt::Box3 below is the dependency-free one.
#ifndef PT_DISTANCE_H
#define PT_DISTANCE_H

#include <cmath>
#include <vector>

#include "Point.h"

#ifdef PT_WITH_CGAL
#include <CGAL/Dimension.h>
#include <CGAL/Kd_tree_rectangle.h>
#endif

namespace pt {
struct Box3 {
  double lo[3], hi[3];
  double min_coord(int a) const { return lo[a]; }
  double max_coord(int a) const { return hi[a]; }
};
}  // namespace pt

#if defined(__clang__)
#define PT_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define PT_NO_CONTRACT   /* gcc: compile with -ffp-contract=off (the build does) */
#endif

struct Distance {
  typedef Point Query_item;
  typedef double FT;
#ifdef PT_WITH_CGAL
  typedef CGAL::Dimension_tag<3> D;
#endif

  double transformed_distance(const Point& a, const Point& b) const {
    PT_NO_CONTRACT
    const double ex = a.ver[0] - b.ver[0], ey = a.ver[1] - b.ver[1], ez = a.ver[2] - b.ver[2];
    return ex * ex + ey * ey + ez * ez;
  }

  template <class Rect>
  double min_distance_to_rectangle(const Point& p, const Rect& box) const {
    PT_NO_CONTRACT
    double acc = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double gap = axis_gap(p.ver[a], box.min_coord(a), box.max_coord(a));
      acc += gap * gap;
    }
    return acc;
  }

  template <class Rect>
  double min_distance_to_rectangle(const Point& p, const Rect& box, std::vector<double>& dists) {
    PT_NO_CONTRACT
    double acc = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double c = p.ver[a];
      if (c < box.min_coord(a) || c > box.max_coord(a)) {   // untouched otherwise, like the reference
        dists[a] = axis_gap(c, box.min_coord(a), box.max_coord(a));
        acc += dists[a] * dists[a];
      }
    }
    return acc;
  }

  template <class Rect>
  double max_distance_to_rectangle(const Point& p, const Rect& box) const {
    PT_NO_CONTRACT
    const double f0 = axis_far(p.ver[0], box.min_coord(0), box.max_coord(0));
    const double f1 = axis_far(p.ver[1], box.min_coord(1), box.max_coord(1));
    const double f2 = axis_far(p.ver[2], box.min_coord(2), box.max_coord(2));
    return f0 * f0 + f1 * f1 + f2 * f2;
  }

  template <class Rect>
  double max_distance_to_rectangle(const Point& p, const Rect& box, std::vector<double>& dists) {
    PT_NO_CONTRACT
    for (int a = 0; a < 3; ++a) dists[a] = axis_far(p.ver[a], box.min_coord(a), box.max_coord(a));
    return dists[0] * dists[0] + dists[1] * dists[1] + dists[2] * dists[2];
  }

  double new_distance(double& dist, double old_off, double new_off, int /*cutting_dimension*/) const {
    PT_NO_CONTRACT
    return dist + new_off * new_off - old_off * old_off;
  }

  double transformed_distance(double d) const { return d * d; }
  double inverse_of_transformed_distance(double d) { return std::sqrt(d); }

 private:
  // distance from c to the interval [lo, hi] (0 inside)
  static double axis_gap(double c, double lo, double hi) { return c < lo ? lo - c : (c > hi ? c - hi : 0.0); }
  // distance from c to the farther end of [lo, hi]
  static double axis_far(double c, double lo, double hi) { return c >= (lo + hi) / 2.0 ? c - lo : hi - c; }
};

#endif  // PT_DISTANCE_H
