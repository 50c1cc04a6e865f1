// Known-answer checks of the build-authored contract headers (include/Point.h, include/Distance.h) against
// the values SURVEY.md 8c records for the reference's headers:
//   sizeof(Point)==80, offsets 0/24/48/64/72; td((0,0,0),(1,2,3))=14;
//   q=(.5,3,.5) vs [0,1]^3: min (3-arg) = 4 with dists = (.,2,.), max = 9.5;
//   the 2-arg min returns the CORRECT 4 here (the reference's typo at Distance.h:20 yields 6).
// Exit code 0 = all good.  Runs on the CPU; built by host/Makefile.
#include <cmath>
#include <cstdio>
#include <vector>

#include "Distance.h"

static int fails = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++fails; } \
  } while (0)

int main() {
  Distance dist;
  Point o(0, 0, 0), p(1, 2, 3), q(0.5, 3, 0.5);
  pt::Box3 unit{{0, 0, 0}, {1, 1, 1}};
  CHECK(dist.transformed_distance(o, p) == 14.0);
  CHECK(dist.min_distance_to_rectangle(q, unit) == 4.0);
  std::vector<double> d(3, -1.0);
  CHECK(dist.min_distance_to_rectangle(q, unit, d) == 4.0);
  CHECK(d[0] == -1.0 && d[1] == 2.0 && d[2] == -1.0);   // axes inside the box are left untouched, like the reference
  CHECK(dist.max_distance_to_rectangle(q, unit) == 9.5);
  CHECK(dist.max_distance_to_rectangle(q, unit, d) == 9.5 && d[0] == 0.5 && d[1] == 3.0 && d[2] == 0.5);
  double base = 10.0;
  CHECK(dist.new_distance(base, 1.0, 3.0, 0) == 18.0);
  CHECK(dist.transformed_distance(3.0) == 9.0);
  CHECK(dist.inverse_of_transformed_distance(9.0) == 3.0);

  Point a(1, 2, 3, 0, 0, 1, 10, 20, 30, 0.25, 0.75), b(1, 2, 3);
  CHECK(a == b && !(a != b));                    // equality is xyz-only
  CHECK(a.r() == 10 && a.g() == 20 && a.b() == 30 && a.u() == 0.25 && a.v() == 0.75 && a.nz() == 1.0);
  a.x() = 5.0;
  CHECK(a.x() == 5.0 && a != b);
  Point z;
  CHECK(z.x() == 0 && z.y() == 0 && z.z() == 0);
  Construct_coord_iterator it;
  CHECK(it(a) == a.ver && it(a, 0) - it(a) == 3);
  CHECK(Point(1, 2, 3, 0, 0, 1, 1, 2, 3).detail() == "1.000000 2.000000 3.000000 0.000000 0.000000 1.000000 1 2 3");
  std::printf(fails ? "contract selftest: %d failure(s)\n" : "contract selftest ok\n", fails);
  return fails ? 1 : 0;
}
