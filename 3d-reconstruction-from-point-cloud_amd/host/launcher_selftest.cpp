// launcher_selftest.cpp -- the process launcher of `pointsTransfer --gpus N` (host/sharded.h: spawn_and_wait) without a GPU:
//   * all children succeed                      -> 0
//   * one child fails while another would run for ten minutes -> the failure's exit code, promptly, and the long runner is gone
//   * a child that ignores SIGTERM              -> killed after the grace period
//   * a child killed by a signal                -> non-zero
// The launcher itself never touches the GPU, so this is the whole of its failure handling.  sharded.h references the C ABI
// (pt_*) only inside functions this program never instantiates a call to at run time; it links against libpt_hip.so like the CLI.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <string>
#include <vector>

#include "sharded.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static bool alive(const std::string& marker) {          // is any process still running whose argv carries the marker?
  const std::string cmd = "ps -eo args | grep -F -- '" + marker + "' | grep -v grep > /dev/null";
  return std::system(cmd.c_str()) == 0;
}

int main() {
  using sharded::spawn_and_wait;
  using V = std::vector<std::vector<std::string>>;
  const std::string tag = "pt_launcher_selftest_" + std::to_string((long)getpid());
  auto t0 = sharded::clk::now();
  CHECK(spawn_and_wait(V{{"/bin/true"}, {"/bin/true"}, {"/bin/sh", "-c", "sleep 0.2"}}) == 0);
  // a rank dies early, its peer would wait for it "inside a collective" for ten minutes
  t0 = sharded::clk::now();
  const std::string nap = "600.0" + std::to_string((long)getpid());        // (a duration nobody else sleeps for: the marker in argv)
  const int rc = spawn_and_wait(V{{"/bin/sleep", nap}, {"/bin/sh", "-c", "sleep 0.3; exit 7"}}, 2.0);
  CHECK(rc == 7);
  CHECK(sharded::since(t0) < 10.0);
  CHECK(!alive("sleep " + nap));
  // a peer that ignores SIGTERM is killed after the grace period (2 s here)
  t0 = sharded::clk::now();
  const int rc2 = spawn_and_wait(V{{"/bin/sh", "-c", "trap '' TERM; while :; do sleep 1; done # " + tag + "_b"}, {"/bin/false"}}, 2.0);
  CHECK(rc2 == 1);
  CHECK(sharded::since(t0) < 15.0);
  CHECK(!alive(tag + "_b"));
  // death by signal counts as failure
  CHECK(spawn_and_wait(V{{"/bin/sh", "-c", "kill -9 $$"}}) != 0);
  // a command that cannot be started
  CHECK(spawn_and_wait(V{{"/nonexistent/binary"}}) != 0);
  // ---- the file rendezvous of the rank processes (sharded.h: write_all, wait_for_file, Pieces, slab_bounds), no GPU involved:
  //      three pieces of a 1000-point cloud, one of them published late by another thread; a piece with a wrong size, a piece that
  //      does not continue the previous one and a piece that never comes are refused
  {
    std::string dir = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/pt_rdv_selftest.XXXXXX";
    CHECK(mkdtemp(&dir[0]) != nullptr);
    CHECK(sharded::free_bytes(dir) > 0);
    const uint64_t counts[3] = {400, 0, 600};
    std::vector<double> xs(1000), ys(1000), zs(1000);
    std::vector<float> nrm(3000, 0.5f);
    std::vector<uint8_t> rgb(3000, 7);
    for (int i = 0; i < 1000; ++i) { xs[(size_t)i] = (double)((i * 37) % 1000); ys[(size_t)i] = 2.0 * i; zs[(size_t)i] = 0.25 * i; }
    auto publish = [&](int r, uint64_t first, uint64_t cnt, const std::string& where) {
      const uint64_t hdr[2] = {first, cnt};
      return sharded::write_all(where + "/piece_" + std::to_string(r) + ".bin",
                                {{hdr, sizeof hdr}, {xs.data() + first, cnt * 8}, {ys.data() + first, cnt * 8}, {zs.data() + first, cnt * 8}, {nrm.data() + 3 * first, cnt * 12}, {rgb.data() + 3 * first, cnt * 3}});
    };
    CHECK(publish(0, 0, counts[0], dir));
    CHECK(publish(1, 400, counts[1], dir));
    std::thread late([&]() { std::this_thread::sleep_for(std::chrono::milliseconds(300)); (void)publish(2, 400, counts[2], dir); });
    {
      sharded::Pieces c;
      CHECK(c.open(dir, 3, 10.0));
      late.join();
      CHECK(c.n == 1000 && c.p.size() == 3);
      CHECK(c.of(0).first == 0 && c.of(399).first == 0 && c.of(400).first == 400 && c.of(999).first == 400);
      bool same = true;
      for (uint64_t g = 0; g < c.n; ++g) { const sharded::Piece& q = c.of(g); same = same && q.x[g - q.first] == xs[(size_t)g] && q.z[g - q.first] == zs[(size_t)g] && q.rgb[3 * (g - q.first)] == 7; }
      CHECK(same);
      int axis = -1;
      std::vector<double> b;
      sharded::slab_bounds(c, 4, axis, b);
      CHECK(axis == 1 && b.size() == 5 && std::isinf(b[0]) && std::isinf(b[4]));       // y spans 0 .. 1998: the longest axis
      CHECK(b[1] <= b[2] && b[2] <= b[3] && b[1] > 300.0 && b[3] < 1700.0);
    }
    { sharded::Pieces c; CHECK(!c.open(dir, 4, 0.2)); }                                 // piece 3 never comes: time-out, no hang
    {
      const uint64_t hdr[2] = {1000, 5};                                                // claims 5 records, carries none
      CHECK(sharded::write_all(dir + "/piece_3.bin", {{hdr, sizeof hdr}}));
      sharded::Pieces c; CHECK(!c.open(dir, 4, 0.2));
      const uint64_t hdr2[2] = {990, 0};                                                // does not continue piece 2
      CHECK(sharded::write_all(dir + "/piece_3.bin", {{hdr2, sizeof hdr2}}));
      sharded::Pieces c2; CHECK(!c2.open(dir, 4, 0.2));
    }
    CHECK(!sharded::write_all(dir + "/no/such/dir/x", {{xs.data(), 8}}));
    for (int r = 0; r < 4; ++r) std::remove((dir + "/piece_" + std::to_string(r) + ".bin").c_str());
    CHECK(rmdir(dir.c_str()) == 0);                                                     // nothing else was left behind (no stray .part file)
  }
  if (fails) return 1;
  std::printf("launcher selftest ok\n");
  return 0;
}
