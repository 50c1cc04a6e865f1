// launcher_selftest.cpp -- the process launcher of `pointsTransfer --gpus N` (host/sharded.h: spawn_and_wait) without a GPU:
//   * all children succeed                      -> 0
//   * one child fails while another would run for ten minutes -> the failure's exit code, promptly, and the long runner is gone
//   * a child that ignores SIGTERM              -> killed after the grace period
//   * a child killed by a signal                -> non-zero
// The launcher itself never touches the GPU, so this is the whole of its failure handling.  sharded.h references the C ABI
// (pt_*) only inside functions this program never instantiates a call to at run time; it links against libpt_hip.so like the CLI.
#include <cstdio>
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
  if (fails) return 1;
  std::printf("launcher selftest ok\n");
  return 0;
}
