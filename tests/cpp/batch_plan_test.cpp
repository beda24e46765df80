// CPU check of the batch driver's partitioning (panorama-opticalflow_amd/tools/batch_plan.hpp): every pair is owned by
// exactly one device, in round-robin order; gather rounds cover all pairs once; slots are unique and ordered.
#include <cstdio>
#include <set>
#include "../../panorama-opticalflow_amd/tools/batch_plan.hpp"
using namespace pano_batch;
int main() {
  for (int ndev = 1; ndev <= 9; ++ndev)
    for (int n = 0; n <= 40; ++n) {
      std::set<int> seen;
      for (int d = 0; d < ndev; ++d) {
        int last = -1;
        for (int p : pairs_for_device(n, d, ndev)) {
          if (p % ndev != d || p <= last || !seen.insert(p).second) { printf("bad ownership n=%d ndev=%d\n", n, ndev); return 1; }
          last = p;
        }
      }
      if ((int)seen.size() != n) { printf("pairs lost n=%d ndev=%d\n", n, ndev); return 1; }
      std::set<int> moved;
      for (int j = 0; j < rounds(n, ndev); ++j)
        for (int r = 0; r < ndev; ++r) {
          const int p = pair_of(j, r, n, ndev);
          if (p < 0) { if (j != rounds(n, ndev) - 1) { printf("idle rank before the last round\n"); return 1; } continue; }
          const Slot s = slot_of_pair(p, ndev);
          if (s.round != j || s.block != r || !moved.insert(p).second) { printf("bad slot\n"); return 1; }
        }
      if ((int)moved.size() != n) { printf("gather rounds do not cover all pairs\n"); return 1; }
    }
  // k pairs per device and round: every pair moves exactly once, in its owner's block, and k = 1 is the plain plan
  for (int ndev = 1; ndev <= 5; ++ndev)
    for (int k = 1; k <= 4; ++k)
      for (int n = 0; n <= 30; ++n) {
        std::set<int> moved;
        for (int j = 0; j < rounds_k(n, ndev, k); ++j)
          for (int r = 0; r < ndev; ++r)
            for (int q = 0; q < k; ++q) {
              const int p = pair_of_k(j, r, q, n, ndev, k);
              if (p < 0) continue;
              if (p % ndev != r || !moved.insert(p).second) { printf("bad k-plan n=%d ndev=%d k=%d\n", n, ndev, k); return 1; }
              if (k == 1 && p != pair_of(j, r, n, ndev)) { printf("k = 1 differs from the plain plan\n"); return 1; }
            }
        if ((int)moved.size() != n) { printf("k-plan loses pairs n=%d ndev=%d k=%d\n", n, ndev, k); return 1; }
        if (pair_of_k(rounds_k(n, ndev, k), 0, 0, n, ndev, k) >= 0) { printf("k-plan needs more rounds\n"); return 1; }
      }
  try { pairs_for_device(4, 3, 3); printf("bad device accepted\n"); return 1; } catch (const std::invalid_argument&) {}
  printf("ok\n");
  return 0;
}
