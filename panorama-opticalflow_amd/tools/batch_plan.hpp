// Partitioning and gather bookkeeping of the multi-GPU batch driver: overlap pairs are independent
// units (CPU/main.cpp:70,82), so pair i simply runs on device i % ndev (static round-robin) with no collective on the
// data path; results travel to rank 0 round by round (round j moves pair j*ndev + r from rank r).  Plain C++ so that the
// CPU test tier can check it (tests/cpp/batch_plan_test.cpp).
#ifndef PANO_BATCH_PLAN_HPP_
#define PANO_BATCH_PLAN_HPP_
#include <stdexcept>
#include <vector>

namespace pano_batch {

inline std::vector<int> pairs_for_device(int n_pairs, int dev, int ndev) {
  if (ndev <= 0 || dev < 0 || dev >= ndev || n_pairs < 0) throw std::invalid_argument("bad device / pair count");
  std::vector<int> v;
  for (int i = dev; i < n_pairs; i += ndev) v.push_back(i);
  return v;
}
inline int rounds(int n_pairs, int ndev) { return ndev > 0 ? (n_pairs + ndev - 1) / ndev : 0; }
// pair handled by `rank` in gather round `round`, or -1 when that rank idles in the last, partial round
inline int pair_of(int round, int rank, int n_pairs, int ndev) { const int p = round * ndev + rank; return p < n_pairs ? p : -1; }
// where pair p's result lands in rank 0's receive area: (round, block index inside the round's ndev blocks)
struct Slot { int round, block; };
inline Slot slot_of_pair(int pair, int ndev) { return Slot{pair / ndev, pair % ndev}; }

// The same with up to k pairs per device and round (throughput mode: a device solves k of its pairs through one set of launches,
// pf_novel_view_batch_dev, and sends them to rank 0 as one block of k strips).  Ownership is unchanged -- pair i still belongs to
// device i % ndev -- a device's j-th round covers its pairs number j*k .. j*k + k - 1.
inline int rounds_k(int n_pairs, int ndev, int k) {
  if (ndev <= 0 || k <= 0) return 0;
  const int per_dev = (n_pairs + ndev - 1) / ndev;
  return (per_dev + k - 1) / k;
}
// pair in slot `slot` (0..k-1) of `rank`'s block in gather round `round`, or -1 when that slot stays empty
inline int pair_of_k(int round, int rank, int slot, int n_pairs, int ndev, int k) {
  const int p = rank + (round * k + slot) * ndev;
  return p < n_pairs ? p : -1;
}

}  // namespace pano_batch
#endif
