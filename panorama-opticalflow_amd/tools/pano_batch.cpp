// pano_batch -- multi-GPU batch driver for independent overlap pairs (BASELINE config 5).
//
// One host thread + one pf_ctx + one RCCL rank per GPU; pair i runs on GPU i % N (static round-robin, no collective on
// the data path: pairs share no state, CPU/main.cpp:70,82).  The only exchange is the gather of the blended strips
// into rank 0's HBM: grouped ncclSend/ncclRecv on the gather stream (pf_dist_gather_async), double-buffered so that the
// gather of round j overlaps the compute of round j+1.  Host code is C++ over the C ABI; no Python, no torch.
//
//   pano_batch -pairs 8 -size 9000x4000 -flow_alg pixflow_search_20 [-gpus N] [-verify 1] [-in_flight K]
//
// -in_flight K (1..8, default 1): with more pairs than GPUs, each GPU solves K of its pairs through ONE set of kernel launches per
// round (pf_novel_view_batch_dev: the exact sweeps of a lone pair leave most of the chip idle) and sends them as one block.
//
// Inputs are synthetic (textured pair with a smooth displacement, alpha holes at the edges) generated on the host per
// pair and uploaded once; the clock covers compute + gather with inputs resident in HBM.  -verify 1 (default) checks every
// gathered strip against its producer's through device-side checksums: nothing is copied to or hashed on the host inside the clock.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/panoflow.h"
#include "batch_plan.hpp"

namespace {

struct Args { int pairs = 8, cols = 9000, rows = 4000, gpus = 0, verify = 1, in_flight = 1; std::string alg = "pixflow_low"; };

bool parse(int argc, char** argv, Args& a) {
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    while (!k.empty() && k[0] == '-') k.erase(0, 1);
    std::string v;
    const size_t eq = k.find('=');
    if (eq != std::string::npos) { v = k.substr(eq + 1); k = k.substr(0, eq); }
    else if (i + 1 < argc) v = argv[++i];
    else return false;
    if (k == "pairs") a.pairs = atoi(v.c_str());
    else if (k == "size") { if (sscanf(v.c_str(), "%dx%d", &a.cols, &a.rows) != 2) return false; }
    else if (k == "flow_alg") a.alg = v;
    else if (k == "gpus") a.gpus = atoi(v.c_str());
    else if (k == "verify") a.verify = atoi(v.c_str());
    else if (k == "in_flight") a.in_flight = atoi(v.c_str());
    else return false;
  }
  return a.pairs > 0 && a.cols > 0 && a.rows > 0 && a.in_flight >= 1 && a.in_flight <= 8;
}

// deterministic synthetic pair: three sinusoid layers per channel, L = T(x + d/2), R = 1.05 T(x - d/2)
void make_pair(int cols, int rows, int seed, std::vector<uint8_t>& L, std::vector<uint8_t>& R, std::vector<float>& blend) {
  L.assign(size_t(cols) * rows * 4, 0); R.assign(size_t(cols) * rows * 4, 0); blend.assign(size_t(cols) * rows, 0.f);
  const double ph = 0.37 * seed;
  auto tex = [&](double u, double v, int c) {
    return 128.0 + 40.0 * std::sin(0.071 * u + 0.013 * v + ph + c) + 35.0 * std::sin(0.019 * u - 0.047 * v + 2.0 * ph + 0.5 * c) +
           25.0 * std::sin(0.23 * u + 0.31 * v + 3.0 * ph + 0.25 * c);
  };
  const int band = cols / 16;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const double dx = 6.0 * std::sin(3 * M_PI * y / rows) + 3.0 * std::cos(4 * M_PI * x / cols), dy = 1.5 * std::sin(6 * M_PI * x / cols);
      const size_t i = (size_t(y) * cols + x) * 4;
      const bool in = x >= band && x < cols - band;
      for (int c = 0; c < 3; ++c) {
        const double l = tex(x + dx / 2, y + dy / 2, c), r = 1.05 * tex(x - dx / 2, y - dy / 2, c);
        L[i + c] = in ? (uint8_t)std::lround(std::fmin(240.0, std::fmax(16.0, l))) : 0;
        R[i + c] = in ? (uint8_t)std::lround(std::fmin(240.0, std::fmax(16.0, r))) : 0;
      }
      L[i + 3] = R[i + 3] = in ? 255 : 0;
      blend[size_t(y) * cols + x] = float(x) / float(cols - 1);
    }
}

struct Shared {
  Args a; int ndev; int max_pct; unsigned char id[128];
  std::vector<pf_ctx*> ctx;                        // one per device, all created (and checked) before any rank enters RCCL
  std::vector<uint64_t> sum_local, sum_gathered;   // per pair: device-side checksum at its producer / at rank 0 after the gather
  std::vector<double> secs;                        // per device
};

// A rank that fails after the communicator exists cannot simply return: its peers would wait for it forever inside a grouped
// send/recv or an all-reduce.  Nothing here is worth a partial result, so any such failure ends the whole process.
[[noreturn]] void die(int dev, const char* what, const char* msg) {
  fprintf(stderr, "[gpu %d] %s: %s\n", dev, what, msg ? msg : "?");
  fflush(stderr);
  _exit(1);
}

void worker(Shared* s, int dev) {
  const Args& a = s->a;
  const int ndev = s->ndev;
  pf_ctx* ctx = s->ctx[dev];
  pf_dist* dist = pf_dist_init(dev, s->id, dev, ndev);
  if (!dist) die(dev, "pf_dist_init", pf_dist_last_error(nullptr));
  const size_t n = size_t(a.cols) * a.rows, ib = n * 4;
  const std::vector<int> mine = pano_batch::pairs_for_device(a.pairs, dev, ndev);
  const int K = a.in_flight;
  const int nrounds = pano_batch::rounds_k(a.pairs, ndev, K);
  // inputs of all my pairs resident in HBM before the clock starts
  std::vector<const uint8_t*> dL(mine.size()), dR(mine.size());
  void* dBlend = pf_dev_alloc(ctx, n * 4);
  void* dOut[2] = {pf_dev_alloc(ctx, ib * K), pf_dev_alloc(ctx, ib * K)};   // a round's K strips, back to back
  void* dRecv[2] = {nullptr, nullptr};
  if (dev == 0) { dRecv[0] = pf_dev_alloc(ctx, ib * K * ndev); dRecv[1] = pf_dev_alloc(ctx, ib * K * ndev); }
  if (!dBlend || !dOut[0] || !dOut[1] || (dev == 0 && (!dRecv[0] || !dRecv[1]))) die(dev, "pf_dev_alloc", pf_last_error(ctx));
  {
    std::vector<uint8_t> L, R; std::vector<float> blend;
    for (size_t k = 0; k < mine.size(); ++k) {
      make_pair(a.cols, a.rows, 1234 + mine[k], L, R, blend);
      void* l = pf_dev_alloc(ctx, ib); void* r = pf_dev_alloc(ctx, ib);
      if (!l || !r || pf_upload(ctx, l, L.data(), ib) || pf_upload(ctx, r, R.data(), ib)) die(dev, "upload", pf_last_error(ctx));
      dL[k] = static_cast<const uint8_t*>(l); dR[k] = static_cast<const uint8_t*>(r);
      if (k == 0 && pf_upload(ctx, dBlend, blend.data(), n * 4)) die(dev, "upload", pf_last_error(ctx));
    }
    if (mine.empty()) { make_pair(a.cols, a.rows, 1, L, R, blend); if (pf_upload(ctx, dBlend, blend.data(), n * 4)) die(dev, "upload", pf_last_error(ctx)); }
  }
  // Verification (-verify 1) never touches the host inside the clock: every strip is checksummed ON THE DEVICE (pf_checksum_dev,
  // ~25 us per 144 MB) at its producer and again in rank 0's receive area after the gather; 8 bytes per strip come back.
  auto consume = [&](int round) {   // rank 0: the blocks of `round` are in dRecv[round % 2]
    if (dev != 0 || !a.verify) return;
    for (int r = 0; r < ndev; ++r)
      for (int q = 0; q < K; ++q) {
        const int p = pano_batch::pair_of_k(round, r, q, a.pairs, ndev, K);
        if (p < 0) continue;
        if (pf_checksum_dev(ctx, static_cast<char*>(dRecv[round % 2]) + (size_t(r) * K + q) * ib, ib, &s->sum_gathered[p])) die(dev, "checksum", pf_last_error(ctx));
      }
  };
  if (pf_dist_barrier(dist)) die(dev, "barrier", pf_dist_last_error(dist));
  const auto t0 = std::chrono::steady_clock::now();
  for (int j = 0; j < nrounds; ++j) {
    char* out = static_cast<char*>(dOut[j % 2]);
    int count = 0;
    while (count < K && pano_batch::pair_of_k(j, dev, count, a.pairs, ndev, K) >= 0) ++count;
    if (count > 0) {
      const size_t k0 = size_t(j) * K;   // my pairs number k0 .. k0 + count - 1 are handled in round j
      const float* blends[8]; uint8_t* outs[8];
      for (int q = 0; q < count; ++q) { blends[q] = static_cast<const float*>(dBlend); outs[q] = reinterpret_cast<uint8_t*>(out + size_t(q) * ib); }
      const int rc = count == 1
          ? pf_novel_view_dev(ctx, dL[k0], dR[k0], a.cols, a.rows, s->max_pct, blends[0], outs[0], nullptr, nullptr)
          : pf_novel_view_batch_dev(ctx, count, &dL[k0], &dR[k0], a.cols, a.rows, s->max_pct, blends, outs, nullptr, nullptr, count);
      if (rc) die(dev, "pf_novel_view", pf_last_error(ctx));
      if (a.verify)
        for (int q = 0; q < count; ++q)
          if (pf_checksum_dev(ctx, outs[q], ib, &s->sum_local[pano_batch::pair_of_k(j, dev, q, a.pairs, ndev, K)])) die(dev, "checksum", pf_last_error(ctx));
    }
    // all ranks take part in every round (empty slots of the last round carry stale strips, which rank 0 ignores);
    // the call first waits for the previous gather, whose receive area the consumer below then owns
    if (pf_dist_gather_async(dist, out, dRecv[j % 2], ib * K)) die(dev, "gather", pf_dist_last_error(dist));
    if (j > 0) consume(j - 1);
  }
  if (pf_dist_wait(dist)) die(dev, "wait", pf_dist_last_error(dist));
  if (nrounds > 0) consume(nrounds - 1);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  s->secs[dev] = dt;
  if (pf_dist_max(dist, &dt)) die(dev, "max", pf_dist_last_error(dist));
  if (dev == 0) s->secs[0] = dt;   // the job's time: the slowest rank's
  pf_dist_destroy(dist);
  for (const uint8_t* p : dL) pf_dev_free(ctx, const_cast<uint8_t*>(p));
  for (const uint8_t* p : dR) pf_dev_free(ctx, const_cast<uint8_t*>(p));
  pf_dev_free(ctx, dBlend); pf_dev_free(ctx, dOut[0]); pf_dev_free(ctx, dOut[1]); pf_dev_free(ctx, dRecv[0]); pf_dev_free(ctx, dRecv[1]);
}

}  // namespace

int main(int argc, char** argv) {
  Shared s;
  if (!parse(argc, argv, s.a)) { fprintf(stderr, "usage: pano_batch -pairs P -size COLSxROWS -flow_alg pixflow_low|pixflow_search_20 [-gpus N] [-verify 0|1] [-in_flight 1..8]\n"); return 2; }
  s.max_pct = pf_max_percentage_by_name(s.a.alg.c_str());
  if (s.max_pct < 0) { fprintf(stderr, "%s\n", pf_last_error(nullptr)); return 1; }
  const int have = pf_device_count();
  if (have <= 0) { fprintf(stderr, "no HIP device (this driver has no CPU path)\n"); return 1; }
  s.ndev = s.a.gpus > 0 ? s.a.gpus : have;
  if (s.ndev > have) { fprintf(stderr, "%d GPUs requested, %d present\n", s.ndev, have); return 1; }
  if (pf_dist_unique_id(s.id)) { fprintf(stderr, "%s\n", pf_dist_last_error(nullptr)); return 1; }
  s.sum_local.assign(s.a.pairs, 0); s.sum_gathered.assign(s.a.pairs, 0); s.secs.assign(s.ndev, 0.0);
  // every context first: a GPU that cannot be used is reported before any rank enters ncclCommInitRank (where the others would hang)
  s.ctx.assign(s.ndev, nullptr);
  for (int d = 0; d < s.ndev; ++d) {
    s.ctx[d] = pf_create(d, s.a.cols, s.a.rows);
    if (!s.ctx[d]) {
      fprintf(stderr, "[gpu %d] pf_create: %s\n", d, pf_last_error(nullptr));
      for (int e = 0; e < d; ++e) pf_destroy(s.ctx[e]);
      return 1;
    }
  }
  std::vector<std::thread> th;
  for (int d = 0; d < s.ndev; ++d) th.emplace_back(worker, &s, d);
  for (auto& t : th) t.join();
  for (pf_ctx* c : s.ctx) pf_destroy(c);
  int bad = 0;
  if (s.a.verify) for (int p = 0; p < s.a.pairs; ++p) if (!s.sum_local[p] || s.sum_local[p] != s.sum_gathered[p]) { fprintf(stderr, "pair %d: gathered strip differs from its producer's\n", p); ++bad; }
  const double mpix = double(s.a.cols) * s.a.rows * s.a.pairs / 1e6;
  printf("{\"tool\": \"pano_batch\", \"gpus\": %d, \"in_flight_per_gpu\": %d, \"pairs\": %d, \"size\": \"%dx%d\", \"flow_alg\": \"%s\", \"seconds\": %.4f, \"Mpix/s\": %.2f, \"verified_pairs\": %d, \"verification\": \"%s\", \"gather\": \"rccl send/recv to rank 0, overlapped\"}\n",
         s.ndev, s.a.in_flight, s.a.pairs, s.a.cols, s.a.rows, s.a.alg.c_str(), s.secs[0], mpix / s.secs[0], s.a.verify ? s.a.pairs - bad : 0, s.a.verify ? "device-side checksums (pf_checksum_dev), no host copies inside the clock" : "off");
  return bad ? 1 : 0;
}
