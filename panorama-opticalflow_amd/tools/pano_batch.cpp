// pano_batch -- multi-GPU batch driver for independent overlap pairs (BASELINE config 5).
//
// One host thread + one pf_ctx + one RCCL rank per GPU; pair i runs on GPU i % N (static round-robin, no collective on
// the data path: pairs share no state, CPU/main.cpp:70,82).  The only exchange is the gather of the blended strips
// into rank 0's HBM: grouped ncclSend/ncclRecv on the gather stream (pf_dist_gather_async), double-buffered so that the
// gather of round j overlaps the compute of round j+1.  Host code is C++ over the C ABI; no Python, no torch.
//
//   pano_batch -pairs 8 -size 9000x4000 -flow_alg pixflow_search_20 [-gpus N] [-verify 1] [-in_flight K] [-golden DIR]
//
// -golden DIR: self-validation against the oracle fixtures (BASELINE config 5).  The inputs of pair p are read from
// DIR/pair_<1234+p>_L.bgra, _R.bgra and DIR/blend.f32 (written by tests/golden/export_dense_inputs.py: the C++ generator below cannot
// reproduce synth.py's torch arithmetic) and, AFTER the clock has stopped, every rank recomputes each of its pairs once more, requires
// the device checksum of the recomputed strip to equal the one recorded for the timed strip, downloads it and compares its SHA-256
// with DIR/dense_<size>.sha256.txt (the oracle's blended strip for that seed).  "golden_pairs_ok" in the JSON line; exit code 1 on a miss.
//
// -in_flight K (1..16, default 1): with more pairs than GPUs, each GPU solves K of its pairs through ONE set of kernel launches per
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
#include <algorithm>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/panoflow.h"
#include "batch_plan.hpp"

namespace {

// ---- SHA-256 (FIPS 180-4), for -golden ----
struct Sha256 {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint8_t buf[64]; size_t fill = 0; uint64_t total = 0;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
        0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
        0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
        0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t(p[4 * i]) << 24) | (uint32_t(p[4 * i + 1]) << 16) | (uint32_t(p[4 * i + 2]) << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    total += n;
    if (fill) { const size_t k = std::min(n, 64 - fill); memcpy(buf + fill, p, k); fill += k; p += k; n -= k; if (fill == 64) { block(buf); fill = 0; } }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) { memcpy(buf, p, n); fill = n; }
  }
  std::string hex() {
    const uint64_t bits = total * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = uint8_t(bits >> (56 - 8 * i));
    update(len, 8);
    char out[65];
    for (int i = 0; i < 8; ++i) snprintf(out + 8 * i, 9, "%08x", h[i]);
    return std::string(out, 64);
  }
};
std::string sha256_of(const void* p, size_t n) { Sha256 s; s.update(p, n); return s.hex(); }

bool read_file(const std::string& path, void* dst, size_t bytes) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t got = fread(dst, 1, bytes, f);
  const bool more = fgetc(f) != EOF;
  fclose(f);
  return got == bytes && !more;
}
// DIR/dense_<cols>x<rows>.sha256.txt: "seed sha_L sha_R sha_blend sha_flow_l2r sha_flow_r2l sha_strip" per line ('#' = comment)
bool golden_line(const std::string& dir, int cols, int rows, int seed, std::string sha[6]) {
  char name[64]; snprintf(name, sizeof name, "/dense_%dx%d.sha256.txt", cols, rows);
  FILE* f = fopen((dir + name).c_str(), "r");
  if (!f) return false;
  char line[1024]; bool found = false;
  while (!found && fgets(line, sizeof line, f)) {
    if (line[0] == '#') continue;
    int sd = 0; char h[6][80];
    if (sscanf(line, "%d %79s %79s %79s %79s %79s %79s", &sd, h[0], h[1], h[2], h[3], h[4], h[5]) == 7 && sd == seed) { for (int i = 0; i < 6; ++i) sha[i] = h[i]; found = true; }
  }
  fclose(f);
  return found;
}

struct Args { int pairs = 8, cols = 9000, rows = 4000, gpus = 0, verify = 1, in_flight = 1; std::string alg = "pixflow_low", golden; };

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
    else if (k == "golden") a.golden = v;
    else if (k == "sha256") {   // self-test of the SHA-256 used by -golden (tests/test_batch_plan.py): prints the digest of a file
      FILE* f = fopen(v.c_str(), "rb");
      if (!f) return false;
      Sha256 h; std::vector<uint8_t> b(1 << 16); size_t n;
      while ((n = fread(b.data(), 1, b.size(), f)) > 0) h.update(b.data(), n);
      fclose(f);
      printf("%s\n", h.hex().c_str());
      exit(0);
    }
    else return false;
  }
  return a.pairs > 0 && a.cols > 0 && a.rows > 0 && a.in_flight >= 1 && a.in_flight <= 16;
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
  std::vector<int> golden_ok;                      // per pair (-golden): 1 = the strip is the oracle fixture's, 0 = it is not
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
      if (a.golden.empty()) make_pair(a.cols, a.rows, 1234 + mine[k], L, R, blend);
      else {
        // the fixture's inputs (written by tests/golden/export_dense_inputs.py), checked against the fixture's input hashes
        L.resize(ib); R.resize(ib); blend.resize(n);
        const std::string base = a.golden + "/pair_" + std::to_string(1234 + mine[k]);
        std::string sha[6];
        if (!read_file(base + "_L.bgra", L.data(), ib) || !read_file(base + "_R.bgra", R.data(), ib) || !read_file(a.golden + "/blend.f32", blend.data(), n * 4)) die(dev, "-golden", "input files missing or of the wrong size (tests/golden/export_dense_inputs.py writes them)");
        if (!golden_line(a.golden, a.cols, a.rows, 1234 + mine[k], sha)) die(dev, "-golden", "no fixture line for this pair's seed");
        if (sha256_of(L.data(), ib) != sha[0] || sha256_of(R.data(), ib) != sha[1] || sha256_of(blend.data(), n * 4) != sha[2]) die(dev, "-golden", "the input files are not the fixture's inputs");
      }
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
      const float* blends[16]; uint8_t* outs[16];
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
  if (!a.golden.empty() && a.verify) {
    // off the clock: recompute each of my pairs once more; same device checksum as the timed strip + the fixture's SHA-256 = the timed
    // strip was the oracle's
    std::vector<uint8_t> host(ib);
    for (size_t k = 0; k < mine.size(); ++k) {
      const int p = mine[k];
      uint64_t sum = 0; std::string sha[6];
      if (pf_novel_view_dev(ctx, dL[k], dR[k], a.cols, a.rows, s->max_pct, static_cast<const float*>(dBlend), static_cast<uint8_t*>(dOut[0]), nullptr, nullptr)) die(dev, "pf_novel_view", pf_last_error(ctx));
      if (pf_checksum_dev(ctx, dOut[0], ib, &sum) || pf_download(ctx, host.data(), dOut[0], ib)) die(dev, "golden check", pf_last_error(ctx));
      s->golden_ok[p] = golden_line(a.golden, a.cols, a.rows, 1234 + p, sha) && sum == s->sum_local[p] && sha256_of(host.data(), ib) == sha[5];
    }
  }
  pf_dist_destroy(dist);
  for (const uint8_t* p : dL) pf_dev_free(ctx, const_cast<uint8_t*>(p));
  for (const uint8_t* p : dR) pf_dev_free(ctx, const_cast<uint8_t*>(p));
  pf_dev_free(ctx, dBlend); pf_dev_free(ctx, dOut[0]); pf_dev_free(ctx, dOut[1]); pf_dev_free(ctx, dRecv[0]); pf_dev_free(ctx, dRecv[1]);
}

}  // namespace

int main(int argc, char** argv) {
  Shared s;
  if (!parse(argc, argv, s.a)) { fprintf(stderr, "usage: pano_batch -pairs P -size COLSxROWS -flow_alg pixflow_low|pixflow_search_20 [-gpus N] [-verify 0|1] [-in_flight 1..16] [-golden DIR]\n"); return 2; }
  s.max_pct = pf_max_percentage_by_name(s.a.alg.c_str());
  if (s.max_pct < 0) { fprintf(stderr, "%s\n", pf_last_error(nullptr)); return 1; }
  const int have = pf_device_count();
  if (have <= 0) { fprintf(stderr, "no HIP device (this driver has no CPU path)\n"); return 1; }
  s.ndev = s.a.gpus > 0 ? s.a.gpus : have;
  if (s.ndev > have) { fprintf(stderr, "%d GPUs requested, %d present\n", s.ndev, have); return 1; }
  if (pf_dist_unique_id(s.id)) { fprintf(stderr, "%s\n", pf_dist_last_error(nullptr)); return 1; }
  s.sum_local.assign(s.a.pairs, 0); s.sum_gathered.assign(s.a.pairs, 0); s.secs.assign(s.ndev, 0.0); s.golden_ok.assign(s.a.pairs, 0);
  if (!s.a.golden.empty() && s.max_pct != 0) { fprintf(stderr, "-golden: the fixtures are pixflow_low solves\n"); return 2; }
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
  int golden = -1;
  if (!s.a.golden.empty() && s.a.verify) { golden = 0; for (int p = 0; p < s.a.pairs; ++p) golden += s.golden_ok[p]; if (golden != s.a.pairs) { fprintf(stderr, "-golden: %d of %d strips are the oracle fixture's\n", golden, s.a.pairs); ++bad; } }
  const double mpix = double(s.a.cols) * s.a.rows * s.a.pairs / 1e6;
  printf("{\"tool\": \"pano_batch\", \"gpus\": %d, \"in_flight_per_gpu\": %d, \"pairs\": %d, \"size\": \"%dx%d\", \"flow_alg\": \"%s\", \"seconds\": %.4f, \"Mpix/s\": %.2f, \"verified_pairs\": %d, \"verification\": \"%s\", \"golden_pairs_ok\": %d, \"gather\": \"rccl send/recv to rank 0, overlapped\"}\n",
         s.ndev, s.a.in_flight, s.a.pairs, s.a.cols, s.a.rows, s.a.alg.c_str(), s.secs[0], mpix / s.secs[0], s.a.verify ? s.a.pairs - bad : 0, s.a.verify ? "device-side checksums (pf_checksum_dev), no host copies inside the clock" : "off", golden);
  return bad ? 1 : 0;
}
