// Multi-GPU side of the C ABI (include/panoflow.h, "pf_dist_*"): the path's ONLY exchange -- the final gather of
// per-pair results to rank 0 -- as grouped ncclSend/ncclRecv on a stream of its own, so that the
// gather of pair k overlaps the compute of pair k+1.  Overlap pairs are independent units (CPU/main.cpp:70,82: no state
// shared between Stitchtools / NovelViewGenerator objects), so there is no collective on the data path itself.
//
// RCCL is bound at run time (dlopen): single-GPU users of libpanoflow.so carry no dependency on it, and a process
// that already loaded a copy (e.g. PyTorch's) keeps using that one.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/panoflow.h"
#include "pf_common.hpp"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

void rccl_load(Rccl& r);
Rccl* rccl() {   // bound once, also when the first callers are several rank threads of one process
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] { rccl_load(r); });
  return &r;
}
void rccl_load(Rccl& r) {
  // A process must hold ONE copy of RCCL (two copies interpose each other's globals): if one is already mapped -- e.g. the
  // one PyTorch ships and loads for torch.distributed -- bind to exactly that file; otherwise load ROCm's.
  std::string loaded;
  dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* out) -> int {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) { *static_cast<std::string*>(out) = info->dlpi_name; return 1; }
    return 0;
  }, &loaded);
  if (!loaded.empty()) r.h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_LOCAL);
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { if (r.h) break; r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
  if (!r.h) { r.err = std::string("cannot load RCCL: ") + dlerror(); return; }
#define PF_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, sym)); if (!r.field) { r.err = std::string("RCCL symbol missing: ") + sym; r.h = nullptr; return; }
  PF_SYM(GetUniqueId, "ncclGetUniqueId") PF_SYM(CommInitRank, "ncclCommInitRank") PF_SYM(CommDestroy, "ncclCommDestroy") PF_SYM(GroupStart, "ncclGroupStart")
  PF_SYM(GroupEnd, "ncclGroupEnd") PF_SYM(Send, "ncclSend") PF_SYM(Recv, "ncclRecv") PF_SYM(AllReduce, "ncclAllReduce") PF_SYM(GetErrorString, "ncclGetErrorString")
#undef PF_SYM
}

thread_local std::string g_derr;
int dfail(int code, const std::string& m) { g_derr = m; return code; }

}  // namespace

struct pf_dist {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  double* d_scalar = nullptr;   // device word for the tiny all-reduce (barrier / max)
  bool pending = false;
  std::string err;
};

#define NCHK(d, expr)                                                                                                          \
  do {                                                                                                                         \
    ncclResult_t r_ = (expr);                                                                                                  \
    if (r_ != ncclSuccess) { (d)->err = std::string(#expr) + " failed: " + rccl()->GetErrorString(r_); g_derr = (d)->err; return PF_ERR_DEVICE; } \
  } while (0)
#define DHIP(d, expr)                                                                                                          \
  do {                                                                                                                         \
    hipError_t e_ = (expr);                                                                                                    \
    if (e_ != hipSuccess) { (d)->err = std::string(#expr) + " failed: " + hipGetErrorString(e_); g_derr = (d)->err; return PF_ERR_DEVICE; } \
  } while (0)

extern "C" {

int pf_dist_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r->h) return dfail(PF_ERR_DEVICE, r->err);
  if (!id128) return dfail(PF_ERR_ARG, "null pointer");
  ncclUniqueId id;
  const ncclResult_t rc = r->GetUniqueId(&id);
  if (rc != ncclSuccess) return dfail(PF_ERR_DEVICE, std::string("ncclGetUniqueId failed: ") + r->GetErrorString(rc));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, sizeof id);
  return 0;
}

pf_dist* pf_dist_init(int device, const void* id128, int rank, int world) {
  Rccl* r = rccl();
  if (!r->h) { dfail(PF_ERR_DEVICE, r->err); return nullptr; }
  if (!id128 || world < 1 || rank < 0 || rank >= world) { dfail(PF_ERR_ARG, "bad rank/world/id"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { dfail(PF_ERR_DEVICE, "hipSetDevice failed"); return nullptr; }
  pf_dist* d = new pf_dist();
  d->device = device; d->rank = rank; d->world = world;
  ncclUniqueId id; memcpy(&id, id128, sizeof id);
  ncclResult_t rc = r->CommInitRank(&d->comm, world, id, rank);
  if (rc != ncclSuccess) { dfail(PF_ERR_DEVICE, std::string("ncclCommInitRank failed: ") + r->GetErrorString(rc)); delete d; return nullptr; }
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->done, hipEventDisableTiming) != hipSuccess ||
      hipMalloc((void**)&d->d_scalar, 64) != hipSuccess) {
    dfail(PF_ERR_DEVICE, "stream/event/alloc for the gather failed");
    r->CommDestroy(d->comm); delete d; return nullptr;
  }
  return d;
}

void pf_dist_destroy(pf_dist* d) {
  if (!d) return;
  hipSetDevice(d->device);
  if (d->stream) hipStreamSynchronize(d->stream);
  if (d->comm) rccl()->CommDestroy(d->comm);
  if (d->done) hipEventDestroy(d->done);
  if (d->stream) hipStreamDestroy(d->stream);
  if (d->d_scalar) hipFree(d->d_scalar);
  delete d;
}

const char* pf_dist_last_error(const pf_dist* d) { return d ? d->err.c_str() : g_derr.c_str(); }

// Gather pattern: every rank sends `bytes` from d_send to rank 0; rank 0 receives rank r's block at d_recv_all + r*bytes
// (its own block too: one grouped call, 7 peers -> 7 distinct xGMI links into rank 0).  Asynchronous: returns once the
// operations are enqueued on the gather stream.  d_send must be complete (every pf_* compute call is synchronous on
// return) and must not be overwritten before pf_dist_wait(); at most one gather is in flight (a second call waits).
int pf_dist_gather_async(pf_dist* d, const void* d_send, void* d_recv_all, size_t bytes) {
  if (!d || !d_send || (d->rank == 0 && !d_recv_all)) return dfail(PF_ERR_ARG, "bad argument");
  Rccl* r = rccl();
  DHIP(d, hipSetDevice(d->device));
  if (d->pending) { DHIP(d, hipEventSynchronize(d->done)); d->pending = false; }
  // One grouped call; a block goes as pieces of at most 256 MiB (a single 1.15 GB send/recv -- eight 9000x4000 strips of the throughput
  // mode -- arrived incomplete with RCCL 2.26: only the first ~600 MB matched, measured with tools/pano_batch -in_flight 8).
  constexpr size_t kPiece = size_t(256) << 20;
  NCHK(d, r->GroupStart());
  ncclResult_t rc = ncclSuccess;
  for (size_t o = 0; o < bytes && rc == ncclSuccess; o += kPiece) {
    const size_t nbytes = bytes - o < kPiece ? bytes - o : kPiece;
    rc = r->Send(static_cast<const char*>(d_send) + o, nbytes, ncclUint8, 0, d->comm, d->stream);
    if (d->rank == 0)
      for (int p = 0; p < d->world && rc == ncclSuccess; ++p)
        rc = r->Recv(static_cast<char*>(d_recv_all) + size_t(p) * bytes + o, nbytes, ncclUint8, p, d->comm, d->stream);
  }
  const ncclResult_t re = r->GroupEnd();   // always: an error inside the group must not leave it open
  if (rc != ncclSuccess || re != ncclSuccess) {
    d->err = std::string("grouped ncclSend/ncclRecv failed: ") + r->GetErrorString(rc != ncclSuccess ? rc : re); g_derr = d->err;
    return PF_ERR_DEVICE;
  }
  DHIP(d, hipEventRecord(d->done, d->stream));
  d->pending = true;
  return 0;
}

int pf_dist_wait(pf_dist* d) {
  if (!d) return dfail(PF_ERR_ARG, "null");
  DHIP(d, hipSetDevice(d->device));
  if (d->pending) { DHIP(d, hipEventSynchronize(d->done)); d->pending = false; }
  return 0;
}

// max over ranks of a host scalar (the job's time is the slowest rank's); doubles as the barrier
int pf_dist_max(pf_dist* d, double* value) {
  if (!d || !value) return dfail(PF_ERR_ARG, "null");
  Rccl* r = rccl();
  DHIP(d, hipSetDevice(d->device));
  if (int e = pf_dist_wait(d)) return e;
  DHIP(d, hipMemcpyAsync(d->d_scalar, value, 8, hipMemcpyHostToDevice, d->stream));
  NCHK(d, r->AllReduce(d->d_scalar, d->d_scalar, 1, ncclDouble, ncclMax, d->comm, d->stream));
  DHIP(d, hipMemcpyAsync(value, d->d_scalar, 8, hipMemcpyDeviceToHost, d->stream));
  DHIP(d, hipStreamSynchronize(d->stream));
  return 0;
}

int pf_dist_barrier(pf_dist* d) { double v = 0; return pf_dist_max(d, &v); }

}  // extern "C"
