// rccl_standin.cpp -- TEST DOUBLE, not a product component and not RCCL.
//
// The GPU pool offers one GPU per box and RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the library's exchange
// (infinitam_amd/csrc/exchange.hip) can never run its N > 1 code there through the real collective library.  This file implements the five
// entry points exchange.hip resolves with dlsym -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString --
// for SEVERAL PROCESSES THAT SHARE ONE GPU, staged through POSIX shared memory on the host, with the stream semantics the caller relies on
// (the gathered table is complete for everything enqueued on `stream` behind the call).  It is loaded instead of librccl.so only when a test
// sets ITM_RCCL_LIBRARY to its path; what it exercises is everything AROUND the collective: rank-major table layout at world > 1, the
// self-check's own-block offset, the ring of batch buffers with a peer that runs at another pace, the bootstrap through a 128-byte id.
// It says nothing about RCCL's own transport, topology or performance.
//
// Fault injection (tests): ITM_STANDIN_ROTATE_RANKS=1 files every rank's block one place further (the table a communicator with a wrong
// rank order would produce): the library's self-check has to notice.
//
// build: hipcc -shared -fPIC -O2 -o librccl_standin.so rccl_standin.cpp -lrt
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 4u << 20;          // per rank and parity: 8 records of 64 KB and room to spare
static double timeout_seconds() {
  static const double t = [] { const char* e = getenv("ITM_STANDIN_TIMEOUT_S"); return (e && atof(e) > 0) ? atof(e) : 60.0; }();
  return t;
}

struct Shared {
  std::atomic<int> attached;
  std::atomic<unsigned long long> written[kMaxRanks];   // number of the last collective whose block of this rank lies in its slot
  std::atomic<unsigned long long> consumed[kMaxRanks];  // number of the last collective of which this rank has read every block
  unsigned char slots[kMaxRanks][2][kSlotBytes];
};
static_assert(std::atomic<unsigned long long>::is_always_lock_free, "counters live in shared memory");

struct Comm {
  Shared* sh = nullptr;
  int n = 0, rank = 0;
  unsigned long long seq = 0;
  bool rotate = false;
};

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <class F> bool wait_until(F&& done) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; !done(); ++spins) {
    if ((spins & 0xffu) == 0xffu) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds()) return false;
      sched_yield();
    }
  }
  return true;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, sizeof id->internal);
  static std::atomic<unsigned> counter{0};
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof id->internal, "/itm_standin_%d_%u_%llx", (int)getpid(), counter.fetch_add(1), (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  char name[129];
  memcpy(name, id.internal, 128); name[128] = 0;
  if (name[0] != '/') return ncclInvalidArgument;
  const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); return ncclSystemError; }     // (a fresh segment reads as zeros: every counter starts at 0)
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm* c = new (std::nothrow) Comm();
  if (!c) { munmap(p, sizeof(Shared)); return ncclSystemError; }
  c->sh = (Shared*)p; c->n = nranks; c->rank = rank;
  const char* rot = getenv("ITM_STANDIN_ROTATE_RANKS");
  c->rotate = rot && rot[0] == '1';
  c->sh->attached.fetch_add(1, std::memory_order_acq_rel);
  const bool all = wait_until([&] { return c->sh->attached.load(std::memory_order_acquire) >= nranks; });
  if (rank == 0) shm_unlink(name);                  // the mappings stay; the name is no longer needed (or never will be)
  if (!all) { munmap(p, sizeof(Shared)); delete c; return ncclSystemError; }
  *comm = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t bytes = sendcount * type_bytes(datatype);
  if (!c || !sendbuff || !recvbuff || bytes == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
  Shared* sh = c->sh;
  const unsigned long long s = ++c->seq;
  const int par = (int)(s & 1u);
  // what was enqueued on the stream before the call has produced the send buffer
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  // my slot of this parity was last used by collective s - 2: every rank must be done reading it
  if (s > 2 && !wait_until([&] { for (int p = 0; p < c->n; ++p) if (sh->consumed[p].load(std::memory_order_acquire) < s - 2) return false; return true; }))
    return ncclSystemError;
  if (hipMemcpy(sh->slots[c->rank][par], sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  sh->written[c->rank].store(s, std::memory_order_release);
  for (int p = 0; p < c->n; ++p) {
    if (!wait_until([&] { return sh->written[p].load(std::memory_order_acquire) >= s; })) return ncclSystemError;
    const int place = c->rotate ? (p + 1) % c->n : p;
    if (hipMemcpy((unsigned char*)recvbuff + (size_t)place * bytes, sh->slots[p][par], bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  sh->consumed[c->rank].store(s, std::memory_order_release);
  return ncclSuccess;      // the table is in place: whatever the caller enqueues on `stream` next sees it
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "stand-in transport: a HIP call failed";
    case ncclSystemError: return "stand-in transport: shared memory or a peer that did not arrive within the time limit";
    case ncclInvalidArgument: return "stand-in transport: invalid argument";
    default: return "stand-in transport: error";
  }
}

}  // extern "C"
