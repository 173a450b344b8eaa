// exchange.hip -- the per-frame exchange of visible-block records between depth streams (BASELINE configs[3]),
// issued from the library: no interpreter and no tensor framework on the per-frame path.
//
// A scene never reads another stream's data, so N streams shard one per GPU with no collective on the fusion path.  What every
// rank publishes per frame is one fixed-size record { float M_d[16]; int32 noVisibleEntries; int32 ids[max_ids] (padded with -1) },
// written on the FRAME stream by a 3 us copy kernel right behind the frame's kernels (itm_export_visible_record).  Every `batch`
// frames the records of the batch are all-gathered with RCCL on a SIDE stream that waits for the last copy.  The batch buffers form
// a ring of eight SLOTS, each with its own send buffer AND its own gathered table; before a slot is written again the HOST checks that
// the collective which last used it has finished (eight batches earlier: the check returns at once unless the host runs that far ahead
// of the GPU) -- the frame stream itself never waits for a collective, and the frame THREAD does not issue one: a thread of the
// exchange's own does (below).  The one event recorded on the frame stream per batch uses a DEVICE-scope release: the default
// system-scope release of hipEventRecord writes back the L2s, and the next frame's kernels then start on cold caches (measured:
// per-frame exchange 8.0 k frames/s against 11.2 k without exchange, whoever performed the collective).  RCCL has no all-gather-v,
// hence the fixed record size.  xGMI is point-to-point, so one 64 KB x batch all-gather per GPU is latency bound; batching trades
// record age (at most `batch` frames) for fewer collectives.
//
// HAND-OFF TO A DEVICE-SIDE CONSUMER ("a per-GPU global visibility table").  itm_exchange_acquire makes a consumer stream
// wait for the newest collective that has been issued and returns ITS slot's table; the slot is the consumer's until it releases it
// (itm_exchange_release, or the next acquire): the ring skips a held slot, and the collective that next uses a released slot waits, on
// the side stream, for the event the release recorded behind the consumer's reads.  So a kernel that reads the table on the consumer
// stream never sees a table that a later collective is writing, however far the frames run ahead (round 4 had ONE gathered table for
// all eight batches in flight and handed out its raw pointer).
//
// RCCL is loaded with dlopen when the first exchange is created: hosts that never exchange do not pay for it, and the library
// has no link-time dependency on it.  The communicator is bootstrapped from a 128-byte id that rank 0 obtains from
// itm_exchange_unique_id and the host distributes by whatever channel it has (a file, MPI, torch.distributed ...).
#include <dlfcn.h>
#include <rccl/rccl.h>              // types and prototypes only: the entry points are resolved with dlsym, nothing links against RCCL

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include "itm_internal.h"

namespace itm {

int g_debug_exchange_device_copy = 0;     // debug key: a ONE-rank exchange replaces ncclAllGather by a device copy (read when the exchange is created)
int g_debug_exchange_corrupt_word = -1;   // debug key: the self-check sees this word of the own block flipped (read when the exchange is created)

constexpr int kRecordHeader = 17;   // 16 floats of pose + the count

// the handful of RCCL entry points used, resolved at run time; their types are the header's own declarations, so a
// signature that drifts in a later RCCL fails to compile here instead of misbehaving on the 8-GPU node
struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  void* lib = nullptr;
  bool ok = false;
};
static_assert(sizeof(ncclUniqueId) == 128, "itm_exchange_unique_id hands out 128 bytes");

static void load_rccl(Rccl& r);
static Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] { load_rccl(r); });      // exchanges may be created from several host threads (one per stream)
  return r;
}
static void load_rccl(Rccl& r) {
  // ITM_RCCL_LIBRARY names the collective library to load instead (a site's own RCCL build; the tests' stand-in transport for several
  // ranks on ONE GPU, tests/cpp/rccl_standin.cpp).  When it is set nothing else is tried -- a path that does not load is an error --
  // and the substitution is announced on stderr: the collective library of a production run is never replaced silently.
  const char* forced = getenv("ITM_RCCL_LIBRARY");
  if (forced && forced[0]) fprintf(stderr, "libitmhip: ITM_RCCL_LIBRARY is set: loading the collective library '%s' instead of librccl.so\n", forced);
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  if (forced && forced[0]) r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
  else for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
  if (!r.lib) return;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
}

// SELF-CHECK (every world size; ITM_EXCHANGE_SELF_CHECK=0 switches it off): behind every collective, on the side stream, the block
// of the gathered table that belongs to THIS rank is compared with the batch buffer this rank sent.  Whatever the other ranks'
// blocks hold cannot be known here, but a communicator that was built with the wrong rank order, a count that disagrees between
// ranks, a buffer reused too early or a collective that did not run all show up in the own block -- so the first run on hardware
// this library has never seen (the pool offers one GPU per box: no collective with more than one rank has ever executed) reports
// corruption instead of numbers.  The words that differ are counted in page-locked host memory, which the frame thread reads at
// the next step: ITM_ERR_DEVICE.  Off the frame stream; one 64 KB compare per collective.
struct SelfCheck { int32_t collectives, mismatchedWords, firstBadWord, pad; };
__global__ void __launch_bounds__(256) exchange_self_check_kernel(const int32_t* __restrict__ sent, const int32_t* __restrict__ gatheredOwn, size_t count,
                                                                  SelfCheck* __restrict__ out, int corruptWord) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) __hip_atomic_fetch_add(&out->collectives, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (i >= count) return;
  int32_t got = gatheredOwn[i];
  if ((long long)i == (long long)corruptWord) got ^= 0x5a5a5a5a;          // test hook: ITM_EXCHANGE_SELF_CHECK_CORRUPT=<word>
  if (got != sent[i]) {
    if (__hip_atomic_fetch_add(&out->mismatchedWords, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) out->firstBadWord = (int32_t)i;
  }
}

// test stand-in for a device-side consumer of the table (itm_debug_checksum): one workgroup, `rounds` passes
__global__ void __launch_bounds__(256) exchange_checksum_kernel(const int32_t* __restrict__ src, size_t words, int rounds, unsigned long long* __restrict__ dst) {
  __shared__ unsigned long long part[256];
  unsigned long long acc = 0;
  for (int r = 0; r < rounds; ++r)
    for (size_t i = threadIdx.x; i < words; i += 256) acc += (unsigned long long)(uint32_t)__builtin_nontemporal_load(src + i) * (unsigned long long)(i % 1021 + 1);
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned long long t = 0; for (int i = 0; i < 256; ++i) t += part[i]; dst[0] = t; }
}

static int rccl_fail(ncclResult_t code, const char* what) {
  char msg[256];
  snprintf(msg, sizeof msg, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(code) : "RCCL error");
  return set_error(ITM_ERR_DEVICE, msg);
}

}  // namespace itm

struct itm_exchange {
  int world = 1, rank = 0, maxIds = 0, batch = 1;
  size_t words = 0;                       // per record
  ncclComm_t comm = nullptr;              // a communicator for EVERY world size, one rank included; null only behind ITM_EXCHANGE_DEVICE_COPY=1 (debug)
  static constexpr int kRing = 8;
  int32_t* buffers[kRing] = {};           // batch records each
  int32_t* gathered[kRing] = {};          // per slot: world x batch records, rank-major, then frame of the batch
  hipStream_t side = nullptr;
  hipEvent_t copied[kRing] = {}, released[kRing] = {}, consumed[kRing] = {};
  // (all guarded by m) which batch a slot's table holds (-1: none yet); the slot a consumer holds (-1: none); slots whose last holder
  // recorded `consumed` behind its reads: the next collective into such a slot waits for that event first; the newest slot whose
  // collective has been put on the side stream
  long long batchOf[kRing];
  bool consumedPending[kRing] = {};
  int held = -1, newest = -1;
  int cur = 0;                            // the slot the frame thread is filling
  long long batchNo = 0;
  // a batch buffer is FREE, then QUEUED (its last record's event is recorded, the issuer has been told), then ISSUED (the collective and
  // its release event are on the side stream), then FREE again once the frame thread has seen that event complete
  enum : int { kFree = 0, kQueued = 1, kIssued = 2 };
  std::atomic<int> state[kRing];
  long long frame = 0;
  // The collectives are issued by a thread of the exchange's own: putting an all-gather on a stream costs the host 60-80 us (RCCL's
  // launch path), more than submitting a whole frame (16 us).  The frame thread records the event behind the record's copy and hands
  // the slot over (issued from the frame thread, as rounds 2-3 did: per-frame exchange 10.1 k frames/s instead of 10.8-10.9 k).
  int device = 0;
  std::thread issuer;
  std::mutex m;
  std::condition_variable cv;
  std::deque<int> queue;
  bool stop = false;
  std::atomic<int> issuerFailed{0};
  std::string issuerMessage;
  itm::SelfCheck* check = nullptr;        // page-locked host memory (mapped); null = self-check off
  itm::SelfCheck* checkDev = nullptr;
  int corruptWord = -1;                   // test hook (debug key ITM_DEBUG_EXCHANGE_CORRUPT_WORD, read when the exchange is created)
};

using namespace itm;

static void free_exchange(itm_exchange* x) {
  if (!x) return;
  if (x->issuer.joinable()) {
    { std::lock_guard<std::mutex> g(x->m); x->stop = true; }
    x->cv.notify_all();
    x->issuer.join();
  }
  if (x->side) (void)hipStreamSynchronize(x->side);
  if (x->comm) rccl().CommDestroy(x->comm);
  for (int b = 0; b < itm_exchange::kRing; ++b) {
    if (x->buffers[b]) (void)hipFree(x->buffers[b]);
    if (x->gathered[b]) (void)hipFree(x->gathered[b]);
    if (x->copied[b]) (void)hipEventDestroy(x->copied[b]);
    if (x->released[b]) (void)hipEventDestroy(x->released[b]);
    if (x->consumed[b]) (void)hipEventDestroy(x->consumed[b]);
  }
  if (x->check) (void)hipHostFree(x->check);
  if (x->side) (void)hipStreamDestroy(x->side);
  delete x;
}

// the collective of batch buffer b and its release event, on the side stream behind the event of the batch's last record
static int issue_collective(itm_exchange* x, int b) {
  const size_t count = x->words * (size_t)x->batch;
  ITM_HIP(hipStreamWaitEvent(x->side, x->copied[b], 0));
  bool waitConsumer;
  { std::lock_guard<std::mutex> g(x->m); waitConsumer = x->consumedPending[b]; x->consumedPending[b] = false; }
  if (waitConsumer) ITM_HIP(hipStreamWaitEvent(x->side, x->consumed[b], 0));      // a consumer's reads of this slot's previous table come first
  if (x->comm) {
    const ncclResult_t nrc = rccl().AllGather(x->buffers[b], x->gathered[b], count, ncclInt32, x->comm, x->side);
    if (nrc) return rccl_fail(nrc, "ncclAllGather");
  } else {
    ITM_HIP(hipMemcpyAsync(x->gathered[b], x->buffers[b], count * 4, hipMemcpyDeviceToDevice, x->side));
  }
  if (x->checkDev) {
    exchange_self_check_kernel<<<(unsigned)((count + 255) / 256), 256, 0, x->side>>>(x->buffers[b], x->gathered[b] + (size_t)x->rank * count, count, x->checkDev, x->corruptWord);
    ITM_LAUNCH_CHECK();
  }
  ITM_HIP(hipEventRecord(x->released[b], x->side));
  { std::lock_guard<std::mutex> g(x->m); x->newest = b; }
  return ITM_OK;
}

static void issuer_main(itm_exchange* x) {
  (void)hipSetDevice(x->device);
  for (;;) {
    int b;
    {
      std::unique_lock<std::mutex> lk(x->m);
      x->cv.wait(lk, [&] { return x->stop || !x->queue.empty(); });
      if (x->queue.empty()) return;                 // (stop: whatever was queued has been issued)
      b = x->queue.front(); x->queue.pop_front();
    }
    if (issue_collective(x, b) != ITM_OK && !x->issuerFailed.load()) {
      const char* msg = itm_last_error();
      x->issuerMessage = msg ? msg : "exchange: the collective could not be issued";
      x->issuerFailed.store(1, std::memory_order_release);
    }
    x->state[b].store(itm_exchange::kIssued, std::memory_order_release);
  }
}

static int self_check_error(const itm_exchange* x) {
  const volatile SelfCheck* c = x->check;
  char msg[320];
  snprintf(msg, sizeof msg, "exchange self-check: after a collective, %d word(s) of rank %d's own block of the gathered table differ from the records the rank sent "
           "(first at word %d; %d collective(s) checked, world %d): the table cannot be trusted", (int)c->mismatchedWords, x->rank, (int)c->firstBadWord, (int)c->collectives, x->world);
  return set_error(ITM_ERR_DEVICE, msg);
}

extern "C" {

int itm_debug_checksum(const int32_t* src, size_t words, int rounds, unsigned long long* dst, itm_stream stream) {
  if (!src || !dst || rounds < 1) return set_error(ITM_ERR_INVALID, "bad argument");
  exchange_checksum_kernel<<<1, 256, 0, as_stream(stream)>>>(src, words, rounds, dst);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_exchange_unique_id(unsigned char id[128]) {
  if (!id) return set_error(ITM_ERR_INVALID, "null argument");
  if (!rccl().ok) return set_error(ITM_ERR_DEVICE, "librccl.so could not be loaded");
  ncclUniqueId u;
  const ncclResult_t rc = rccl().GetUniqueId(&u);
  if (rc) return rccl_fail(rc, "ncclGetUniqueId");
  memcpy(id, u.internal, 128);
  return ITM_OK;
}

int itm_exchange_create(int world, int rank, const unsigned char id[128], int max_ids, int batch, itm_exchange** out) {
  if (!out || world < 1 || rank < 0 || rank >= world || max_ids < 0 || batch < 1) return set_error(ITM_ERR_INVALID, "bad argument");
  if (world > 1 && !id) return set_error(ITM_ERR_INVALID, "a communicator of more than one rank needs the unique id of rank 0");
  itm_exchange* x = new (std::nothrow) itm_exchange();
  if (!x) return set_error(ITM_ERR_DEVICE, "out of host memory");
  x->world = world; x->rank = rank; x->maxIds = max_ids; x->batch = batch;
  x->words = (size_t)kRecordHeader + (size_t)max_ids;
  for (int b = 0; b < itm_exchange::kRing; ++b) { x->state[b].store(itm_exchange::kFree); x->batchOf[b] = -1; }
  (void)hipGetDevice(&x->device);
  const size_t batchBytes = x->words * (size_t)batch * 4;
  hipError_t e = hipStreamCreateWithFlags(&x->side, hipStreamNonBlocking);
  // (device-scope release: a system-scope one writes back the L2s the next frame's kernels are working in; A/B in DESIGN.md section 6)
  const unsigned evFlags = hipEventDisableTiming | (unsigned)hipEventReleaseToDevice;
  for (int b = 0; b < itm_exchange::kRing && e == hipSuccess; ++b) {
    e = hipMalloc((void**)&x->buffers[b], batchBytes);
    if (e == hipSuccess) e = hipMemset(x->buffers[b], 0xFF, batchBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&x->gathered[b], batchBytes * (size_t)world);
    if (e == hipSuccess) e = hipMemset(x->gathered[b], 0xFF, batchBytes * (size_t)world);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->copied[b], evFlags);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->released[b], evFlags);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->consumed[b], evFlags);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { free_exchange(x); return hip_fail(e, "exchange buffers", __FILE__, __LINE__); }
  // One code path for every world size: a single rank gets a communicator too (its id made here) and runs the same ncclAllGather
  // as eight do.  Debug key ITM_DEBUG_EXCHANGE_DEVICE_COPY replaces the one-rank collective by a device copy (a test pins both to the same table).
  const bool deviceCopy = world == 1 && g_debug_exchange_device_copy != 0;
  if (!deviceCopy) {
    if (!rccl().ok) { free_exchange(x); return set_error(ITM_ERR_DEVICE, "librccl.so could not be loaded"); }
    ncclUniqueId u;
    if (id) memcpy(u.internal, id, 128);
    else {
      const ncclResult_t rc = rccl().GetUniqueId(&u);
      if (rc) { free_exchange(x); return rccl_fail(rc, "ncclGetUniqueId"); }
    }
    const ncclResult_t rc = rccl().CommInitRank(&x->comm, world, u, rank);
    if (rc) { x->comm = nullptr; free_exchange(x); return rccl_fail(rc, "ncclCommInitRank"); }
  }
  {
    const char* sc = getenv("ITM_EXCHANGE_SELF_CHECK");
    if (!(sc && sc[0] == '0')) {
      void* h = nullptr; void* d = nullptr;
      if (hipHostMalloc(&h, sizeof(SelfCheck), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        memset(h, 0, sizeof(SelfCheck));
        x->check = (SelfCheck*)h; x->checkDev = (SelfCheck*)d;
      } else { if (h) (void)hipHostFree(h); (void)hipGetLastError(); }
    }
    x->corruptWord = g_debug_exchange_corrupt_word;
  }
  x->issuer = std::thread(issuer_main, x);
  *out = x;
  return ITM_OK;
}

int itm_exchange_destroy(itm_exchange* x) { free_exchange(x); return ITM_OK; }

int itm_exchange_step(itm_exchange* x, const itm_render_state* rs, const float M_d[16], itm_stream frame_stream) {
  if (!x || !rs || !M_d) return set_error(ITM_ERR_INVALID, "null argument");
  if (!rs->scene) return set_error(ITM_ERR_INVALID, "the render state's scene has been destroyed");
  { const int rc = enter_scene(rs->scene, rs); if (rc) return rc; }       // engine calls recorded on the render state are launched first (pending.hip)
  hipStream_t fs = as_stream(frame_stream);
  const int slot = (int)(x->frame % x->batch);
  if (x->issuerFailed.load(std::memory_order_acquire)) return set_error(ITM_ERR_DEVICE, x->issuerMessage);
  if (x->check && ((volatile SelfCheck*)x->check)->mismatchedWords) return self_check_error(x);
  if (slot == 0) {
    // the next slot of the ring that no consumer holds (at most one is held: the ring of eight always has another)
    int b;
    { std::lock_guard<std::mutex> g(x->m); b = (x->cur + (x->batchNo ? 1 : 0)) % itm_exchange::kRing; if (b == x->held) b = (b + 1) % itm_exchange::kRing; x->cur = b; }
    if (x->state[b].load(std::memory_order_acquire) != itm_exchange::kFree) {
      // the collective that used this slot eight batches ago must have let go of it.  Polled, not hipEventSynchronize: the blocking
      // wait of the runtime was measured at ~60 ms per call in a process whose other threads keep the cores busy (bench.py with a gloo
      // control plane: 33 frames/s), a query is a load.  The wait returns at once unless the host runs that far ahead of the GPU.
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0;; ++spins) {
        if (x->state[b].load(std::memory_order_acquire) == itm_exchange::kIssued) {
          const hipError_t q = hipEventQuery(x->released[b]);
          if (q == hipSuccess) break;
          if (q != hipErrorNotReady) return hip_fail(q, "exchange: collective", __FILE__, __LINE__);
        }
        if ((spins & 0x3ffu) == 0x3ffu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0)
          return set_error(ITM_ERR_DEVICE, "exchange: the collective that used this slot eight batches ago has not finished");
        __builtin_ia32_pause();
      }
      x->state[b].store(itm_exchange::kFree, std::memory_order_relaxed);
    }
  }
  const int b = x->cur;
  int rc = itm_export_visible_record(rs, M_d, x->maxIds, x->buffers[b] + (size_t)slot * x->words, frame_stream);
  if (rc) return rc;
  if (slot == x->batch - 1) {
    ITM_HIP(hipEventRecord(x->copied[b], fs));
    x->state[b].store(itm_exchange::kQueued, std::memory_order_release);
    { std::lock_guard<std::mutex> g(x->m); x->batchOf[b] = x->batchNo; x->queue.push_back(b); }
    x->cv.notify_one();
    ++x->batchNo;
  }
  ++x->frame;
  return ITM_OK;
}

// waits until every collective queued so far has been put on the side stream
static void wait_issued(itm_exchange* x) {
  for (int b = 0; b < itm_exchange::kRing; ++b)
    while (x->state[b].load(std::memory_order_acquire) == itm_exchange::kQueued) std::this_thread::yield();
}
// the newest slot whose collective is on the side stream (all queued collectives issued first); -1 if none yet
static int newest_issued(itm_exchange* x) {
  wait_issued(x);
  std::lock_guard<std::mutex> g(x->m);
  return x->newest;
}

static int release_locked(itm_exchange* x, hipStream_t consumer) {
  if (x->held < 0) return ITM_OK;
  ITM_HIP(hipEventRecord(x->consumed[x->held], consumer));      // behind everything the consumer stream was given so far
  x->consumedPending[x->held] = true;
  x->held = -1;
  return ITM_OK;
}

int itm_exchange_acquire(itm_exchange* x, itm_stream consumer_stream, const int32_t** table, long long* first_frame) {
  if (!x || !table) return set_error(ITM_ERR_INVALID, "null argument");
  if (x->issuerFailed.load(std::memory_order_acquire)) return set_error(ITM_ERR_DEVICE, x->issuerMessage);
  if (x->check && ((volatile SelfCheck*)x->check)->mismatchedWords) return self_check_error(x);
  // The slot is chosen and marked as held under ONE lock: chosen outside it, a frame thread that advanced the ring by seven
  // batches in between could re-use that slot before `held` names it, and the next collective would overwrite a table the consumer
  // is still reading.  (acquire / release belong to the thread that calls itm_exchange_step, or to a consumer the host serialises with it.)
  wait_issued(x);
  std::lock_guard<std::mutex> g(x->m);
  const int b = x->newest;
  { const int rc = release_locked(x, as_stream(consumer_stream)); if (rc) return rc; }       // "release on the next acquire"
  if (b < 0) { *table = nullptr; if (first_frame) *first_frame = -1; return ITM_OK; }        // no collective has been issued yet
  ITM_HIP(hipStreamWaitEvent(as_stream(consumer_stream), x->released[b], 0));                // the table is complete (and self-checked) behind this event
  x->held = b;
  *table = x->gathered[b];
  if (first_frame) *first_frame = x->batchOf[b] * (long long)x->batch;
  return ITM_OK;
}

int itm_exchange_release(itm_exchange* x, itm_stream consumer_stream) {
  if (!x) return set_error(ITM_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> g(x->m);
  return release_locked(x, as_stream(consumer_stream));
}

int itm_exchange_info(const itm_exchange* x, int* world, int* rank, int* max_ids, int* batch, const void** gathered_device) {
  if (!x) return set_error(ITM_ERR_INVALID, "null argument");
  if (world) *world = x->world;
  if (rank) *rank = x->rank;
  if (max_ids) *max_ids = x->maxIds;
  if (batch) *batch = x->batch;
  // (the table of the newest collective issued so far, or NULL.  A bare pointer: whoever reads it on the device without
  // itm_exchange_acquire races with the collectives of later batches -- kept for hosts that read after itm_exchange_self_check)
  if (gathered_device) { const int b = newest_issued(const_cast<itm_exchange*>(x)); *gathered_device = b >= 0 ? x->gathered[b] : nullptr; }
  return ITM_OK;
}

int itm_exchange_table(itm_exchange* x, int32_t* dst_host, size_t words) {
  if (!x || !dst_host) return set_error(ITM_ERR_INVALID, "null argument");
  const size_t all = x->words * (size_t)x->batch * (size_t)x->world;
  if (words < all) return set_error(ITM_ERR_INVALID, "destination too small for world x batch records");
  // every queued collective must have been put on the side stream before it is drained
  const int b = newest_issued(x);
  if (x->issuerFailed.load(std::memory_order_acquire)) return set_error(ITM_ERR_DEVICE, x->issuerMessage);
  ITM_HIP(hipStreamSynchronize(x->side));     // the table is written by the collectives on the side stream
  if (b < 0) { memset(dst_host, 0xFF, all * 4); return ITM_OK; }      // no collective yet: the "nothing gathered" pattern of a fresh exchange
  ITM_HIP(hipMemcpy(dst_host, x->gathered[b], all * 4, hipMemcpyDeviceToHost));
  if (x->check && ((volatile SelfCheck*)x->check)->mismatchedWords) return self_check_error(x);
  return ITM_OK;
}

// {collectives checked, words of the own block that differed}; synchronises the side stream.  checked == 0 with the self-check
// enabled means no collective has completed yet.
int itm_exchange_self_check(itm_exchange* x, int* collectives_checked, int* mismatched_words) {
  if (!x) return set_error(ITM_ERR_INVALID, "null argument");
  for (int b = 0; b < itm_exchange::kRing; ++b)
    while (x->state[b].load(std::memory_order_acquire) == itm_exchange::kQueued) std::this_thread::yield();
  ITM_HIP(hipStreamSynchronize(x->side));
  if (collectives_checked) *collectives_checked = x->check ? (int)((volatile SelfCheck*)x->check)->collectives : -1;
  if (mismatched_words) *mismatched_words = x->check ? (int)((volatile SelfCheck*)x->check)->mismatchedWords : 0;
  return ITM_OK;
}

}  // extern "C"
