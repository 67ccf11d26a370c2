// The collectives of the multi-GPU solve behind one small interface.
//
// The reference is single-threaded (Main_Calibration/bundle_adjustment_manager.cpp:90-92 sets no num_threads); sharding the
// points over GPUs is this implementation's own: every rank eliminates its own points, the packed reduced
// camera system is summed over the ranks, every rank factors the identical sum.  ba_solver.hip issues exactly three kinds of
// collective — sum of doubles, max of doubles, min of ints, all in place — through `Comm`:
//
//   RcclComm      ncclAllReduce over xGMI, one process per GPU (the product).
//   LoopbackComm  the ranks are solver objects of ONE process on ONE GPU, each driven by its own host thread: a contribution
//                 is copied to a staging slot, the ranks meet at a host barrier, every rank adds all slots in rank order (the
//                 same bits on every rank, like a ring all-reduce).  This exists so that every line of the N > 1 schedule except
//                 ncclAllReduce itself — the sharded upload, the agreement collectives at set-up, three collectives per LM
//                 step, the summed stall flag and the common fallback — runs on the one-GPU boxes of this pool
//                 (tests/test_gpu_loopback.py).  The ranks take turns on the device: a rank holds the group's token while it
//                 launches, and hands it over only inside a collective, after its own streams have drained — so that the
//                 in-kernel waits of one rank's step (several-workgroup Cholesky, persistent tiles) never compete for
//                 residency with another rank's kernels, which on separate GPUs they never would.
//   ShmComm       the ranks are PROCESSES of one host (round 5): contributions staged through a POSIX shared-memory segment, a
//                 barrier of atomics in it, every rank adds all slots in rank order on the host and copies the sum back.  Slow by
//                 construction (two trips over PCIe per collective) — it exists so that bench.py's real launcher, the unique-id
//                 bootstrap's twin and one process per rank run end to end where only ONE GPU is visible, which RCCL refuses
//                 (two ranks of a communicator on one device).  Sequential multi-GPU schedule only (AllowsResidentWaiters).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace rsba {

class Comm {
 public:
  virtual ~Comm() {}
  virtual int nranks() const = 0;
  virtual const char* kind() const = 0;
  // collectives between GroupStart and GroupEnd go out together; every call is in place and returns false on failure
  virtual bool GroupStart() = 0;
  virtual bool GroupEnd() = 0;
  virtual bool SumDoubles(double* buf, size_t n, hipStream_t st) = 0;
  virtual bool MaxDoubles(double* buf, size_t n, hipStream_t st) = 0;
  virtual bool MinInts(int* buf, size_t n, hipStream_t st) = 0;
  // a rank brackets every stretch of host code that launches on the device with Enter / Leave (no-ops over RCCL)
  virtual void Enter() {}
  virtual void Leave() {}
  // a rank that gives up (error return) says so, so that the others do not wait for it forever
  virtual void Abort() {}
  // false: the ranks are PROCESSES sharing one device (ShmComm) — kernels that wait resident for a flag which only another
  // process' collective can raise would hold the CUs that process needs: the pipelined multi-GPU schedule is not offered
  virtual bool AllowsResidentWaiters() const { return true; }
  // host-side wait for a stream that carries one of this communicator's collectives; false: the communicator failed (RcclComm
  // polls its asynchronous error state with a bounded wait; the host-staged communicators have completed by then anyway)
  virtual bool WaitStream(hipStream_t st) { return hipStreamSynchronize(st) == hipSuccess; }
};

struct CommScope {
  explicit CommScope(Comm* c) : c_(c) { if (c_) c_->Enter(); }
  ~CommScope() { if (c_) c_->Leave(); }
  CommScope(const CommScope&) = delete;
  CommScope& operator=(const CommScope&) = delete;
 private:
  Comm* c_;
};

// ------------------------------------------------------------------------------------------------ RCCL
class RcclComm : public Comm {
 public:
  // One communicator per unique id and process (an id can be used for one ncclCommInitRank only); solvers created later with
  // the same id share it.  world_size <= 1 with id == nullptr: a 1-rank communicator (RSBA_FORCE_COMM=1: the collective path
  // on a single GPU).
  static std::shared_ptr<Comm> Create(int world_size, int rank, const void* id128) {
    std::lock_guard<std::mutex> lk(Mu());
    auto& comms = Comms();
    ncclUniqueId id;
    if (world_size > 1) memcpy(&id, id128, sizeof(id));
    else if (ncclGetUniqueId(&id) != ncclSuccess) return nullptr;
    const std::string key = world_size > 1 ? std::string((const char*)&id, sizeof(id)) : std::string("single");
    auto it = comms.find(key);
    if (it != comms.end()) {
      if (!static_cast<RcclComm*>(it->second.get())->dead_) return it->second;
      // an aborted communicator: its id is spent (one ncclCommInitRank per id — the ranks whose communicator is still alive would
      // hand out their cached one and this rank would hang in the bootstrap).  Several ranks: the caller needs a fresh id; one rank:
      // the id above is fresh anyway.
      if (world_size > 1) { fprintf(stderr, "rsba: the communicator of this id was aborted; a new job needs a new id\n"); return nullptr; }
      comms.erase(it);
    }
    ncclComm_t c = nullptr;
    if (ncclCommInitRank(&c, world_size > 1 ? world_size : 1, id, world_size > 1 ? rank : 0) != ncclSuccess) return nullptr;
    std::shared_ptr<Comm> p(new RcclComm(c));
    comms.emplace(key, p);
    return p;
  }
  // Communicators live as long as the process (an id serves one ncclCommInitRank, and bench.py creates several solvers on one id)
  // unless the host finalises them: rsba_comm_finalize -> FinalizeAll destroys every live one (ncclCommDestroy) while the HIP
  // runtime is still up.  Nothing is destroyed from a static destructor at exit: the runtime may be gone by then.
  static void FinalizeAll() {
    std::lock_guard<std::mutex> lk(Mu());
    for (auto& kv : Comms()) { RcclComm* c = static_cast<RcclComm*>(kv.second.get()); if (c->c_ && !c->dead_) { (void)ncclCommDestroy(c->c_); c->dead_ = true; c->c_ = nullptr; } }
    Comms().clear();
  }
  int nranks() const override { int n = 0; return !dead_ && ncclCommCount(c_, &n) == ncclSuccess ? n : 0; }
  const char* kind() const override { return "rccl"; }
  bool GroupStart() override { return Live() && Ok(ncclGroupStart()); }
  bool GroupEnd() override { return Live() && Ok(ncclGroupEnd()) && Healthy(); }
  bool SumDoubles(double* b, size_t n, hipStream_t st) override { return Live() && Ok(ncclAllReduce(b, b, n, ncclDouble, ncclSum, c_, st)) && Healthy(); }
  bool MaxDoubles(double* b, size_t n, hipStream_t st) override { return Live() && Ok(ncclAllReduce(b, b, n, ncclDouble, ncclMax, c_, st)) && Healthy(); }
  bool MinInts(int* b, size_t n, hipStream_t st) override { return Live() && Ok(ncclAllReduce(b, b, n, ncclInt32, ncclMin, c_, st)) && Healthy(); }
  // A rank that gives up, or one that finds its peers gone (WaitStream below): ncclCommAbort tears the communicator down and
  // frees every kernel of it that is still waiting for a peer, so that the process can leave with an error code instead of
  // hanging in a collective forever (round 4: Abort was a no-op here and communicators were never destroyed).
  void Abort() override { if (c_ && !dead_) { dead_ = true; (void)ncclCommAbort(c_); } }
  // Waits for `st` (which carries a collective of this communicator) with the communicator's health polled beside it: a rank
  // that died leaves its peers inside ncclAllReduce, the stream never completes, and hipStreamSynchronize would wait for ever.
  // RSBA_COMM_TIMEOUT_S (default 120) bounds the wait; an asynchronous RCCL error or the time-out aborts the communicator.
  bool WaitStream(hipStream_t st) override {
    static const double budget = getenv("RSBA_COMM_TIMEOUT_S") ? atof(getenv("RSBA_COMM_TIMEOUT_S")) : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) return !dead_;
      if (q != hipErrorNotReady) { (void)hipGetLastError(); Abort(); return false; }
      if ((spin & 63) == 63) sched_yield();   // (a wait of up to two minutes: not a core spinning flat out)
      if ((spin & 1023) == 1023) {
        if (!Healthy()) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budget) {
          fprintf(stderr, "rsba: a collective did not complete within %.0f s (a peer rank is gone?); aborting the communicator\n", budget);
          Abort();
          return false;
        }
      }
    }
  }
 private:
  explicit RcclComm(ncclComm_t c) : c_(c) {}
  static std::mutex& Mu() { static std::mutex mu; return mu; }
  static std::map<std::string, std::shared_ptr<Comm>>& Comms() { static auto* m = new std::map<std::string, std::shared_ptr<Comm>>(); return *m; }
  bool Live() const { if (dead_) fprintf(stderr, "rsba: the RCCL communicator was aborted\n"); return !dead_; }
  // ncclCommGetAsyncError: errors of collectives already enqueued (a peer's death, a transport failure) surface here, not in the
  // return code of the enqueue
  bool Healthy() {
    if (dead_) return false;
    ncclResult_t async = ncclSuccess;
    if (ncclCommGetAsyncError(c_, &async) != ncclSuccess || (async != ncclSuccess && async != ncclInProgress)) {
      fprintf(stderr, "rsba: RCCL asynchronous error %s; aborting the communicator\n", ncclGetErrorString(async));
      Abort();
      return false;
    }
    return true;
  }
  static bool Ok(ncclResult_t r) {
    if (r != ncclSuccess) fprintf(stderr, "rsba: RCCL error %s\n", ncclGetErrorString(r));
    return r == ncclSuccess;
  }
  ncclComm_t c_;
  bool dead_ = false;
};

// ------------------------------------------------------------------------------------------------ loopback
// out[i] = op over the ranks, in rank order, of slot[r][i]
template <typename T, int kOp /* 0 sum, 1 max, 2 min */>
__global__ void k_loopback_reduce(T* __restrict__ out, size_t n, const T* const* __restrict__ slots, size_t offset, int nranks) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T a = slots[0][offset + i];
    for (int r = 1; r < nranks; ++r) {
      const T b = slots[r][offset + i];
      a = kOp == 0 ? a + b : (kOp == 1 ? (b > a ? b : a) : (b < a ? b : a));
    }
    out[i] = a;
  }
}

class LoopbackGroup {
 public:
  explicit LoopbackGroup(int n) : n_(n) {
    for (int p = 0; p < 2; ++p) { slot_[p].assign(n, nullptr); cap_[p].assign(n, 0); table_[p] = nullptr; }
  }
  ~LoopbackGroup() {
    for (int p = 0; p < 2; ++p) {
      for (void* q : slot_[p]) if (q) (void)hipFree(q);
      if (table_[p]) (void)hipFree(table_[p]);
    }
  }
  int n() const { return n_; }
  // the device token
  void Lock() { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return !held_; }); held_ = true; }
  void Unlock() { { std::lock_guard<std::mutex> lk(mu_); held_ = false; } cv_.notify_all(); }
  // all ranks meet; false: somebody gave up, or nobody came for two minutes
  bool Barrier() {
    std::unique_lock<std::mutex> lk(mu_);
    if (aborted_) return false;
    const long gen = gen_;
    if (++arrived_ == n_) { arrived_ = 0; ++gen_; cv_.notify_all(); return true; }
    const bool ok = cv_.wait_for(lk, std::chrono::seconds(120), [&] { return gen_ != gen || aborted_; });
    if (!ok) { aborted_ = true; cv_.notify_all(); }
    return ok && !aborted_;
  }
  void Abort() { { std::lock_guard<std::mutex> lk(mu_); aborted_ = true; } cv_.notify_all(); }
  // rank r's staging slot of the given parity, at least `bytes` long (grown by its owner, who holds the token); the table of all
  // ranks' slots as the reduce kernel reads it lives in device memory, rewritten by whoever finds it stale after the barrier
  bool Reserve(int parity, int r, size_t bytes) {
    if (cap_[parity][r] >= bytes) return true;
    std::lock_guard<std::mutex> lk(mu_);
    if (slot_[parity][r]) (void)hipFree(slot_[parity][r]);
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipMalloc(&slot_[parity][r], want) != hipSuccess) { slot_[parity][r] = nullptr; cap_[parity][r] = 0; return false; }
    cap_[parity][r] = want;
    ++table_version_[parity];
    return true;
  }
  void* Slot(int parity, int r) const { return slot_[parity][r]; }
  // device array of the n slot pointers (call after the barrier, with the token held)
  const void* const* Table(int parity) {
    std::lock_guard<std::mutex> lk(mu_);
    if (table_[parity] == nullptr && hipMalloc((void**)&table_[parity], n_ * sizeof(void*)) != hipSuccess) return nullptr;
    if (table_written_[parity] != table_version_[parity]) {
      if (hipMemcpy(table_[parity], slot_[parity].data(), n_ * sizeof(void*), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
      table_written_[parity] = table_version_[parity];
    }
    return (const void* const*)table_[parity];
  }
 private:
  const int n_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool held_ = false, aborted_ = false;
  int arrived_ = 0;
  long gen_ = 0;
  std::vector<void*> slot_[2];
  std::vector<size_t> cap_[2];
  void** table_[2];
  long table_version_[2] = {0, 0}, table_written_[2] = {-1, -1};
};

class LoopbackComm : public Comm {
 public:
  // 128-byte ids that name a loopback group start with this tag (rsba_comm_loopback_id)
  static const char* Magic() { return "rsba-loopback-v1"; }
  static bool IsLoopbackId(const void* id128) { return id128 && memcmp(id128, Magic(), strlen(Magic())) == 0; }
  static void NewId(void* out128) {
    static std::mutex mu;
    static unsigned long counter = 0;
    std::lock_guard<std::mutex> lk(mu);
    memset(out128, 0, 128);
    snprintf((char*)out128, 128, "%s#%lu", Magic(), ++counter);
  }
  static std::shared_ptr<Comm> Create(int world_size, int rank, const void* id128) {
    static std::mutex mu;
    static std::map<std::string, std::weak_ptr<LoopbackGroup>> groups;   // a group lives as long as one of its ranks' solvers
    std::shared_ptr<LoopbackGroup> g;
    {
      std::lock_guard<std::mutex> lk(mu);
      const std::string key((const char*)id128, 128);
      g = groups[key].lock();
      if (!g) { g = std::make_shared<LoopbackGroup>(world_size); groups[key] = g; }
    }
    if (g->n() != world_size || rank < 0 || rank >= world_size) return nullptr;
    return std::shared_ptr<Comm>(new LoopbackComm(g, rank));
  }
  int nranks() const override { return g_->n(); }
  const char* kind() const override { return "loopback"; }
  void Enter() override { if (depth_++ == 0) g_->Lock(); }
  void Leave() override { if (--depth_ == 0) g_->Unlock(); }
  void Abort() override { g_->Abort(); }
  bool GroupStart() override { grouped_ = true; return true; }
  bool GroupEnd() override { grouped_ = false; return Flush(); }
  bool SumDoubles(double* b, size_t n, hipStream_t st) override { return Add(b, n, sizeof(double), 0, st); }
  bool MaxDoubles(double* b, size_t n, hipStream_t st) override { return Add(b, n, sizeof(double), 1, st); }
  bool MinInts(int* b, size_t n, hipStream_t st) override { return Add(b, n, sizeof(int), 2, st); }
 private:
  struct Op { void* buf; size_t n, elem, offset; int op; hipStream_t st; };
  LoopbackComm(std::shared_ptr<LoopbackGroup> g, int rank) : g_(std::move(g)), rank_(rank) {}
  bool Add(void* b, size_t n, size_t elem, int op, hipStream_t st) {
    const size_t off = (bytes_ + 15) & ~(size_t)15;
    ops_.push_back(Op{b, n, elem, off, op, st});
    bytes_ = off + n * elem;
    return grouped_ ? true : Flush();
  }
  // contribution -> staging slot; own streams drained; the device handed over; all ranks meet; the sums, on the caller's streams
  bool Flush() {
    if (ops_.empty()) return true;
    bool ok = depth_ > 0;   // (a collective outside Enter / Leave is a bug of the caller)
    const int par = parity_;
    parity_ ^= 1;
    ok = ok && g_->Reserve(par, rank_, bytes_);
    char* slot = (char*)g_->Slot(par, rank_);
    for (const Op& o : ops_) ok = ok && hipMemcpyAsync(slot + o.offset, o.buf, o.n * o.elem, hipMemcpyDeviceToDevice, o.st) == hipSuccess;
    for (const Op& o : ops_) ok = ok && hipStreamSynchronize(o.st) == hipSuccess;
    // (and the previous collective's sums, which read the OTHER ranks' slots of the other parity: a rank that passes this
    //  barrier may write that slot again in its next collective)
    for (hipStream_t st : last_streams_) ok = ok && hipStreamSynchronize(st) == hipSuccess;
    last_streams_.clear();
    for (const Op& o : ops_) last_streams_.push_back(o.st);
    if (!ok) g_->Abort();
    g_->Unlock();
    const bool met = g_->Barrier();
    g_->Lock();
    ok = ok && met;
    const void* const* table = ok ? g_->Table(par) : nullptr;
    ok = ok && table != nullptr;
    for (const Op& o : ops_) {
      if (!ok) break;
      const int blocks = (int)std::min<size_t>(1024, (o.n + 255) / 256);
      const size_t eo = o.offset / o.elem;
      if (o.elem == sizeof(double) && o.op == 0) k_loopback_reduce<double, 0><<<blocks, 256, 0, o.st>>>((double*)o.buf, o.n, (const double* const*)table, eo, g_->n());
      else if (o.elem == sizeof(double)) k_loopback_reduce<double, 1><<<blocks, 256, 0, o.st>>>((double*)o.buf, o.n, (const double* const*)table, eo, g_->n());
      else k_loopback_reduce<int, 2><<<blocks, 256, 0, o.st>>>((int*)o.buf, o.n, (const int* const*)table, eo, g_->n());
      ok = hipGetLastError() == hipSuccess;
    }
    ops_.clear();
    bytes_ = 0;
    if (!ok) fprintf(stderr, "rsba: loopback collective failed on rank %d\n", rank_);
    return ok;
  }
  std::shared_ptr<LoopbackGroup> g_;
  const int rank_;
  int depth_ = 0, parity_ = 0;
  bool grouped_ = false;
  std::vector<Op> ops_;
  std::vector<hipStream_t> last_streams_;
  size_t bytes_ = 0;
};

// ------------------------------------------------------------------------------------------------ shared memory (processes of one host)
struct ShmHeader {
  std::atomic<int> world;        // set by the first rank to attach (a fresh segment is zero-filled), checked by the others
  std::atomic<long> slot_bytes;
  std::atomic<int> arrived;
  std::atomic<long> generation;
  std::atomic<int> aborted;
  char pad[64];
};

class ShmComm : public Comm {
 public:
  static const char* Magic() { return "rsba-shm-v1#"; }
  static bool IsShmId(const void* id128) { return id128 && memcmp(id128, Magic(), strlen(Magic())) == 0; }
  static bool NewId(const char* name, void* out128) {
    if (!name || !*name || strlen(name) > 80) return false;
    for (const char* c = name; *c; ++c) if (!(isalnum((unsigned char)*c) || *c == '_' || *c == '-' || *c == '.')) return false;
    memset(out128, 0, 128);
    snprintf((char*)out128, 128, "%s%s", Magic(), name);
    return true;
  }
  // slot_bytes: the largest payload a collective (or a group of them) of this solver will carry
  static std::shared_ptr<Comm> Create(int world_size, int rank, const void* id128, size_t slot_bytes) {
    if (world_size < 2 || rank < 0 || rank >= world_size) return nullptr;
    char name[128];
    snprintf(name, sizeof(name), "/rsba_%s", (const char*)id128 + strlen(Magic()));
    slot_bytes = (slot_bytes + 4095) & ~(size_t)4095;
    const size_t total = 4096 + (size_t)2 * world_size * slot_bytes;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { perror("rsba: shm_open"); return nullptr; }
    if (ftruncate(fd, (off_t)total) != 0) { perror("rsba: ftruncate"); close(fd); return nullptr; }
    void* base = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { perror("rsba: mmap"); return nullptr; }
    std::shared_ptr<ShmComm> c(new ShmComm(base, total, world_size, rank, slot_bytes, name));
    ShmHeader* h = c->hdr();
    int w0 = 0; long b0 = 0;
    if (!h->world.compare_exchange_strong(w0, world_size) && w0 != world_size) { fprintf(stderr, "rsba: shm group %s has %d ranks, not %d\n", name, w0, world_size); return nullptr; }
    if (!h->slot_bytes.compare_exchange_strong(b0, (long)slot_bytes) && b0 != (long)slot_bytes) { fprintf(stderr, "rsba: shm group %s: slot sizes differ between ranks\n", name); return nullptr; }
    // everybody attached: the name can go (the segment lives as long as it is mapped); a stale segment of a crashed run with the
    // same name would carry its counters — the caller makes names unique per run
    if (!c->Barrier()) return nullptr;
    if (rank == 0) (void)shm_unlink(name);
    return c;
  }
  ~ShmComm() override { if (base_) munmap(base_, total_); if (host_) (void)hipHostFree(host_); }
  int nranks() const override { return world_; }
  const char* kind() const override { return "shm"; }
  bool AllowsResidentWaiters() const override { return false; }
  void Abort() override { hdr()->aborted.store(1); }
  bool GroupStart() override { grouped_ = true; return true; }
  bool GroupEnd() override { grouped_ = false; return Flush(); }
  bool SumDoubles(double* b, size_t n, hipStream_t st) override { return Add(b, n, sizeof(double), 0, st); }
  bool MaxDoubles(double* b, size_t n, hipStream_t st) override { return Add(b, n, sizeof(double), 1, st); }
  bool MinInts(int* b, size_t n, hipStream_t st) override { return Add(b, n, sizeof(int), 2, st); }
 private:
  struct Op { void* buf; size_t n, elem, offset; int op; hipStream_t st; };
  ShmComm(void* base, size_t total, int world, int rank, size_t slot_bytes, const char* name)
      : base_(base), total_(total), world_(world), rank_(rank), slot_bytes_(slot_bytes), name_(name) {}
  ShmHeader* hdr() const { return (ShmHeader*)base_; }
  char* Slot(int parity, int r) const { return (char*)base_ + 4096 + ((size_t)parity * world_ + r) * slot_bytes_; }
  // all ranks meet (sense by generation); false: somebody gave up, or nobody came for RSBA_COMM_TIMEOUT_S (120 s)
  bool Barrier() {
    static const double budget = getenv("RSBA_COMM_TIMEOUT_S") ? atof(getenv("RSBA_COMM_TIMEOUT_S")) : 120.0;
    ShmHeader* h = hdr();
    if (h->aborted.load()) return false;
    const long gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == world_) { h->arrived.store(0); h->generation.fetch_add(1); return true; }
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0; h->generation.load() == gen; ++spin) {
      if (h->aborted.load()) return false;
      if ((spin & 63) == 63) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budget) { h->aborted.store(1); fprintf(stderr, "rsba: shm rank %d waited %.0f s at a barrier; giving up\n", rank_, budget); return false; }
      }
    }
    return !h->aborted.load();
  }
  bool Add(void* b, size_t n, size_t elem, int op, hipStream_t st) {
    const size_t off = (bytes_ + 15) & ~(size_t)15;
    ops_.push_back(Op{b, n, elem, off, op, st});
    bytes_ = off + n * elem;
    return grouped_ ? true : Flush();
  }
  template <typename T>
  static void Reduce(T* out, const std::vector<const char*>& slots, size_t off, size_t n, int op) {
    for (size_t i = 0; i < n; ++i) {
      T a = ((const T*)(slots[0] + off))[i];
      for (size_t r = 1; r < slots.size(); ++r) { const T b = ((const T*)(slots[r] + off))[i]; a = op == 0 ? a + b : (op == 1 ? (b > a ? b : a) : (b < a ? b : a)); }
      out[i] = a;
    }
  }
  // contribution -> this rank's slot (device to host); all ranks meet; every rank adds the slots in rank order (identical bits
  // everywhere, like a ring all-reduce) and copies the sums back, on the caller's streams' behalf (they are drained here)
  bool Flush() {
    if (ops_.empty()) return true;
    const int par = parity_;
    parity_ ^= 1;
    bool ok = bytes_ <= slot_bytes_;
    if (!ok) fprintf(stderr, "rsba: shm collective of %zu bytes exceeds the group's slots (%zu)\n", bytes_, slot_bytes_);
    if (ok && host_cap_ < bytes_) {
      if (host_) (void)hipHostFree(host_);
      host_cap_ = bytes_ + bytes_ / 2 + 4096;
      if (hipHostMalloc((void**)&host_, host_cap_, hipHostMallocDefault) != hipSuccess) { host_ = nullptr; host_cap_ = 0; ok = false; }
    }
    for (const Op& o : ops_) ok = ok && hipStreamSynchronize(o.st) == hipSuccess;
    for (const Op& o : ops_) ok = ok && hipMemcpy(host_ + o.offset, o.buf, o.n * o.elem, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) memcpy(Slot(par, rank_), host_, bytes_);
    if (!ok) Abort();
    ok = Barrier() && ok;
    if (ok) {
      std::vector<const char*> slots(world_);
      for (int r = 0; r < world_; ++r) slots[r] = Slot(par, r);
      for (const Op& o : ops_) {
        if (o.elem == sizeof(double)) Reduce<double>((double*)(host_ + o.offset), slots, o.offset, o.n, o.op);
        else Reduce<int>((int*)(host_ + o.offset), slots, o.offset, o.n, o.op);
        ok = ok && hipMemcpy(o.buf, host_ + o.offset, o.n * o.elem, hipMemcpyHostToDevice) == hipSuccess;
      }
    }
    ops_.clear();
    bytes_ = 0;
    if (!ok) fprintf(stderr, "rsba: shm collective failed on rank %d\n", rank_);
    return ok;
  }
  void* base_;
  size_t total_;
  const int world_, rank_;
  const size_t slot_bytes_;
  std::string name_;
  int parity_ = 0;
  bool grouped_ = false;
  std::vector<Op> ops_;
  size_t bytes_ = 0;
  char* host_ = nullptr;
  size_t host_cap_ = 0;
};

}  // namespace rsba
