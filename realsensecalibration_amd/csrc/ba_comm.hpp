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
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

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
    static std::mutex mu;
    static std::map<std::string, std::shared_ptr<Comm>> comms;
    std::lock_guard<std::mutex> lk(mu);
    ncclUniqueId id;
    if (world_size > 1) memcpy(&id, id128, sizeof(id));
    else if (ncclGetUniqueId(&id) != ncclSuccess) return nullptr;
    const std::string key = world_size > 1 ? std::string((const char*)&id, sizeof(id)) : std::string("single");
    auto it = comms.find(key);
    if (it != comms.end()) return it->second;
    ncclComm_t c = nullptr;
    if (ncclCommInitRank(&c, world_size > 1 ? world_size : 1, id, world_size > 1 ? rank : 0) != ncclSuccess) return nullptr;
    std::shared_ptr<Comm> p(new RcclComm(c));
    comms.emplace(key, p);
    return p;
  }
  int nranks() const override { int n = 0; return ncclCommCount(c_, &n) == ncclSuccess ? n : 0; }
  const char* kind() const override { return "rccl"; }
  bool GroupStart() override { return Ok(ncclGroupStart()); }
  bool GroupEnd() override { return Ok(ncclGroupEnd()); }
  bool SumDoubles(double* b, size_t n, hipStream_t st) override { return Ok(ncclAllReduce(b, b, n, ncclDouble, ncclSum, c_, st)); }
  bool MaxDoubles(double* b, size_t n, hipStream_t st) override { return Ok(ncclAllReduce(b, b, n, ncclDouble, ncclMax, c_, st)); }
  bool MinInts(int* b, size_t n, hipStream_t st) override { return Ok(ncclAllReduce(b, b, n, ncclInt32, ncclMin, c_, st)); }
 private:
  explicit RcclComm(ncclComm_t c) : c_(c) {}
  static bool Ok(ncclResult_t r) {
    if (r != ncclSuccess) fprintf(stderr, "rsba: RCCL error %s\n", ncclGetErrorString(r));
    return r == ncclSuccess;
  }
  ncclComm_t c_;   // (never destroyed: communicators live as long as the process, like the map that holds them)
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

}  // namespace rsba
