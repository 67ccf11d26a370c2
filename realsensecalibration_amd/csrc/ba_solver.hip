// Device-resident Levenberg-Marquardt solve of the point model: the replacement for the
// ceres::Solve call of /root/reference/Test1_BundleAdjustment/main.cpp:82-87 (and, through BAManager,
// Main_Calibration/bundle_adjustment_manager.cpp:90-95) with DENSE_SCHUR.
//
// The trust-region rules are Ceres 1.14's (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc;
// SURVEY.md Appendix A): accept when rho > 1e-3, radius /= max(1/3, 1-(2rho-1)^3) on success,
// radius /= decrease_factor (2, 4, 8, ...) on failure, and the three tolerances tested in Ceres' order,
// with the candidate DISCARDED when the function/parameter tolerance fires.
//
// Multi-GPU: each rank holds all cameras and a contiguous block of points with their observations.
// Per iteration one RCCL all-reduce (sum) of the packed reduced system [S | gc | corr | diagU | scalars],
// one (max) of the gradient bound, and one tiny (sum) of the candidate scalars.  Every rank then
// factors the identical system, so no broadcast is needed and all ranks take identical decisions.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ba_comm.hpp"
#include "ba_marker_kernels.hpp"
#include "ba_marker_schur.hpp"
#include "ba_point_kernels.hpp"
#include "ba_cholesky_large.hpp"
#include "ba_cholesky_multi.hpp"
#include "ba_cholesky_diag.hpp"
#include "ba_cholesky_tiles.hpp"
#include "ba_problem.hpp"
#include "ba_schur_tiled.hpp"
#include "ba_solver.hpp"

namespace rsba {

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      fprintf(stderr, "rsba: HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); \
      return RSBA_ERR_HIP;                                                                  \
    }                                                                                       \
  } while (0)
#define COMMCHK(expr)                                                                       \
  do {                                                                                      \
    if (!(expr)) {                                                                          \
      fprintf(stderr, "rsba: collective failed at %s:%d\n", __FILE__, __LINE__);            \
      return RSBA_ERR_COMM;                                                                 \
    }                                                                                       \
  } while (0)

int DeviceCount() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------------------------------------
class KernelTimer {
 public:
  // mode 0: off, 1: every kernel, 2: only the two kernels the roofline is quoted on, and only every fourth LM step (the
  // host-side cost of the event records sits on the critical path between two launches: ~40 us per step when every
  // step was recorded, 5% of the step).  Events come from a pool: creating one per record costs more than recording it;
  // they are timing-only (hipEventDisableSystemFence): a default event writes back and invalidates the caches when it
  // completes, between two kernels of the iteration that was measured at ~40 us per recorded kernel.
  void Enable(int mode) { mode_ = mode; on_ = false; }
  bool enabled() const { return mode_ != 0 && sampled_; }
  bool all_kernels() const { return mode_ == 1; }   // the kernels that wait inside (solve, back-substitution) are only timed then
  void NextStep() { ++step_; sampled_ = mode_ == 1 || (mode_ == 2 && (step_ & 3) == 0); }
  static bool Major(const char* n) { return strcmp(n, "k_schur_tiles") == 0 || strcmp(n, "k_linearize_schur_ref") == 0; }
  void Begin(const char* name, hipStream_t s) {
    on_ = sampled_ && (mode_ == 1 || (mode_ == 2 && Major(name)));
    if (!on_) return;
    Pending p; p.name = name; p.a = Get(); p.b = Get();
    if (p.a == nullptr || p.b == nullptr) {   // no event could be created: this launch is simply not timed
      if (p.a) pool_.push_back(p.a);
      if (p.b) pool_.push_back(p.b);
      on_ = false;
      return;
    }
    (void)hipEventRecord(p.a, s);
    pending_.push_back(p);
  }
  void End(hipStream_t s) { if (!on_) return; (void)hipEventRecord(pending_.back().b, s); }
  void Collect() {
    for (auto& p : pending_) {
      float ms = 0;
      if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        auto& st = stats_[p.name]; st.first += 1; st.second += ms;
      }
      pool_.push_back(p.a); pool_.push_back(p.b);
    }
    pending_.clear();
  }
  void Add(const char* name, double ms) { auto& st = stats_[name]; st.first += 1; st.second += ms; }
  void Reset() { Collect(); stats_.clear(); for (hipEvent_t e : pool_) (void)hipEventDestroy(e); pool_.clear(); }
  void Reserve(int n) { while ((int)pool_.size() < n) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) break; pool_.push_back(e); } }
  // (the events of the last run are read HERE, when somebody asks — not at the end of the run, where ~5 us per pair were part of
  //  what a caller times around rsba_solver_run)
  const std::map<std::string, std::pair<int64_t, double>>& Stats() { Collect(); return stats_; }
  // the start of a run: a caller that never asks for the statistics would grow the list of pending pairs, and the event pool with
  // it, run after run — beyond 4096 pairs they are collected here (not always: ~5 us per pair would land in what the caller times)
  void BeginRun() { if (on_ && pending_.size() > 4096) Collect(); }
 private:
  hipEvent_t Get() {
    if (pool_.empty()) Reserve(64);
    if (pool_.empty()) return nullptr;
    hipEvent_t e = pool_.back(); pool_.pop_back();
    return e;
  }
  struct Pending { const char* name; hipEvent_t a, b; };
  bool on_ = false, sampled_ = true;
  int mode_ = 0;
  int64_t step_ = -1;
  std::vector<Pending> pending_;
  std::vector<hipEvent_t> pool_;
  std::map<std::string, std::pair<int64_t, double>> stats_;
};

// roctx ranges around the stages of an LM step (K1 residual + cost, K2 linearise, K3 Schur accumulate, K4
// reduced solve, K5 back-substitution, K6 LM bookkeeping, K7 RCCL), for `rocprofv3 --marker-trace`.  Opt-in (RSBA_ROCTX=1):
// the library is looked up at run time (librocprofiler-sdk-roctx.so, then libroctx64.so), nothing is linked, and without
// the variable a range is one branch.  The ranges bracket the host-side LAUNCHES; the kernels carry their own names.
class RoctxRange {
 public:
  explicit RoctxRange(const char* name) { if (Api().push) { Api().push(name); on_ = true; } }
  ~RoctxRange() { End(); }
  void End() { if (on_) { Api().pop(); on_ = false; } }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
 private:
  struct Fns { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
  static const Fns& Api() {
    static const Fns fns = [] {
      Fns f;
      const char* e = getenv("RSBA_ROCTX");
      if (!e || atoi(e) == 0) return f;
      for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
        if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
          f.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
          f.pop = (int (*)())dlsym(h, "roctxRangePop");
          if (f.push && f.pop) return f;
          f = Fns();
        }
      }
      fprintf(stderr, "rsba: RSBA_ROCTX set but no roctx library could be loaded\n");
      return f;
    }();
    return fns;
  }
  bool on_ = false;
};

template <typename T>
static int DevAlloc(T** p, size_t n) { return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? RSBA_OK : RSBA_ERR_HIP; }

}  // namespace rsba

using namespace rsba;

// RSBA_MC_TRACE: [workgroup][panel 0..15][8] stamps of the diagonal-chain factorisation — up to eight workgroups and the border's
static constexpr size_t kMcTraceWords = (8 + 1) * 16 * 8;

struct rsba_solver {
  rsba_problem* prob = nullptr;
  rsba_options opt;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // Pipelined solve (single GPU, 17..64 cameras): the one-workgroup Cholesky is launched FIRST, on its own hardware
  // queue restricted to one CU (sB), and waits inside the kernel for ready flags; the pair kernel (on the main stream)
  // publishes each camera group's columns as its last workgroups reduce them, the self tiles are the first blocks of the same launch.
  bool pipelined = false;
  int test_stall = 0;        // RSBA_TEST_STALL=1: the Cholesky waits for a tag nobody publishes, =2: the back-substitution does,
                             // =3: a diagonal tile of the persistent tiled factorisation waits for a hand-over nobody writes
                             // (both exercise the fallback to the sequential schedule); =4: the DOUBLE fault — the first pipelined
                             // step stalls as with 1, and its sequential repeat reports a stalled multi-workgroup factorisation once
                             // (the one-workgroup fallback; later steps are NOT forced sequential, so the re-enable logic is what runs)
  bool test_seq_stall_fired = false;
  // max_solver_time_in_seconds with several ranks: every rank its own clock would let the ranks part (one stops, the others hang in a
  // collective).  Rank 0's clock decides: what it says when it LAUNCHES a step rides in small_red[6] through that step's all-reduce of the
  // candidate scalars (the other ranks add 0), and every rank reads the same RES_TIME_UP behind the step (MinimizeLoop).
  bool share_clock = false;
  double time_up = 0.0;
  bool pipeline_off = false; // the one-workgroup fallback was taken with a bordered work list (or any other state the pipelined schedule's
                             // gates do not describe): the solver stays sequential for good
  int step_tag = 0;
  int inject_stall_step = 0; // RSBA_TEST_STALL_STEP=k (with a communicator; RSBA_TEST_STALL_RANK=r: on that rank only): step k reports a
                             // stalled factorisation on this rank — the flag is summed over the ranks and ALL of them repeat the step
  bool tiles_small = false;  // RSBA_TILES_SMALL=1 (opt-in, round 5): the persistent TILED factorisation + chain back-substitution (ba_cholesky_tiles.hpp)
                             // for 17 .. 64 cameras in the SEQUENTIAL schedule: 130 + 26 us against the diagonal-chain kernel's 170 at 64 cameras
                             // (0.436 against 0.449 ms per step).  Opt-in because its roundings are not the pipelined schedule's kernel's: the two
                             // schedules — and a step repeated sequentially after a stall — would no longer add the same bits.
  int chol_wgs = 1;          // > 1: the reduced system is factored by this many workgroups (ba_cholesky_multi.hpp)
  bool chol_diag = false;    // ... with the diagonal chain in workgroup 0 (ba_cholesky_diag.hpp; RSBA_CHOL_DIAG=0: blocks dealt round-robin, ba_cholesky_multi.hpp)
  int border_cols = 0;       // > 0 (RSBA_BORDER=1; 33 .. 64 cameras, no communicator): the diagonal-chain kernel factors the leading 96 B columns and one more
                             // workgroup forms the last camera group as their border (ba_cholesky_border.hpp); both schedules, so that they add the same bits
  int* mc_flags = nullptr;   // tdone[16] | strip_ready[16] | wg_done[8] | error (error[12]: the leading system is through)
  int* tc_flags = nullptr;   // persistent tiled factorisation (more than 64 cameras): tdone[np] | xdone[np][nrt] | error
  int* tc_map = nullptr;     // ... which tile each of its workgroups takes (TileOrder)
  double* tc_hand = nullptr; // ... and the private hand-over buffers of its diagonal chain (TileCholFlags::hand), two sets
  int tc_launches = 0;       //     used by launch parity
  double* tc_xs = nullptr;   // k_backsub_chain: x as it is solved | the helpers' slices of y (two sets each, ba_cholesky_tiles.hpp)
  double* tc_ys = nullptr;
  int tc_bs_launches = 0;
  int tc_np = 0, tc_nrt = 0, tc_tiles = 0;   // 0 tiles: the multi-launch path
  double* mc_dg = nullptr;         // look-ahead sums and unsolved blocks handed over between the workgroups of k_reduced_system_solve_diag
  long long* mc_trace = nullptr;   // RSBA_MC_TRACE=1: stamps of the latest multi-workgroup factorisation
  hipStream_t sB = nullptr;
  // Multi-GPU pipeline: the stage flags the Cholesky waits on are published on the communication stream sR, each after
  // the RCCL all-reduce of that stage's row slab of S (k_wait_stage / k_set_flag, ba_schur_tiled.hpp)
  bool pipelined_mg = false;
  bool pipe_serial = false;            // RSBA_PIPELINE=2: the pipelined schedule's kernels launched one after the other (counter collection)
  hipEvent_t ev_serial[2] = {nullptr, nullptr};
  hipEvent_t ev_tiles = nullptr;       // tile pipeline (more than 64 cameras): the side stream's solve -> the main stream's back-substitution
  size_t tc_nflags = 0, tc_hand_doubles = 0, tc_xs_doubles = 0, tc_ys_doubles = 0;   // sizes of the tiled factorisation's flags and hand-over buffers (reset after a stall)
  hipStream_t sR = nullptr;
  int* ready_global = nullptr;
  long long* chol_waited = nullptr;   // device: ticks the pipelined Cholesky spent waiting for its columns (cumulative)
  long long chol_waited_seen = 0, backsub_waited_seen = 0;   // [1]: the back-substitution's wait for the solve
  long long trace_prev_post = 0;   // RSBA_TRACE=1: device time at which the previous step posted its result
  // the trust-region state the device needs to take the step's decision itself (LmNext, ba_point_kernels.hpp): set by
  // MinimizeLoop before every step; dec: the device's decision block; dec_step: this step queued a damping kernel on it
  double lm_decrease_factor = 2.0;
  double* dec = nullptr;
  bool dec_step = false;
  // Launch-ahead (single GPU, pipelined, up to 64 cameras): the NEXT step's factorisation and Schur kernel are queued behind this
  // step's back-substitution and the damping kernel, on the device's decision, before the host has this step's result (see
  // launch_ahead in PointsStep).  ahead_ok: MinimizeLoop says another step may follow (the iteration limit is not reached);
  // ahead_inflight / ahead_tag: such a pair is queued, with that step tag; ahead_state / ahead_radius: what the device decided
  // (from the result block): the next PointsStep uses the pair if that is what the host asks for, else waits for it and
  // launches its own.
  bool ahead_ok = false, ahead_inflight = false;
  int ahead_tag = 0, ahead_state = 0, ahead_x = 0;   // ahead_x: which buffers held x when the pair was launched
  double ahead_radius = 0.0;
  double* cam_backup = nullptr;    // [6 C + C x CC_STRIDE]: see AheadSel
  // pipelined schedule: steps that timed out (the step is repeated sequentially; the third time-out ends the pipelined
  // schedule for this solver), and "the next pipelined step waits until the factorisation is resident" (first step of a
  // run, first step after a time-out)
  int pipe_stalls = 0;
  int other_stalls = 0;      // stalls of the multi-workgroup / tiled factorisation in the sequential schedule (each is a permanent fallback)
  bool pipe_check_resident = false;
  long long* trace = nullptr;   // RSBA_TRACE=1: 64 wall-clock stamps of the pipelined step
  // RSBA_TRACE=3: the stamps of the last 256 steps in a ring (trace points at the current step's 64 slots), nothing is
  // copied or printed per step — the step's own timing is not disturbed; TraceRingDump prints the gaps at the end of a run
  long long* trace_base = nullptr;
  int trace_ring = 0, trace_ring_first_tag = 0;
  std::chrono::steady_clock::time_point host_t[4];
  // RSBA_HOSTPROF=1: host-side time of a step's phases, summed over the run and printed at its end (no device work, no print per step)
  bool hostprof = false;
  double hp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long hp_n = 0;
  std::chrono::steady_clock::time_point hp_t, hp_result;
  long long* wg_trace = nullptr;  // RSBA_TRACE=2: per-block stamps of the Schur kernel, dumped to RSBA_TRACE_FILE
  long long* bs_wg = nullptr;     // RSBA_TRACE=4: per-workgroup stamps of the point back-substitution, dumped to RSBA_TRACE_FILE
  std::shared_ptr<Comm> comm;   // RCCL, or the one-GPU loopback group (ba_comm.hpp); null: single GPU
  KernelTimer timer;
  std::vector<rsba_iteration> iters;
  double final_cost = 0, final_sumsq = 0, setup_seconds = 0;
  rsba_summary last_summary{};   // of the latest rsba_solver_run (rsba_solver_full_report)
  bool has_run = false;

  // ---- point model
  int C = 0, P = 0, nc = 0;
  int64_t N = 0;
  int max_views = 0;
  RedLayout L{0};
  std::vector<int64_t> order;       // sorted position -> original observation index
  std::vector<int> pt_perm;         // device position of a point -> its index in the problem (empty: identity); BalancedPointOrder
  double *obs_u = nullptr, *obs_v = nullptr, *intr = nullptr;
  // sliced-ELL copy of the observations for the point-centric kernels (ObsSliced)
  int *sl_row_ptr = nullptr, *sl_cam = nullptr;
  double2* sl_uv = nullptr;
  ObsSliced sliced() const { return ObsSliced{sl_row_ptr, sl_uv, sl_cam}; }
  int *obs_cam = nullptr, *pt_ptr = nullptr;
  double *cam[2] = {nullptr, nullptr}, *pts[2] = {nullptr, nullptr}, *camc[2] = {nullptr, nullptr};
  double* cam_free = nullptr;   // [C] 1.0 / 0.0: constant cameras (nullptr when there are none)
  unsigned char* pt_const = nullptr;   // [P] != 0: constant point blocks, in the solver's internal point order (nullptr: none)
  double *cam0 = nullptr, *pts0 = nullptr;  // uploaded initial state (rsba_solver_run restarts from it)
  double *scale_c = nullptr, *scale_p = nullptr;
  double *W = nullptr;     // working copy of the reduced system for the multi-launch Cholesky (nc > RSBA_CHOL_MAXN)
  int* chol_ok = nullptr;
  double *red = nullptr, *A = nullptr, *S_copy = nullptr, *rhs_copy = nullptr, *dcam = nullptr;
  double* red_tri = nullptr;   // the all-reduce payload as lower triangle + vectors (several ranks, more than 64 cameras: TriangularPayload)
  double *block_scal = nullptr, *block_part = nullptr, *small_red = nullptr, *gmax = nullptr, *res = nullptr;
  double* res_host = nullptr;  // pinned, coherent: the last kernel of a step posts res[] here, sequence number in the last slot
  double res_seq = 0.0;
  int grid_lin = 0, grid_pts = 0;
  int cur = 0;
  TiledSchur tiled;
  bool fused_lin = true;     // RSBA_FUSED_LIN=0: every step runs the full point pass (the round-1 schedule)

  // ---- marker-chain model
  MarkerDevice marker;
  MarkerSchurDevice marker_schur;   // time blocks eliminated: the marker-chain model at scale
  bool eliminate_times = false;
};

namespace rsba {

// doubles of the triangular all-reduce payload (lower triangle of S, then g_c, corr, diag U, scalars: k_pack_lower below)
__host__ __device__ inline size_t TriSize(int n) { return (size_t)n * (n + 1) / 2 + 3 * (size_t)n + 8; }

// ------------------------------------------------------------------------------------------------
// Static structure of the tiled Schur kernel: visibility bitsets, tiles, segments.
// ------------------------------------------------------------------------------------------------
// Word bounds of the segments one PAIR tile's points are cut into (ns + 1 values, in 64-point mask words): the same for
// every pair tile.  Shared by TiledSchur::Build and by the point ordering below (whose units are these segments' chunks).
static int DeviceCUs() {
  // (asked once per device: the step asks for it as well)
  static std::mutex mu;
  static std::map<int, int> cached;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cached.find(dev);
  if (it != cached.end()) return it->second;
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  cached[dev] = cus;
  return cus;
}
static std::vector<int> SegmentBounds(int nW, int ns, double taper = 1.0) {
  // Tapered segments (taper > 1: the first of a tile is that many times as long as the last).  A launch is over when its last
  // block is, so long blocks first, short ones last.  At 64 cameras, pipelined, it measured nothing (a stage is only ~1-2
  // "rounds" of slots deep and two blocks share a CU's VALU, so a block's duration follows its CU-mate more than its own
  // length): 1.  Above 64 cameras (sparse pair segments, handed out by point range: the last ranges are the short ones)
  // the launch is 2.6 rounds of ~110 us blocks deep and its tail was 100 us long: 4 (419 -> 385 us at the config-5 shard).
  std::vector<int> bound(ns + 1, 0);
  if (taper == 1.0 && ns >= 1) {
    // as many segments as whole runs of k chunks give (PairSegmentsPerTile's and the self segments' rounding): cut exactly there — an even
    // split of the words (1563 words into 196: 7.97 each) drifts off the chunk boundaries, and a segment that straddles one stages two chunks
    const int k = std::max(1, (int)std::lround((double)nW / ns / RSBA_CW));
    if ((nW + RSBA_CW * k - 1) / (RSBA_CW * k) == ns) {
      for (int i = 0; i <= ns; ++i) bound[i] = std::min(nW, i * RSBA_CW * k);
      return bound;
    }
  }
  const double hi = 2.0 * taper / (taper + 1.0), lo = 2.0 - hi;
  double cum = 0.0;
  for (int i = 0; i < ns; ++i) { cum += ns > 1 ? hi - (hi - lo) * i / (ns - 1) : 1.0; bound[i + 1] = (int)std::llround(nW * cum / ns); }
  bound[ns] = nW;
  bool ok = true;
  for (int i = 0; i < ns; ++i) ok = ok && bound[i + 1] > bound[i];
  if (!ok) for (int i = 0; i <= ns; ++i) bound[i] = (int)((int64_t)nW * i / ns);
  return bound;
}
// The taper of a pair tile's segments and whether its pair segments walk the sparse hit lists — one place for both
// TiledSchur::Build and the point ordering, whose balancing units must be the units the kernel really synchronises on.
// (above 64 cameras in EITHER schedule: the pipelined one there — round 4, the tiled factorisation beside the Schur kernel — runs the
//  same sparse, tapered pair segments in stage order)
static double PairSegmentTaper(int C, bool staged) { (void)staged; return 6 * C > RSBA_CHOL_MAXN ? 4.0 : 1.0; }
static bool SparsePairSegments(int C, bool staged) {
  (void)staged;
  static const bool sparse_on = !(getenv("RSBA_SPARSE_PAIRS") && atoi(getenv("RSBA_SPARSE_PAIRS")) == 0);
  return sparse_on && 6 * C > RSBA_CHOL_MAXN;
}
static int PairSegmentsPerTile(int C, int P, bool staged) {
  const int ngroups = (C + RSBA_TG - 1) / RSBA_TG;
  int npair_tiles = 0;
  for (int ga = 0; ga < ngroups; ++ga) for (int gb = ga; gb < ngroups; ++gb) if (!(ga == gb && std::min(RSBA_TG, C - RSBA_TG * ga) < 2)) ++npair_tiles;
  // measured at 64 cameras: the pipelined schedule likes shorter workgroups (a stage ends with its last one), the
  // sequential one fewer partial sums
  // (RSBA_SEG_TARGET: the same number through the whole-chunk rounding below, which RSBA_SEG_PER_CU bypasses)
  const int seg_per_cu = getenv("RSBA_SEG_PER_CU") ? atoi(getenv("RSBA_SEG_PER_CU")) : getenv("RSBA_SEG_TARGET") ? atoi(getenv("RSBA_SEG_TARGET")) : (6 * C > RSBA_CHOL_MAXN ? 6 : (staged ? 8 : 4));
  const int target = seg_per_cu * DeviceCUs();
  const int nW = (P + 63) / 64;
  int ns = std::max(1, std::min((int)std::lround((double)target / std::max(1, npair_tiles)), nW));
  // Up to 64 cameras a segment is walked in chunks of RSBA_CW words, and a ragged last chunk (9.5 words per segment = a full
  // chunk and 98 points) is a staging round trip and a barrier for a handful of hits per lane: segments of WHOLE chunks —
  // as many chunks as the target length is nearest to.  64 cameras x 125k points (a rank's shard of config 4): 245 segments of one
  // chunk instead of 205 of 9.5 words, 0.437 against 0.471 ms per iteration (round 4); 100k points: 7.6 words, unchanged.
  // (round 6: also when the target length is only NEAR a chunk — three quarters of one or more: at 100k points the target was 7.6 words,
  //  205 ragged segments a tile, each staging a chunk's buffers for 488 points; 196 of exactly one chunk: 0.3398 - 0.3409 ms per step
  //  against 0.3431 - 0.3445, two alternating runs each, RSBA_SEG_TARGET=6 against the default on one box)
  if (!getenv("RSBA_SEG_PER_CU") && !SparsePairSegments(C, staged) && 4 * (long)nW > 3L * RSBA_CW * ns) {
    const int k = std::max(1, (int)std::lround((double)nW / ns / RSBA_CW));
    ns = std::max(1, (nW + RSBA_CW * k - 1) / (RSBA_CW * k));
  }
  return ns;
}

// ------------------------------------------------------------------------------------------------
// Point order for the tiled Schur kernel.
//
// A lane of a pair tile walks the points its two cameras share, chunk by chunk (RSBA_CHUNK points of one segment), and
// a wavefront runs as many trips per chunk as its busiest lane: with the points in file order the hit counts of the
// 64 pairs of a wave are Binomial(~490, (k/C)^2) — at 64 cameras x 20 views the busiest lane has 38 % more hits than the
// mean, i.e. 26 % of the lane-trips of the dominant kernel are masked off.  Which point sits in which chunk is free (points are
// independent given the cameras), so the points are dealt to the chunks such that every camera PAIR gets about the same
// number of shared points in every chunk: greedily, each point (in a fixed pseudo-random order) goes to the best of a
// few candidate chunks, "best" = fewest points so far that share a pair with it, relative to the chunk's fill.
// Measured on the 64 x 100k x 20 problem: lane utilisation of the pair tiles 74 % -> 86 % for ~0.5 s of single-threaded
// set-up; the dealing runs as 8 independent streams on up to 8 host threads.
// The permutation is internal: parameters are uploaded / downloaded through it, nothing the caller sees changes order.
// Deterministic (fixed seed): two solvers of the same problem add in the same order.
// Returns perm (position -> original point); empty = keep the file order.
// ------------------------------------------------------------------------------------------------
static std::vector<int> BalancedPointOrder(int C, int P, bool staged, const std::vector<int>& ptr, const std::vector<int>& cam) {
  const char* env = getenv("RSBA_BALANCE");
  const int mode = env ? atoi(env) : 1;
  if (mode == 0 || C < 2 || P < 4 * RSBA_CHUNK) return {};
  const int nW = (P + 63) / 64;
  const std::vector<int> bound = SegmentBounds(nW, PairSegmentsPerTile(C, P, staged), PairSegmentTaper(C, staged));
  // units: what the lanes of a pair tile's wavefront synchronise on — the 512-point chunks of the masked search, or, with the
  // sparse hit lists (more than 64 cameras), the WHOLE pair segment: its lists run as many trips as the longest of a
  // wavefront's 64, over all of the segment's points.  (Until round 4 the units were cut with taper 1 whatever Build used,
  // and into chunks whatever the pair segments walked: above 64 cameras the balance was computed against the wrong partition.)
  const bool whole_segments = SparsePairSegments(C, staged);
  std::vector<int> ubeg, ucap;
  for (size_t i = 0; i + 1 < bound.size(); ++i) {
    const int step = whole_segments ? std::max(1, bound[i + 1] - bound[i]) : RSBA_CW;
    for (int w = bound[i]; w < bound[i + 1]; w += step) {
      const int we = std::min(w + step, bound[i + 1]);
      ubeg.push_back(64 * w);
      ucap.push_back(std::min(64 * we, P) - 64 * w);
    }
  }
  const int nu = (int)ubeg.size();
  // (the pair counters, 2 bytes per unit and camera pair: at most 64 MB of host memory — beyond that the file order is kept)
  if (nu < 2 || (double)nu * C * C > 3.2e7) return {};
  const int64_t N = ptr[P];
  const double pairs_per_point = N > 0 ? 0.5 * ((double)N / P) * ((double)N / P) : 1.0;
  // candidates per point: bounded work (~6e8 counter reads), at least 2, at most 32 (or RSBA_BALANCE = number)
  int D = mode > 1 ? mode : (int)std::max(2.0, std::min(32.0, 6e8 / (std::max(1.0, pairs_per_point) * P)));
  D = std::min(D, nu);
  std::vector<uint16_t> cnt((size_t)nu * C * C, 0);   // [unit][a][b], a < b
  std::vector<int> fill(nu, 0), unit_of(P, 0);
  // fixed pseudo-random visiting order (splitmix64)
  auto mix = [](uint64_t& st) { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  std::vector<int> visit(P);
  {
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (int j = 0; j < P; ++j) visit[j] = j;
    for (int j = P - 1; j > 0; --j) std::swap(visit[j], visit[(size_t)(mix(st) % (uint64_t)(j + 1))]);
  }
  // Independent streams: the units are cut into kStreams contiguous ranges, the visiting order into runs of matching
  // capacity, and every stream deals its run to its own units.  The number of streams is a constant (not the number of
  // threads that happen to run them): the order, and with it every sum of the solve, is the same on every machine.
  constexpr int kStreams = 8;
  const int nstream = nu >= 4 * kStreams ? kStreams : 1;
  std::vector<int> su(nstream + 1, 0), sp(nstream + 1, 0);
  for (int t = 0; t < nstream; ++t) {
    su[t + 1] = (int)((int64_t)nu * (t + 1) / nstream);
    int capsum = 0;
    for (int g = su[t]; g < su[t + 1]; ++g) capsum += ucap[g];
    sp[t + 1] = sp[t] + capsum;
  }
  if (sp[nstream] != P) return {};   // cannot happen: the capacities add up to P
  std::atomic<int> failed{0};
  auto run_stream = [&](int t) {
    const int g0 = su[t], ng = su[t + 1] - su[t];
    const int Dt = std::min(D, ng);
    uint64_t st = 0xD1B54A32D192ED03ull * (uint64_t)(t + 1);
    int open_from = g0;   // all units of the stream below are full
    for (int q = sp[t]; q < sp[t + 1]; ++q) {
      const int j = visit[q];
      const int b = ptr[j], k = ptr[j + 1] - b;
      const int* cj = cam.data() + b;
      int best = -1; double best_s = 0.0;
      auto score = [&](int g) {
        const uint16_t* c = &cnt[(size_t)g * C * C];
        long sum = 0;
        for (int x = 0; x < k; ++x) { const uint16_t* row = c + (size_t)cj[x] * C; for (int y = x + 1; y < k; ++y) sum += row[cj[y]]; }
        const double sc = (double)sum / (double)(fill[g] + 1);
        if (best < 0 || sc < best_s) { best = g; best_s = sc; }
      };
      if (ng <= D) {
        // few enough units in the stream: all of them that still have room (first minimum wins)
        for (int g = g0; g < g0 + ng; ++g) if (fill[g] < ucap[g]) score(g);
      } else {
        for (int d = 0, tries = 0; d < Dt || best < 0; ++tries) {
          int g;
          if (tries < 4 * Dt) { g = g0 + (int)(mix(st) % (uint64_t)ng); if (fill[g] >= ucap[g]) continue; }
          else { while (open_from < g0 + ng && fill[open_from] >= ucap[open_from]) ++open_from; g = open_from; if (g >= g0 + ng) break; }
          ++d;
          score(g);
          if (tries >= 4 * Dt) break;
        }
      }
      if (best < 0) { failed = 1; return; }
      uint16_t* c = &cnt[(size_t)best * C * C];
      for (int x = 0; x < k; ++x) { uint16_t* row = c + (size_t)cj[x] * C; for (int y = x + 1; y < k; ++y) if (row[cj[y]] != 0xFFFF) ++row[cj[y]]; }
      unit_of[j] = best; ++fill[best];
    }
  };
  {
    // (as many threads as this process may run on — its affinity mask, not the machine's core count; the streams, and with
    //  them the order, do not depend on it)
    cpu_set_t aff;
    const int ncpu = sched_getaffinity(0, sizeof(aff), &aff) == 0 ? CPU_COUNT(&aff) : 1;
    const int nthreads = std::max(1, std::min<int>(nstream, ncpu));
    std::atomic<int> next_stream{0};
    auto worker = [&]() { for (int t = next_stream++; t < nstream; t = next_stream++) run_stream(t); };
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  if (failed) return {};
  // positions: the points of a unit in ascending original order
  std::vector<int> next(ubeg), perm(P, -1);
  for (int j = 0; j < P; ++j) perm[next[unit_of[j]]++] = j;
  for (int q = 0; q < P; ++q) if (perm[q] < 0) return {};
  // Inside every unit (round 5): which 64-point WORD a point sits in decides which half of a DIAGONAL tile's workgroup finds its
  // hits — the tile's 120 pairs sit in both halves, one walks the even words of a chunk, the other the odd ones (PairSegment,
  // PairSegmentSparse) — and the dealing above balances whole units only.  So the unit's points are dealt to its even / odd words
  // such that every pair of cameras of ONE group gets about the same number of shared points in either half (greedy, in the
  // unit's order: the half where the point's diagonal pairs have fewer hits so far, relative to the half's fill).  Measured offline
  // on the 64 x 100k x 20 problem (the kernel's lane -> pair map replayed on the host): lane utilisation of the diagonal tiles'
  // hit loops 69.8 % -> 75.3 %, of all pair tiles 81.9 % -> 83.6 %.
  // The pass walks every point's camera pairs twice, O(P k^2) on one thread: bounded like the dealing above (~6e8 pair visits:
  // 64 views x 1M points would add seconds of set-up, which count against max_solver_time_in_seconds) — beyond that the words stay as
  // dealt, which costs speed only.
  static const bool parity_on = !(getenv("RSBA_BALANCE_PARITY") && atoi(getenv("RSBA_BALANCE_PARITY")) == 0);
  if (parity_on && 2.0 * pairs_per_point * P <= 6e8) {
    std::vector<uint16_t> hc((size_t)C * C * 2);
    std::vector<int> half[2];
    for (int g = 0; g < nu; ++g) {
      const int p0 = ubeg[g], np = ucap[g], nwu = (np + 63) / 64;
      if (nwu < 2) continue;
      int cap[2] = {0, 0};
      for (int w = 0; w < nwu; ++w) cap[w & 1] += std::min(64, np - 64 * w);
      std::fill(hc.begin(), hc.end(), (uint16_t)0);
      half[0].clear(); half[1].clear();
      for (int q = p0; q < p0 + np; ++q) {
        const int j = perm[q], b = ptr[j], k = ptr[j + 1] - b;
        const int* cj = cam.data() + b;
        long sc[2] = {0, 0};
        for (int x = 0; x < k; ++x) for (int y = x + 1; y < k; ++y) if (cj[x] / RSBA_TG == cj[y] / RSBA_TG) { const uint16_t* c = &hc[((size_t)cj[x] * C + cj[y]) * 2]; sc[0] += c[0]; sc[1] += c[1]; }
        int h;
        if ((int)half[0].size() >= cap[0]) h = 1;
        else if ((int)half[1].size() >= cap[1]) h = 0;
        else h = sc[0] * (long)(half[1].size() + 1) <= sc[1] * (long)(half[0].size() + 1) ? 0 : 1;
        half[h].push_back(j);
        for (int x = 0; x < k; ++x) for (int y = x + 1; y < k; ++y) if (cj[x] / RSBA_TG == cj[y] / RSBA_TG) { uint16_t& c = hc[((size_t)cj[x] * C + cj[y]) * 2 + h]; if (c != 0xFFFF) ++c; }
      }
      size_t i0 = 0, i1 = 0;
      for (int w = 0; w < nwu; ++w) { const int n = std::min(64, np - 64 * w); for (int l = 0; l < n; ++l) perm[p0 + 64 * w + l] = (w & 1) ? half[1][i1++] : half[0][i0++]; }
    }
  }
  return perm;
}

int TiledSchur::Build(int C_, int P_, const std::vector<int>& pt_ptr, const std::vector<int>& obs_cam, const std::vector<double>& u, const std::vector<double>& v,
                      const std::vector<int>& sliced_q, bool staged, bool bordered) {
  C = C_; P = P_;
  ngroups = (C + RSBA_TG - 1) / RSBA_TG;
  // Stage of a tile.  Plain: the camera group of its columns — self tile g and the pair tiles (g, g' >= g): what the left-looking
  // factorisation needs for group g's panels.  With the last group Bg as a BORDER (ba_cholesky_border.hpp): the leading system's
  // tiles first — stage g < Bg: self tile g, pair tiles (g, g') with g' < Bg —, then the border's rows, tile (g, Bg) = stage
  // Bg + g, and last the border's own self and pair tile, stage 2 Bg.
  const int Bg = bordered ? ngroups - 1 : -1;
  auto stage_of = [&](int ga, int gb, bool self) { return !bordered ? ga : (ga == Bg ? 2 * Bg : (!self && gb == Bg ? Bg + ga : ga)); };
  nwords = ((P + 63) / 64 + RSBA_CW - 1) / RSBA_CW * RSBA_CW;
  nchunks = nwords / RSBA_CW;
  const int ncam = ngroups * RSBA_TG;
  const int64_t N = pt_ptr[P];
  std::vector<unsigned long long> mask((size_t)ncam * nwords, 0ull);
  std::vector<int> cptr(ncam + 1, 0);
  for (int j = 0; j < P; ++j)
    for (int q = pt_ptr[j]; q < pt_ptr[j + 1]; ++q) { mask[(size_t)obs_cam[q] * nwords + (j >> 6)] |= 1ull << (j & 63); cptr[obs_cam[q] + 1]++; }
  for (int c = 0; c < ncam; ++c) cptr[c + 1] += cptr[c];
  std::vector<int> prefix((size_t)ncam * nwords, 0), cmpos(std::max<int64_t>(N, 1), 0);
  for (int c = 0; c < ncam; ++c) { int run = 0; for (int w = 0; w < nwords; ++w) { prefix[(size_t)c * nwords + w] = run; run += __builtin_popcountll(mask[(size_t)c * nwords + w]); } }
  {
    // camera-major position of every (sorted) observation; a camera seeing the same point twice keeps file order
    std::vector<int> fill(cptr.begin(), cptr.end() - 1);
    for (int j = 0; j < P; ++j) for (int q = pt_ptr[j]; q < pt_ptr[j + 1]; ++q) cmpos[q] = fill[obs_cam[q]]++;
  }
  // tiles: pair tiles (ga <= gb) and one self tile per group.  A workgroup's time per chunk is set by its
  // busiest lane, which is the same for diagonal and off-diagonal pair tiles (~10% of the points) and ~1/5 of
  // that for self tiles (31% of the points dealt to 16 lanes): weights 1 and 1/4.
  std::vector<int> tab; std::vector<double> wt;
  for (int ga = 0; ga < ngroups; ++ga) for (int gb = ga; gb < ngroups; ++gb) {
    if (ga == gb && std::min(RSBA_TG, C - RSBA_TG * ga) < 2) continue;  // a 1-camera group has no off-diagonal pair
    tab.push_back(ga); tab.push_back(gb); tab.push_back(0); wt.push_back(1.0);
  }
  for (int ga = 0; ga < ngroups; ++ga) { tab.push_back(ga); tab.push_back(ga); tab.push_back(1); wt.push_back(0.25); }
  ntiles = (int)wt.size();
  const int cus = DeviceCUs();
  // the pair tiles share `target` workgroups, same number for every tile; the self tiles (much lighter) get 2 per CU in
  // total.  (Sizing each stage's workgroups to whole rounds of slots was tried for the pipelined schedule: no gain, and
  // the two schedules would no longer add in the same order.)
  int npair_tiles = 0;
  nstages = bordered ? 2 * Bg + 1 : ngroups;
  // the order the stages are worked through: a border's rows of group g, tile (g, Bg), right behind the leading system's stage g —
  // the border's workgroup then has one group's time for them (everything it needs of the leading factor is there by then)
  std::vector<int> stage_order;
  if (bordered) { for (int g = 0; g < Bg; ++g) { stage_order.push_back(g); stage_order.push_back(Bg + g); } stage_order.push_back(2 * Bg); }
  else for (int g = 0; g < nstages; ++g) stage_order.push_back(g);
  std::vector<int> stage_of_tile(ntiles, 0), tiles_of_stage(nstages, 0);
  for (int t = 0; t < ntiles; ++t) stage_of_tile[t] = stage_of(tab[3 * t], tab[3 * t + 1], tab[3 * t + 2] != 0);
  for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2]) { ++npair_tiles; ++tiles_of_stage[stage_of_tile[t]]; }
  std::vector<SchurSeg> sg; std::vector<int> tsp(ntiles + 1, 0);
  ngrp = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool self = tab[3 * t + 2] != 0;
    const int nW = (P + 63) / 64;  // mask words that hold points
    // (more than 64 cameras: at most 16 self segments per tile — two reduction groups, no reducer workgroups)
    int ns_self = std::max(1, std::min((2 * cus + ngroups - 1) / ngroups, nW));   // (two self segments per CU over all tiles: the target before the rounding below)
    // (round 6: the self segments too in WHOLE chunks where their target length is three quarters of a chunk or more — they walk the chunk
    //  buffers like the pair segments do.  100k points: 98 segments of two chunks a tile instead of 128 of 12.2 words, 0.3357 - 0.3383 ms per
    //  step against 0.3423; 66 of three 0.3377 - 0.3395, 196 of one 0.3475 - 0.3493 — two alternating runs each on one box)
    if (!SparsePairSegments(C, staged) && 4 * (long)nW > 3L * RSBA_CW * ns_self) {
      const int k = std::max(1, (int)std::lround((double)nW / ns_self / RSBA_CW));
      ns_self = std::max(1, (nW + RSBA_CW * k - 1) / (RSBA_CW * k));
    }
    const int ns = self ? (6 * C > RSBA_CHOL_MAXN ? std::min(ns_self, 16) : ns_self) : PairSegmentsPerTile(C, P, staged);
    const std::vector<int> bound = SegmentBounds(nW, ns, self ? 1.0 : PairSegmentTaper(C, staged));
    for (int i = 0; i < ns; ++i) {
      SchurSeg e; memset(&e, 0, sizeof(e));
      e.ga = tab[3 * t]; e.gb = tab[3 * t + 1]; e.self = tab[3 * t + 2];
      e.word_begin = bound[i]; e.word_end = bound[i + 1];
      sg.push_back(e);
    }
    tsp[t + 1] = (int)sg.size();
    {
      // reduction tree of this tile: groups of RSBA_GRP consecutive segments, but the last segments in groups of 2, 2,
      // 1, 1, 1, 1: a tile is over when its last group has been added, and that group is usually one of the last in order.
      // Up to 64 cameras (both schedules: they add in the same order) the groups are smaller: the last arriver of a group adds it
      // through ONE compute unit (~30 GB/s beside the rest of the kernel: 22 us for eight partial blocks, measured), and a
      // group whose last segment ends late in its stage is what the stage's flag waits for
      // (measured at 64 cameras x 100k points, ms per iteration: groups of 8: 0.411, 6: 0.408, 5: 0.407, 4: 0.406, 3: 0.412, 2: 0.428
      //  — the reducers then read twice the group sums and fall behind)
      // (round 6, beside three factorisation workgroups: 3 and 8 no different from 4 — HISTORY.md)
      const int GRP = 6 * C > RSBA_CHOL_MAXN ? RSBA_GRP : RSBA_GRP_SMALL;
      const int s0 = tsp[t], ns_t = tsp[t + 1] - tsp[t], g0 = ngrp;
      std::vector<int> gsize;
      {
        int left = ns_t;
        // (a group of one segment goes straight into the group sum, SegmentOut in ba_schur_tiled.hpp; four of them and two pairs
        //  at the end: 0.4057 ms against 0.4087 with 1, 1, 2, 4 and 0.409 with eight or more single segments)
        const std::vector<int> tail = {1, 1, 1, 1, 2, 2};
        std::vector<int> last;
        if (ns_t >= 4 * GRP) for (size_t k = 0; k < tail.size() && left > tail[k]; ++k) { last.push_back(tail[k]); left -= tail[k]; }
        while (left > 0) { const int g = std::min(GRP, left); gsize.push_back(g); left -= g; }
        for (int k = (int)last.size() - 1; k >= 0; --k) gsize.push_back(last[k]);
      }
      const int ng = (int)gsize.size();
      int i = 0;
      for (int g = 0; g < ng; ++g) {
        for (int k = 0; k < gsize[g]; ++k, ++i) {
          SchurSeg& e = sg[s0 + i];
          e.tile = t; e.grp = g0 + g; e.grp_seg0 = s0 + i - k; e.grp_nseg = gsize[g];
          e.tile_grp0 = g0; e.tile_ngrp = ng;
          e.stage = stage_of_tile[t]; e.stage_ntiles = tiles_of_stage[stage_of_tile[t]] + 1;   // (set for good below: the arrivals at the stage's counter)
        }
      }
      ngrp += ng;
    }
  }
  nseg = (int)sg.size();
  // reducer workgroups (see GroupReduce in ba_schur_tiled.hpp): entries behind the compute segments
  std::vector<std::vector<int>> red_of_tile(ntiles);
  for (int t = 0; t < ntiles; ++t) {
    const bool self = tab[3 * t + 2] != 0;
    const SchurSeg first = sg[tsp[t]];
    // a pair tile: one reducer per 3 x 3 quadrant of the pairs' blocks, each finishing its own (ReducerQuadrant); a self tile:
    // slices of its 42 components, the last reducer finishes the tile
    // ... a self tile: one reducer per set of components the K factors do not couple (ReducerSelfSet: six sets of six or nine)
    const int nred = first.tile_ngrp <= RSBA_DIRECT_GROUPS ? 0 : (self ? RSBA_SELF_SETS : 4);
    for (int q = tsp[t]; q < tsp[t + 1]; ++q) sg[q].nred = nred;
    for (int r = 0; r < nred; ++r) {
      SchurSeg e = first;
      e.self = self ? 3 : 2; e.nred = nred;
      e.word_begin = r; e.word_end = r + (self ? (r == 1 || r == 5 ? 9 : 6) : 9);   // (set / quadrant r and its number of components)
      red_of_tile[t].push_back((int)sg.size());
      sg.push_back(e);
    }
  }
  // arrivals at a stage's counter: its self tile, and per pair tile the finisher — or each of the four quadrant reducers
  {
    std::vector<int> arrivals(nstages, 0);
    self_arrivals = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int n_t = sg[tsp[t]].nred != 0 ? sg[tsp[t]].nred : 1;
      arrivals[stage_of_tile[t]] += n_t;
      if (tab[3 * t + 2]) self_arrivals += n_t;
    }
    for (auto& e : sg) e.stage_ntiles = arrivals[e.stage];
  }
  nblocks = (int)sg.size();
  nsync = ngrp + 2 * ntiles + RSBA_MAX_STAGES + 2;   // [ngrp] group members | [ntiles] groups done | [RSBA_MAX_STAGES] stage arrivals, [1] self tiles | [ntiles] spare
  // Block order of the launch.  Pipelined: stage by stage — the stage's self tile, its pair tiles, then their reducers —
  // so that camera group g's columns are complete as early as possible.  Sequential schedule: every pair tile first and
  // the (much shorter) self workgroups last, where they fill the tail of the last round of pair workgroups (at 256
  // cameras a pair workgroup runs 220 us and the launch is ~3 rounds deep), reducers behind everything.
  std::vector<int> border; border.reserve(nblocks);
  // More than 64 cameras: the pair segments' hit lists (PairSegmentSparse, ba_schur_tiled.hpp).  RSBA_SPARSE_PAIRS=0: the
  // masked search of the 512-point chunks, as below 65 cameras.
  const bool sparse = SparsePairSegments(C, staged);
  std::vector<int> red_pending;
  if (staged) {
    for (int g : stage_order) {
      std::vector<int> tiles_g;   // the stage's self tile, then its pair tiles
      for (int t = 0; t < ntiles; ++t) if (tab[3 * t + 2] && stage_of_tile[t] == g) tiles_g.push_back(t);
      for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2] && stage_of_tile[t] == g) tiles_g.push_back(t);
      // the stage's tiles interleaved by position, so that the stage as a whole runs long blocks first, short ones last
      // and all its tiles end together
      // (the self tile's segments ahead of the pair tiles', so that its slower finish — seven reducers, a tile sum — ends early: 0.400
      //  against 0.3975 ms, the stage's pair segments then all sit at its end)
      std::vector<std::pair<double, int>> ord;
      for (int t : tiles_g) for (int q = tsp[t]; q < tsp[t + 1]; ++q) ord.push_back({(q - tsp[t] + 0.5) / (tsp[t + 1] - tsp[t]), q});
      std::stable_sort(ord.begin(), ord.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
      // The stage's reducers are drawn RSBA_RED_DELAY entries into the NEXT stage's compute entries, not right behind their own
      // stage's: a reducer holds a workgroup slot (its registers, 72 KB of LDS) from its first poll to the tile's last group, and
      // right behind the stage's last compute entries — which have 40 us to run — that was 46 - 63 us of waiting on 64 slots a launch,
      // 3.5 % of the slot time.  Everything a reducer waits for still has its ticket before it (no deadlock), and who runs what does not
      // change a sum.  Round 5, one box, six alternating runs each: 0 (as before) 0.3464 - 0.3523 (mean 0.3486), 250: 0.3446 - 0.3469
      // (0.3457), 300: 0.3459 - 0.3482 (0.3467), 350: 0.3452 - 0.3573 (0.3485): later than ~300 the stage's flag waits for them.
      static const int red_delay = getenv("RSBA_RED_DELAY") ? atoi(getenv("RSBA_RED_DELAY")) : 250;
      int k = 0;
      for (const auto& o : ord) {
        if (k++ == red_delay) { for (int q : red_pending) border.push_back(q); red_pending.clear(); }
        border.push_back(o.second);
      }
      for (int q : red_pending) border.push_back(q);
      red_pending.clear();
      for (int t : tiles_g) for (int q : red_of_tile[t]) red_pending.push_back(q);
      if (red_delay <= 0) { for (int q : red_pending) border.push_back(q); red_pending.clear(); }
    }
    for (int q : red_pending) border.push_back(q);
  } else if (sparse) {
    // More than 64 cameras (sparse pair segments, PairSegmentSparse): the short self segments first — behind the pair segments
    // their reducers sat in 112 of the 512 slots for 80 us each, waiting for them (there are no reducers any more: at most
    // 16 self segments per tile, two groups, finished by the last arrival) — then the pair segments by POINT RANGE (segment i
    // of every tile, then segment i + 1, ...): the workgroups running at any time gather their point records from a few
    // neighbouring ranges (pair segment 133 -> 110 us)
    for (int q = 0; q < nseg; ++q) if (sg[q].self == 1) border.push_back(q);
    for (int t = 0; t < ntiles; ++t) if (tab[3 * t + 2]) for (int q : red_of_tile[t]) border.push_back(q);
    int ns_pair = 0;
    for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2]) ns_pair = std::max(ns_pair, tsp[t + 1] - tsp[t]);
    for (int i = 0; i < ns_pair; ++i) for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2] && i < tsp[t + 1] - tsp[t]) border.push_back(tsp[t] + i);
    for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2]) for (int q : red_of_tile[t]) border.push_back(q);
  } else {
    for (int q = 0; q < nseg; ++q) border.push_back(q);   // segments are stored pair tiles first, self tiles after them
    for (int t = 0; t < ntiles; ++t) for (int q : red_of_tile[t]) border.push_back(q);
  }
  // First step of a run (pipelined): every self tile ahead of the pair tiles, their reducers right behind them — every
  // camera's diag U is then known early (ready[9]), the factorisation forms its Jacobi scale and is gated stage by stage like
  // in every other iteration instead of waiting for the last stage.
  std::vector<int> border_first;
  if (staged) {
    std::vector<std::pair<double, int>> ord;
    for (int t = 0; t < ntiles; ++t) if (tab[3 * t + 2]) for (int q = tsp[t]; q < tsp[t + 1]; ++q) ord.push_back({(q - tsp[t] + 0.5) / (tsp[t + 1] - tsp[t]), q});
    std::stable_sort(ord.begin(), ord.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
    for (const auto& o : ord) border_first.push_back(o.second);
    for (int t = 0; t < ntiles; ++t) if (tab[3 * t + 2]) for (int q : red_of_tile[t]) border_first.push_back(q);
    for (int g : stage_order) {
      std::vector<int> pair_g;
      for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2] && stage_of_tile[t] == g) pair_g.push_back(t);
      std::vector<std::pair<double, int>> op;
      for (int t : pair_g) for (int q = tsp[t]; q < tsp[t + 1]; ++q) op.push_back({(q - tsp[t] + 0.5) / (tsp[t + 1] - tsp[t]), q});
      std::stable_sort(op.begin(), op.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
      for (const auto& o : op) border_first.push_back(o.second);
      for (int t : pair_g) for (int q : red_of_tile[t]) border_first.push_back(q);
    }
    if ((int)border_first.size() != nblocks) border_first.clear();   // (cannot happen)
  }
  std::vector<int> border_self;
  for (int q = 0; q < nseg; ++q) if (sg[q].self == 1) border_self.push_back(q);
  for (int t = 0; t < ntiles; ++t) if (tab[3 * t + 2]) for (int q : red_of_tile[t]) border_self.push_back(q);
  nblocks_self = (int)border_self.size();
  nseg_pair = 0;
  for (int q = 0; q < nseg; ++q) if (!sg[q].self) ++nseg_pair;  // pair tiles come first, self tiles after them, reducers last
  grid_pp = std::max(1, std::min((P + 255) / 256, 2048));
  int rc;
  if ((rc = DevAlloc(&cam_mask, mask.size())) || (rc = DevAlloc(&segs, (size_t)nblocks)) ||
      (rc = DevAlloc(&ptdata, (size_t)P * RSBA_PT_STRIDE)) ||
      (rc = DevAlloc(&partial, (size_t)nseg * RSBA_PART * 256)) || (rc = DevAlloc(&grp_sum, (size_t)std::max(ngrp, 1) * RSBA_PART * 256)) ||
      (rc = DevAlloc(&sync_cnt, (size_t)nsync)) || (rc = DevAlloc(&grp_flag, (size_t)ngrp)) || (rc = DevAlloc(&block_seg, (size_t)nblocks)) || (rc = DevAlloc(&segs_ordered, (size_t)nblocks)) || (rc = DevAlloc(&segs_ordered_first, (size_t)nblocks)) || (rc = DevAlloc(&segs_ordered_self, (size_t)nblocks_self)) || (rc = DevAlloc(&small_flag, 1)) ||
      (rc = DevAlloc(&tree_error, 2)) || (rc = DevAlloc(&ready, 64)) || (rc = DevAlloc(&block_scal, (size_t)4 * std::max(grid_pp, 2 * cus))) ||
      (rc = DevAlloc(&cam_prefix, prefix.size())) || (rc = DevAlloc(&cam_ptr, cptr.size())) || (rc = DevAlloc(&cm_pos, sliced_q.size())) ||
      (rc = DevAlloc(&sq_cm2[0], cmpos.size())) || (rc = DevAlloc(&sq_cm2[1], cmpos.size())) ||
      (rc = DevAlloc(&lin2[0], (size_t)P * RSBA_LIN_STRIDE)) || (rc = DevAlloc(&lin2[1], (size_t)P * RSBA_LIN_STRIDE)) ||
      (rc = DevAlloc(&u_cm, cmpos.size())) || (rc = DevAlloc(&v_cm, cmpos.size())))
    return rc;
  {
    std::vector<double> ucm(cmpos.size(), 0.0), vcm(cmpos.size(), 0.0);
    for (int64_t q = 0; q < N; ++q) { ucm[cmpos[q]] = u[q]; vcm[cmpos[q]] = v[q]; }
    HIPCHK(hipMemcpy(u_cm, ucm.data(), ucm.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(v_cm, vcm.data(), vcm.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemset(sync_cnt, 0, (size_t)nsync * sizeof(int)));
  HIPCHK(hipMemset(grp_flag, 0, (size_t)std::max(ngrp, 1) * sizeof(int)));
  HIPCHK(hipMemset(tree_error, 0, 2 * sizeof(int)));   // [0] error flag, [1] ticket counter of the Schur kernel
  HIPCHK(hipMemset(ready, 0, 64 * sizeof(int)));   // [0] unused, [1 + g] stage g published, RSBA_READY_*: all self tiles, the solve's done flag, its started counter
  HIPCHK(hipMemcpy(block_seg, border.data(), border.size() * sizeof(int), hipMemcpyHostToDevice));
  {
    for (size_t q = 0; q < sg.size(); ++q) sg[q].index = (int)q;
    std::vector<SchurSeg> ord(border.size()), ord_self(std::max<size_t>(border_self.size(), 1));
    for (size_t b = 0; b < border.size(); ++b) ord[b] = sg[border[b]];
    for (size_t b = 0; b < border_self.size(); ++b) ord_self[b] = sg[border_self[b]];
    HIPCHK(hipMemcpy(segs_ordered, ord.data(), ord.size() * sizeof(SchurSeg), hipMemcpyHostToDevice));
    has_first_order = !border_first.empty();
    if (has_first_order) {
      std::vector<SchurSeg> ordf(border_first.size());
      for (size_t b = 0; b < border_first.size(); ++b) ordf[b] = sg[border_first[b]];
      HIPCHK(hipMemcpy(segs_ordered_first, ordf.data(), ordf.size() * sizeof(SchurSeg), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(segs_ordered_self, ord_self.data(), ord_self.size() * sizeof(SchurSeg), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(small_flag, 0, sizeof(int)));
  }
  HIPCHK(hipMemcpy(cam_mask, mask.data(), mask.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  if (sparse && nseg_pair > 0) {
    const auto th0 = std::chrono::steady_clock::now();
    std::vector<int> tile_of((size_t)ngroups * ngroups, -1);
    for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2]) tile_of[(size_t)tab[3 * t] * ngroups + tab[3 * t + 1]] = t;
    const int nW = (P + 63) / 64;
    // word -> segment of a pair tile (the same bounds for every pair tile)
    std::vector<int> seg_of_word(nW, 0);
    {
      int t0 = -1;
      for (int t = 0; t < ntiles; ++t) if (!tab[3 * t + 2]) { t0 = t; break; }
      for (int q = tsp[t0]; q < tsp[t0 + 1]; ++q) for (int w = sg[q].word_begin; w < sg[q].word_end && w < nW; ++w) seg_of_word[w] = q - tsp[t0];
    }
    // (thread of the workgroup that owns pair (a, b) of point j: see PairSegmentSparse)
    auto owner = [&](int a, int b, int j, int* seg, int* tid) {
      const int ga = a / RSBA_TG, gb = b / RSBA_TG, ia = a - RSBA_TG * ga, ib = b - RSBA_TG * gb;
      const int t = tile_of[(size_t)ga * ngroups + gb];
      const int w = j >> 6;
      *seg = tsp[t] + seg_of_word[w];
      if (ga != gb) { *tid = ia * RSBA_TG + ib; return; }
      const int dt = ia * 15 - ia * (ia - 1) / 2 + ib - ia - 1;           // index of (ia, ib), ia < ib, in kDiagPair
      const int half = (w - sg[*seg].word_begin) & 1;
      *tid = half * 128 + dt;
    };
    std::vector<unsigned> count((size_t)nseg_pair * 256, 0u);
    for (int j = 0; j < P; ++j)
      for (int qa = pt_ptr[j]; qa < pt_ptr[j + 1]; ++qa)
        for (int qb = qa + 1; qb < pt_ptr[j + 1]; ++qb) {
          int seg, tid;
          owner(obs_cam[qa], obs_cam[qb], j, &seg, &tid);
          ++count[(size_t)seg * 256 + tid];
        }
    std::vector<unsigned> off((size_t)nseg_pair * 4, 0u);
    std::vector<int> trips((size_t)nseg_pair * 4, 0);
    size_t entries = 0;
    for (int q = 0; q < nseg_pair; ++q)
      for (int wv = 0; wv < 4; ++wv) {
        unsigned m = 0;
        for (int l = 0; l < 64; ++l) m = std::max(m, count[(size_t)q * 256 + wv * 64 + l]);
        off[(size_t)q * 4 + wv] = (unsigned)entries; trips[(size_t)q * 4 + wv] = (int)m;
        entries += (size_t)m * 64;
      }
    if (entries < (size_t)1 << 32) {
      std::vector<unsigned> h(3 * std::max<size_t>(entries, 1), RSBA_HIT_NONE);
      std::fill(count.begin(), count.end(), 0u);
      for (int j = 0; j < P; ++j)
        for (int qa = pt_ptr[j]; qa < pt_ptr[j + 1]; ++qa)
          for (int qb = qa + 1; qb < pt_ptr[j + 1]; ++qb) {
            int seg, tid;
            owner(obs_cam[qa], obs_cam[qb], j, &seg, &tid);
            const unsigned n = count[(size_t)seg * 256 + tid]++;
            const size_t e = (size_t)off[(size_t)seg * 4 + (tid >> 6)] + (size_t)n * 64 + (tid & 63);
            h[3 * e] = (unsigned)j; h[3 * e + 1] = (unsigned)cmpos[qa]; h[3 * e + 2] = (unsigned)cmpos[qb];
          }
      if ((rc = DevAlloc(&hits, h.size())) || (rc = DevAlloc(&hit_off, off.size())) || (rc = DevAlloc(&hit_trips, trips.size()))) return rc;
      HIPCHK(hipMemcpy(hits, h.data(), h.size() * sizeof(unsigned), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(hit_off, off.data(), off.size() * sizeof(unsigned), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(hit_trips, trips.data(), trips.size() * sizeof(int), hipMemcpyHostToDevice));
      hit_entries = entries;
      if (getenv("RSBA_DEBUG")) {
        size_t nh = 0; for (unsigned c2 : count) nh += c2;
        fprintf(stderr, "rsba: hit lists of %d pair segments: %zu hits in %zu entries (%.0f %% of the lane-trips), %.1f MB, built in %.3f s\n", nseg_pair, nh, entries,
                100.0 * nh / std::max<size_t>(entries, 1), h.size() * 4e-6, std::chrono::duration<double>(std::chrono::steady_clock::now() - th0).count());
      }
    }
  }
  HIPCHK(hipMemcpy(segs, sg.data(), sg.size() * sizeof(SchurSeg), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cam_prefix, prefix.data(), prefix.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cam_ptr, cptr.data(), cptr.size() * sizeof(int), hipMemcpyHostToDevice));
  {
    std::vector<int> cmsl(std::max<size_t>(sliced_q.size(), 1), 0);
    for (size_t e = 0; e < sliced_q.size(); ++e) cmsl[e] = sliced_q[e] >= 0 ? cmpos[sliced_q[e]] : 0;
    HIPCHK(hipMemcpy(cm_pos, cmsl.data(), sliced_q.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  return RSBA_OK;
}

void TiledSchur::Free() {
  void* ptrs[] = {cam_mask, segs, ptdata, partial, grp_sum, tree_error, sync_cnt, grp_flag, ready, block_seg, segs_ordered, segs_ordered_first, segs_ordered_self, small_flag, block_scal, cam_prefix, cam_ptr, cm_pos, sq_cm2[0], sq_cm2[1], lin2[0], lin2[1], u_cm, v_cm, hits, hit_off, hit_trips};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  cam_mask = nullptr; hits = nullptr; hit_off = nullptr; hit_trips = nullptr;
}

static void FreeSolver(rsba_solver* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  s->timer.Reset();
  void* ptrs[] = {s->obs_u, s->obs_v, s->intr, s->obs_cam, s->pt_ptr, s->sl_row_ptr, s->sl_cam, s->sl_uv, s->cam[0], s->cam[1], s->pts[0], s->pts[1], s->camc[0], s->camc[1],
                  s->cam0, s->pts0, s->scale_c, s->scale_p, s->red, s->A, s->W, s->chol_ok, s->S_copy, s->rhs_copy, s->dcam, s->block_scal,
                  s->block_part, s->small_red, s->gmax, s->res, s->dec, s->cam_backup, s->red_tri, s->cam_free, s->pt_const, s->mc_flags, s->mc_dg, s->tc_flags, s->tc_map, s->tc_hand, s->tc_xs, s->tc_ys};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  s->tiled.Free();
  s->marker.Free();
  s->marker_schur.Free();
  for (hipEvent_t e : s->ev_serial) if (e) (void)hipEventDestroy(e);
  if (s->ev_tiles) (void)hipEventDestroy(s->ev_tiles);
  if (s->res_host) (void)hipHostFree(s->res_host);
  if (s->trace_base) (void)hipFree(s->trace_base);
  if (s->wg_trace) (void)hipFree(s->wg_trace);
  if (s->bs_wg) (void)hipFree(s->bs_wg);
  if (s->chol_waited) (void)hipFree(s->chol_waited);
  if (s->sB) (void)hipStreamDestroy(s->sB);
  if (s->sR) (void)hipStreamDestroy(s->sR);
  if (s->ready_global) (void)hipFree(s->ready_global);
  if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

// RSBA_DEBUG=1: synchronise and report after every launch (bisecting device faults).
static bool DebugSync(hipStream_t st, const char* what) {
  static const bool on = getenv("RSBA_DEBUG") != nullptr;
  if (!on) return true;
  hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, "rsba[debug] %s: %s\n", what, hipGetErrorString(e));
  return e == hipSuccess;
}

static IterParams MakeIterParams(const rsba_options& o, double radius, bool first) {
  IterParams ip;
  ip.radius = radius; ip.min_lm_diagonal = o.min_lm_diagonal; ip.max_lm_diagonal = o.max_lm_diagonal;
  ip.huber_delta = o.huber_delta > 0.0 ? (o.loss_type == RSBA_LOSS_CAUCHY ? -o.huber_delta : o.huber_delta) : 0.0;   // signed: see LossAndScale
  ip.first = first ? 1 : 0; ip.jacobi_scaling = o.jacobi_scaling;
  return ip;
}

// The pipelined schedule needs the solver's two streams to run CONCURRENTLY: the factorisation, launched first on sB,
// spins on flags of the Schur kernel behind it on the main stream.  HIP promises no such thing: streams are dealt onto a
// few hardware queues, and two streams of one queue run one after the other (seen after stream churn: 1 solver lifetime in
// ~100 at 64 cameras, more at 33) — the factorisation then waits out its stall budget (0.5 s) before the solver falls back
// to the sequential schedule.  So ask once, at set-up: a waiter on sB and the setter of its flag behind it on the main stream.
__global__ void k_probe_wait(int* flag, long long budget_ticks, int* seen) {
  const long long t0 = wall_clock64();
  int ok = 0;
  while (wall_clock64() - t0 < budget_ticks) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) { ok = 1; break; }
    __builtin_amdgcn_s_sleep(16);
  }
  *seen = ok;
}
__global__ void k_set_double(double* p, double v) { *p = v; }
__global__ void k_probe_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// true: a kernel on `main` starts while one on `side` is running
static bool StreamsRunConcurrently(hipStream_t side, hipStream_t main) {
  int* d = nullptr;
  if (hipMalloc((void**)&d, 2 * sizeof(int)) != hipSuccess) return false;
  int seen = 0;
  bool ok = hipMemset(d, 0, 2 * sizeof(int)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (ok) {
    k_probe_wait<<<1, 1, 0, side>>>(d, 200000 /* 2 ms */, d + 1);
    k_probe_set<<<1, 1, 0, main>>>(d);
    ok = hipStreamSynchronize(side) == hipSuccess && hipStreamSynchronize(main) == hipSuccess &&
         hipMemcpy(&seen, d + 1, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d);
  return ok && seen == 1;
}

// ------------------------------------------------------------------------------------------------
// Upload of the point model: observations are re-ordered by (point, camera) so that one point's
// records are contiguous; the permutation is kept so nothing the caller sees changes order.
// ------------------------------------------------------------------------------------------------
// Streams and events of the pipelined solve.  The Cholesky stream is created with a CU mask — of ALL CUs: what it buys is a
// hardware queue of its own (a waiting kernel must never sit in front of its producers in a shared queue; probed below).
// Round 1 masked it down to bit 0 (one CU per XCD) "to reserve a CU"; a mask reserves nothing — the main stream's kernels
// run on those CUs too — and it has a failure mode: when one of the eight CUs is still busy at dispatch, that workgroup of
// the factorisation waits, the Schur kernel keeps the CU full, and the back-substitution launched behind it fills it with
// workgroups that spin on the factorisation's flag — which now can never start.  Seen as 8-14 stalls (0.5 s each, then the
// sequential schedule for good) per 60 solver lifetimes at 33 cameras, none with the full mask, same speed at 64 cameras.
// RSBA_PIPELINE=0 switches the pipeline off, =3 restores the one-bit mask.
static bool SetupPipeline(rsba_solver* s) {
  const char* env = getenv("RSBA_PIPELINE");
  const int mode = env ? atoi(env) : 1;
  if (mode == 0) return false;
  if (mode == 2) {
    s->pipe_serial = hipEventCreateWithFlags(&s->ev_serial[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s->ev_serial[1], hipEventDisableTiming) == hipSuccess;
    if (!s->pipe_serial) return false;
  }
  s->test_stall = getenv("RSBA_TEST_STALL") ? std::max(1, atoi(getenv("RSBA_TEST_STALL"))) : 0;
  const bool mg = s->comm != nullptr;
  // Multi-GPU pipeline: opt-in (RSBA_PIPELINE_MG=1).  Round 3: the same kernels as the single-GPU pipeline (diagonal-workgroup
  // factorisation reading each stage's all-reduced row slab transposed, the back-substitution's last workgroup leaving the
  // rank's sums for the all-reduce, the decision taken by k_publish_result) — with a 1-rank communicator 0.477 ms per LM
  // iteration at 64 cameras x 100k points against 0.707 (round 2), 0.478 for the sequential multi-GPU schedule and 0.430
  // without a communicator.  What is left of the gap is not in the kernels: the next step's factorisation and Schur kernel
  // START ~35 us later after the posted result than on a process with two hardware queues, whatever the host does (it has
  // launched both 7 us after the result in either case, RSBA_HOSTPROF=1) — the same 0.430 -> 0.470 ms is measured WITHOUT a
  // communicator when two more hardware queues merely exist in the process.  It could not be run on
  // several GPUs in this environment, so it stays opt-in.
  // Round 4: ON by default with a communicator (RSBA_PIPELINE_MG=0: the sequential multi-GPU schedule) — tools/stress_pipeline.py
  // passes under loopback groups of 2, 4 and 8 ranks at 64 cameras, and with a 1-rank communicator it is the faster schedule
  // whatever the queue pool (0.405 ms with GPU_MAX_HW_QUEUES=8, 0.473 with the runtime's default pool, sequential 0.4725).  The one
  // setting it cannot live with is a pool of THREE hardware queues (the waiting kernels land behind each other: every step times
  // out): then the sequential schedule is used, and the library says so once.
  if (mg) {
    static std::once_flag warned;
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int nq = q ? atoi(q) : 0;
    if (getenv("RSBA_PIPELINE_MG") && atoi(getenv("RSBA_PIPELINE_MG")) == 0) return false;
    // Round 5 (ADVICE, medium): over REAL RCCL with more than one rank the pipelined schedule is opt-in (RSBA_PIPELINE_MG=1) until
    // an N >= 2 run on hardware is on record — RCCL's own kernels then need CUs and hardware queues beside the resident waiters
    // (k_wait_stage, the gated factorisation), and every stall costs ten stall budgets before the step is repeated.  Loopback and
    // shared-memory groups and 1-rank communicators, where it has run thousands of steps, keep it as their default.
    if (strcmp(s->comm->kind(), "rccl") == 0 && s->comm->nranks() > 1 && !(getenv("RSBA_PIPELINE_MG") && atoi(getenv("RSBA_PIPELINE_MG")) == 1)) return false;
    if (!s->comm->AllowsResidentWaiters()) return false;
    if (nq == 3) {
      std::call_once(warned, [] { fprintf(stderr, "rsba: GPU_MAX_HW_QUEUES=3 puts the pipelined multi-GPU schedule's waiting kernels behind each other; "
                                                   "using the sequential schedule (set GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises)\n"); });
      return false;
    }
    if (nq != 8 && nq != 1 && nq != 2)
      std::call_once(warned, [nq] { fprintf(stderr, "rsba: a communicator exists and GPU_MAX_HW_QUEUES is %s: the pipelined multi-GPU schedule starts each "
                                                     "step's kernels ~40 us late with the runtime's default pool of hardware queues; set GPU_MAX_HW_QUEUES=8 before the "
                                                     "HIP runtime initialises (INTEGRATION.md)\n", nq ? "not 1, 2 or 8" : "unset"); });
  }
  if (s->C <= RSBA_TG) return false;        // one camera group: nothing to overlap
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, s->device) != hipSuccess) return false;
  if (s->nc > RSBA_CHOL_MAXN) {
    // More than 64 cameras (round 4): the persistent TILED factorisation beside the Schur kernel — OPT-IN (RSBA_PIPELINE_TILES=1),
    // because it is slower than the sequential schedule: 1.04 against 0.955 ms per iteration at 256 cameras x 62.5k points.  One
    // resident workgroup per 64 x 64 tile is 325 of the chip's 512 workgroup slots from the first panel on; the elimination runs on
    // the 187 left (a hit loop alone on its SIMDs is as fast as two, so that is ~73 % of its rate, and the holes retiring tiles
    // leave only fit a Schur workgroup since the tiles ask for as much LDS), its stages come at 160 / 250 / 290 ... us instead of
    // 60 / 100 / ..., and the chain of tile columns, which needs ~18 us per column alone, needs ~27 beside the hit loops: the
    // factorisation ends ~250 us behind the last stage (HISTORY.md, round 4).  What an overlap above 64 cameras needs is a
    // factorisation with a small resident footprint (left-looking, a tile column at a time).  One process / one GPU only, every
    // tile resident at once, at most RSBA_MAX_STAGES camera groups to gate on.
#ifdef RSBA_EXPERIMENTAL
    const char* et = getenv("RSBA_PIPELINE_TILES");
#else
    const char* et = nullptr;   // (-DRSBA_EXPERIMENTAL builds only: slower than the sequential step, and it hung one run of the suite)
#endif
    const char* ec = getenv("RSBA_CHOL_TILES");
    const int m = MultiCholPadded(s->nc), nrt = (m + 1 + 63) / 64, ntiles = nrt * (nrt + 1) / 2;
    if (mg || !(et && atoi(et) == 1) || (ec && atoi(ec) == 0) || ntiles > 2 * prop.multiProcessorCount || (s->C + RSBA_TG - 1) / RSBA_TG > RSBA_MAX_STAGES) return false;
  }
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  std::vector<uint32_t> mask(words, 0xffffffffu);
  // A side stream with a hardware queue of its own (hipExtStreamCreateWithCUMask, all CUs) that passes `probe`; a few attempts:
  // every new stream is dealt onto the next hardware queue.  (Streams from HIP's pool — hipStreamCreateWithFlags — were tried
  // for the multi-GPU pipeline, round 3: they pass the probe at set-up and are dealt onto other queues later; once in ~30 steps
  // the communication stream then sat behind the waiting factorisation and the step timed out.)
  auto side_stream = [&](hipStream_t* out, const std::function<bool(hipStream_t)>& probe) {
    std::vector<hipStream_t> rejected;
    bool got = false;
    for (int attempt = 0; attempt < 6 && !got; ++attempt) {
      if (hipExtStreamCreateWithCUMask(out, words, mask.data()) != hipSuccess) { *out = nullptr; break; }
      got = probe(*out);
      if (!got) { rejected.push_back(*out); *out = nullptr; }
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    return got;
  };
  // (RSBA_PIPELINE=2 launches the kernels one after the other: nothing to probe — and under counter collection, which serialises
  //  kernels, the probe's waiter would never see its flag)
  bool ok = side_stream(&s->sB, [&](hipStream_t c) { return s->pipe_serial || StreamsRunConcurrently(c, s->stream); });
  if (!ok && getenv("RSBA_DEBUG")) fprintf(stderr, "rsba: the side stream does not run beside the main stream, solve not pipelined\n");
  ok = ok && hipMalloc((void**)&s->chol_waited, 2 * sizeof(long long)) == hipSuccess && hipMemset(s->chol_waited, 0, 2 * sizeof(long long)) == hipSuccess;
  if (ok && s->nc > RSBA_CHOL_MAXN) ok = hipEventCreateWithFlags(&s->ev_tiles, hipEventDisableTiming) == hipSuccess;
  if (ok && mg) {
    // The communication stream carries kernels that WAIT (k_wait_stage, for the Schur kernel on the main stream) and kernels
    // others wait for (k_set_flag, for the factorisation on sB): it must run beside both, probed in every direction that
    // occurs.  (Seen when it shared sB's queue: the first stage's exchange sat behind the factorisation that was waiting for
    // it, and every pipelined step timed out.)
    const bool got = side_stream(&s->sR, [&](hipStream_t c) {
      return StreamsRunConcurrently(s->sB, c) && StreamsRunConcurrently(c, s->stream) && StreamsRunConcurrently(s->stream, c);
    });
    ok = got && hipMalloc((void**)&s->ready_global, 16 * sizeof(int)) == hipSuccess && hipMemset(s->ready_global, 0, 16 * sizeof(int)) == hipSuccess;
    s->pipelined_mg = ok;
  }
  if (!ok) { (void)hipGetLastError(); if (getenv("RSBA_DEBUG")) fprintf(stderr, "rsba: CU-masked stream unavailable, solve not pipelined\n"); }
  return ok;
}

static int UploadPoints(rsba_solver* s) {
  const rsba_problem& p = *s->prob;
  s->C = p.num_cameras; s->P = p.num_points; s->N = p.num_observations; s->nc = 6 * s->C; s->L = RedLayout{s->nc};
  const int C = s->C, P = s->P; const int64_t N = s->N;
  s->tiles_small = getenv("RSBA_TILES_SMALL") && atoi(getenv("RSBA_TILES_SMALL")) != 0 && s->nc <= RSBA_CHOL_MAXN && C > RSBA_TG;
  if (std::find(p.camera_constant.begin(), p.camera_constant.end(), (uint8_t)1) != p.camera_constant.end()) {
    std::vector<double> fr(C, 1.0);
    for (int c = 0; c < C && c < (int)p.camera_constant.size(); ++c) if (p.camera_constant[c]) fr[c] = 0.0;
    if (hipMalloc((void**)&s->cam_free, C * sizeof(double)) != hipSuccess ||
        hipMemcpy(s->cam_free, fr.data(), C * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
  }
  std::vector<int> ptr(P + 1, 0);
  for (int64_t i = 0; i < N; ++i) ptr[p.point_index[i] + 1]++;
  int maxk = 0;
  for (int j = 0; j < P; ++j) { maxk = std::max(maxk, ptr[j + 1]); ptr[j + 1] += ptr[j]; }
  s->max_views = maxk;
  std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
  s->order.resize(N);
  for (int64_t i = 0; i < N; ++i) s->order[fill[p.point_index[i]]++] = i;
  for (int j = 0; j < P; ++j)
    std::stable_sort(s->order.begin() + ptr[j], s->order.begin() + ptr[j + 1],
                     [&](int64_t a, int64_t b) { return p.camera_index[a] < p.camera_index[b]; });
  std::vector<double> u(N), v(N);
  std::vector<int> cam(N);
  for (int64_t q = 0; q < N; ++q) { const int64_t i = s->order[q]; u[q] = p.observations[2 * i]; v[q] = p.observations[2 * i + 1]; cam[q] = p.camera_index[i]; }
  int rc;
  if (s->opt.schur_impl != 0) {
    // the tiled kernel keeps one visibility bit per (camera, point): a camera observing the same point
    // twice needs the reference kernel (still on the GPU)
    bool dup = false;
    for (int j = 0; j < P && !dup; ++j) for (int q = ptr[j] + 1; q < ptr[j + 1]; ++q) if (cam[q] == cam[q - 1]) { dup = true; break; }
    if (dup) s->opt.schur_impl = 0;
  }
  const bool any_const_point = std::find(p.point_constant.begin(), p.point_constant.end(), (uint8_t)1) != p.point_constant.end();
  if (any_const_point && s->opt.schur_impl == 0) {
    fprintf(stderr, "rsba: constant point blocks need the tiled Schur kernel (schur_impl != 0, no duplicate observations)\n");
    return RSBA_ERR_UNSUPPORTED;
  }
  if (s->opt.schur_impl != 0) {
    if (getenv("RSBA_TRACE") && atoi(getenv("RSBA_TRACE")) == 2 && s->nc > RSBA_CHOL_MAXN) {
      if (hipMalloc((void**)&s->wg_trace, 3 * 65536 * sizeof(long long)) != hipSuccess) return RSBA_ERR_HIP;   // (the block timeline alone)
    }
    if (getenv("RSBA_TRACE") && s->nc <= RSBA_CHOL_MAXN) {   // the 32 slots are laid out for at most four stages (64 cameras)
      // diagnostics: wall-clock stamps of the step (1) and of every block of the Schur kernel (2)
      s->trace_ring = atoi(getenv("RSBA_TRACE")) == 3 ? 256 : 0;
      const size_t nslot = 64 * (size_t)std::max(1, s->trace_ring);
      if (hipMalloc((void**)&s->trace, nslot * sizeof(long long)) != hipSuccess || hipMemset(s->trace, 0, nslot * sizeof(long long)) != hipSuccess) return RSBA_ERR_HIP;
      s->trace_base = s->trace;
      if (atoi(getenv("RSBA_TRACE")) == 2 && hipMalloc((void**)&s->wg_trace, 3 * 65536 * sizeof(long long)) != hipSuccess) return RSBA_ERR_HIP;
      // (4: every workgroup of the point back-substitution: past the solve's flag, tables staged, pass at x done, block sums out)
      if (atoi(getenv("RSBA_TRACE")) == 4 && (hipMalloc((void**)&s->bs_wg, 4 * 1024 * sizeof(long long)) != hipSuccess || hipMemset(s->bs_wg, 0, 4 * 1024 * sizeof(long long)) != hipSuccess)) return RSBA_ERR_HIP;
    }
    s->pipelined = SetupPipeline(s);
    s->fused_lin = !(getenv("RSBA_FUSED_LIN") && atoi(getenv("RSBA_FUSED_LIN")) == 0);
  }
  {
    // Multi-GPU: the pipelined schedule issues other collectives than the sequential one, so the ranks have to agree.
    // Every rank with a communicator takes part in this one all-reduce (min), whatever its own answer was (a shard
    // with duplicate observations runs schur_impl 0 and cannot pipeline); it is also the communicator's first
    // collective, so connection set-up happens here and not inside a step.
    const char* e1 = getenv("RSBA_PIPELINE"); const char* e2 = getenv("RSBA_PIPELINE_MG");
    const bool mg_possible = s->comm && !(e1 && atoi(e1) == 0) && !(e2 && atoi(e2) == 0);
    if (mg_possible) {
      int h = s->pipelined ? 1 : 0, *d = nullptr;
      if ((rc = DevAlloc(&d, 1))) return rc;
      HIPCHK(hipMemcpy(d, &h, sizeof(int), hipMemcpyHostToDevice));
      COMMCHK(s->comm->MinInts(d, 1, s->stream));
      COMMCHK(s->comm->WaitStream(s->stream));
      HIPCHK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
      (void)hipFree(d);
      if (!h) { s->pipelined = false; s->pipelined_mg = false; }
    }
  }
  if (s->opt.schur_impl != 0) {
    // chunk-balanced point order for the tiled kernel (BalancedPointOrder): everything below is laid out in it
    const auto tb0 = std::chrono::steady_clock::now();
    std::vector<int> perm = BalancedPointOrder(C, P, s->pipelined, ptr, cam);
    if (!perm.empty()) {
      std::vector<int> ptr2(P + 1, 0), cam2(N);
      std::vector<double> u2(N), v2(N);
      std::vector<int64_t> order2(N);
      for (int jn = 0; jn < P; ++jn) ptr2[jn + 1] = ptr2[jn] + (ptr[perm[jn] + 1] - ptr[perm[jn]]);
      for (int jn = 0; jn < P; ++jn) {
        const int b0 = ptr[perm[jn]], n = ptr[perm[jn] + 1] - b0, d0 = ptr2[jn];
        for (int t = 0; t < n; ++t) { u2[d0 + t] = u[b0 + t]; v2[d0 + t] = v[b0 + t]; cam2[d0 + t] = cam[b0 + t]; order2[d0 + t] = s->order[b0 + t]; }
      }
      ptr.swap(ptr2); cam.swap(cam2); u.swap(u2); v.swap(v2); s->order.swap(order2);
      s->pt_perm.swap(perm);
    }
    if (getenv("RSBA_DEBUG")) fprintf(stderr, "rsba: point order %s in %.3f s\n", s->pt_perm.empty() ? "kept" : "balanced over the Schur kernel's chunks",
                                      std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count());
  }

  // sliced-ELL layout: slice = 64 consecutive points, as wide as its widest point
  const int nslices = (P + 63) / 64;
  std::vector<int> sl_ptr(nslices + 1, 0);
  for (int sl = 0; sl < nslices; ++sl) {
    int w = 0;
    for (int j = 64 * sl; j < std::min(P, 64 * sl + 64); ++j) w = std::max(w, ptr[j + 1] - ptr[j]);
    sl_ptr[sl + 1] = sl_ptr[sl] + w;
  }
  const size_t sl_elems = (size_t)sl_ptr[nslices] * 64;
  std::vector<int> sl_q(sl_elems, -1), sl_cam(std::max<size_t>(sl_elems, 1), -1);
  std::vector<double> sl_uv(std::max<size_t>(2 * sl_elems, 2), 0.0);
  for (int j = 0; j < P; ++j)
    for (int q = ptr[j]; q < ptr[j + 1]; ++q) {
      const size_t e = ((size_t)sl_ptr[j >> 6] + (q - ptr[j])) * 64 + (j & 63);
      sl_q[e] = q; sl_cam[e] = cam[q]; sl_uv[2 * e] = u[q]; sl_uv[2 * e + 1] = v[q];
    }

  if ((rc = DevAlloc(&s->obs_u, N)) || (rc = DevAlloc(&s->obs_v, N)) || (rc = DevAlloc(&s->obs_cam, N)) || (rc = DevAlloc(&s->pt_ptr, P + 1)) ||
      (rc = DevAlloc(&s->sl_row_ptr, nslices + 1)) || (rc = DevAlloc(&s->sl_cam, sl_elems)) || (rc = DevAlloc(&s->sl_uv, sl_elems)) ||
      (rc = DevAlloc(&s->intr, 4 * C)) || (rc = DevAlloc(&s->cam[0], 6 * C)) || (rc = DevAlloc(&s->cam[1], 6 * C)) || (rc = DevAlloc(&s->cam0, 6 * C)) ||
      (rc = DevAlloc(&s->pts[0], 3 * (size_t)P)) || (rc = DevAlloc(&s->pts[1], 3 * (size_t)P)) || (rc = DevAlloc(&s->pts0, 3 * (size_t)P)) ||
      (rc = DevAlloc(&s->camc[0], CC_STRIDE * C)) || (rc = DevAlloc(&s->camc[1], CC_STRIDE * C)) || (rc = DevAlloc(&s->scale_c, 6 * C)) ||
      (rc = DevAlloc(&s->scale_p, 3 * (size_t)P)) || (rc = DevAlloc(&s->red, s->L.size())) || (rc = DevAlloc(&s->A, (size_t)(MultiCholPadded(s->nc) + 2) * MultiCholPadded(s->nc) + (size_t)6 * RSBA_TG * s->nc /* the border's rows of L behind a smaller leading factor, ba_cholesky_border.hpp */)) || (rc = DevAlloc(&s->W, (s->nc > RSBA_CHOL_MAXN || s->tiles_small) ? (size_t)(s->nc + 1) * s->nc : 1)) ||
      (rc = DevAlloc(&s->chol_ok, 3)) ||
      (rc = DevAlloc(&s->S_copy, (size_t)s->nc * s->nc)) || (rc = DevAlloc(&s->rhs_copy, s->nc)) || (rc = DevAlloc(&s->dcam, s->nc)) ||
      (rc = DevAlloc(&s->small_red, 8)) || (rc = DevAlloc(&s->gmax, 2)) || (rc = DevAlloc(&s->res, RES_SIZE)) || (rc = DevAlloc(&s->dec, 4)) ||
      (rc = DevAlloc(&s->cam_backup, (size_t)s->C * (6 + CC_STRIDE))))
    return rc;
  if (s->comm) {
    // triangular all-reduce payload: by default where bytes bound the collective (more than 64 cameras), RSBA_TRI_PAYLOAD=1 / 0 forces / disables it
    const char* tp = getenv("RSBA_TRI_PAYLOAD");
    if (tp ? atoi(tp) != 0 : s->nc > RSBA_CHOL_MAXN) { if ((rc = DevAlloc(&s->red_tri, TriSize(s->nc)))) return rc; }
  }
  HIPCHK(hipMemset(s->res, 0, RES_SIZE * sizeof(double)));   // (not every path writes every field: RES_STALL above 64 cameras)
  HIPCHK(hipMemset(s->dec, 0, 4 * sizeof(double)));          // (dec[3]: whose decision it is — no step's yet)
  HIPCHK(hipMemset(s->small_red, 0, 8 * sizeof(double)));
  HIPCHK(hipMemset(s->chol_ok, 0, 3 * sizeof(int)));   // [0] Cholesky status, [1] arrival counter of the back-substitution's blocks, [2] its wait timed out
  HIPCHK(hipMemcpy(s->obs_u, u.data(), N * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->obs_v, v.data(), N * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->obs_cam, cam.data(), N * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->pt_ptr, ptr.data(), (P + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->sl_row_ptr, sl_ptr.data(), (nslices + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->sl_cam, sl_cam.data(), sl_elems * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->sl_uv, sl_uv.data(), sl_elems * sizeof(double2), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->intr, p.intrinsics.data(), 4 * C * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->cam0, p.parameters.data(), 6 * C * sizeof(double), hipMemcpyHostToDevice));
  if (s->pt_perm.empty()) {
    HIPCHK(hipMemcpy(s->pts0, p.parameters.data() + 6 * C, 3 * (size_t)P * sizeof(double), hipMemcpyHostToDevice));
  } else {
    std::vector<double> xp(3 * (size_t)P);
    const double* src = p.parameters.data() + 6 * C;
    for (int jn = 0; jn < P; ++jn) { const size_t o = 3 * (size_t)s->pt_perm[jn]; xp[3 * (size_t)jn] = src[o]; xp[3 * (size_t)jn + 1] = src[o + 1]; xp[3 * (size_t)jn + 2] = src[o + 2]; }
    HIPCHK(hipMemcpy(s->pts0, xp.data(), xp.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (any_const_point) {
    // constant point blocks, in the internal point order; they need the kept linearisation (k_fix_const_lin, ba_point_kernels.hpp)
    if (!s->fused_lin) { fprintf(stderr, "rsba: constant point blocks need the kept point linearisation (RSBA_FUSED_LIN=0 is set)\n"); return RSBA_ERR_UNSUPPORTED; }
    std::vector<unsigned char> pc(P, 0);
    for (int jn = 0; jn < P; ++jn) { const int j = s->pt_perm.empty() ? jn : s->pt_perm[jn]; pc[jn] = j < (int)p.point_constant.size() && p.point_constant[j] ? 1 : 0; }
    if ((rc = DevAlloc(&s->pt_const, (size_t)P))) return rc;
    HIPCHK(hipMemcpy(s->pt_const, pc.data(), (size_t)P, hipMemcpyHostToDevice));
  }
  // launch geometry: fixed grids (deterministic second-stage reductions depend only on these)
  s->grid_lin = std::max(1, std::min((P + 3) / 4, 2048));
  s->grid_pts = std::max(1, std::min((P + 255) / 256, 2048));
  if ((rc = DevAlloc(&s->block_scal, 4 * (size_t)std::max(s->grid_lin, 4096))) || (rc = DevAlloc(&s->block_part, 8 * (size_t)std::max(s->grid_pts, 2 * DeviceCUs())))) return rc;
  // dynamic LDS above 48 KB has to be asked for, once
  if (s->nc <= RSBA_CHOL_MAXN) {
    const size_t lds_c = std::max((size_t)4 * 1024, CholeskyLdsDoubles(s->nc)) * sizeof(double);
    if (lds_c > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_reduced_system_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
  } else {
    HIPCHK(hipFuncSetAttribute((const void*)k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CholStepLdsDoubles() * sizeof(double))));
  }
  {
    // several workgroups for the reduced system: 32 to 64 cameras (padded to whole 32-wide panels), full symmetric S
    // (the diagonal-chain kernel, one workgroup for the diagonal + five for the rows below by default: 0.469 against 0.474 ms per
    //  LM iteration with three row workgroups — fewer blocks per workgroup, more K slices per block; up to 8; the round-robin kernel
    //  RSBA_CHOL_DIAG=0: four, up to RSBA_MC_MAXG)
    const char* e = getenv("RSBA_CHOL_WGS");
#ifdef RSBA_EXPERIMENTAL
    const bool want_diag = !(getenv("RSBA_CHOL_DIAG") && atoi(getenv("RSBA_CHOL_DIAG")) == 0);
#else
    const bool want_diag = true;   // (the round-robin kernel, RSBA_CHOL_DIAG=0, exists in -DRSBA_EXPERIMENTAL builds only)
#endif
    // (the last camera group as a border — three camera groups or more, the tiled Schur kernel, one rank; RSBA_BORDER=0: every camera group
    //  through the diagonal-chain kernel — leaves that kernel nine panels instead of twelve at 64 cameras: ONE diagonal and THREE row
    //  workgroups then (round 6; six alternating runs on one box: 0.3452 - 0.3470 ms per step against 0.3489 - 0.3513 with five row
    //  workgroups, the driver's twenty-step command 0.3485 against 0.3527; 5: 0.3478 - 0.3483, 7 / 8: 0.351 - 0.353 — every workgroup of
    //  the factorisation holds a CU's whole LDS for the length of the step, and the row workgroups' traffic shares the Schur kernel's paths))
    static const bool border_env = !(getenv("RSBA_BORDER") && atoi(getenv("RSBA_BORDER")) == 0);
    const int ngroups_c = (C + RSBA_TG - 1) / RSBA_TG;
    const bool border_ok = border_env && want_diag && !s->comm && ngroups_c >= 3;
    const int want = e ? atoi(e) : (want_diag ? (border_ok ? 3 : 6) : 4);
    if (want > 1 && s->opt.schur_impl != 0 && s->nc >= 6 * RSBA_PB && s->nc <= RSBA_CHOL_MAXN) {
      s->chol_wgs = std::min(want, want_diag ? 8 : RSBA_MC_MAXG);
      if ((rc = DevAlloc(&s->mc_flags, 64))) return rc;
      HIPCHK(hipMemset(s->mc_flags, 0, 64 * sizeof(int)));
      // RSBA_CHOL_DIAG=0: round-robin kernel.  A row workgroup of the diagonal-chain kernel keeps the look-ahead sums of at
      // most four blocks
      const int np_d = MultiCholPadded(s->nc) / RSBA_PB;
      const int np_rule = border_ok ? 3 * (ngroups_c - 1) : np_d;   // panels of the system the diagonal-chain kernel factors (the border's workgroup has the rest)
      s->chol_diag = want_diag && s->chol_wgs >= 2 && (np_rule - 2 + s->chol_wgs - 2) / (s->chol_wgs - 1) <= 4;
#ifdef RSBA_EXPERIMENTAL
      if (!s->chol_diag) s->chol_wgs = std::min(s->chol_wgs, RSBA_MC_MAXG);
#else
      if (!s->chol_diag) s->chol_wgs = 1;   // (more blocks per row workgroup than the diagonal-chain kernel keeps sums for: one workgroup)
#endif
      if (s->chol_diag && (rc = DevAlloc(&s->mc_dg, (size_t)2 * (np_d + 1) * 1024))) return rc;   // look-ahead sums | blocks as handed over
      {
        // the last camera group as a border: three camera groups or more, the tiled Schur kernel, one rank
        if (border_ok && s->chol_diag) s->border_cols = 6 * RSBA_TG * (ngroups_c - 1);
      }
      HIPCHK(hipFuncSetAttribute((const void*)k_reduced_system_solve_diag<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(std::max(DiagCholLdsDoubles(s->nc), BorderLdsDoubles(s->nc)) * sizeof(double))));
      HIPCHK(hipFuncSetAttribute((const void*)k_reduced_system_solve_diag<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(DiagCholLdsDoubles(s->nc) * sizeof(double))));
      // (RSBA_CHOL_WGS allows eight workgroups, and the border's is one more: nine rows of [16][8] stamps)
      if (getenv("RSBA_MC_TRACE")) { if ((rc = DevAlloc(&s->mc_trace, kMcTraceWords))) return rc; HIPCHK(hipMemset(s->mc_trace, 0, kMcTraceWords * sizeof(long long))); }
#ifdef RSBA_EXPERIMENTAL
      HIPCHK(hipFuncSetAttribute((const void*)k_reduced_system_solve_multi, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(MultiCholLdsDoubles(s->nc) * sizeof(double))));
#endif
    }
    // more than 64 cameras: one resident workgroup per 64 x 64 tile, if they all fit on the chip at once
    const char* e2 = getenv("RSBA_CHOL_TILES");
    if ((s->nc > RSBA_CHOL_MAXN || s->tiles_small) && !(e2 && atoi(e2) == 0)) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, s->device) == hipSuccess) {
        const int m = MultiCholPadded(s->nc), nrt = (m + 1 + 63) / 64, ntiles = nrt * (nrt + 1) / 2;
        if (ntiles <= 2 * prop.multiProcessorCount) {
          s->tc_np = m / RSBA_PB; s->tc_nrt = nrt; s->tc_tiles = ntiles;
          const size_t nflags = (size_t)s->tc_np * (nrt + 2) + 1;   // tdone | xdone | error | xdone of the back-substitution
          s->tc_nflags = nflags;
          if ((rc = DevAlloc(&s->tc_flags, nflags))) return rc;
          HIPCHK(hipMemset(s->tc_flags, 0, nflags * sizeof(int)));
          {
            const char* e3 = getenv("RSBA_TILE_ORDER");   // 0: row by row (the pairing of tiles on shared CUs left to chance)
            if (!(e3 && atoi(e3) == 0)) {
              const std::vector<int> order = TileOrder(nrt, prop.multiProcessorCount);
              if ((rc = DevAlloc(&s->tc_map, order.size()))) return rc;
              HIPCHK(hipMemcpy(s->tc_map, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
            }
          }
          s->tc_hand_doubles = (size_t)2 * nrt * kTileHandDoubles;
          if ((rc = DevAlloc(&s->tc_hand, (size_t)2 * nrt * kTileHandDoubles))) return rc;
          HIPCHK(hipMemset(s->tc_hand, 0xff, (size_t)2 * nrt * kTileHandDoubles * sizeof(double)));   // the sentinel everywhere
          if (getenv("RSBA_MC_TRACE")) {
            // (RSBA_TILES_SMALL: the diagonal-chain kernel was given a buffer above, and still stamps its [9][16][8] words on the steps it
            //  runs — one buffer large enough for both)
            const size_t words = std::max((size_t)(nrt + 1) * 24 + ntiles, s->mc_trace ? kMcTraceWords : (size_t)0);
            if (s->mc_trace) { (void)hipFree(s->mc_trace); s->mc_trace = nullptr; }
            if ((rc = DevAlloc(&s->mc_trace, words))) return rc;
            HIPCHK(hipMemset(s->mc_trace, 0, words * sizeof(long long)));
          }
          {
            const int H = (s->tc_np + 2) / 3;
            // k_backsub_chain's hand-overs, two sets each, the sentinel everywhere: x and the helpers' slices of y
            s->tc_xs_doubles = (size_t)2 * m; s->tc_ys_doubles = (size_t)2 * H * 96;
            if ((rc = DevAlloc(&s->tc_xs, (size_t)2 * m)) || (rc = DevAlloc(&s->tc_ys, (size_t)2 * H * 96))) return rc;
            HIPCHK(hipMemset(s->tc_xs, 0xff, (size_t)2 * m * sizeof(double)));
            HIPCHK(hipMemset(s->tc_ys, 0xff, (size_t)2 * H * 96 * sizeof(double)));
          }
          HIPCHK(hipFuncSetAttribute((const void*)k_chol_tiles_persistent, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)std::max(TileCholLdsDoubles() * sizeof(double), (size_t)RSBA_SCHUR_LDS_BYTES + 64)));
        }
      }
    }
  }
  if (s->pipelined && s->nc > RSBA_CHOL_MAXN && s->tc_tiles == 0) s->pipelined = false;   // (the persistent tiles did not fit after all)
  if (s->opt.schur_impl != 0) {
    rc = s->tiled.Build(C, P, ptr, cam, u, v, sl_q, s->pipelined, s->border_cols > 0);
    if (rc != RSBA_OK) return rc;
  } else if (maxk > 64) {
    fprintf(stderr, "rsba: schur_impl=0 handles at most 64 views per point (problem has %d)\n", maxk);
    return RSBA_ERR_UNSUPPORTED;
  }
  return RSBA_OK;
}

// Device point array (3P, in the solver's internal point order) -> dst in the problem's own order.
static int DownloadPoints(const rsba_solver* s, const double* dev, double* dst) {
  const size_t P = (size_t)s->P;
  if (s->pt_perm.empty()) return hipMemcpy(dst, dev, 3 * P * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? RSBA_OK : RSBA_ERR_HIP;
  std::vector<double> xp(3 * P);
  if (hipMemcpy(xp.data(), dev, 3 * P * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return RSBA_ERR_HIP;
  for (size_t jn = 0; jn < P; ++jn) { const size_t o = 3 * (size_t)s->pt_perm[jn]; dst[o] = xp[3 * jn]; dst[o + 1] = xp[3 * jn + 1]; dst[o + 2] = xp[3 * jn + 2]; }
  return RSBA_OK;
}

// The uploaded start back into the working state: cameras and points in ONE launch (two hipMemcpyAsync cost the host 25 - 30 us at
// the start of every run — the runtime's copy path — where a kernel launch is 3)
__global__ void __launch_bounds__(256) k_reset_state(double* __restrict__ cam, const double* __restrict__ cam0, size_t ncam,
                                                     double* __restrict__ pts, const double* __restrict__ pts0, size_t npts) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ncam + npts; i += (size_t)gridDim.x * 256) {
    if (i < ncam) cam[i] = cam0[i];
    else pts[i - ncam] = pts0[i - ncam];
  }
}

static int ResetPoints(rsba_solver* s) {
  const size_t ncam = 6 * (size_t)s->C, npts = 3 * (size_t)s->P;
  k_reset_state<<<(unsigned)std::min<size_t>((ncam + npts + 255) / 256, 2048), 256, 0, s->stream>>>(s->cam[0], s->cam0, ncam, s->pts[0], s->pts0, npts);
  HIPCHK(hipGetLastError());
  s->cur = 0;
  s->tiled.lin_valid = false;   // a run starts with a full point pass at the uploaded point
  s->tiled.pt_valid = false;
  return RSBA_OK;
}


void TiledSchur::LaunchPointPass(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st) {
  const int x = s->cur;
  const bool stage = (size_t)C * RSBA_CC_LDS * sizeof(double) <= 56 * 1024;
  const size_t lds = stage ? (size_t)C * RSBA_CC_LDS * sizeof(double) : 0;
  T.Begin("k_point_pass", st);
  if (stage)
    k_point_pass<true><<<grid_pp, 256, lds, st>>>(C, P, s->sliced(), s->camc[x], s->pts[x], s->scale_p, ptdata,
                                                  block_scal, cm_pos, sq_cm2[x], lin2[x], small_flag, ip);
  else
    k_point_pass<false><<<grid_pp, 256, lds, st>>>(C, P, s->sliced(), s->camc[x], s->pts[x], s->scale_p, ptdata,
                                                   block_scal, cm_pos, sq_cm2[x], lin2[x], small_flag, ip);
  T.End(st);
  pt_valid = true; pt_state = x; pt_radius = ip.radius; scal_blocks = grid_pp;
}

void TiledSchur::LaunchPointDamp(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st) {
  const int x = s->cur;
  T.Begin("k_point_damp", st);
  k_point_damp<<<grid_pp, 256, 0, st>>>(P, s->pts[x], s->scale_p, lin2[x], ptdata, block_scal, C, s->camc[x], small_flag, ip, s->trace);
  T.End(st);
  pt_valid = true; pt_state = x; pt_radius = ip.radius; scal_blocks = grid_pp;
}

// The point side of a step: a full pass over the observation records only when x has no linearisation yet (the first
// step of a run, or RSBA_FUSED_LIN=0); otherwise the kept one is damped with this step's radius.
static void LaunchPointSide(TiledSchur& ts, rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st) {
  RoctxRange rr("K1+K2 point side: residuals, cost, point blocks (or their damping)");
  if (ts.lin_valid && !ip.first) {
    // (nothing to launch when the previous step's back-substitution has left exactly these blocks: k_backsub_candidate_proj)
    if (!(ts.pt_valid && ts.pt_state == s->cur && ts.pt_radius == ip.radius)) ts.LaunchPointDamp(s, ip, T, st);
  } else {
    ts.LaunchPointPass(s, ip, T, st);
  }
}

// The Schur elimination launch: self segments first, then the pair tiles stage by stage; results land in s->red.
// tag != 0: the ready flags are published for a Cholesky that is already waiting (pipelined schedule).
// Workgroups of a launch of the Schur kernel.  More than 64 cameras (the sparse instance): as many as the chip holds at once (two
// per CU), each drawing tickets until the work list is through; otherwise one per entry.
static int SchurGrid(int entries, bool sparse) {
  // (RSBA_RESIDENT builds: resident up to 64 cameras too — a few slots fewer than the chip holds: the factorisation's six workgroups, 160 KB
  //  of LDS each, are resident beside the kernel, and a workgroup that found no slot would take its tickets late)
  if (RSBA_RESIDENT != 0 && !sparse) { static const int spare = getenv("RSBA_RESIDENT_SPARE") ? atoi(getenv("RSBA_RESIDENT_SPARE")) : 12; return std::min(entries, std::max(1, 2 * DeviceCUs() - spare)); }
  return sparse ? std::min(entries, 2 * DeviceCUs()) : entries;
}

// ------------------------------------------------------------------------------------------------
// Several ranks, more than 64 cameras: the all-reduce payload as LOWER TRIANGLE + vectors (SURVEY 8e priced the
// triangular 9.4 MB at 256 cameras; the full square S || g_c || corr || diag U || scalars is 18.9 MB, and an
// all-reduce over a ring of point-to-point xGMI links is bound by bytes per link).  S is written symmetric to the bit by
// the Schur kernel's finishers (every block and its mirror from the same registers), so summing the lower triangle over
// the ranks and mirroring it gives exactly the bits the full all-reduce gives.  k_pack_lower: row i of the triangle =
// S[i][0..i] behind each other, then the 3 n + 8 doubles behind S as they are; k_unpack_lower writes both halves back.
// Up to 64 cameras the payload is 1.19 MB and the collective latency-bound: two more launches would cost more than the
// bytes save (RSBA_TRI_PAYLOAD=1 forces the packed payload there too, =0 disables it; tests/test_gpu_loopback.py).
// ------------------------------------------------------------------------------------------------
// sym_full == 0 (schur_impl 0: k_linearize_schur_ref accumulates the UPPER block triangle only, the lower half of S stays at the
// memset's zeros): element (i, j), j <= i, of the triangle is read from its mirror (j, i), which that kernel does fill.  A group may
// mix ranks of both kinds (a shard with duplicate observations runs schur_impl 0): each packs its own S correctly, the sum is right,
// and k_unpack_lower writes both halves, which the consumers of either kind accept.
__global__ void __launch_bounds__(256) k_pack_lower(int n, const double* __restrict__ red, RedLayout L, double* __restrict__ tri, int sym_full) {
  const size_t nt = (size_t)n * (n + 1) / 2, total = nt + 3 * (size_t)n + 8;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
    if (k >= nt) { tri[k] = red[L.gc() + (k - nt)]; continue; }
    // row of element k of the triangle: the largest i with i (i + 1) / 2 <= k (the root in floating point, corrected in integers)
    size_t i = (size_t)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > k) --i;
    while ((i + 1) * (i + 2) / 2 <= k) ++i;
    const size_t j = k - i * (i + 1) / 2;
    tri[k] = sym_full ? red[L.S() + i * (size_t)n + j] : red[L.S() + j * (size_t)n + i];
  }
}
__global__ void __launch_bounds__(256) k_unpack_lower(int n, const double* __restrict__ tri, double* __restrict__ red, RedLayout L) {
  const size_t nn = (size_t)n * n, nt = (size_t)n * (n + 1) / 2, total = nn + 3 * (size_t)n + 8;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
    if (k >= nn) { red[L.gc() + (k - nn)] = tri[nt + (k - nn)]; continue; }
    const size_t i = k / n, j = k - i * n, hi = i > j ? i : j, lo = i > j ? j : i;
    red[L.S() + k] = tri[hi * (hi + 1) / 2 + lo];
  }
}

// The instance of the Schur kernel that holds the pair segments this problem runs: sparse hit lists above 64 cameras, the masked
// search up to 64.
static void LaunchSchurInstance(const SchurArgs& a, int grid, bool loss, hipStream_t st) {
#define RSBA_LAUNCH_SCHUR(LOSS, MODE) k_schur_tiles<LOSS, MODE><<<grid, 256, 0, st>>>(a.ticket, a.ticket_base, a.total, a.small_flag, a.segs_ordered, a)
  if (a.hits != nullptr) { if (loss) RSBA_LAUNCH_SCHUR(true, RSBA_PAIRS_SPARSE); else RSBA_LAUNCH_SCHUR(false, RSBA_PAIRS_SPARSE); }
  else if (loss) RSBA_LAUNCH_SCHUR(true, RSBA_PAIRS_MASKED);
  else RSBA_LAUNCH_SCHUR(false, RSBA_PAIRS_MASKED);
#undef RSBA_LAUNCH_SCHUR
}

static SchurArgs MakeSchurArgs(TiledSchur& ts, rsba_solver* s, int tag) {
  const int x = s->cur;
  SchurArgs a;
  a.C = ts.C; a.P = ts.P; a.nwords = ts.nwords; a.camc = s->camc[x]; a.cam_free = s->cam_free; a.segs = ts.segs; a.cam_mask = ts.cam_mask; a.ptdata = ts.ptdata;
  a.cam_prefix = ts.cam_prefix; a.cam_ptr = ts.cam_ptr; a.sq_cm = ts.sq_cm2[x]; a.u_cm = ts.u_cm; a.v_cm = ts.v_cm; a.partial = ts.partial;
  a.grp_sum = ts.grp_sum; a.sync_cnt = ts.sync_cnt; a.ngrp = ts.ngrp; a.ntiles = ts.ntiles; a.segs_ordered = ts.segs_ordered; a.small_flag = ts.small_flag; a.last_group = ts.ngroups - 1;
  a.tree_error = ts.error_flag ? ts.error_flag : ts.tree_error; a.ticket = ts.tree_error + 1;
  a.ready = ts.ready; a.tag = tag; a.self_only = 0; a.red = s->red; a.L = s->L; a.nblocks_pp = ts.scal_blocks; a.block_scal = ts.block_scal; a.gmax_p = s->gmax;
  a.trace = s->trace; a.wg_trace = s->wg_trace;
  a.grp_flag = ts.grp_flag; a.epoch = ++ts.epoch;
  a.hits = ts.hits; a.hit_off = ts.hit_off; a.hit_trips = ts.hit_trips;
  a.all_self = 0; a.self_arrivals = ts.self_arrivals;
  return a;
}

void TiledSchur::LaunchTiles(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st, int tag, bool first_staged, bool ahead, long long* ahead_trace) {
  RoctxRange rr("K2+K3 camera-side rows + Schur elimination into the reduced system");
  SchurArgs a = MakeSchurArgs(*this, s, tag);
  if (first_staged) { a.segs_ordered = segs_ordered_first; a.all_self = 1; }
  if (ahead) {
    // the NEXT step's elimination, queued before this step's outcome is known: state by the device's decision (SchurArgs::dec)
    const int c = 1 - s->cur;
    a.dec = s->dec; a.camc_alt = s->camc[c]; a.sq_cm_alt = sq_cm2[c];
    a.trace = ahead_trace; a.wg_trace = nullptr;
  }
  const bool sparse = a.hits != nullptr;   // (resident workgroups: as many as the chip holds, each drawing tickets until the list is through)
  const int grid = SchurGrid(nblocks, sparse);
  a.total = nblocks; a.ticket_base = ticket_base; ticket_base += (unsigned)(nblocks + ((sparse || RSBA_RESIDENT != 0) ? grid : 0));
  T.Begin("k_schur_tiles", st);
  LaunchSchurInstance(a, grid, ip.huber_delta != 0.0, st);
  T.End(st);
}

void TiledSchur::LaunchSelfOnly(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st) {
  RoctxRange rr("K2 camera gradient only (self tiles)");
  SchurArgs a = MakeSchurArgs(*this, s, 0);
  a.segs_ordered = segs_ordered_self; a.self_only = 1; a.trace = nullptr; a.wg_trace = nullptr;
  const int grid = SchurGrid(nblocks_self, false);   // (the self-only pass runs the masked instance at any size)
  a.total = nblocks_self; a.ticket_base = ticket_base; ticket_base += (unsigned)(nblocks_self + (RSBA_RESIDENT != 0 ? grid : 0));
  T.Begin("k_schur_tiles(self only)", st);
  a.hits = nullptr;   // (one entry per workgroup; a self segment is the same code in every instance)
  LaunchSchurInstance(a, grid, ip.huber_delta != 0.0, st);
  T.End(st);
}

int TiledSchur::Launch(rsba_solver* s, const IterParams& ip, KernelTimer& T) {
  hipStream_t st = s->stream;
  LaunchPointSide(*this, s, ip, T, st);
  LaunchTiles(s, ip, T, st, 0);
  return RSBA_OK;
}

// Wait for the result block of the step (or gradient evaluation) just enqueued: poll the sequence word the last kernel
// posts (PostToHost).  After 2 s of polling the stream that POSTS the result is synchronised (the in-kernel waits give up
// after RSBA_STALL_TICKS, ten times that in the multi-GPU pipeline, so the synchronisation returns); a dead queue cannot
// spin us forever.
// RSBA_TRACE=3: the step-to-step gaps of the run's last steps, from the ring of device stamps (no per-step copies).
static void TraceRingDump(rsba_solver* s) {
  if (!s->trace_ring || !s->trace_base || s->trace_ring_first_tag == 0) return;
  if (hipStreamSynchronize(s->stream) != hipSuccess) return;
  std::vector<long long> h((size_t)64 * s->trace_ring);
  if (hipMemcpy(h.data(), s->trace_base, h.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return;
  const int last = s->step_tag, first = std::max(s->trace_ring_first_tag + 1, last - s->trace_ring + 2);
  std::vector<double> period, head, tail, damp_gap;
  for (int t = first; t <= last; ++t) {
    const long long* a = &h[(size_t)64 * ((t - 1) % s->trace_ring)];
    const long long* b = &h[(size_t)64 * (t % s->trace_ring)];
    if (a[24] == 0 || b[24] == 0 || a[29] == 0) continue;
    period.push_back((b[24] - a[24]) * 0.01);          // first Schur block of one step -> of the next
    head.push_back((b[24] - a[29]) * 0.01);            // result posted -> first Schur block of the next step
    tail.push_back((a[29] - a[15]) * 0.01);            // the solve's end -> result posted (back-substitution)
    // result posted -> the damping kernel's end: queued behind the step on the device's decision it stamps into the step's own
    // window, launched by the host for the next step into that one's
    if (a[31] > a[29]) damp_gap.push_back((a[31] - a[29]) * 0.01);
    else if (b[31] > a[29]) damp_gap.push_back((b[31] - a[29]) * 0.01);
  }
  auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  fprintf(stderr, "rsba[ring] %zu steps: period %.1f us = first tile block -> solve end ... -> result posted (%.1f after the solve) -> next first tile block %.1f"
                  " (damping kernel done %.1f after the post)\n", period.size(), med(period), med(tail), med(head), med(damp_gap));
  s->trace_ring_first_tag = 0;
}

// Host-side wait for everything the solver has in flight (the stall handlers, before they reset flags and repeat a step).  With a
// communicator every stream that may carry a collective is waited for through Comm::WaitStream — bounded, the communicator's health
// polled beside it — so that a peer that is gone ends this rank with RSBA_ERR_COMM instead of leaving it in hipDeviceSynchronize for
// ever.
static int SyncSolver(rsba_solver* s) {
  if (!s->comm) { HIPCHK(hipDeviceSynchronize()); return RSBA_OK; }
  for (hipStream_t st : {s->stream, s->sB, s->sR}) {
    if (st == nullptr) continue;
    if (!s->comm->WaitStream(st)) return RSBA_ERR_COMM;
  }
  return RSBA_OK;
}

static int WaitResult(rsba_solver* s, hipStream_t posting) {
  s->res_seq += 1.0;
  volatile double* seq = s->res_host + (RES_SIZE - 1);
  auto t_poll = std::chrono::steady_clock::now();
  while (*seq != s->res_seq) {
    __builtin_ia32_pause();
    if (std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(2)) {
      // (with a communicator the stream may sit in a collective whose peer is gone: a bounded wait that polls the communicator's
      //  health and aborts it, ba_comm.hpp — the process then leaves with RSBA_ERR_COMM instead of hanging)
      if (s->comm) { if (!s->comm->WaitStream(posting)) return RSBA_ERR_COMM; }
      else HIPCHK(hipStreamSynchronize(posting));
      if (*seq != s->res_seq) { fprintf(stderr, "rsba: step finished without posting its result\n"); return RSBA_ERR_HIP; }
      t_poll = std::chrono::steady_clock::now();
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return RSBA_OK;
}

// One "solve at radius": linearise at x, reduce, factor, back-substitute, evaluate the candidate.
// On return res_host holds the RES_* block (host has synchronised).
// A factorisation + Schur kernel launched ahead that nobody will consume (the host stopped, or asks for something else): they
// run to their end on the device's decision — the step tag they used is skipped.
static int DrainAhead(rsba_solver* s) {
  if (!s->ahead_inflight) return RSBA_OK;
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->sB) HIPCHK(hipStreamSynchronize(s->sB));
  ++s->step_tag;
  s->ahead_inflight = false;
  if (s->ahead_state != s->ahead_x && s->cur == s->ahead_x) {
    // the device had accepted the step, the host has not (it stopped on a tolerance or the time limit, which the device does not
    // test): the factorisation launched ahead has written its candidate where x is — x's cameras and constants back (AheadSel)
    const size_t nc6 = 6 * (size_t)s->C, ncc = (size_t)s->C * CC_STRIDE;
    HIPCHK(hipMemcpyAsync(s->cam[s->ahead_x], s->cam_backup, nc6 * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->camc[s->ahead_x], s->cam_backup + nc6, ncc * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
  }
  return RSBA_OK;
}

static int PointsStep(rsba_solver* s, double radius, bool first, bool keep_system_copy) {
  const int C = s->C, P = s->P, x = s->cur, c = 1 - s->cur;
  hipStream_t st = s->stream;
  IterParams ip = MakeIterParams(s->opt, radius, first);
  ip.cam_free = s->cam_free;
  ip.pt_const = s->pt_const;
  KernelTimer& T = s->timer;

  if (s->trace) s->host_t[0] = std::chrono::steady_clock::now();
  auto hp = [&](int k) { if (s->hostprof) { const auto now = std::chrono::steady_clock::now(); s->hp_sum[k] += std::chrono::duration<double, std::micro>(now - s->hp_t).count(); s->hp_t = now; } };
  if (s->hostprof) { s->hp_t = std::chrono::steady_clock::now(); if (s->hp_n++ > 0) s->hp_sum[0] += std::chrono::duration<double, std::micro>(s->hp_t - s->hp_result).count(); }
  T.NextStep();
  DebugSync(st, "enter PointsStep");
  // camera constants at x: only the first step of a run computes them; afterwards x is either unchanged (rejected step)
  // or the former candidate, whose constants k_reduced_system_solve already wrote into camc[c] before the swap
  if (first) {
    T.Begin("k_camera_constants", st);
    k_camera_constants<<<(C + 63) / 64, 64, 0, st>>>(C, s->cam[x], s->intr, s->camc[x]);
    T.End(st);
  }
  DebugSync(st, "k_camera_constants");
  // (above 64 cameras the pipelined schedule needs the resident tiles building their own entries of the system, the chain
  //  back-substitution, and — on the first step of a run — the work list that runs every self tile first)
  static const int fuse_env = getenv("RSBA_SYS_FUSED") ? atoi(getenv("RSBA_SYS_FUSED")) : 1;
  static const int bsm = getenv("RSBA_BACKSUB_MULTI") ? atoi(getenv("RSBA_BACKSUB_MULTI")) : 2;
  static const bool first_staged_env = !(getenv("RSBA_FIRST_STAGED") && atoi(getenv("RSBA_FIRST_STAGED")) == 0);
  const bool tiles_ok = s->nc <= RSBA_CHOL_MAXN ||
                        (s->tc_tiles > 0 && fuse_env != 0 && bsm >= 2 && s->tc_xs != nullptr && s->ev_tiles != nullptr &&
                         (!first || (first_staged_env && s->tiled.has_first_order)));
  const bool pipe = s->pipelined && s->opt.schur_impl != 0 && !keep_system_copy && tiles_ok;
  const bool pipe_tiles = pipe && s->nc > RSBA_CHOL_MAXN;
  // the persistent tiled factorisation and the chain back-substitution behind it (more than 64 cameras), on stream sx
  auto launch_tile_solve = [&](hipStream_t sx, const TileGate& gate) {
    const int n = s->nc;
    TileSysSource src;
    src.fused = 1; src.red = s->red; src.L = s->L; src.scale_c = s->scale_c; src.ip = ip; src.sym_full = s->opt.schur_impl != 0 ? 1 : 0;
    // (gated: a tile asks for as much LDS as a Schur workgroup — 81 KB instead of its 76 — so that the hole a retiring tile leaves is
    //  one a Schur workgroup fits into: LDS is allocated contiguously, and behind 76 KB tiles the elimination ran on the 187 slots
    //  the tiles had left it from the first block to the last, 710 us instead of 340)
    const size_t tile_lds = gate.ready != nullptr ? std::max(TileCholLdsDoubles() * sizeof(double), (size_t)RSBA_SCHUR_LDS_BYTES + 64) : TileCholLdsDoubles() * sizeof(double);
    T.Begin("k_chol_tiles_persistent", sx);
    k_chol_tiles_persistent<<<s->tc_tiles, 256, tile_lds, sx>>>(
        n, s->W, s->A, s->chol_ok, TileCholFlags{s->tc_flags, s->tc_flags + s->tc_np, s->tc_flags + (size_t)s->tc_np * (s->tc_nrt + 1), s->tc_nrt, s->tc_hand, s->tc_launches++ & 1, s->test_stall == 3 ? 1 : 0, s->tc_map, s->mc_trace},
        s->step_tag, s->res, src, gate);
    T.End(sx);
    const int nblk = s->tc_np, H = (nblk + 2) / 3;
    int* fl = s->tc_flags + (size_t)s->tc_np * (s->tc_nrt + 1);   // [error | xdone ...]
    T.Begin("k_backsub_chain", sx);
    k_backsub_chain<<<1 + H, 256, 0, sx>>>(
        C, s->red, s->L, s->A, s->tc_xs, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, s->chol_ok, s->cam_free,
        s->tc_ys, fl, s->tc_bs_launches++ & 1);
    T.End(sx);
  };
  // the factorisation + Schur kernel the previous step launched ahead (launch_ahead, below): they are this step's if they are what
  // it would launch — the state and radius of the device's decision are the host's — else they run out unused
  bool ahead_hit = false;
  if (s->ahead_inflight) {
    ahead_hit = pipe && !pipe_tiles && !first && !s->pipelined_mg && !s->pipe_serial && s->ahead_state == x && s->ahead_radius == radius &&
                s->ahead_tag == s->step_tag + 1;
    if (!ahead_hit) { const int rcd = DrainAhead(s); if (rcd != RSBA_OK) return rcd; }
    s->ahead_inflight = false;
  }
  ++s->step_tag;
  if (s->trace_ring) {
    if (s->trace_ring_first_tag == 0) s->trace_ring_first_tag = s->step_tag;
    s->trace = s->trace_base + 64 * (size_t)(s->step_tag % s->trace_ring);
  }
  // impl 0 accumulates into `red` with atomics; the tiled path writes every entry of it exactly once
  if (s->opt.schur_impl == 0) HIPCHK(hipMemsetAsync(s->red, 0, s->L.size() * sizeof(double), st));
  DebugSync(st, "memset red");

  if (s->opt.schur_impl == 0) {
    const bool stage = (size_t)C * (RSBA_ACC_PER_CAM + RSBA_CC_LDS) * sizeof(double) <= 96 * 1024;
    const size_t lds = (size_t)C * (RSBA_ACC_PER_CAM + (stage ? RSBA_CC_LDS : 0)) * sizeof(double);
    T.Begin("k_linearize_schur_ref", st);
    if (stage)
      k_linearize_schur_ref<true><<<s->grid_lin, 256, lds, st>>>(C, P, s->obs_u, s->obs_v, s->obs_cam, s->pt_ptr, s->camc[x], s->pts[x], s->scale_p,
                                                                 s->red, s->L, s->gmax, s->block_scal, ip);
    else
      k_linearize_schur_ref<false><<<s->grid_lin, 256, lds, st>>>(C, P, s->obs_u, s->obs_v, s->obs_cam, s->pt_ptr, s->camc[x], s->pts[x], s->scale_p,
                                                                  s->red, s->L, s->gmax, s->block_scal, ip);
    T.End(st);
    T.Begin("k_finish_linearize", st);
    k_finish_linearize<<<1, 256, 0, st>>>(s->grid_lin, s->block_scal, s->red, s->L, s->gmax);
    T.End(st);
  } else if (!pipe) {
    int rc = s->tiled.Launch(s, ip, T);
    if (rc != RSBA_OK) return rc;
  } else if (pipe_tiles) {
    // Pipelined above 64 cameras (round 4).  The persistent tiled factorisation goes out FIRST on the side stream — one resident
    // workgroup per 64 x 64 tile, 76 KB of LDS and one wavefront per SIMD each, so that a Schur workgroup (81 KB, one wavefront
    // per SIMD) fits beside every one of them — and every tile sleeps until the Schur kernel, launched behind it on the main
    // stream with the stage-ordered work list, has published the camera group(s) of its columns (TileGate).  A sleeping tile
    // issues nothing; a hit loop alone on its SIMDs runs at 98 % of what two of them reach together (tools/probes/hit_probe.hip),
    // so the elimination loses little, and the factorisation — a latency chain of ~18 us per tile column — ends a few tile
    // columns behind the last stage instead of starting there.  The chain back-substitution follows it on the side stream; the
    // point back-substitution waits for both behind a stream event.
    TiledSchur& ts = s->tiled;
    const int tag = s->step_tag;
    LaunchPointSide(ts, s, ip, T, st);
    if (s->trace) s->host_t[1] = std::chrono::steady_clock::now();
    RoctxRange rr_k4("K4 reduced camera system: tiled Cholesky + solve (launched ahead, gated on the Schur stages)");
    const bool first_staged = ip.first;   // (tiles_ok: the first-step work list exists)
    int* resident_word = (ip.first || s->pipe_check_resident) ? reinterpret_cast<int*>(s->res_host + RES_SIZE) : nullptr;
    s->pipe_check_resident = false;
    TileGate gate;
    gate.ready = ts.ready; gate.tag = (s->test_stall == 1 || (s->test_stall == 4 && s->pipe_stalls == 0)) ? tag + 1 : tag; gate.cols = 6 * RSBA_TG;
    gate.all_diag = first_staged ? ts.ready + RSBA_READY_ALLDIAG : nullptr;
    gate.started_cnt = ts.ready + RSBA_READY_STARTED; gate.started_host = resident_word; gate.started_need = s->tc_tiles;
    gate.waited = T.all_kernels() ? s->chol_waited : nullptr;
    if (s->pipe_serial) {
      // RSBA_PIPELINE=2: the same kernels one after the other (counter collection)
      ts.LaunchTiles(s, ip, T, st, tag, first_staged);
      HIPCHK(hipEventRecord(s->ev_serial[0], st));
      HIPCHK(hipStreamWaitEvent(s->sB, s->ev_serial[0], 0));
      gate.started_host = nullptr;
      launch_tile_solve(s->sB, gate);
    } else {
      // The tiles go out BEHIND the point side (a stream event, no host wait).  Launched beside it — as the 64-camera factorisation is —
      // the first step of every run but a solver's first deadlocked: with all 325 tiles resident and asleep, the point pass on the
      // main stream did not complete (and the Schur kernel queued behind it never started) until the tiles' waits ran out, 0.5 s
      // later; in a solver's very first step the side stream's first launch starts ~150 us late, i.e. behind the point pass anyway.
      // (Measured with device stamps: first Schur block 500 001 us after tile (0, 0)'s start; with the event: no stall in 3 x 5 runs.)
      HIPCHK(hipEventRecord(s->ev_tiles, st));
      HIPCHK(hipStreamWaitEvent(s->sB, s->ev_tiles, 0));
      launch_tile_solve(s->sB, gate);
      if (resident_word != nullptr) {
        volatile int* w = resident_word;
        const auto t_res = std::chrono::steady_clock::now();
        while (*w != gate.tag && std::chrono::steady_clock::now() - t_res < std::chrono::milliseconds(20)) __builtin_ia32_pause();
      }
      hp(1);
      ts.LaunchTiles(s, ip, T, st, tag, first_staged);
      hp(2);
    }
    HIPCHK(hipEventRecord(s->ev_tiles, s->sB));
    HIPCHK(hipStreamWaitEvent(st, s->ev_tiles, 0));
    rr_k4.End();
  } else if (ahead_hit) {
    // launched by the previous step (launch_ahead): damping kernel, factorisation (side stream) and Schur kernel are queued or running
    hp(1); hp(2);
  } else {
    // Pipelined.  Camera group g's columns of the reduced system are complete once the pair tiles (g, g' >= g) are
    // reduced, and the left-looking Cholesky needs nothing else for its panels 3g..3g+2.  The Cholesky kernel goes out
    // first (it takes the reserved CU while the chip is idle and sleeps on the ready flags), then the point pass, then
    // the pair kernel (stage by stage in block order) with the self tiles beside it.
    TiledSchur& ts = s->tiled;
    const int n = s->nc, tag = s->step_tag;
    const size_t lds_c = std::max((size_t)4 * 1024, CholeskyLdsDoubles(n)) * sizeof(double);
    // the point pass first: it is the head of the critical path; the Cholesky (which must be resident before the
    // Schur kernel fills the chip) goes out while it runs
    LaunchPointSide(ts, s, ip, T, st);
    if (s->trace) s->host_t[1] = std::chrono::steady_clock::now();
    const bool mg = s->pipelined_mg;
    // multi-GPU: the gates open on the flags the communication stream publishes after each stage's all-reduce, the
    // panels are read from the (all-reduced) row slab of their own group, and the waits may last as long as the slowest rank
    RoctxRange rr_k4("K4 reduced camera system: Cholesky + solve (launched ahead, gated on the Schur stages)");
    // First step of a run: the Schur kernel is not launched before every workgroup of the factorisation is resident (its
    // last workgroup to start writes the tag into a pinned word, StageGate).  The first launch on the side stream was
    // measured to start ~150 us behind the main stream's kernels; a factorisation that arrives when the chip is full does
    // not find a CU with enough LDS until the Schur kernel AND the back-substitution — whose workgroups hold their CUs
    // while they wait for the solve — have left: with few points the step timed out (and fell back to the sequential
    // schedule for good) in three of four solver lifetimes.  Later steps launch the factorisation ~5 us ahead on an idle
    // stream.
    const int gate_tag = (s->test_stall == 1 || (s->test_stall == 4 && s->pipe_stalls == 0)) ? tag + 1 : tag;
    // the first step of a run gated stage by stage too (diagonal-workgroup factorisation only): the Schur kernel runs every
    // self tile first and publishes ready[9] when all cameras' diag U are written (RSBA_FIRST_STAGED=0: wait for all stages)
    static const bool first_staged_on = !(getenv("RSBA_FIRST_STAGED") && atoi(getenv("RSBA_FIRST_STAGED")) == 0);
    const bool first_staged = ip.first && first_staged_on && ts.has_first_order && s->chol_wgs > 1 && !mg && s->chol_diag;
    const int* all_diag = first_staged ? ts.ready + RSBA_READY_ALLDIAG : nullptr;
    int* resident_word = (ip.first || s->pipe_check_resident) ? reinterpret_cast<int*>(s->res_host + RES_SIZE) : nullptr;
    s->pipe_check_resident = false;
    // RSBA_PIPELINE=2 — for counter collection (rocprofv3 --pmc serialises kernels, and a factorisation that waits inside the
    // kernel for a Schur kernel queued behind it would only run out its budget): the SAME kernels with the same work list,
    // stage flags and publication fences, launched one after the other — Schur kernel, then (stream events) the factorisation,
    // whose gates are all up by then, then the back-substitution.  Nothing overlaps; it is not a schedule anybody should time.
    const bool serial = s->pipe_serial && !mg;
    auto launch_factorisation = [&]() -> int {
      T.Begin("k_reduced_system_solve", s->sB);
      // multi-GPU: the gates open on the flags k_stage_unpack publishes behind each stage's all-reduce, and a wait may last as
      // long as the slowest rank
      const int* gate_ready = mg ? s->ready_global : ts.ready;
      const long long gate_budget = mg ? 10 * RSBA_STALL_TICKS : 0;
      if (s->chol_wgs > 1 && s->chol_diag) {
        const int wgs = s->chol_wgs + (s->border_cols > 0 ? 1 : 0);   // (the border's workgroup is the launch's last)
        const size_t lds_d = (s->border_cols > 0 ? std::max(DiagCholLdsDoubles(s->border_cols), BorderLdsDoubles(n)) : DiagCholLdsDoubles(n)) * sizeof(double);
        const StageGate sg{gate_ready, gate_tag, 6 * RSBA_TG, ts.ready + RSBA_READY_SOLVED, T.all_kernels() ? s->chol_waited : nullptr, s->trace, gate_budget, ts.ready + RSBA_READY_STARTED, resident_word, wgs, all_diag, mg ? 1 : 0};
        const DiagCholFlags df{s->mc_flags, s->mc_flags + 16, s->mc_flags + 32, s->mc_flags + 48, s->mc_dg, s->mc_dg + (size_t)(MultiCholPadded(s->nc) / RSBA_PB + 1) * 1024};
        if (mg) k_reduced_system_solve_diag<true><<<s->chol_wgs, 512, DiagCholLdsDoubles(n) * sizeof(double), s->sB>>>(
            C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, ip, s->chol_ok, sg, df, tag, s->mc_trace);
        else k_reduced_system_solve_diag<false><<<wgs, 512, lds_d, s->sB>>>(
            C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, ip, s->chol_ok, sg, df, tag, s->mc_trace, AheadSel{}, s->border_cols);
      }
#ifdef RSBA_EXPERIMENTAL
      else if (s->chol_wgs > 1 && !mg)   // (the round-robin kernel has no transposed source: multi-GPU, it is the one-workgroup kernel)
        k_reduced_system_solve_multi<<<s->chol_wgs, 512, MultiCholLdsDoubles(n) * sizeof(double), s->sB>>>(
            C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, ip, s->chol_ok,
            StageGate{gate_ready, gate_tag, 6 * RSBA_TG, ts.ready + RSBA_READY_SOLVED, T.all_kernels() ? s->chol_waited : nullptr, s->trace, gate_budget, ts.ready + RSBA_READY_STARTED, resident_word, s->chol_wgs},
            MultiCholFlags{s->mc_flags, s->mc_flags + 16, s->mc_flags + 32, s->mc_flags + 48}, tag, s->mc_trace);
#endif
      else
      k_reduced_system_solve<<<1, 512, lds_c, s->sB>>>(C, s->red, s->L, s->A, nullptr, nullptr, s->scale_c, s->cam[x], s->cam[c], s->intr,
                                                       s->camc[c], s->dcam, s->gmax, s->res, ip, mg ? 2 : 1,
                                                       s->chol_ok, StageGate{gate_ready, gate_tag, 6 * RSBA_TG, ts.ready + RSBA_READY_SOLVED,
                                                                             T.all_kernels() ? s->chol_waited : nullptr, s->trace, gate_budget,
                                                                             ts.ready + RSBA_READY_STARTED, resident_word, 1});
      T.End(s->sB);
      rr_k4.End();
      if (resident_word != nullptr) {
        volatile int* w = resident_word;
        const auto t_res = std::chrono::steady_clock::now();
        while (*w != gate_tag && std::chrono::steady_clock::now() - t_res < std::chrono::milliseconds(20)) __builtin_ia32_pause();
      }
      return RSBA_OK;
    };
    if (serial) {
      ts.LaunchTiles(s, ip, T, st, tag, first_staged);
      HIPCHK(hipEventRecord(s->ev_serial[0], st));
      HIPCHK(hipStreamWaitEvent(s->sB, s->ev_serial[0], 0));
      { const int rcf = launch_factorisation(); if (rcf != RSBA_OK) return rcf; }
      HIPCHK(hipEventRecord(s->ev_serial[1], s->sB));
      HIPCHK(hipStreamWaitEvent(st, s->ev_serial[1], 0));
      hp(1); hp(2);
    } else {
    { const int rcf = launch_factorisation(); if (rcf != RSBA_OK) return rcf; }
    hp(1);   // point side + factorisation launched
    ts.LaunchTiles(s, ip, T, st, tag, first_staged);
    hp(2);   // Schur kernel launched
    }
    if (mg) {
      // communication stream: stage by stage, as the Schur kernel publishes them locally — the row slab of S of the
      // stage's camera group and the group's ranges of g_c, rhs correction and diag U, as one grouped collective; the
      // scalars and max |g_p| ride with the last stage
      const RedLayout& L = s->L;
      for (int g = 0; g < ts.nstages; ++g) {
        const int r0 = 6 * RSBA_TG * g, r1 = std::min(6 * RSBA_TG * (g + 1), n), rows = r1 - r0;
        k_wait_stage<<<1, 64, 0, s->sR>>>(ts.ready + 1 + g, tag, ts.error_flag ? ts.error_flag : ts.tree_error);
        COMMCHK(s->comm->GroupStart());
        COMMCHK(s->comm->SumDoubles(s->red + L.S() + (size_t)r0 * n, (size_t)rows * n, s->sR));
        COMMCHK(s->comm->SumDoubles(s->red + L.gc() + r0, rows, s->sR));
        COMMCHK(s->comm->SumDoubles(s->red + L.corr() + r0, rows, s->sR));
        COMMCHK(s->comm->SumDoubles(s->red + L.diagU() + r0, rows, s->sR));
        if (g == ts.nstages - 1) {
          COMMCHK(s->comm->SumDoubles(s->red + L.scal(), 8, s->sR));
          COMMCHK(s->comm->MaxDoubles(s->gmax, 1, s->sR));
        }
        COMMCHK(s->comm->GroupEnd());
        k_set_flag<<<1, 64, 0, s->sR>>>(s->ready_global + 1 + g, tag);
      }
    }
  }
  hp(3);   // the stages' collectives issued
  HIPCHK(hipGetLastError());
  DebugSync(st, "linearize+schur");

  if (s->comm && !(pipe && s->pipelined_mg)) {
    RoctxRange rr_k7("K7 RCCL all-reduce of the reduced system");
    // one group: the sum of the packed reduced system and the max of the point-gradient bound go out as one launch
    if (s->red_tri != nullptr) {
      const int n = s->nc;
      const int grid_tri = std::max(1, std::min(2 * DeviceCUs(), (int)((TriSize(n) + 255) / 256)));
      k_pack_lower<<<grid_tri, 256, 0, st>>>(n, s->red, s->L, s->red_tri, s->opt.schur_impl != 0 ? 1 : 0);
      COMMCHK(s->comm->GroupStart());
      COMMCHK(s->comm->SumDoubles(s->red_tri, TriSize(n), st));
      COMMCHK(s->comm->MaxDoubles(s->gmax, 1, st));
      COMMCHK(s->comm->GroupEnd());
      k_unpack_lower<<<grid_tri, 256, 0, st>>>(n, s->red_tri, s->red, s->L);
    } else {
      COMMCHK(s->comm->GroupStart());
      COMMCHK(s->comm->SumDoubles(s->red, s->L.size(), st));
      COMMCHK(s->comm->MaxDoubles(s->gmax, 1, st));
      COMMCHK(s->comm->GroupEnd());
    }
  }

  RoctxRange rr_k4s(pipe ? "K4 (already launched)" : "K4 reduced camera system: Cholesky + solve");
  if (pipe) {
    // issued at the top of the step (see below): nothing left to launch here
  } else if (s->nc <= RSBA_CHOL_MAXN && !(s->tiles_small && s->tc_tiles > 0 && !keep_system_copy)) {
    const size_t lds_c = std::max((size_t)4 * 1024, CholeskyLdsDoubles(s->nc)) * sizeof(double);
    T.Begin("k_reduced_system_solve", st);
    if (s->chol_wgs > 1 && !keep_system_copy && s->chol_diag)
      k_reduced_system_solve_diag<false><<<s->chol_wgs + (s->border_cols > 0 ? 1 : 0), 512,
                                           (s->border_cols > 0 ? std::max(DiagCholLdsDoubles(s->border_cols), BorderLdsDoubles(s->nc)) : DiagCholLdsDoubles(s->nc)) * sizeof(double), st>>>(
          C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, ip, s->chol_ok,
          StageGate{nullptr, 0, 0, nullptr, nullptr, nullptr, 0}, DiagCholFlags{s->mc_flags, s->mc_flags + 16, s->mc_flags + 32, s->mc_flags + 48, s->mc_dg, s->mc_dg + (size_t)(MultiCholPadded(s->nc) / RSBA_PB + 1) * 1024},
          s->step_tag, s->mc_trace, AheadSel{}, s->border_cols);
#ifdef RSBA_EXPERIMENTAL
    else if (s->chol_wgs > 1 && !keep_system_copy)
      k_reduced_system_solve_multi<<<s->chol_wgs, 512, MultiCholLdsDoubles(s->nc) * sizeof(double), st>>>(
          C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, ip, s->chol_ok,
          StageGate{nullptr, 0, 0, nullptr, nullptr, nullptr, 0}, MultiCholFlags{s->mc_flags, s->mc_flags + 16, s->mc_flags + 32, s->mc_flags + 48},
          s->step_tag, s->mc_trace);
#endif
    else
    k_reduced_system_solve<<<1, 512, lds_c, st>>>(C, s->red, s->L, s->A, keep_system_copy ? s->S_copy : nullptr,
                                                  keep_system_copy ? s->rhs_copy : nullptr, s->scale_c, s->cam[x], s->cam[c], s->intr,
                                                  s->camc[c], s->dcam, s->gmax, s->res, ip, s->opt.schur_impl != 0 ? 1 : 0,
                                                  s->chol_ok, StageGate{nullptr, 0, 0, nullptr, nullptr, nullptr});
    T.End(st);
  } else {
    // more than 64 cameras: right-looking factorisation over the whole chip, one launch per 32-wide panel
    const int n = s->nc;
    // the resident tiles build their entries of the system themselves (TileSysSource); k_sys_build only where its output is
    // wanted for itself (the copies of the system a caller asked for) or the multi-launch factorisation reads it
    const bool fused = s->tc_tiles > 0 && !keep_system_copy && fuse_env != 0;
    if (!fused) {
      T.Begin("k_sys_build", st);
      k_sys_build<<<n + 1, 256, 0, st>>>(s->red, s->L, s->W, keep_system_copy ? s->S_copy : nullptr, keep_system_copy ? s->rhs_copy : nullptr,
                                         s->scale_c, ip, s->opt.schur_impl != 0 ? 1 : 0, s->chol_ok);
      T.End(st);
    }
    if (s->tc_tiles > 0) {
      TileSysSource src;
      if (fused) { src.fused = 1; src.red = s->red; src.L = s->L; src.scale_c = s->scale_c; src.ip = ip; src.sym_full = s->opt.schur_impl != 0 ? 1 : 0; }
      T.Begin("k_chol_tiles_persistent", st);
      k_chol_tiles_persistent<<<s->tc_tiles, 256, TileCholLdsDoubles() * sizeof(double), st>>>(
          n, s->W, s->A, s->chol_ok, TileCholFlags{s->tc_flags, s->tc_flags + s->tc_np, s->tc_flags + (size_t)s->tc_np * (s->tc_nrt + 1), s->tc_nrt, s->tc_hand, s->tc_launches++ & 1, s->test_stall == 3 ? 1 : 0, s->tc_map, s->mc_trace},
          s->step_tag, s->res, src, TileGate{});
      T.End(st);
    } else {
    const size_t lds_s = CholStepLdsDoubles() * sizeof(double);
    T.Begin("k_chol_step(all panels)", st);
    for (int kb = 0; kb < n; kb += RSBA_PB) {
      const int r0 = kb + std::min(RSBA_PB, n - kb);
      const int nrt = (n + 1 - r0 + RSBA_CT - 1) / RSBA_CT;
      k_chol_step<<<nrt * (nrt + 1) / 2, 256, lds_s, st>>>(n, kb, s->W, s->A, s->chol_ok);
    }
    T.End(st);
    }
    // RSBA_BACKSUB_MULTI: 2 (default) the chain in one workgroup with helpers for the far strips (k_backsub_chain), 1 the chain
    // passed from owner to owner (k_backsub_multi, round 2), 0 one workgroup for everything (k_chol_finish)
    if (s->tc_tiles > 0 && bsm >= 2 && s->tc_xs != nullptr) {
      const int nblk = s->tc_np, H = (nblk + 2) / 3;
      int* fl = s->tc_flags + (size_t)s->tc_np * (s->tc_nrt + 1);   // [error | xdone ...]
      T.Begin("k_backsub_chain", st);
      k_backsub_chain<<<1 + H, 256, 0, st>>>(
          C, s->red, s->L, s->A, s->tc_xs, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, s->chol_ok, s->cam_free,
          s->tc_ys, fl, s->tc_bs_launches++ & 1);
      T.End(st);
    } else if (s->tc_tiles > 0 && bsm >= 1) {
      // block back-substitution on several workgroups (three 32-column blocks each); x goes to row n of W (free by now)
      const int nblk = s->tc_np, G = (nblk + RSBA_BSM_BPG - 1) / RSBA_BSM_BPG;
      int* fl = s->tc_flags + (size_t)s->tc_np * (s->tc_nrt + 1);   // [error | xdone ...]
      T.Begin("k_backsub_multi", st);
      k_backsub_multi<<<G, 256, 0, st>>>(C, s->red, s->L, s->A, s->W + (size_t)n * n, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam,
                                         s->gmax, s->res, s->chol_ok, s->cam_free, fl + 1, fl, s->step_tag);
      T.End(st);
    } else {
    const size_t lds_f = std::max((size_t)4 * 1024, (size_t)((n + 63) & ~63) + 3 * RSBA_PB * RSBA_PLD + 64) * sizeof(double);
    T.Begin("k_chol_finish", st);
    k_chol_finish<<<1, 1024, lds_f, st>>>(C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res,
                                          s->chol_ok, s->cam_free);
    T.End(st);
    }
  }
  rr_k4s.End();
  RoctxRange rr_k5("K5 back-substitution + candidate (K1 at the candidate, fused re-linearisation)");
  DebugSync(st, "k_reduced_system_solve");
  if (s->comm && s->inject_stall_step != 0 && s->step_tag == s->inject_stall_step) {
    // (test hook: what a factorisation whose in-kernel wait ran out of its budget leaves in the result block)
    k_set_double<<<1, 1, 0, st>>>(s->res + RES_STALL, 1.0);
    s->inject_stall_step = 0;
  }
  T.Begin("k_backsub_candidate", st);
  // single GPU: the kernel's last workgroup finishes the step (sums, result block, post to the host); with a communicator the
  // projective kernel's last workgroup leaves the rank's sums for the all-reduce (FusedLin::sums_only), the other kernels their
  // per-block sums for k_finish_candidate
  // (RSBA_BACKSUB_PROJ=0: the form that reads the camera constants themselves, for comparison)
  static const bool proj_form = !(getenv("RSBA_BACKSUB_PROJ") && atoi(getenv("RSBA_BACKSUB_PROJ")) == 0);
  const bool use_proj = s->opt.schur_impl != 0 && s->fused_lin && proj_form && C <= 256;
  const bool comm_tail = s->comm != nullptr;
  int* fin_cnt = (comm_tail && !use_proj) ? nullptr : s->chol_ok + 1;
  LmNext lm_publish;   // with a communicator: the decision is k_publish_result's, behind the all-reduce
  int grid_bs = s->grid_pts;   // workgroups of the back-substitution = blocks of partial sums the finisher adds
  {
    const size_t lds_b = (size_t)C * (2 * RSBA_CC_LDS + 6) * sizeof(double);
    const bool fused = s->opt.schur_impl != 0 && s->fused_lin;
    const FusedLin fl0 = fused ? FusedLin{s->tiled.lin2[x], s->tiled.lin2[c], s->tiled.cm_pos, s->tiled.sq_cm2[c], s->trace} : FusedLin{nullptr, nullptr, nullptr, nullptr, s->trace};
    FusedLin fl0b = fl0; fl0b.bs_wg = s->bs_wg;
    const FusedLin& fl = fl0b;
    const int* solve_done = pipe && !pipe_tiles ? s->tiled.ready + RSBA_READY_SOLVED : nullptr;   // (tile pipeline: a stream event orders the back-substitution behind the solve)
    const int solve_tag = (s->test_stall && (s->test_stall != 4 || s->pipe_stalls == 0)) ? s->step_tag + s->test_stall * s->test_stall : s->step_tag;
    long long* waited = pipe && T.all_kernels() ? s->chol_waited + 1 : nullptr;
#define RSBA_BACKSUB_ARGS C, P, s->sliced(), s->camc[x], s->camc[c], s->dcam, s->pts[x], s->pts[c], s->scale_p, s->block_part, ip, fin_cnt, s->small_red, \
                          s->res, s->res_host, s->res_seq + 1.0, solve_done, solve_tag, waited, s->chol_ok + 2, fl
    if (use_proj) {
      // slices of 64 points dealt to two workgroups per CU (k_backsub_candidate_proj); ten slots of a lane's observation
      // records in registers, ten (<= 64 cameras) or six (<= 128) more in LDS, both tables + records <= 72 KB per workgroup;
      // up to 256 cameras one workgroup per CU (tables 80 KB + ten LDS slots)
      // (16 fewer than the chip holds: the factorisation's workgroups are resident beside this kernel, and a workgroup that
      //  has to wait for a slot starts behind the solve, without its records.  Measured again in round 5 with the border's seven
      //  workgroups: 2 x CUs - 0 / 8 / 14 / 16 / 32: 0.3506 - 0.3516 / 0.3470 - 0.3482 / 0.3491 / 0.3486 - 0.3493 / 0.3465 - 0.3482 ms per
      //  step — nothing beyond the run-to-run spread but the full grid, which is slower)
      // (more than 128 cameras: the tables alone are 80 KB, one workgroup per CU)
      grid_bs = std::max(1, std::min(C <= 128 ? 2 * DeviceCUs() - 16 : DeviceCUs() - 8, (P + 63) / 64));   // (round 6, four factorisation workgroups: 0 / 8 / 16 / 32 fewer no different)
      FusedLin fl = fl0b;
      // single GPU: the workgroup that completes the result block takes the step's decision as well, and the damping
      // kernel of the NEXT step is queued right here, behind this kernel, on that decision (LmNext; RSBA_DECIDED_DAMP=0:
      // the host launches it once it has decided itself)
      static const bool decided = !(getenv("RSBA_DECIDED_DAMP") && atoi(getenv("RSBA_DECIDED_DAMP")) == 0);
      s->dec_step = decided && fin_cnt != nullptr;
      if (s->dec_step) {
        LmNext lm;
        lm.dec = s->dec; lm.radius = ip.radius; lm.decrease_factor = s->lm_decrease_factor;
        lm.min_relative_decrease = s->opt.min_relative_decrease; lm.max_radius = s->opt.max_trust_region_radius;
        if (comm_tail) lm_publish = lm; else fl.lm = lm;
      }
      fl.sums_only = comm_tail ? 1 : 0;
      // (more than 48 KB of dynamic LDS has to be asked for, once per kernel)
      // <cameras, slots in registers, slots in LDS, loss>: the robust instances keep fewer records in registers — the loss'
      // square roots and the store of sqrt(rho') need the registers, and a spilled value is a trip to memory per slot
#define RSBA_BS_INSTANCES(X) X(64, 10, 10, false) X(64, RSBA_BS_REG_LOSS, 11, true) X(128, 10, 6, false) X(128, RSBA_BS_REG_LOSS, 7, true) \
                             X(256, 10, 10, false) X(256, 10, 10, true)
      static const bool lds_attr_set = []() {
        bool ok = true;
#define RSBA_X(cp, rg, ld, ls) ok = hipFuncSetAttribute((const void*)k_backsub_candidate_proj<cp, rg, ld, ls>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BacksubProjLds<cp, ld>::kBytes) == hipSuccess && ok;
        RSBA_BS_INSTANCES(RSBA_X)
#undef RSBA_X
        return ok;
      }();
      if (!lds_attr_set) return RSBA_ERR_HIP;
      const bool loss = ip.huber_delta != 0.0;
      const int cp = C <= 64 ? 64 : (C <= 128 ? 128 : 256);
#define RSBA_X(cpad, rg, ld, ls) if (cp == cpad && loss == ls) k_backsub_candidate_proj<cpad, rg, ld, ls><<<grid_bs, 256, BacksubProjLds<cpad, ld>::kBytes, st>>>(RSBA_BACKSUB_ARGS);
      RSBA_BS_INSTANCES(RSBA_X)
#undef RSBA_X
#undef RSBA_BS_INSTANCES
    } else if (lds_b <= 60 * 1024) {
      if (fused) k_backsub_candidate<true, true><<<s->grid_pts, 256, lds_b, st>>>(RSBA_BACKSUB_ARGS);
      else k_backsub_candidate<true, false><<<s->grid_pts, 256, lds_b, st>>>(RSBA_BACKSUB_ARGS);
    } else {
      if (fused) k_backsub_candidate<false, true><<<s->grid_pts, 256, 0, st>>>(RSBA_BACKSUB_ARGS);
      else k_backsub_candidate<false, false><<<s->grid_pts, 256, 0, st>>>(RSBA_BACKSUB_ARGS);
    }
#undef RSBA_BACKSUB_ARGS
    // a completed step leaves lin2[x] (point pass or an earlier candidate) and lin2[c] (this candidate) in place
    if (fused) s->tiled.lin_valid = true;
  }
  T.End(st);
  // constant point blocks: the candidate's linearisation records the back-substitution has just written say "infinitely stiff, no
  // gradient" again before anybody reads them (k_fix_const_lin, ba_point_kernels.hpp); the damping kernels know the blocks themselves
  if (s->pt_const != nullptr && s->tiled.lin2[c] != nullptr)
    k_fix_const_lin<<<std::max(1, std::min((P + 255) / 256, 1024)), 256, 0, st>>>(P, s->pt_const, s->tiled.lin2[c], RSBA_LIN_STRIDE);
  auto queue_damping = [&]() {
    // the damping kernel of the NEXT step, on the device's decision (LmNext)
    TiledSchur& ts = s->tiled;
    IterParams ipn = ip; ipn.first = 0;
    T.Begin("k_point_damp", st);
    k_point_damp<<<ts.grid_pp, 256, 0, st>>>(P, s->pts[x], s->scale_p, ts.lin2[x], ts.ptdata, ts.block_scal, C, s->camc[x], ts.small_flag, ipn, s->trace,
                                             s->dec, s->pts[c], ts.lin2[c], s->camc[c]);
    T.End(st);
    ts.pt_valid = false;   // (until the decision is in: below)
  };
  // Launch-ahead.  Between the result of one step and the first Schur workgroup of the next lay the host: the result's trip over
  // PCIe, the decision, two launches — 23 us of a 390 us step at 64 cameras, the chip idle but for the damping kernel.  The device
  // takes the decision itself (DecideStep, the host's arithmetic), so the NEXT step's factorisation (side stream: resident and asleep
  // until the decision is in, AheadSel) and Schur kernel (main stream, behind the damping kernel; state by SchurArgs::dec) are queued
  // right here, before this step's result exists.  The host takes the same decision from the same numbers and, in the next
  // PointsStep, finds its launches done (ahead_hit) — or stops (a tolerance, the time limit) and lets them run out (DrainAhead); the
  // step that reaches the iteration limit launches nothing ahead (MinimizeLoop: ahead_ok).
  // OPT-IN (RSBA_LAUNCH_AHEAD=1).  Measured at 64 cameras (round 4, device stamps, 99 steps): result posted -> first Schur workgroup
  // 13.1 us without, 12.1 us with — the host's share was already hidden behind the damping kernel (7.3 us) it queues on the device's
  // decision, what is left is that kernel and the dispatch behind it; 0.3885 vs 0.3890 ms per step, inside the run-to-run spread.
  // Round 5, with the border factorisation: 14.2 -> 13.2 us, 0.3523 - 0.3531 against 0.3516 - 0.3529 ms per step.
  // Against that microsecond per step stands a whole unused elimination (0.26 ms) at the end of every run a tolerance ends.
#ifdef RSBA_EXPERIMENTAL
  static const bool ahead_env = getenv("RSBA_LAUNCH_AHEAD") && atoi(getenv("RSBA_LAUNCH_AHEAD")) != 0;
#else
  static const bool ahead_env = false;   // (-DRSBA_EXPERIMENTAL builds only: worth 1 us a step, costs an unused elimination at the end of a run)
#endif
  auto launch_ahead = [&]() {
    TiledSchur& ts = s->tiled;
    const int n = s->nc, atag = s->step_tag + 1;
    IterParams ipn = ip; ipn.first = 0;
    long long* tr = s->trace_ring ? s->trace_base + 64 * (size_t)(atag % s->trace_ring) : nullptr;
    const StageGate sg{ts.ready, atag, 6 * RSBA_TG, ts.ready + RSBA_READY_SOLVED, nullptr, tr, 0, ts.ready + RSBA_READY_STARTED, nullptr, s->chol_wgs + (s->border_cols > 0 ? 1 : 0), nullptr, 0};
    const DiagCholFlags df{s->mc_flags, s->mc_flags + 16, s->mc_flags + 32, s->mc_flags + 48, s->mc_dg, s->mc_dg + (size_t)(MultiCholPadded(s->nc) / RSBA_PB + 1) * 1024};
    AheadSel ah; ah.dec = s->dec; ah.seq = s->res_seq + 1.0; ah.camc_x = s->camc[x]; ah.cam_backup = s->cam_backup; ah.camc_backup = s->cam_backup + 6 * (size_t)C;
    T.Begin("k_reduced_system_solve", s->sB);
    const int wgs = s->chol_wgs + (s->border_cols > 0 ? 1 : 0);   // (with the border's workgroup: ba_cholesky_border.hpp)
    const size_t lds_d = (s->border_cols > 0 ? std::max(DiagCholLdsDoubles(s->border_cols), BorderLdsDoubles(n)) : DiagCholLdsDoubles(n)) * sizeof(double);
    k_reduced_system_solve_diag<false><<<wgs, 512, lds_d, s->sB>>>(
        C, s->red, s->L, s->A, s->scale_c, s->cam[x], s->cam[c], s->intr, s->camc[c], s->dcam, s->gmax, s->res, ipn, s->chol_ok, sg, df, atag, s->mc_trace, ah, s->border_cols);
    T.End(s->sB);
    ts.LaunchTiles(s, ipn, T, st, atag, false, true, tr);
    s->ahead_inflight = true; s->ahead_tag = atag; s->ahead_x = x;
  };
  if (s->dec_step && !comm_tail) {
    queue_damping();
    if (ahead_env && s->ahead_ok && pipe && !pipe_tiles && !s->pipelined_mg && !s->pipe_serial && s->chol_wgs > 1 && s->chol_diag && !keep_system_copy &&
        !T.all_kernels() && !(s->trace && !s->trace_ring) && !s->wg_trace && !s->test_stall)
      launch_ahead();
  }
  rr_k5.End();
  DebugSync(st, "k_backsub_candidate");
  if (comm_tail) {
    const bool mg = pipe && s->pipelined_mg;
    if (!use_proj) {
      T.Begin("k_finish_candidate", st);
      // res_stall: with a communicator the stall flag of the factorisation (multi-workgroup / persistent tiles: in-kernel
      // waits) and of the pipeline rides in small_red[5] and is SUMMED over the ranks in every schedule, so that all ranks
      // take the same fallback below and keep issuing the same collectives
      k_finish_candidate<<<1, 256, 0, st>>>(grid_bs, s->block_part, s->small_red, nullptr, nullptr, 0.0, s->res, s->trace, mg ? s->chol_ok + 2 : nullptr);
      T.End(st);
    }
    // the candidate's sums and the stall flags — on the main stream, right behind the kernel that formed them — then the
    // result block, the decision and the post to the host
    if (s->share_clock) k_set_double<<<1, 1, 0, st>>>(s->small_red + 6, s->time_up);   // (rank 0's "out of time" word, MinimizeLoop)
    COMMCHK(s->comm->SumDoubles(s->small_red, 8, st));
    k_publish_result<<<1, 64, 0, st>>>(s->small_red, s->res, s->res_host, s->res_seq + 1.0, 1, s->trace, lm_publish);
    if (s->dec_step) queue_damping();
  }
  HIPCHK(hipGetLastError());
  hp(4);   // everything launched
  if (s->trace) s->host_t[2] = std::chrono::steady_clock::now();
  RoctxRange rr_k6("K6 LM bookkeeping: wait for the step's result block");
  { const int rcw = WaitResult(s, st); if (rcw != RSBA_OK) return rcw; }
  hp(5);   // result in
  if (s->hostprof) s->hp_result = s->hp_t;
  if (s->trace && !s->trace_ring) {
    const auto now = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "rsba[host] since previous result %.1f us | enter -> point pass launched %.1f -> all launched %.1f -> result %.1f\n",
            us(s->host_t[3], s->host_t[0]), us(s->host_t[0], s->host_t[1]), us(s->host_t[1], s->host_t[2]), us(s->host_t[2], now));
    s->host_t[3] = now;
  }
  if (s->dec_step) {
    // the damping kernel queued behind the step has damped (or is damping) the state and radius of the device's decision
    TiledSchur& ts = s->tiled;
    const double* r = s->res_host;
    ts.pt_valid = r[RES_DEC_GO] != 0.0; ts.pt_state = r[RES_DEC_ACCEPT] != 0.0 ? c : x; ts.pt_radius = r[RES_DEC_RADIUS]; ts.scal_blocks = ts.grid_pp;
    s->ahead_state = ts.pt_state; s->ahead_radius = ts.pt_radius;
    s->dec_step = false;
  }
  if (pipe && T.all_kernels()) {
    // the Cholesky's event span includes the time it slept on the ready flags: record that separately
    long long w[2] = {0, 0};
    HIPCHK(hipMemcpy(w, s->chol_waited, sizeof(w), hipMemcpyDeviceToHost));
    T.Add(pipe_tiles ? "k_chol_tiles_persistent:waiting" : "k_reduced_system_solve:waiting", (w[0] - s->chol_waited_seen) * 1e-5);
    T.Add("k_backsub_candidate:waiting", (w[1] - s->backsub_waited_seen) * 1e-5);
    s->chol_waited_seen = w[0]; s->backsub_waited_seen = w[1];
  }
  if (s->trace && !s->trace_ring && s->opt.schur_impl != 0) {
    long long h[64];
    HIPCHK(hipMemcpy(h, s->trace, sizeof(h), hipMemcpyDeviceToHost));
    const long long t0 = h[24];
    fprintf(stderr, "rsba[trace] us since first tile block: chol start %.1f ready0 seen %.1f | gates (wait..pass)", (h[0] - t0) * 0.01, (h[1] - t0) * 0.01);
    for (int g = 0; g < std::min(s->tiled.nstages, 4); ++g) fprintf(stderr, " %.1f..%.1f", (h[2 + 2 * g] - t0) * 0.01, (h[3 + 2 * g] - t0) * 0.01);
    fprintf(stderr, " factored %.1f back-substituted %.1f end %.1f | published: self %.1f stages", (h[13] - t0) * 0.01, (h[14] - t0) * 0.01, (h[15] - t0) * 0.01, (h[16] - t0) * 0.01);
    for (int g = 0; g < s->tiled.nstages; ++g) fprintf(stderr, " %.1f", (h[17 + g] - t0) * 0.01);
    if (s->comm) fprintf(stderr, " | after the solve: candidate sums start +%.1f, publish +%.1f", (h[26] - h[15]) * 0.01, (h[27] - h[15]) * 0.01);
    // the tail of the step: back-substitution past the solve's flag, result posted; and this step's head: the previous result
    // posted -> damping kernel's first workgroup -> its last -> first Schur block
    fprintf(stderr, " | backsub past the flag %.1f posted %.1f | head: previous post -> damp start %.1f -> damp end %.1f -> first tile block %.1f",
            (h[28] - t0) * 0.01, (h[29] - t0) * 0.01, (h[30] - s->trace_prev_post) * 0.01, (h[31] - h[30]) * 0.01, (t0 - h[31]) * 0.01);
    fprintf(stderr, " | backsub wg 0 after the flag: staged +%.1f pass at x +%.1f pass at the candidate +%.1f block sums +%.1f; last workgroup arrives +%.1f",
            (h[32] - h[28]) * 0.01, (h[33] - h[28]) * 0.01, (h[34] - h[28]) * 0.01, (h[35] - h[28]) * 0.01, (h[36] - h[28]) * 0.01);
    s->trace_prev_post = h[29];
    fprintf(stderr, "\n");
  }
  if (s->bs_wg && s->step_tag == 8) {
    std::vector<long long> w(4 * 1024);
    HIPCHK(hipMemcpy(w.data(), s->bs_wg, w.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long t0 = 0;
    for (int b = 0; b < grid_bs && b < 1024; ++b) if (w[4 * b] != 0 && (t0 == 0 || w[4 * b] < t0)) t0 = w[4 * b];
    if (FILE* f = fopen(getenv("RSBA_TRACE_FILE") ? getenv("RSBA_TRACE_FILE") : "bswg.txt", "w")) {
      for (int b = 0; b < grid_bs && b < 1024; ++b)
        fprintf(f, "%d %.2f %.2f %.2f %.2f\n", b, (w[4 * b] - t0) * 0.01, (w[4 * b + 1] - t0) * 0.01, (w[4 * b + 2] - t0) * 0.01, (w[4 * b + 3] - t0) * 0.01);
      fclose(f);
    }
  }
  if (s->wg_trace && s->step_tag == 5) {
    // RSBA_TRACE=2: one step's block timeline of the Schur kernel — list position, segment, tile, type, stage, words, start,
    // sums stored, end (us after the first block to start); tools/schur_timeline_summary.py
    const int nb = s->tiled.nblocks;
    std::vector<long long> w((size_t)3 * nb);
    std::vector<SchurSeg> hs(nb); std::vector<int> bs(nb);
    HIPCHK(hipMemcpy(w.data(), s->wg_trace, w.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hs.data(), s->tiled.segs, nb * sizeof(SchurSeg), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(bs.data(), s->tiled.block_seg, nb * sizeof(int), hipMemcpyDeviceToHost));
    long long t0 = w[0];
    for (int b = 0; b < nb; ++b) t0 = std::min(t0, w[3 * b]);
    if (FILE* f = fopen(getenv("RSBA_TRACE_FILE") ? getenv("RSBA_TRACE_FILE") : "wgtrace.txt", "w")) {
      for (int b = 0; b < nb; ++b) {
        const SchurSeg& e = hs[bs[b]];
        fprintf(f, "%d %d %d %d %d %d %.2f %.2f %.2f\n", b, bs[b], e.tile, e.self, e.stage, e.word_end - e.word_begin, (w[3 * b] - t0) * 0.01,
                (w[3 * b + 2] - t0) * 0.01, (w[3 * b + 1] - t0) * 0.01);
      }
      fclose(f);
    }
  }
  if (!pipe && s->tc_tiles > 0 && s->res_host[RES_STALL] != 0.0) {
    fprintf(stderr, "rsba: persistent tiled Cholesky stalled; using the multi-launch factorisation\n");
    ++s->other_stalls;
    { const int rcs = SyncSolver(s); if (rcs != RSBA_OK) return rcs; }
    HIPCHK(hipMemset(s->res, 0, RES_SIZE * sizeof(double)));
    s->tc_tiles = 0;
    return PointsStep(s, radius, first, keep_system_copy);
  }
  if (!pipe && s->chol_wgs > 1 && (s->res_host[RES_STALL] != 0.0 || (s->test_stall == 4 && s->pipe_stalls > 0 && !s->test_seq_stall_fired))) {
    // the workgroups of the factorisation did not run side by side (cannot happen on an idle stream): one workgroup then
    fprintf(stderr, "rsba: multi-workgroup Cholesky stalled; using one workgroup\n");
    s->test_seq_stall_fired = true;
    ++s->other_stalls;
    { const int rcs = SyncSolver(s); if (rcs != RSBA_OK) return rcs; }
    HIPCHK(hipMemset(s->mc_flags, 0, 64 * sizeof(int)));
    HIPCHK(hipMemset(s->res, 0, RES_SIZE * sizeof(double)));
    s->chol_wgs = 1;
    if (s->border_cols > 0) {
      // The Schur work list was built for the border (2 Bg + 1 stages, permuted: stage g holds the leading tiles only, tile (g, Bg) is
      // stage Bg + g, the border's own tile stage 2 Bg).  The one-workgroup factorisation's gates read ready[1 + g] as "all columns of
      // camera group g": pipelined again it would factor panels whose border rows are still being accumulated, silently.  No border,
      // and no pipelined schedule for this solver any more (a kernel boundary needs no stages).
      s->border_cols = 0;
      s->pipeline_off = true;
      s->pipelined = false; s->pipelined_mg = false;
    }
    s->tiled.pt_valid = false;
    return PointsStep(s, radius, first, keep_system_copy);
  }
  if (pipe && (s->res_host[RES_STALL] != 0.0 || s->res_host[RES_WAIT_TIMEOUT] != 0.0)) {
    // the waiting Cholesky never saw its columns (its producers were not running beside it): nothing of x has been
    // touched, so repeat the step with the plain schedule and stay there
    fprintf(stderr, "rsba: pipelined solve stalled (step %d: %s); falling back to the sequential schedule%s\n", s->step_tag,
            s->res_host[RES_STALL] == 2.0 ? "the factorisation gave up waiting for every camera's diag U" :
            s->res_host[RES_STALL] >= 1000.0 || s->res_host[RES_STALL] == 5.0 || s->res_host[RES_STALL] == 6.0 ? "the block back-substitution gave up waiting for a hand-over" :
            s->res_host[RES_STALL] >= 10.0 ? "a tile of the factorisation gave up waiting for its stage" :
            s->res_host[RES_STALL] != 0.0 ? "the factorisation gave up waiting for its columns" : "the back-substitution gave up waiting for the solve",
            s->pipe_stalls + 1 < 3 && (!s->test_stall || s->test_stall == 4) ? " for this step" : "");
    { const int rcs = SyncSolver(s); if (rcs != RSBA_OK) return rcs; }
    HIPCHK(hipMemset(s->tiled.sync_cnt, 0, (size_t)s->tiled.nsync * sizeof(int)));
    HIPCHK(hipMemset(s->tiled.tree_error + 1, 0, sizeof(int)));
    s->tiled.ticket_base = 0;
    if (s->mc_flags) HIPCHK(hipMemset(s->mc_flags, 0, 64 * sizeof(int)));
    if (s->tc_tiles > 0) {
      // the tiled factorisation gave up half-way: its flags, and the hand-over buffers whose slots are their own flags, back to
      // their initial state
      HIPCHK(hipMemset(s->tc_flags, 0, s->tc_nflags * sizeof(int)));
      HIPCHK(hipMemset(s->tc_hand, 0xff, s->tc_hand_doubles * sizeof(double)));
      if (s->tc_xs) HIPCHK(hipMemset(s->tc_xs, 0xff, s->tc_xs_doubles * sizeof(double)));
      if (s->tc_ys) HIPCHK(hipMemset(s->tc_ys, 0xff, s->tc_ys_doubles * sizeof(double)));
      HIPCHK(hipMemset(s->tiled.ready + RSBA_READY_STARTED, 0, sizeof(int)));
    }
    // the repeat runs sequentially; a solver that has timed out three times stays there
    ++s->pipe_stalls;
    const bool was_mg = s->pipelined_mg;
    s->pipelined = false; s->pipelined_mg = false;
    s->tiled.pt_valid = false;
    const int rc_rep = PointsStep(s, radius, first, keep_system_copy);
    if (s->pipe_stalls < 3 && (!s->test_stall || s->test_stall == 4) && !s->pipeline_off) { s->pipelined = true; s->pipelined_mg = was_mg; s->pipe_check_resident = true; }
    return rc_rep;
  }
  return RSBA_OK;
}

// Cost and gradient at the current x without a solve: what Ceres' HandleSuccessfulStep evaluates at the accepted point.
// Only needed when the run ends right after an accepted step (otherwise the next step's linearisation delivers them).
// Leaves res_host[RES_COST_X] and res_host[RES_GMAX] filled.
static int PointsGradient(rsba_solver* s, double radius) {
  const int C = s->C, P = s->P, x = s->cur;
  hipStream_t st = s->stream;
  IterParams ip = MakeIterParams(s->opt, radius, false);
  ip.cam_free = s->cam_free;
  ip.pt_const = s->pt_const;
  KernelTimer& T = s->timer;
  T.NextStep();
  { const int rcd = DrainAhead(s); if (rcd != RSBA_OK) return rcd; }
  if (s->opt.schur_impl == 0) {
    HIPCHK(hipMemsetAsync(s->red, 0, s->L.size() * sizeof(double), st));
    const bool stage = (size_t)C * (RSBA_ACC_PER_CAM + RSBA_CC_LDS) * sizeof(double) <= 96 * 1024;
    const size_t lds = (size_t)C * (RSBA_ACC_PER_CAM + (stage ? RSBA_CC_LDS : 0)) * sizeof(double);
    if (stage)
      k_linearize_schur_ref<true><<<s->grid_lin, 256, lds, st>>>(C, P, s->obs_u, s->obs_v, s->obs_cam, s->pt_ptr, s->camc[x], s->pts[x], s->scale_p,
                                                                 s->red, s->L, s->gmax, s->block_scal, ip);
    else
      k_linearize_schur_ref<false><<<s->grid_lin, 256, lds, st>>>(C, P, s->obs_u, s->obs_v, s->obs_cam, s->pt_ptr, s->camc[x], s->pts[x], s->scale_p,
                                                                  s->red, s->L, s->gmax, s->block_scal, ip);
    k_finish_linearize<<<1, 256, 0, st>>>(s->grid_lin, s->block_scal, s->red, s->L, s->gmax);
  } else {
    LaunchPointSide(s->tiled, s, ip, T, st);
    s->tiled.LaunchSelfOnly(s, ip, T, st);
  }
  if (s->comm) {
    // g_c and the scalars are sums over the ranks' shards, max |g_p| a maximum
    COMMCHK(s->comm->GroupStart());
    COMMCHK(s->comm->SumDoubles(s->red + s->L.gc(), (size_t)s->nc, st));
    COMMCHK(s->comm->SumDoubles(s->red + s->L.scal(), 8, st));
    COMMCHK(s->comm->MaxDoubles(s->gmax, 1, st));
    COMMCHK(s->comm->GroupEnd());
  }
  k_gradient_result<<<1, 256, 0, st>>>(s->red, s->L, s->gmax, s->res, s->res_host, s->res_seq + 1.0);
  HIPCHK(hipGetLastError());
  return WaitResult(s, st);
}

// ------------------------------------------------------------------------------------------------
// TrustRegionMinimizer, host side.  `step(radius, first)` must leave res[] filled.
// ------------------------------------------------------------------------------------------------
// `eval()` must leave res[RES_COST_X] and res[RES_GMAX] of the CURRENT point filled, without a solve.
//
// Order of the tests is Ceres 1.14's (trust_region_minimizer.cc): after an
// accepted step the new point is re-linearised and ITS gradient decides the gradient-tolerance stop before any further
// solve is interpreted.  Here a step() call linearises, solves and evaluates the candidate in one go, so the new point's
// cost and gradient arrive with the NEXT step(): they are written back into the accepted iteration's row, and when the
// gradient test fires that step's solve and candidate are discarded.  When the run ends right after an accepted step
// (iteration limit), eval() supplies them.
template <typename StepFn, typename AcceptFn, typename EvalFn>
static int MinimizeLoop(rsba_solver* s, rsba_summary* sum, StepFn step, AcceptFn accept, EvalFn eval) {
  const rsba_options& o = s->opt;
  s->iters.clear();
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  int invalid_run = 0;
  bool first = true;
  bool x_moved = false;   // the latest row is an accepted step whose cost / gradient at the new point are not in yet
  size_t printed = 0;     // rows of the progress table already on stdout
  double x_cost = 0, gmax = 0, x_norm = 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto t_iter = t_start;
  sum->num_successful_steps = sum->num_unsuccessful_steps = 0;
  sum->termination_type = RSBA_NO_CONVERGENCE; sum->stop_reason = RSBA_STOP_NONE;
  // Ceres' progress table (bundle_adjustment_manager.cpp:92 minimizer_progress_to_stdout): a row goes out once it is final
  auto flush_rows = [&]() {
    if (!o.minimizer_progress_to_stdout) { printed = s->iters.size(); return; }
    for (; printed < s->iters.size(); ++printed) {
      const rsba_iteration& r = s->iters[printed];
      if (printed == 0) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter  iter_time  total_time\n");
      printf("%4d % 8e   % 3.2e   % 3.2e  % 3.2e  % 3.2e % 3.2e     % 4d   % 3.2e   % 3.2e\n", r.iteration, r.cost, r.cost_change, r.gradient_max_norm,
             r.step_norm, r.relative_decrease, r.trust_region_radius, r.linear_solver_iterations, r.iteration_time_in_seconds, r.cumulative_time_in_seconds);
    }
  };
  auto push_row = [&](rsba_iteration& it) {
    const auto now = std::chrono::steady_clock::now();
    it.iteration_time_in_seconds = std::chrono::duration<double>(now - t_iter).count();
    it.cumulative_time_in_seconds = std::chrono::duration<double>(now - t_start).count();
    t_iter = now;
    s->iters.push_back(it);
  };
  auto finish = [&](int term, int reason) {
    sum->termination_type = term; sum->stop_reason = reason; sum->final_cost = x_cost;
    sum->num_iterations = (int)s->iters.size() - 1;
    flush_rows();
    return RSBA_OK;
  };
  // the accepted point's own cost and gradient (evaluated by the linearisation that follows the acceptance)
  auto settle_moved = [&]() {
    rsba_iteration& last = s->iters.back();
    last.cost = x_cost; last.gradient_max_norm = gmax;
    x_moved = false;
  };
  // "Maximum solver time reached" (trust_region_minimizer.cc MaxSolverTimeReached): wall time since the minimiser started PLUS
  // the preprocessor's (here: the upload, setup_seconds), tested in front of the iteration limit — and for the first time right
  // behind iteration 0, so that a budget of zero ends the run with no step taken, as Ceres does
  const auto t_loop0 = std::chrono::steady_clock::now();
  auto clock_says = [&]() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count() + s->setup_seconds >= o.max_solver_time_in_seconds;
  };
  // Several ranks: RANK 0's clock, as of the launch of the latest step, all-reduced with that step's candidate scalars (RES_TIME_UP) —
  // the same answer on every rank, one step behind the clock (a limit is a request to stop soon, not a deadline)
  s->share_clock = s->comm && s->comm->nranks() > 1 && o.max_solver_time_in_seconds < 1e9;
  auto out_of_time = [&]() { return s->share_clock ? s->res_host[RES_TIME_UP] != 0.0 : clock_says(); };
  for (;;) {
    // checks of FinalizeIterationAndCheckIfMinimizerCanContinue for the previous iteration
    if (!first) {
      const bool at_limit = s->iters.back().iteration >= o.max_num_iterations;
      const bool timed_out = out_of_time();
      if (x_moved && (at_limit || timed_out || radius < o.min_trust_region_radius)) {
        int rc = eval();
        if (rc != RSBA_OK) return rc;
        x_cost = s->res_host[RES_COST_X]; gmax = s->res_host[RES_GMAX];
        settle_moved();
      }
      // FinalizeIterationAndCheckIfMinimizerCanContinue's order: solver time, iterations, gradient, radius
      if (timed_out) return finish(RSBA_NO_CONVERGENCE, RSBA_STOP_MAX_TIME);
      if (at_limit) return finish(RSBA_NO_CONVERGENCE, RSBA_STOP_MAX_ITERATIONS);
      if (!x_moved && gmax <= o.gradient_tolerance) return finish(RSBA_CONVERGENCE, RSBA_STOP_GRADIENT);
      if (radius < o.min_trust_region_radius) return finish(RSBA_CONVERGENCE, RSBA_STOP_MIN_RADIUS);
    }
    s->lm_decrease_factor = decrease_factor;
    // (another step may follow this one unless it reaches the iteration limit: PointsStep may launch its head ahead)
    s->ahead_ok = (first ? 1 : s->iters.back().iteration + 1) < o.max_num_iterations;
    if (s->share_clock) s->time_up = (o.rank == 0 && clock_says()) ? 1.0 : 0.0;
    int rc = step(radius, first);
    if (rc != RSBA_OK) return rc;
    const double* r = s->res_host;
    x_cost = r[RES_COST_X]; gmax = r[RES_GMAX]; x_norm = std::sqrt(r[RES_XNORM2]);
    if (first) {
      // iteration 0: the evaluation at the starting point
      rsba_iteration it0{}; it0.iteration = 0; it0.cost = x_cost; it0.gradient_max_norm = gmax; it0.trust_region_radius = radius;
      push_row(it0);
      sum->initial_cost = x_cost;
      if (!std::isfinite(x_cost)) { sum->final_cost = x_cost; sum->termination_type = RSBA_FAILURE; sum->stop_reason = RSBA_STOP_INITIAL_FAILURE; sum->num_iterations = 0; flush_rows(); return RSBA_OK; }
      first = false;
      if (gmax <= o.gradient_tolerance) return finish(RSBA_CONVERGENCE, RSBA_STOP_GRADIENT);
      if (out_of_time()) return finish(RSBA_NO_CONVERGENCE, RSBA_STOP_MAX_TIME);   // (this step's solve and candidate are discarded)
      if (o.max_num_iterations <= 0) return finish(RSBA_NO_CONVERGENCE, RSBA_STOP_MAX_ITERATIONS);
      if (radius < o.min_trust_region_radius) return finish(RSBA_CONVERGENCE, RSBA_STOP_MIN_RADIUS);
    } else if (x_moved) {
      settle_moved();
      if (gmax <= o.gradient_tolerance) return finish(RSBA_CONVERGENCE, RSBA_STOP_GRADIENT);   // this step's solve and candidate are discarded
    }
    flush_rows();
    rsba_iteration it{};
    it.iteration = s->iters.back().iteration + 1;
    it.linear_solver_iterations = 1;   // a direct solve (Ceres reports 1 for DENSE_SCHUR)
    it.cost = x_cost; it.gradient_max_norm = gmax;
    const bool solved = r[RES_CHOL_OK] != 0.0 && std::isfinite(r[RES_MCC]) && std::isfinite(r[RES_STEP2]);
    const double mcc = r[RES_MCC];
    it.step_is_valid = solved && mcc > 0.0;
    if (!it.step_is_valid) {
      ++invalid_run;
      ++sum->num_unsuccessful_steps;
      if (invalid_run >= o.max_num_consecutive_invalid_steps) { it.trust_region_radius = radius; push_row(it); return finish(RSBA_FAILURE, RSBA_STOP_INVALID_STEPS); }
      radius /= decrease_factor; decrease_factor *= 2.0;
      it.trust_region_radius = radius;
      push_row(it);
      continue;
    }
    invalid_run = 0;
    const double cand_cost = r[RES_COST_C];
    it.step_norm = std::sqrt(r[RES_STEP2]);
    it.trust_region_radius = radius;
    if (it.step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { push_row(it); return finish(RSBA_CONVERGENCE, RSBA_STOP_PARAMETER); }
    it.cost_change = x_cost - cand_cost;
    if (std::fabs(it.cost_change) <= o.function_tolerance * x_cost) { push_row(it); return finish(RSBA_CONVERGENCE, RSBA_STOP_FUNCTION); }
    it.relative_decrease = it.cost_change / mcc;
    if (it.relative_decrease > o.min_relative_decrease) {
      accept();
      x_cost = cand_cost;  // re-evaluated by the next linearisation (settle_moved); kept for the summary if we stop here
      s->final_sumsq = r[RES_SUMSQ_C];
      // (2 rho - 1)^3: Cube (ba_math.hpp) — pow's bits (trust_region_minimizer.cc calls pow), and the same sequence of exact
      // operations the device runs when it takes this decision for the damping kernel it has queued (DecideStep)
      const double t = 2.0 * it.relative_decrease - 1.0;
      radius = radius / std::max(1.0 / 3.0, 1.0 - Cube(t));
      radius = std::min(o.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      it.step_is_successful = 1; it.cost = cand_cost; it.trust_region_radius = radius;
      ++sum->num_successful_steps;
      x_moved = true;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0;
      it.trust_region_radius = radius;
      ++sum->num_unsuccessful_steps;
    }
    push_row(it);
  }
}

}  // namespace rsba

// ================================================================================================
// C ABI (declared in include/rsba.h)
// ================================================================================================
extern "C" {

int rsba_comm_unique_id(void* out128) {
  if (!out128) return RSBA_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return RSBA_ERR_COMM;
  memcpy(out128, &id, sizeof(id));
  return RSBA_OK;
}

int rsba_comm_loopback_id(void* out128) {
  if (!out128) return RSBA_ERR_ARG;
  rsba::LoopbackComm::NewId(out128);
  return RSBA_OK;
}

int rsba_comm_shm_id(const char* name, void* out128) {
  if (!out128 || !rsba::ShmComm::NewId(name, out128)) return RSBA_ERR_ARG;
  return RSBA_OK;
}

void rsba_comm_finalize(void) { rsba::RcclComm::FinalizeAll(); }

int rsba_solver_comm_nranks(const rsba_solver* s) {
  if (!s) return 0;
  if (!s->comm) return 1;
  return s->comm->nranks();
}

int rsba_solver_schedule_info(const rsba_solver* s, rsba_schedule_info* out) {
  if (!s || !out) return RSBA_ERR_ARG;
  memset(out, 0, sizeof(*out));
  out->schedule = s->pipelined_mg ? 2 : (s->pipelined ? 1 : 0);
  out->stalls = s->pipe_stalls + s->other_stalls;
  out->fallbacks = (s->pipe_stalls >= 3 ? 1 : 0) + s->other_stalls;
  out->comm_nranks = s->comm ? s->comm->nranks() : 1;
  out->chol_workgroups = s->nc > RSBA_CHOL_MAXN ? s->tc_tiles : s->chol_wgs + (s->border_cols > 0 ? 1 : 0);   // (the border's workgroup: ba_cholesky_border.hpp)
  out->schur_impl = s->opt.schur_impl;
  strncpy(out->comm_kind, s->comm ? s->comm->kind() : "none", sizeof(out->comm_kind) - 1);
  return RSBA_OK;
}

int rsba_solver_create(rsba_problem* p, const rsba_options* o, rsba_solver** out) {
  if (!p || !out) return RSBA_ERR_ARG;
  if (rsba::DeviceCount() <= 0) return RSBA_ERR_NO_DEVICE;
  rsba_options opt;
  if (o) opt = *o; else rsba_options_default(&opt);
  if (opt.world_size > 1 && (!opt.comm_unique_id || opt.rank < 0 || opt.rank >= opt.world_size)) return RSBA_ERR_ARG;
  if (opt.world_size > 1 && p->model != RSBA_MODEL_POINTS) return RSBA_ERR_UNSUPPORTED;  // marker-chain: replicas only
  if (!(opt.max_solver_time_in_seconds >= 0.0)) return RSBA_ERR_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  rsba_solver* s = new rsba_solver();
  s->prob = p; s->opt = opt;
  if (opt.device >= 0) { if (hipSetDevice(opt.device) != hipSuccess) { delete s; return RSBA_ERR_HIP; } }
  if (hipGetDevice(&s->device) != hipSuccess) { delete s; return RSBA_ERR_HIP; }
  if (opt.stream) s->stream = (hipStream_t)opt.stream;
  else { if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return RSBA_ERR_HIP; } s->own_stream = true; }
  s->timer.Enable(opt.profile_kernels);
  if (opt.profile_kernels) s->timer.Reserve(256);
  int rc = RSBA_OK;
  if (opt.world_size > 1 || getenv("RSBA_FORCE_COMM")) {
    // RCCL (one communicator per unique id and process; RSBA_FORCE_COMM=1: a 1-rank communicator, so that the collective
    // path can be exercised on a single GPU), or — an id from rsba_comm_loopback_id — the ranks of one process on one GPU
    // or — rsba_comm_shm_id — processes of one host staging through shared memory (the slots sized for this problem's payload)
    const size_t nc6 = 6 * (size_t)std::max(p->num_cameras, 1);
    s->comm = opt.world_size > 1 && rsba::LoopbackComm::IsLoopbackId(opt.comm_unique_id)
                  ? rsba::LoopbackComm::Create(opt.world_size, opt.rank, opt.comm_unique_id)
              : opt.world_size > 1 && rsba::ShmComm::IsShmId(opt.comm_unique_id)
                  ? rsba::ShmComm::Create(opt.world_size, opt.rank, opt.comm_unique_id, (nc6 * nc6 + 3 * nc6 + 64) * sizeof(double))
                  : rsba::RcclComm::Create(opt.world_size, opt.rank, opt.comm_unique_id);
    if (!s->comm) { rsba::FreeSolver(s); return RSBA_ERR_COMM; }
  }
  std::shared_ptr<rsba::Comm> comm_keep = s->comm;   // (outlives s on the error paths below)
  rsba::CommScope device_turn(comm_keep.get());
  if (s->comm && getenv("RSBA_TEST_STALL_STEP") && (!getenv("RSBA_TEST_STALL_RANK") || atoi(getenv("RSBA_TEST_STALL_RANK")) == opt.rank))
    s->inject_stall_step = atoi(getenv("RSBA_TEST_STALL_STEP"));
  s->hostprof = getenv("RSBA_HOSTPROF") != nullptr;
  if (hipHostMalloc((void**)&s->res_host, (RES_SIZE + 8) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { rsba::FreeSolver(s); return RSBA_ERR_HIP; }
  memset(s->res_host, 0, (RES_SIZE + 8) * sizeof(double));   // [RES_SIZE]: the factorisation's "resident" word (StageGate)
  if (p->model == RSBA_MODEL_POINTS) {
    rc = rsba::UploadPoints(s);
    if (rc == RSBA_OK && s->tiled.tree_error) s->tiled.error_flag = reinterpret_cast<int*>(s->res_host + RES_SIZE + 4);   // (zeroed above)
  } else {
    const bool any_const_block = std::find(p->block_constant.begin(), p->block_constant.end(), (uint8_t)1) != p->block_constant.end();
    s->eliminate_times = !any_const_block && rsba::MarkerSchurDevice::Wanted(*p, opt.schur_impl);   // (constant blocks: the dense path)
    rc = s->eliminate_times ? s->marker_schur.Upload(*p) : s->marker.Upload(*p);
    if (s->eliminate_times && rc == RSBA_ERR_UNSUPPORTED) {
      // duplicate detections, no camera / marker block at all, or a time wider than the kernel's LDS: the dense path is general
      s->marker_schur.Free();
      s->eliminate_times = false;
      rc = s->marker.Upload(*p);
    }
  }
  if (rc == RSBA_OK && hipDeviceSynchronize() != hipSuccess) rc = RSBA_ERR_HIP;
  if (rc != RSBA_OK) { if (comm_keep) comm_keep->Abort(); rsba::FreeSolver(s); return rc; }
  s->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *out = s;
  return RSBA_OK;
}

int rsba_solver_run(rsba_solver* s, rsba_summary* sum_out) {
  if (!s) return RSBA_ERR_ARG;
  rsba_summary sum; memset(&sum, 0, sizeof(sum));
  sum.setup_seconds = s->setup_seconds;
  if (hipSetDevice(s->device) != hipSuccess) return RSBA_ERR_HIP;
  int rc;
  rsba::CommScope device_turn(s->comm.get());
  s->timer.BeginRun();
  struct AbortOnError { rsba_solver* s; int* rc; ~AbortOnError() { if (*rc != RSBA_OK && s->comm) s->comm->Abort(); } } abort_guard{s, &rc};
  if (s->prob->model == RSBA_MODEL_POINTS) {
    const auto tr0 = std::chrono::steady_clock::now();
    if ((rc = rsba::ResetPoints(s)) != RSBA_OK) return rc;
    const auto tr1 = std::chrono::steady_clock::now();
    if (hipStreamSynchronize(s->stream) != hipSuccess) return RSBA_ERR_HIP;
    const auto t0 = std::chrono::steady_clock::now();
    double cur_radius = s->opt.initial_trust_region_radius;
    rc = rsba::MinimizeLoop(s, &sum, [&](double radius, bool first) { cur_radius = radius; return rsba::PointsStep(s, radius, first, false); },
                            [&]() { s->cur = 1 - s->cur; }, [&]() { return rsba::PointsGradient(s, cur_radius); });
    if (rc == RSBA_OK) rc = rsba::DrainAhead(s);   // (a run that a tolerance ended: the step launched ahead runs out)
    sum.minimizer_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("RSBA_RUNPROF")) {
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      fprintf(stderr, "rsba[runprof] reset issued %.1f us | its sync %.1f | loop %.1f (%d iterations) ", us(tr0, tr1), us(tr1, t0), 1e6 * sum.minimizer_seconds, sum.num_iterations);
      for (size_t i = 0; i < s->iters.size() && i < 3; ++i) fprintf(stderr, "| it %zu: %.1f ", i, 1e6 * s->iters[i].iteration_time_in_seconds);
      if (s->iters.size() > 1) fprintf(stderr, "| last it: %.1f", 1e6 * s->iters.back().iteration_time_in_seconds);
      fprintf(stderr, "\n");
    }
    rsba::TraceRingDump(s);
    if (s->hostprof && s->hp_n > 1) {
      const double n = (double)s->hp_n;
      fprintf(stderr, "rsba[hostprof] %ld steps, host us per step: previous result -> enter %.1f | -> point side + factorisation launched %.1f | -> Schur kernel launched %.1f | -> stage collectives issued %.1f | -> all launched %.1f | -> result %.1f\n",
              s->hp_n, s->hp_sum[0] / (n - 1), s->hp_sum[1] / n, s->hp_sum[2] / n, s->hp_sum[3] / n, s->hp_sum[4] / n, s->hp_sum[5] / n);
      for (double& v : s->hp_sum) v = 0.0; s->hp_n = 0;
    }
    if (rc == RSBA_OK && s->tiled.tree_error) {
      // a reducer workgroup of the Schur kernel gave up waiting for its tile (cannot happen by construction): the sums
      // it produced are garbage, so is the result
      // (the flag sits in the host-mapped result block: every kernel that could raise it has completed before the last result
      //  was posted; a blocking 4-byte copy here was ~20 us of every run)
      int bad = 0;
      if (s->tiled.error_flag) bad = *(volatile int*)s->tiled.error_flag;
      else if (hipMemcpy(&bad, s->tiled.tree_error, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
      if (bad) {
        fprintf(stderr, "rsba: in-kernel reduction timed out\n");
        rc = RSBA_ERR_HIP;
      }
    }
  } else {
    if ((rc = s->eliminate_times ? s->marker_schur.Reset(s->stream) : s->marker.Reset(s->stream)) != RSBA_OK) return rc;
    if (hipStreamSynchronize(s->stream) != hipSuccess) return RSBA_ERR_HIP;
    const auto t0 = std::chrono::steady_clock::now();
    double mk_radius = s->opt.initial_trust_region_radius;
    if (s->eliminate_times)
      rc = rsba::MinimizeLoop(s, &sum, [&](double radius, bool first) { mk_radius = radius; return s->marker_schur.Step(s->stream, s->opt, radius, first, s->res_host, s->timer); },
                              [&]() { s->marker_schur.Accept(); },
                              // (the marker-chain step is small: a full step at the accepted point, of which only cost and gradient are used)
                              [&]() { return s->marker_schur.Step(s->stream, s->opt, mk_radius, false, s->res_host, s->timer); });
    else
      rc = rsba::MinimizeLoop(s, &sum, [&](double radius, bool first) { mk_radius = radius; return s->marker.Step(s->stream, s->opt, radius, first, s->res_host, s->timer); },
                              [&]() { s->marker.Accept(); },
                              [&]() { return s->marker.Step(s->stream, s->opt, mk_radius, false, s->res_host, s->timer); });
    sum.minimizer_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  s->final_cost = sum.final_cost;
  s->last_summary = sum; s->has_run = true;
  if (sum_out) *sum_out = sum;
  return rc;
}

int rsba_solver_configure_run(rsba_solver* s, int32_t max_num_iterations, int32_t profile_kernels) {
  if (!s || max_num_iterations < 0 || profile_kernels < 0 || profile_kernels > 2) return RSBA_ERR_ARG;
  if (hipSetDevice(s->device) != hipSuccess) return RSBA_ERR_HIP;
  rsba::CommScope device_turn(s->comm.get());   // (event synchronisation: a loopback rank touches the device on its turn only)
  s->opt.max_num_iterations = max_num_iterations;
  s->opt.profile_kernels = profile_kernels;
  s->timer.Reset();
  s->timer.Enable(profile_kernels);
  if (profile_kernels) s->timer.Reserve(256);
  return RSBA_OK;
}

int rsba_solver_download(rsba_solver* s) {
  if (!s) return RSBA_ERR_ARG;
  if (hipSetDevice(s->device) != hipSuccess) return RSBA_ERR_HIP;
  rsba::CommScope device_turn(s->comm.get());   // (blocking copies are not limited to this rank's streams: on its turn only)
  rsba_problem& p = *s->prob;
  if (p.model == RSBA_MODEL_POINTS) {
    if (hipMemcpy(p.parameters.data(), s->cam[s->cur], 6 * s->C * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return RSBA_ERR_HIP;
    return rsba::DownloadPoints(s, s->pts[s->cur], p.parameters.data() + 6 * s->C);
  }
  return s->eliminate_times ? s->marker_schur.Download(&p) : s->marker.Download(&p);
}

int rsba_solver_iterations(const rsba_solver* s, rsba_iteration* out, int32_t capacity) {
  if (!s || (!out && capacity > 0)) return 0;
  const int n = std::min<int>((int)s->iters.size(), capacity);
  for (int i = 0; i < n; ++i) out[i] = s->iters[i];
  return n;
}

int rsba_solver_kernel_stats(const rsba_solver* s_in, rsba_kernel_stat* out, int32_t capacity) {
  if (!s_in) return 0;
  // (the accessor collects the pending event pairs of the last run: it synchronises events on the solver's device, and a
  //  loopback rank touches the device on its turn only)
  rsba_solver* s = const_cast<rsba_solver*>(s_in);
  if (hipSetDevice(s->device) != hipSuccess) return 0;
  rsba::CommScope device_turn(s->comm.get());
  int n = 0;
  for (const auto& kv : s->timer.Stats()) {
    if (n >= capacity) break;
    memset(&out[n], 0, sizeof(out[n]));
    strncpy(out[n].name, kv.first.c_str(), sizeof(out[n].name) - 1);
    out[n].launches = kv.second.first; out[n].total_ms = kv.second.second;
    ++n;
  }
  return n;
}

// ceres::Solver::Summary::FullReport() (printed by bundle_adjustment_manager.cpp:95): same sections, the values this
// solver has.  Returns the length the full text needs (snprintf semantics); buf may be NULL with capacity 0.
int rsba_solver_full_report(const rsba_solver* s, char* buf, int32_t capacity) {
  if (!s || !s->has_run || capacity < 0 || (!buf && capacity > 0)) return -1;
  const rsba_problem& p = *s->prob;
  const rsba_summary& m = s->last_summary;
  long long blocks0, params0, blocks1, params1, rblocks, residuals;
  if (p.model == RSBA_MODEL_POINTS) {
    int nconst = 0;
    for (uint8_t c : p.camera_constant) nconst += c ? 1 : 0;
    blocks0 = (long long)p.num_cameras + p.num_points; params0 = 6LL * p.num_cameras + 3LL * p.num_points;
    int nconst_pt = 0;
    for (uint8_t c : p.point_constant) nconst_pt += c ? 1 : 0;
    blocks1 = blocks0 - nconst - nconst_pt; params1 = params0 - 6LL * nconst - 3LL * nconst_pt;
    rblocks = p.num_observations; residuals = 2 * rblocks;
  } else {
    const int nb = p.num_cameras + p.num_times + p.num_markers;
    std::vector<char> used(nb, 0);
    for (int64_t i = 0; i < p.num_observations; ++i) {
      if (p.uses_camera(i)) used[p.camera_block(i)] = 1;
      used[p.time_block(i)] = 1;
      if (p.uses_marker(i)) used[p.marker_block(i)] = 1;
    }
    int nu = 0; for (char u : used) nu += u;
    int nconst = 0;
    for (int b = 0; b < nb && b < (int)p.block_constant.size(); ++b) nconst += (used[b] && p.block_constant[b]) ? 1 : 0;
    // blocks no residual touches never enter the ceres::Problem (camera 0 / marker 0 of the reference's wiring); constant ones leave the reduced program
    blocks0 = nu; params0 = 6LL * nu; blocks1 = nu - nconst; params1 = 6LL * (nu - nconst);
    rblocks = p.num_observations; residuals = 8 * rblocks;
  }
  static const char* kTerm[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
  char why[192] = "";
  const rsba_iteration* last = s->iters.empty() ? nullptr : &s->iters.back();
  switch (m.stop_reason) {
    case RSBA_STOP_FUNCTION:
      if (last) snprintf(why, sizeof(why), "Function tolerance reached. |cost_change|/cost: %e <= %e", std::fabs(last->cost_change) / std::max(last->cost + last->cost_change, 1e-300), s->opt.function_tolerance);
      break;
    case RSBA_STOP_PARAMETER: snprintf(why, sizeof(why), "Parameter tolerance reached. Relative step_norm <= %e", s->opt.parameter_tolerance); break;
    case RSBA_STOP_GRADIENT:
      if (last) snprintf(why, sizeof(why), "Gradient tolerance reached. Gradient max norm: %e <= %e", last->gradient_max_norm, s->opt.gradient_tolerance);
      break;
    case RSBA_STOP_MAX_ITERATIONS: snprintf(why, sizeof(why), "Maximum number of iterations reached. Number of iterations: %d.", m.num_iterations); break;
    case RSBA_STOP_MIN_RADIUS: snprintf(why, sizeof(why), "Minimum trust region radius reached. Trust region radius <= %e", s->opt.min_trust_region_radius); break;
    case RSBA_STOP_INVALID_STEPS: snprintf(why, sizeof(why), "Number of successive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d", s->opt.max_num_consecutive_invalid_steps); break;
    case RSBA_STOP_INITIAL_FAILURE: snprintf(why, sizeof(why), "Residual and Jacobian evaluation failed."); break;
    case RSBA_STOP_MAX_TIME: snprintf(why, sizeof(why), "Maximum solver time reached. Total solver time: %e >= %e.", m.minimizer_seconds + s->setup_seconds, s->opt.max_solver_time_in_seconds); break;   // (the value out_of_time() tests: Ceres counts the preprocessor's time)
    default: break;
  }
  hipDeviceProp_t prop; memset(&prop, 0, sizeof(prop));
  (void)hipGetDeviceProperties(&prop, s->device);
  const int n = snprintf(buf, (size_t)capacity,
      "\nSolver Summary (rsba %d.%02d, %s)\n\n"
      "%-32s%12s%25s\n"
      "%-32s%12lld%25lld\n%-32s%12lld%25lld\n%-32s%12lld%25lld\n%-32s%12lld%25lld\n\n"
      "%-25s%19s\n%-25s%19s\n%-25s%19s\n\n"
      "%-32s%12s%25s\n%-32s%12s%25s\n%-32s%12d%25d\n\n"
      "Cost:\n%-25s%19.6e\n%-25s%19.6e\n%-25s%19.6e\n\n"
      "%-32s%12d\n%-32s%12d\n%-32s%12d\n\n"
      "Time (in seconds):\n%-25s%19.6f\n%-25s%19.6f\n%-25s%19.6f\n\n"
      "Termination: %25s (%s)\n",
      RSBA_VERSION / 100, RSBA_VERSION % 100, prop.gcnArchName[0] ? prop.gcnArchName : "HIP device",
      "", "Original", "Reduced",
      "Parameter blocks", blocks0, blocks1, "Parameters", params0, params1, "Residual blocks", rblocks, rblocks, "Residuals", residuals, residuals,
      "Minimizer", "TRUST_REGION", "Dense linear algebra library", "HIP", "Trust region strategy", "LEVENBERG_MARQUARDT",
      "", "Given", "Used", "Linear solver", "DENSE_SCHUR", "DENSE_SCHUR", "GPUs", std::max(s->opt.world_size, 1), std::max(s->opt.world_size, 1),
      "Initial", m.initial_cost, "Final", m.final_cost, "Change", m.initial_cost - m.final_cost,
      "Minimizer iterations", m.num_successful_steps + m.num_unsuccessful_steps + 1, "Successful steps", m.num_successful_steps + 1, "Unsuccessful steps", m.num_unsuccessful_steps,
      "Preprocessor", m.setup_seconds, "Minimizer", m.minimizer_seconds, "Total", m.setup_seconds + m.minimizer_seconds,
      kTerm[std::min(std::max(m.termination_type, 0), 2)], why);
  return n;
}

int rsba_solver_final_costs(const rsba_solver* s, double* cost, double* sum_sq) {
  if (!s) return RSBA_ERR_ARG;
  if (cost) *cost = s->final_cost;
  if (sum_sq) *sum_sq = s->final_sumsq;
  return RSBA_OK;
}

void rsba_solver_destroy(rsba_solver* s) {
  // (hipFree and the diagnostics' blocking copies wait for the whole device: a loopback rank frees on its turn only, so that an
  //  early finisher cannot stall another rank's in-kernel waits into their budget)
  std::shared_ptr<rsba::Comm> comm_keep = s ? s->comm : nullptr;
  rsba::CommScope device_turn(comm_keep.get());
  if (s && s->mc_trace && s->tc_tiles > 0) {
    // diagnostic: the diagonal tiles' chain of the latest persistent tiled factorisation, microseconds since its first stamp.
    // 0 last update entered | 10 sub-diagonal rows there | 11 T there | 12 T in LDS | 1 X formed | 2 update done, first half |
    // 3 factored | 4 T, X published | 5 own update | 6 second half | 7 factored | 8 T published | 9 (sub-diagonal tile) rows handed over
    std::vector<long long> h((size_t)(s->tc_nrt + 1) * 24 + s->tc_tiles);
    if (hipMemcpy(h.data(), s->mc_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
      // (stamps of an earlier launch survive where this one wrote none: only what lies within 10 ms of the latest counts)
      long long tmax = 0, t0 = 0;
      for (size_t i = 0; i < (size_t)(s->tc_nrt + 1) * 24; ++i) tmax = std::max(tmax, h[i]);
      for (size_t i = 0; i < (size_t)(s->tc_nrt + 1) * 24; ++i) { if (h[i] < tmax - 1000000) h[i] = 0; if (h[i] && (!t0 || h[i] < t0)) t0 = h[i]; }
      static const int order[24] = {19, 20, 21, 22, 23, 16, 13, 14, 15, 17, 18, 9, 0, 10, 11, 12, 1, 2, 3, 4, 5, 6, 7, 8};   // 13..15: the sub-diagonal tile of this row: L11 there, solved, X rows there
      for (int J = 0; J < s->tc_nrt; ++J) {
        fprintf(stderr, "rsba[tc] tile %2d:", J);
        for (int k : order) { const long long v = h[(size_t)J * 24 + k]; fprintf(stderr, " %7.2f", v ? (v - t0) / 100.0 : -1.0); }
        fprintf(stderr, "\n");
      }
      // which tiles share a CU (HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 on gfx9; XCC_ID)
      std::map<long long, std::vector<int>> where;
      std::vector<int> tiles_of = TileOrder(s->tc_nrt, s->tc_map ? DeviceCUs() : 1 << 30);
      for (int t = 0; t < s->tc_tiles; ++t) {
        const long long v = h[(size_t)(s->tc_nrt + 1) * 24 + t];
        if (v) where[((v >> 32) << 16) | ((v >> 8) & 0xff)].push_back(t);
      }
      for (const auto& kv : where) if (kv.second.size() > 1) {
        fprintf(stderr, "rsba[tc] cu %llx:", (unsigned long long)kv.first);
        for (int t : kv.second) fprintf(stderr, " (%d,%d)", tiles_of[t] >> 8, tiles_of[t] & 255);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "rsba[tc] %zu CUs hold the %d tiles\n", where.size(), s->tc_tiles);
    }
    (void)hipFree(s->mc_trace);
  } else if (s && s->mc_trace) {
    // diagnostic: per workgroup and panel, microseconds since the kernel's first stamp
    std::vector<long long> h(kMcTraceWords);
    if (hipMemcpy(h.data(), s->mc_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
      long long t0 = 0; for (long long v : h) if (v && (!t0 || v < t0)) t0 = v;
      for (int w = 0; w < s->chol_wgs; ++w) for (int p = 0; p < s->nc / 32; ++p) {
        fprintf(stderr, "rsba[mc] wg %d panel %2d:", w, p);
        for (int k = 0; k < 8; ++k) { const long long v = h[((size_t)w * 16 + p) * 8 + k]; fprintf(stderr, " %7.2f", v ? (v - t0) / 100.0 : -1.0); }
        fprintf(stderr, "\n");
      }
      if (s->border_cols > 0) {   // the border's workgroup: BorderWorkgroup's stamps (ba_cholesky_border.hpp)
        fprintf(stderr, "rsba[mc] border:");
        for (int k = 0; k < 30; ++k) { const long long v = h[(size_t)s->chol_wgs * 16 * 8 + k]; fprintf(stderr, " %d:%.2f", k, v ? (v - t0) / 100.0 : -1.0); }
        fprintf(stderr, "\n");
      }
    }
    (void)hipFree(s->mc_trace);
  }
#ifdef RSBA_PROFILE_PHASES
  { long long h[16]; if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rsba::g_phase_cycles), sizeof(h)) == hipSuccess) { fprintf(stderr, "rsba[phases]"); for (int i = 0; i < 16; ++i) fprintf(stderr, " %lld", h[i]); fprintf(stderr, "\n"); } }
  { long long h[16]; if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rsba::g_mt_cycles), sizeof(h)) == hipSuccess) { fprintf(stderr, "rsba[mt-phases]"); for (int i = 0; i < 16; ++i) fprintf(stderr, " %lld", h[i]); fprintf(stderr, "\n"); } }
#endif
  rsba::FreeSolver(s);
}

int rsba_solve(rsba_problem* p, const rsba_options* o, rsba_summary* summary) {
  rsba_solver* s = nullptr;
  int rc = rsba_solver_create(p, o, &s);
  if (rc != RSBA_OK) return rc;
  rc = rsba_solver_run(s, summary);
  if (rc == RSBA_OK) rc = rsba_solver_download(s);
  rsba_solver_destroy(s);
  return rc;
}

int rsba_points_linearize_and_step(rsba_problem* p, const rsba_options* o, double radius, double* S, double* rhs, double* delta, double* scalars) {
  if (!p || p->model != RSBA_MODEL_POINTS) return RSBA_ERR_ARG;
  rsba_solver* s = nullptr;
  int rc = rsba_solver_create(p, o, &s);
  if (rc != RSBA_OK) return rc;
  rc = rsba::ResetPoints(s);
  if (rc == RSBA_OK) rc = rsba::PointsStep(s, radius, true, true);
  if (rc == RSBA_OK) {
    const int nc = s->nc;
    if (S && hipMemcpy(S, s->S_copy, (size_t)nc * nc * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RSBA_ERR_HIP;
    if (rhs && hipMemcpy(rhs, s->rhs_copy, nc * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RSBA_ERR_HIP;
    if (delta) {
      std::vector<double> xc(6 * s->C + 3 * (size_t)s->P);
      if (hipMemcpy(xc.data(), s->cam[1 - s->cur], 6 * s->C * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RSBA_ERR_HIP;
      if (rsba::DownloadPoints(s, s->pts[1 - s->cur], xc.data() + 6 * s->C) != RSBA_OK) rc = RSBA_ERR_HIP;
      for (size_t i = 0; i < xc.size(); ++i) delta[i] = xc[i] - p->parameters[i];
    }
    if (scalars) {
      const double* r = s->res_host;
      scalars[0] = r[RES_COST_X]; scalars[1] = r[RES_MCC]; scalars[2] = r[RES_GMAX]; scalars[3] = r[RES_CHOL_OK];
      scalars[4] = r[RES_COST_C]; scalars[5] = std::sqrt(r[RES_STEP2]); scalars[6] = std::sqrt(r[RES_XNORM2]); scalars[7] = r[RES_SUMSQ_C];
    }
  }
  rsba_solver_destroy(s);
  return rc;
}

int rsba_points_linearize_payload(rsba_problem* p, const rsba_options* o, double radius, double* payload, int64_t capacity, int64_t* count) {
  if (!p || p->model != RSBA_MODEL_POINTS || !count) return RSBA_ERR_ARG;
  rsba_solver* s = nullptr;
  int rc = rsba_solver_create(p, o, &s);
  if (rc != RSBA_OK) return rc;
  *count = (int64_t)s->L.size() + 1;
  if (payload) {
    if (capacity < *count) rc = RSBA_ERR_ARG;
    if (rc == RSBA_OK) rc = rsba::ResetPoints(s);
    if (rc == RSBA_OK) rc = rsba::PointsStep(s, radius, true, true);   // keeps `red` as the linearisation left it
    if (rc == RSBA_OK && hipMemcpy(payload, s->red, s->L.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RSBA_ERR_HIP;
    if (rc == RSBA_OK && hipMemcpy(payload + s->L.size(), s->gmax, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RSBA_ERR_HIP;
  }
  rsba_solver_destroy(s);
  return rc;
}

int rsba_reprojection_error(rsba_problem* p, const rsba_options* o, double* error, double* rms) {
  if (!p) return RSBA_ERR_ARG;
  rsba_solver* s = nullptr;
  rsba_options opt; if (o) opt = *o; else rsba_options_default(&opt);
  opt.world_size = 1; opt.comm_unique_id = nullptr; opt.huber_delta = 0.0;
  int rc = rsba_solver_create(p, &opt, &s);
  if (rc != RSBA_OK) return rc;
  double sumsq = 0; int64_t npts = 0;
  if (p->model == RSBA_MODEL_POINTS) {
    rc = rsba::ResetPoints(s);
    if (rc == RSBA_OK) {
      k_camera_constants<<<(s->C + 63) / 64, 64, 0, s->stream>>>(s->C, s->cam[0], s->intr, s->camc[0]);
      k_cost_only<<<s->grid_pts, 256, 0, s->stream>>>(s->P, s->sliced(), s->camc[0], s->pts[0], s->block_part, 0.0);
      k_finish_candidate<<<1, 256, 0, s->stream>>>(s->grid_pts, s->block_part, s->small_red, nullptr, nullptr, 0.0, nullptr);
      double h[8];
      if (hipMemcpyAsync(h, s->small_red, 8 * sizeof(double), hipMemcpyDeviceToHost, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) rc = RSBA_ERR_HIP;
      sumsq = h[4]; npts = p->num_observations;
    }
  } else {
    rc = s->eliminate_times ? s->marker_schur.SumSquares(s->stream, &sumsq) : s->marker.SumSquares(s->stream, &sumsq);
    npts = 4 * p->num_observations;
  }
  if (rc == RSBA_OK) {
    if (error) *error = sumsq / 2.0;                                   // reprojection_check.cpp:81
    if (rms) *rms = std::sqrt((sumsq / 2.0) * 2.0 / (npts * 2.0));     // reprojection_check.cpp:101
  }
  rsba_solver_destroy(s);
  return rc;
}

int rsba_reprojection_check_files(const char* correspondence_txt, const char* point3d_txt, const char* camera_transform_xml,
                                  const double* intrinsics, double* error, double* rms) {
  rsba_problem* p = nullptr;
  int rc = rsba::LoadReprojectionCheck(correspondence_txt, point3d_txt, camera_transform_xml, intrinsics, &p);
  if (rc != RSBA_OK) return rc;
  rc = rsba_reprojection_error(p, nullptr, error, rms);
  delete p;
  return rc;
}

}  // extern "C"
