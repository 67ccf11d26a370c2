// Offline lab for the Schur kernel's point order (lane utilisation of the pair tiles' hit loops).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include <chrono>
using namespace std;
static const int C = 64, K = 20, TG = 16, CH = 512;
int P;
vector<uint8_t> vis;   // P x K
static inline uint64_t mix(uint64_t& st) { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// wave of pair (a<b) in its tile, and whether the tile is diagonal
static inline void pair_wave(int a, int b, int* tile, int* wave, bool* diag) {
  const int ga = a / TG, gb = b / TG, ia = a % TG, ib = b % TG;
  *tile = ga * 4 + gb; *diag = ga == gb;
  if (ga != gb) { *wave = ia >> 2; return; }
  const int dt = ia * 15 - ia * (ia - 1) / 2 + ib - ia - 1;
  *wave = dt >> 6;
}
// evaluate: perm[pos] = point; chunks of CH positions
double evaluate(const vector<int>& perm, bool verbose) {
  const int nu = (P + CH - 1) / CH;
  long long hits = 0, lanetrips = 0;
  vector<uint16_t> cnt((size_t)C * C * 2);
  long long hits_d = 0, lt_d = 0;
  for (int u = 0; u < nu; ++u) {
    fill(cnt.begin(), cnt.end(), 0);
    const int p0 = u * CH, p1 = min(P, p0 + CH);
    for (int pos = p0; pos < p1; ++pos) {
      const uint8_t* cj = &vis[(size_t)perm[pos] * K];
      const int par = ((pos - p0) >> 6) & 1;
      for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) cnt[((size_t)cj[x] * C + cj[y]) * 2 + par]++;
    }
    // per tile, wave
    for (int ga = 0; ga < 4; ++ga) for (int gb = ga; gb < 4; ++gb) {
      if (ga != gb) {
        for (int w = 0; w < 4; ++w) {
          int mx = 0; long long h = 0;
          for (int ia = 4 * w; ia < 4 * w + 4; ++ia) for (int ib = 0; ib < 16; ++ib) { const int a = 16 * ga + ia, b = 16 * gb + ib; const int c = cnt[((size_t)a * C + b) * 2] + cnt[((size_t)a * C + b) * 2 + 1]; mx = max(mx, c); h += c; }
          hits += h; lanetrips += 64LL * mx;
        }
      } else {
        for (int half = 0; half < 2; ++half) for (int w = 0; w < 2; ++w) {
          int mx = 0; long long h = 0;
          for (int ia = 0; ia < 16; ++ia) for (int ib = ia + 1; ib < 16; ++ib) {
            const int dt = ia * 15 - ia * (ia - 1) / 2 + ib - ia - 1;
            if ((dt >> 6) != w) continue;
            const int a = 16 * ga + ia, b = 16 * ga + ib; const int c = cnt[((size_t)a * C + b) * 2 + half]; mx = max(mx, c); h += c;
          }
          hits += h; lanetrips += 64LL * mx; hits_d += h; lt_d += 64LL * mx;
        }
      }
    }
  }
  if (verbose) printf("  hits %lld lane-trips %lld utilisation %.2f %%  (diagonal tiles alone %.2f %% of %lld lane-trips)\n", hits, lanetrips, 100.0 * hits / lanetrips, 100.0 * hits_d / lt_d, lt_d);
  return (double)lanetrips;
}

// the product's greedy (BalancedPointOrder, single stream for the lab)
vector<int> greedy(int D) {
  const int nu = (P + CH - 1) / CH;
  vector<int> ucap(nu), ubeg(nu);
  for (int u = 0; u < nu; ++u) { ubeg[u] = u * CH; ucap[u] = min(P, u * CH + CH) - u * CH; }
  vector<uint16_t> cnt((size_t)nu * C * C, 0);
  vector<int> fillv(nu, 0), unit_of(P, 0), visit(P);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  iota(visit.begin(), visit.end(), 0);
  for (int j = P - 1; j > 0; --j) swap(visit[j], visit[(size_t)(mix(st) % (uint64_t)(j + 1))]);
  const int kStreams = 8;
  vector<int> su(kStreams + 1, 0), sp(kStreams + 1, 0);
  for (int t = 0; t < kStreams; ++t) { su[t + 1] = (int)((int64_t)nu * (t + 1) / kStreams); int cs = 0; for (int g = su[t]; g < su[t + 1]; ++g) cs += ucap[g]; sp[t + 1] = sp[t] + cs; }
  for (int t = 0; t < kStreams; ++t) {
    const int g0 = su[t], ng = su[t + 1] - su[t]; const int Dt = min(D, ng);
    uint64_t s2 = 0xD1B54A32D192ED03ull * (uint64_t)(t + 1);
    int open_from = g0;
    for (int q = sp[t]; q < sp[t + 1]; ++q) {
      const int j = visit[q]; const uint8_t* cj = &vis[(size_t)j * K];
      int best = -1; double best_s = 0;
      auto score = [&](int g) { const uint16_t* c = &cnt[(size_t)g * C * C]; long sum = 0; for (int x = 0; x < K; ++x) { const uint16_t* row = c + (size_t)cj[x] * C; for (int y = x + 1; y < K; ++y) sum += row[cj[y]]; } const double sc = (double)sum / (double)(fillv[g] + 1); if (best < 0 || sc < best_s) { best = g; best_s = sc; } };
      if (ng <= D) { for (int g = g0; g < g0 + ng; ++g) if (fillv[g] < ucap[g]) score(g); }
      else for (int d = 0, tries = 0; d < Dt || best < 0; ++tries) { int g; if (tries < 4 * Dt) { g = g0 + (int)(mix(s2) % (uint64_t)ng); if (fillv[g] >= ucap[g]) continue; } else { while (open_from < g0 + ng && fillv[open_from] >= ucap[open_from]) ++open_from; g = open_from; if (g >= g0 + ng) break; } ++d; score(g); if (tries >= 4 * Dt) break; }
      uint16_t* c = &cnt[(size_t)best * C * C];
      for (int x = 0; x < K; ++x) { uint16_t* row = c + (size_t)cj[x] * C; for (int y = x + 1; y < K; ++y) ++row[cj[y]]; }
      unit_of[j] = best; ++fillv[best];
    }
  }
  vector<int> next(ubeg), perm(P, -1);
  for (int j = 0; j < P; ++j) perm[next[unit_of[j]]++] = j;
  return perm;
}

// within every chunk: the points dealt to the even / odd 64-point words so that every DIAGONAL pair (both cameras of one group) gets
// about the same number of hits in either half (the diagonal tiles' two workgroup halves walk the even / odd words)
vector<int> parity_balance(const vector<int>& perm) {
  const int nu = (P + CH - 1) / CH;
  vector<int> out(P);
  vector<uint16_t> cnt((size_t)C * C * 2);
  for (int u = 0; u < nu; ++u) {
    const int p0 = u * CH, p1 = min(P, p0 + CH), np = p1 - p0;
    const int nwords = (np + 63) / 64;
    int cap[2] = {0, 0};
    for (int w = 0; w < nwords; ++w) cap[w & 1] += min(64, np - 64 * w);
    fill(cnt.begin(), cnt.end(), 0);
    vector<int> half[2];
    for (int pos = p0; pos < p1; ++pos) {
      const int j = perm[pos]; const uint8_t* cj = &vis[(size_t)j * K];
      long sc[2] = {0, 0};
      for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) if (cj[x] / TG == cj[y] / TG) { const uint16_t* c = &cnt[((size_t)cj[x] * C + cj[y]) * 2]; sc[0] += c[0]; sc[1] += c[1]; }
      int h;
      if ((int)half[0].size() >= cap[0]) h = 1; else if ((int)half[1].size() >= cap[1]) h = 0;
      else h = (sc[0] * (long)(half[1].size() + 1) <= sc[1] * (long)(half[0].size() + 1)) ? 0 : 1;
      half[h].push_back(j);
      for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) if (cj[x] / TG == cj[y] / TG) cnt[((size_t)cj[x] * C + cj[y]) * 2 + h]++;
    }
    // positions: half 0 fills the even words, half 1 the odd ones
    size_t i0 = 0, i1 = 0;
    for (int w = 0; w < nwords; ++w) { const int n = min(64, np - 64 * w); for (int l = 0; l < n; ++l) out[p0 + 64 * w + l] = (w & 1) ? half[1][i1++] : half[0][i0++]; }
  }
  return out;
}

// ---- local search on the chunk assignment: objective = sum over (chunk, wave) of the busiest lane's hit count (x 64 = lane-trips)
struct Refine {
  int nu; vector<int> unit_of, pos_in; vector<vector<int>> members; vector<uint16_t> cnt;   // cnt[u][a][b], a < b, whole chunk (parity ignored here)
  // lane lists per wave slot: wl[ws] = list of pair indices a*C+b; wave slot of pair: ws_of[a*C+b]
  vector<int> ws_of; vector<vector<int>> wl; int nws;
  vector<int> wmax;   // [u][ws]
  void init(const vector<int>& perm) {
    nu = (P + CH - 1) / CH; unit_of.assign(P, 0); members.assign(nu, {}); cnt.assign((size_t)nu * C * C, 0);
    for (int pos = 0; pos < P; ++pos) { const int u = pos / CH; unit_of[perm[pos]] = u; members[u].push_back(perm[pos]); }
    ws_of.assign(C * C, -1); wl.clear();
    int id = 0; vector<int> key(C * C, -1);
    for (int a = 0; a < C; ++a) for (int b = a + 1; b < C; ++b) { int t, w; bool d; pair_wave(a, b, &t, &w, &d); key[a * C + b] = t * 4 + w; }
    vector<int> remap(64, -1);
    for (int a = 0; a < C; ++a) for (int b = a + 1; b < C; ++b) { const int k = key[a * C + b]; if (remap[k] < 0) { remap[k] = id++; wl.push_back({}); } ws_of[a * C + b] = remap[k]; wl[remap[k]].push_back(a * C + b); }
    nws = id;
    for (int j = 0; j < P; ++j) { const uint8_t* cj = &vis[(size_t)j * K]; uint16_t* c = &cnt[(size_t)unit_of[j] * C * C]; for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) c[cj[x] * C + cj[y]]++; }
    wmax.assign((size_t)nu * nws, 0);
    for (int u = 0; u < nu; ++u) for (int w = 0; w < nws; ++w) wmax[(size_t)u * nws + w] = wave_max(u, w);
  }
  int wave_max(int u, int w) const { const uint16_t* c = &cnt[(size_t)u * C * C]; int m = 0; for (int pr : wl[w]) m = max(m, (int)c[pr]); return m; }
  long long total() const { long long t = 0; for (int v : wmax) t += v; return t; }
  void apply(int j, int u, int d) { const uint8_t* cj = &vis[(size_t)j * K]; uint16_t* c = &cnt[(size_t)u * C * C]; for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) c[cj[x] * C + cj[y]] += d; }
  // objective change of swapping j (in u) with i (in v); leaves the counts swapped iff accepted
  bool try_swap(int j, int i) {
    const int u = unit_of[j], v = unit_of[i];
    if (u == v) return false;
    apply(j, u, -1); apply(i, u, +1); apply(i, v, -1); apply(j, v, +1);
    long long d = 0; static vector<int> nu_max, nv_max; nu_max.resize(nws); nv_max.resize(nws);
    for (int w = 0; w < nws; ++w) { nu_max[w] = wave_max(u, w); nv_max[w] = wave_max(v, w); d += nu_max[w] - wmax[(size_t)u * nws + w] + nv_max[w] - wmax[(size_t)v * nws + w]; }
    if (d < 0) {
      for (int w = 0; w < nws; ++w) { wmax[(size_t)u * nws + w] = nu_max[w]; wmax[(size_t)v * nws + w] = nv_max[w]; }
      unit_of[j] = v; unit_of[i] = u;
      auto& mu = members[u]; auto& mv = members[v];
      *find(mu.begin(), mu.end(), j) = i; *find(mv.begin(), mv.end(), i) = j;
      return true;
    }
    apply(j, u, +1); apply(i, u, -1); apply(i, v, +1); apply(j, v, -1);
    return false;
  }
  vector<int> perm() const { vector<int> out; out.reserve(P); for (int u = 0; u < nu; ++u) { vector<int> m = members[u]; sort(m.begin(), m.end()); for (int j : m) out.push_back(j); } return out; }
};

int main(int argc, char** argv) {
  FILE* f = fopen("/tmp/lab/cfg3_vis.bin", "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); P = (int)(sz / K); vis.resize(sz); if (fread(vis.data(), 1, sz, f) != (size_t)sz) return 1; fclose(f);
  vector<int> id(P); iota(id.begin(), id.end(), 0);
  printf("file order:\n"); evaluate(id, true);
  auto t0 = chrono::steady_clock::now();
  vector<int> g = greedy(30);
  printf("greedy D=30 (%.2f s):\n", chrono::duration<double>(chrono::steady_clock::now() - t0).count()); evaluate(g, true);
  vector<int> gp = parity_balance(g);
  printf("greedy + parity balance inside the chunks:\n"); evaluate(gp, true);
  const long trials = argc > 1 ? atol(argv[1]) : 200000;
  Refine R; R.init(g);
  printf("refine: objective %lld wave-trips\n", R.total());
  uint64_t st = 12345; long acc = 0;
  auto t1 = chrono::steady_clock::now();
  for (long t = 0; t < trials; ++t) {
    // a wave slot at random, its busiest lane, a point of the chunk that both cameras see, and a partner from another chunk that lacks the pair
    const int u = (int)(mix(st) % (uint64_t)R.nu), w = (int)(mix(st) % (uint64_t)R.nws);
    const uint16_t* c = &R.cnt[(size_t)u * C * C]; int best = -1, m = -1;
    for (int pr : R.wl[w]) if ((int)c[pr] > m) { m = c[pr]; best = pr; }
    const int a = best / C, b = best % C;
    const auto& mu = R.members[u]; int j = -1;
    for (int tries = 0; tries < 64 && j < 0; ++tries) { const int cand = mu[(size_t)(mix(st) % (uint64_t)mu.size())]; const uint8_t* cj = &vis[(size_t)cand * K]; bool ha = false, hb = false; for (int x = 0; x < K; ++x) { ha |= cj[x] == a; hb |= cj[x] == b; } if (ha && hb) j = cand; }
    if (j < 0) continue;
    const int v = (int)(mix(st) % (uint64_t)R.nu); if (v == u) continue;
    const auto& mv = R.members[v]; int i = -1;
    for (int tries = 0; tries < 16 && i < 0; ++tries) { const int cand = mv[(size_t)(mix(st) % (uint64_t)mv.size())]; const uint8_t* ci = &vis[(size_t)cand * K]; bool ha = false, hb = false; for (int x = 0; x < K; ++x) { ha |= ci[x] == a; hb |= ci[x] == b; } if (!(ha && hb)) i = cand; }
    if (i < 0) continue;
    acc += R.try_swap(j, i) ? 1 : 0;
    if ((t + 1) % 50000 == 0) printf("  %ld trials, %ld accepted, objective %lld, %.1f s\n", t + 1, acc, R.total(), chrono::duration<double>(chrono::steady_clock::now() - t1).count());
  }
  vector<int> r = parity_balance(R.perm());
  printf("greedy + %ld targeted swaps + parity balance:\n", trials); evaluate(r, true);
  // ---- quadratic potential: sum over chunks and pairs of count^2, random swaps, strict descent (a swap costs ~800 operations)
  {
    const long qtrials = argc > 2 ? atol(argv[2]) : 3000000;
    Refine Q; Q.init(g);
    auto t2 = chrono::steady_clock::now();
    uint64_t s3 = 777; long qacc = 0;
    vector<char> has(C);
    for (long t = 0; t < qtrials; ++t) {
      const int u = (int)(mix(s3) % (uint64_t)Q.nu), v = (int)(mix(s3) % (uint64_t)Q.nu); if (u == v) continue;
      auto& mu = Q.members[u]; auto& mv = Q.members[v];
      const size_t ju = (size_t)(mix(s3) % (uint64_t)mu.size()), iv = (size_t)(mix(s3) % (uint64_t)mv.size());
      const int j = mu[ju], i = mv[iv];
      const uint8_t* cj = &vis[(size_t)j * K]; const uint8_t* ci = &vis[(size_t)i * K];
      const uint16_t* cu = &Q.cnt[(size_t)u * C * C]; const uint16_t* cv = &Q.cnt[(size_t)v * C * C];
      // delta of sum c^2: pairs of j leave u and enter v, pairs of i leave v and enter u (pairs both points have cancel)
      long d = 0;
      fill(has.begin(), has.end(), 0); for (int x = 0; x < K; ++x) has[ci[x]] = 1;
      for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) { const int pr = cj[x] * C + cj[y]; if (has[cj[x]] && has[cj[y]]) continue; d += 2 * ((long)cv[pr] - (long)cu[pr] + 1); }
      fill(has.begin(), has.end(), 0); for (int x = 0; x < K; ++x) has[cj[x]] = 1;
      for (int x = 0; x < K; ++x) for (int y = x + 1; y < K; ++y) { const int pr = ci[x] * C + ci[y]; if (has[ci[x]] && has[ci[y]]) continue; d += 2 * ((long)cu[pr] - (long)cv[pr] + 1); }
      if (d < 0) { Q.apply(j, u, -1); Q.apply(i, u, +1); Q.apply(i, v, -1); Q.apply(j, v, +1); mu[ju] = i; mv[iv] = j; Q.unit_of[j] = v; Q.unit_of[i] = u; ++qacc; }
      if ((t + 1) % 1000000 == 0) printf("  quadratic: %ld trials, %ld accepted, %.1f s\n", t + 1, qacc, chrono::duration<double>(chrono::steady_clock::now() - t2).count());
    }
    vector<int> q = parity_balance(Q.perm());
    printf("greedy + %ld quadratic swaps + parity balance:\n", qtrials); evaluate(q, true);
  }
  return 0;
}
