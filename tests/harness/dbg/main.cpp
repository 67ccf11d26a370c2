#include <cstdio>
#include <vector>
#include "../../../include/rsba.h"
int main() {
  const int C = 2, P = 8; const int N = 16;
  std::vector<int> cam(N), pt(N); std::vector<double> obs(2 * N, 300.0), par(6 * C + 3 * P, 0.1), intr(4 * C, 600.0);
  for (int i = 0; i < N; ++i) { cam[i] = i % 2; pt[i] = i / 2; obs[2 * i] = 300 + i; obs[2 * i + 1] = 200 + 2 * i; }
  for (int c = 0; c < C; ++c) { par[6 * c + 5] = 3.0; intr[4 * c + 2] = 320; intr[4 * c + 3] = 240; }
  for (int j = 0; j < P; ++j) { par[6 * C + 3 * j] = 0.1 * j - 0.4; par[6 * C + 3 * j + 1] = 0.05 * j; par[6 * C + 3 * j + 2] = 0.2; }
  rsba_problem* p = nullptr;
  printf("create %d\n", rsba_problem_create_points(C, P, N, cam.data(), pt.data(), obs.data(), par.data(), intr.data(), &p));
  rsba_options o; rsba_options_default(&o);
  double scal[8];
  int rc = rsba_points_linearize_and_step(p, &o, 1e4, nullptr, nullptr, nullptr, scal);
  printf("step rc=%d cost=%g ok=%g\n", rc, scal[0], scal[3]);
  return 0;
}
