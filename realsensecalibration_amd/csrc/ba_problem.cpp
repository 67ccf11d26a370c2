// File formats either side of the hot path, without OpenCV.
//   in : correspondence.txt   (writer /root/reference/Main_Calibration/correspondencer.cpp:207-282,
//                              reader /root/reference/Main_Calibration/bundle_adjustment.cpp:132-187)
//        two_cam_data.txt     (writer Test1_ReprojectionError/main.cpp:162-183,
//                              reader Test1_BundleAdjustment/bundle_adjustmenter.cpp:55-85)
//        <serial>.xml         (my_io.cpp:5-31)
//   out: Camera_Transform.xml, mat{i}.txt, point3d.txt (bundle_adjustment_manager.cpp:98-175)
#include "ba_problem.hpp"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>
#include <algorithm>

namespace rsba {

namespace {

// Whitespace token reader with the failure semantics of Test2's fscanfOrDie (-> RSBA_ERR_FORMAT).
class Tokens {
 public:
  explicit Tokens(const std::string& text) : s_(text), pos_(0) {}
  bool NextInt(int64_t* v) {
    const char* b; if (!Next(&b)) return false;
    char* e = nullptr; long long x = strtoll(b, &e, 10);
    if (e == b) return false;
    *v = x; pos_ = (size_t)(e - s_.c_str()); return true;
  }
  bool NextDouble(double* v) {
    const char* b; if (!Next(&b)) return false;
    char* e = nullptr; double x = strtod(b, &e);
    if (e == b) return false;
    *v = x; pos_ = (size_t)(e - s_.c_str()); return true;
  }
 private:
  bool Next(const char** b) {
    while (pos_ < s_.size() && isspace((unsigned char)s_[pos_])) ++pos_;
    if (pos_ >= s_.size()) return false;
    *b = s_.c_str() + pos_; return true;
  }
  std::string s_; size_t pos_;
};

bool ReadAll(const char* path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::stringstream ss; ss << f.rdbuf(); *out = ss.str(); return true;
}

}  // namespace

int LoadPointsFile(const char* path, const double* intrinsics4, rsba_problem** out) {
  if (!path || !intrinsics4 || !out) return RSBA_ERR_ARG;
  std::string text;
  if (!ReadAll(path, &text)) return RSBA_ERR_IO;
  // first line: `C P` (reference, one observation per point) or `C P N` (extended)
  const size_t eol = text.find('\n');
  int header_tokens = 0;
  { std::stringstream hs(text.substr(0, eol)); std::string t; while (hs >> t) ++header_tokens; }
  Tokens tk(text);
  int64_t C = 0, P = 0, N = 0;
  if (!tk.NextInt(&C) || !tk.NextInt(&P)) return RSBA_ERR_FORMAT;
  if (header_tokens >= 3) { if (!tk.NextInt(&N)) return RSBA_ERR_FORMAT; } else N = P;  // bundle_adjustmenter.cpp:64
  if (C <= 0 || P <= 0 || N <= 0) return RSBA_ERR_FORMAT;
  rsba_problem* p = new rsba_problem();
  p->model = RSBA_MODEL_POINTS; p->num_cameras = (int32_t)C; p->num_points = (int32_t)P; p->num_observations = N;
  p->camera_index.resize(N); p->point_index.resize(N); p->observations.resize(2 * N);
  p->parameters.resize(6 * C + 3 * P); p->intrinsics.resize(4 * C);
  for (int64_t i = 0; i < N; ++i) {
    int64_t c, j;
    if (!tk.NextInt(&c) || !tk.NextInt(&j) || !tk.NextDouble(&p->observations[2 * i]) || !tk.NextDouble(&p->observations[2 * i + 1]) ||
        c < 0 || c >= C || j < 0 || j >= P) { delete p; return RSBA_ERR_FORMAT; }
    p->camera_index[i] = (int32_t)c; p->point_index[i] = (int32_t)j;
  }
  for (size_t i = 0; i < p->parameters.size(); ++i) if (!tk.NextDouble(&p->parameters[i])) { delete p; return RSBA_ERR_FORMAT; }
  for (int64_t c = 0; c < C; ++c) memcpy(&p->intrinsics[4 * c], intrinsics4, 4 * sizeof(double));  // Test1 main.cpp:73-74
  *out = p;
  return RSBA_OK;
}

int LoadCorrespondence(const char* path, int32_t model, double marker_side, const double* intrinsics, rsba_problem** out) {
  if (!path || !intrinsics || !out) return RSBA_ERR_ARG;
  if (model != RSBA_MODEL_MARKER_CHAIN && model != RSBA_MODEL_MARKER_CHAIN_TEST2) return RSBA_ERR_ARG;
  std::string text;
  if (!ReadAll(path, &text)) return RSBA_ERR_IO;
  Tokens tk(text);
  int64_t T, C, M, N;
  if (!tk.NextInt(&T) || !tk.NextInt(&C) || !tk.NextInt(&M) || !tk.NextInt(&N)) return RSBA_ERR_FORMAT;
  if (T <= 0 || C <= 0 || M <= 0 || N <= 0) return RSBA_ERR_FORMAT;
  rsba_problem* p = new rsba_problem();
  p->model = model; p->marker_side = marker_side;
  p->num_times = (int32_t)T; p->num_cameras = (int32_t)C; p->num_markers = (int32_t)M; p->num_observations = N;
  p->obs_per_time_camera.resize(T * C);
  for (int64_t t = 0; t < T; ++t) {
    int64_t tmp;  // leading time id is read and discarded (bundle_adjustment.cpp:160-161)
    if (!tk.NextInt(&tmp)) { delete p; return RSBA_ERR_FORMAT; }
    for (int64_t c = 0; c < C; ++c) { int64_t v; if (!tk.NextInt(&v)) { delete p; return RSBA_ERR_FORMAT; } p->obs_per_time_camera[t * C + c] = (int32_t)v; }
  }
  p->time_index.resize(N); p->camera_index.resize(N); p->marker_index.resize(N); p->observations.resize(8 * N);
  for (int64_t i = 0; i < N; ++i) {
    int64_t t, c, m;
    if (!tk.NextInt(&t) || !tk.NextInt(&c) || !tk.NextInt(&m) || t < 0 || t >= T || c < 0 || c >= C || m < 0 || m >= M) { delete p; return RSBA_ERR_FORMAT; }
    p->time_index[i] = (int32_t)t; p->camera_index[i] = (int32_t)c; p->marker_index[i] = (int32_t)m;
    for (int j = 0; j < 8; ++j) if (!tk.NextDouble(&p->observations[8 * i + j])) { delete p; return RSBA_ERR_FORMAT; }
  }
  p->parameters.resize(6 * (C + T + M));
  for (size_t i = 0; i < p->parameters.size(); ++i) if (!tk.NextDouble(&p->parameters[i])) { delete p; return RSBA_ERR_FORMAT; }
  p->intrinsics.assign(intrinsics, intrinsics + 4 * C);
  *out = p;
  return RSBA_OK;
}

// Minimal reader for the one node the reference reads: <intrinsics type_id="opencv-matrix"> 3x3 doubles.
int ReadIntrinsicsXml(const char* path, double* out4) {
  if (!path || !out4) return RSBA_ERR_ARG;
  std::string text;
  if (!ReadAll(path, &text)) return RSBA_ERR_IO;
  size_t a = text.find("<intrinsics");
  if (a == std::string::npos) return RSBA_ERR_FORMAT;
  size_t d = text.find("<data>", a);
  size_t e = text.find("</data>", a);
  if (d == std::string::npos || e == std::string::npos || e < d) return RSBA_ERR_FORMAT;
  Tokens tk(text.substr(d + 6, e - d - 6));
  double k[9];
  for (int i = 0; i < 9; ++i) if (!tk.NextDouble(&k[i])) return RSBA_ERR_FORMAT;
  out4[0] = k[0]; out4[1] = k[4]; out4[2] = k[2]; out4[3] = k[5];  // bundle_adjustment.h:66-69
  return RSBA_OK;
}

// One <name type_id="opencv-matrix"> node of an OpenCV FileStorage XML: rows, cols and the doubles of <data>.
static int ReadXmlMatrix(const std::string& text, const std::string& name, int* rows, int* cols, std::vector<double>* data) {
  const size_t a = text.find("<" + name + " ");
  if (a == std::string::npos) return RSBA_ERR_FORMAT;
  const size_t end = text.find("</" + name + ">", a);
  if (end == std::string::npos) return RSBA_ERR_FORMAT;
  auto field = [&](const char* open, const char* close, std::string* out) {
    const size_t b = text.find(open, a), e = text.find(close, a);
    if (b == std::string::npos || e == std::string::npos || e < b || e > end) return false;
    *out = text.substr(b + strlen(open), e - b - strlen(open));
    return true;
  };
  std::string r, c, d;
  if (!field("<rows>", "</rows>", &r) || !field("<cols>", "</cols>", &c) || !field("<data>", "</data>", &d)) return RSBA_ERR_FORMAT;
  *rows = atoi(r.c_str()); *cols = atoi(c.c_str());
  if (*rows <= 0 || *cols <= 0 || *rows * *cols > 64) return RSBA_ERR_FORMAT;
  Tokens tk(d);
  data->resize((size_t)*rows * *cols);
  for (double& v : *data) if (!tk.NextDouble(&v)) return RSBA_ERR_FORMAT;
  return RSBA_OK;
}

// Rotation matrix (row-major) -> angle-axis, the inverse of Rodrigues above (what cv::projectPoints does with the 3x3
// "rvec" reprojection_check.cpp:65-69 hands it).
void RotationToAngleAxis(const double R[9], double aa[3]) {
  const double tr = R[0] + R[4] + R[8];
  double c = 0.5 * (tr - 1.0);
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  const double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];   // 2 sin(theta) * axis
  const double s2 = std::sqrt(ax * ax + ay * ay + az * az);
  const double theta = std::atan2(0.5 * s2, c);
  if (s2 < 1e-12) {
    if (c > 0.0) { aa[0] = 0.5 * ax; aa[1] = 0.5 * ay; aa[2] = 0.5 * az; return; }
    // theta ~ pi: axis from the diagonal of (R + I) / 2
    double v[3] = {std::sqrt(std::max(0.0, 0.5 * (R[0] + 1.0))), std::sqrt(std::max(0.0, 0.5 * (R[4] + 1.0))), std::sqrt(std::max(0.0, 0.5 * (R[8] + 1.0)))};
    if (R[1] + R[3] < 0.0) v[1] = -v[1];
    if (R[2] + R[6] < 0.0) v[2] = -v[2];
    for (int k = 0; k < 3; ++k) aa[k] = theta * v[k];
    return;
  }
  const double f = theta / s2;
  aa[0] = f * ax; aa[1] = f * ay; aa[2] = f * az;
}

// ReprojectionCheck::Reproject's inputs (reprojection_check.cpp:5-66) as a point-model problem: cameras from
// Camera_Transform.xml (R{i} 3x3 as Main writes it, or the 3x1 rvec of the Test2 variant, and t{i}), one point per
// row of point3d.txt (the 6-digit 3D corners, `4N T C` header and T count rows first), one observation per point: the
// corner of correspondence.txt rounded to float32 as the reference holds it (Point2f), camera from the same row.
int LoadReprojectionCheck(const char* correspondence_txt, const char* point3d_txt, const char* camera_transform_xml,
                          const double* intrinsics /* 4 per camera */, rsba_problem** out) {
  if (!correspondence_txt || !point3d_txt || !camera_transform_xml || !intrinsics || !out) return RSBA_ERR_ARG;
  rsba_problem* corr = nullptr;
  // the wiring model does not matter here: only the observation rows are used
  std::string text;
  if (!ReadAll(correspondence_txt, &text)) return RSBA_ERR_IO;
  int64_t T = 0, C = 0, M = 0, N = 0;
  { Tokens tk(text); if (!tk.NextInt(&T) || !tk.NextInt(&C) || !tk.NextInt(&M) || !tk.NextInt(&N)) return RSBA_ERR_FORMAT; }
  if (C <= 0 || N <= 0) return RSBA_ERR_FORMAT;
  int rc = LoadCorrespondence(correspondence_txt, RSBA_MODEL_MARKER_CHAIN, 0.0, intrinsics, &corr);
  if (rc != RSBA_OK) return rc;
  std::string p3, xml;
  if (!ReadAll(point3d_txt, &p3) || !ReadAll(camera_transform_xml, &xml)) { delete corr; return RSBA_ERR_IO; }
  Tokens tk(p3);
  int64_t npts = 0, t3 = 0, c3 = 0;
  if (!tk.NextInt(&npts) || !tk.NextInt(&t3) || !tk.NextInt(&c3) || npts != 4 * N || c3 != C) { delete corr; return RSBA_ERR_FORMAT; }
  for (int64_t t = 0; t < t3; ++t) for (int64_t k = 0; k <= C; ++k) { int64_t v; if (!tk.NextInt(&v)) { delete corr; return RSBA_ERR_FORMAT; } }
  rsba_problem* p = new rsba_problem();
  p->model = RSBA_MODEL_POINTS;
  p->num_cameras = (int32_t)C; p->num_points = (int32_t)npts; p->num_observations = npts;
  p->parameters.assign(6 * C + 3 * npts, 0.0);
  for (int64_t c = 0; c < C; ++c) {
    int rr, cc; std::vector<double> Rm, tv;
    if (ReadXmlMatrix(xml, "R" + std::to_string(c), &rr, &cc, &Rm) != RSBA_OK || ReadXmlMatrix(xml, "t" + std::to_string(c), &rr, &cc, &tv) != RSBA_OK ||
        tv.size() != 3 || (Rm.size() != 9 && Rm.size() != 3)) { delete corr; delete p; return RSBA_ERR_FORMAT; }
    double aa[3];
    if (Rm.size() == 9) RotationToAngleAxis(Rm.data(), aa); else { aa[0] = Rm[0]; aa[1] = Rm[1]; aa[2] = Rm[2]; }
    for (int k = 0; k < 3; ++k) { p->parameters[6 * c + k] = aa[k]; p->parameters[6 * c + 3 + k] = tv[k]; }
  }
  for (int64_t j = 0; j < 3 * npts; ++j) if (!tk.NextDouble(&p->parameters[6 * C + j])) { delete corr; delete p; return RSBA_ERR_FORMAT; }
  p->camera_index.resize(npts); p->point_index.resize(npts); p->observations.resize(2 * npts);
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < 4; ++k) {
      const int64_t j = 4 * i + k;
      p->camera_index[j] = corr->camera_index[i]; p->point_index[j] = (int32_t)j;
      p->observations[2 * j] = (double)(float)corr->observations[8 * i + 2 * k];          // Point2f
      p->observations[2 * j + 1] = (double)(float)corr->observations[8 * i + 2 * k + 1];
    }
  p->intrinsics.assign(intrinsics, intrinsics + 4 * C);
  delete corr;
  *out = p;
  return RSBA_OK;
}

void Rodrigues(const double rvec[3], double R[9]) {
  const double theta = std::sqrt(rvec[0] * rvec[0] + rvec[1] * rvec[1] + rvec[2] * rvec[2]);
  if (theta < DBL_EPSILON) { R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1; return; }
  const double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c;
  const double x = rvec[0] / theta, y = rvec[1] / theta, z = rvec[2] / theta;
  R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
  R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
  R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

void AngleAxisRotatePointHost(const double aa[3], const double pt[3], double out[3]) {
  const double t2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  double r[3];
  if (t2 > DBL_EPSILON) {
    const double th = std::sqrt(t2), c = std::cos(th), s = std::sin(th), it = 1.0 / th;
    const double w[3] = {aa[0] * it, aa[1] * it, aa[2] * it};
    const double x[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - c);
    for (int k = 0; k < 3; ++k) r[k] = pt[k] * c + x[k] * s + w[k] * tmp;
  } else {
    r[0] = pt[0] + aa[1] * pt[2] - aa[2] * pt[1];
    r[1] = pt[1] + aa[2] * pt[0] - aa[0] * pt[2];
    r[2] = pt[2] + aa[0] * pt[1] - aa[1] * pt[0];
  }
  out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

// BALProblem::getPoint3dCoordinates (bundle_adjustment.cpp:89-130)
void MarkerCorners3d(const rsba_problem& p, double* out) {
  const double h = p.marker_side / 2;
  const double corner[4][3] = {{-h, h, 0}, {h, h, 0}, {h, -h, 0}, {-h, -h, 0}};
  for (int64_t i = 0; i < p.num_observations; ++i) {
    const double* tim = &p.parameters[6 * p.time_block(i)];
    const double* mar = &p.parameters[6 * p.marker_block(i)];
    for (int j = 0; j < 4; ++j) {
      double q[3];
      AngleAxisRotatePointHost(mar, corner[j], q);
      q[0] += mar[3]; q[1] += mar[4]; q[2] += mar[5];
      AngleAxisRotatePointHost(tim, q, q);
      q[0] += tim[3]; q[1] += tim[4]; q[2] += tim[5];
      out[12 * i + 3 * j] = q[0]; out[12 * i + 3 * j + 1] = q[1]; out[12 * i + 3 * j + 2] = q[2];
    }
  }
}

namespace {
void WriteMatNode(FILE* f, const char* name, int idx, int rows, int cols, const double* v) {
  fprintf(f, "<%s%d type_id=\"opencv-matrix\">\n  <rows>%d</rows>\n  <cols>%d</cols>\n  <dt>d</dt>\n  <data>\n   ", name, idx, rows, cols);
  for (int i = 0; i < rows * cols; ++i) {
    if (v[i] == std::floor(v[i]) && std::fabs(v[i]) < 1e15) fprintf(f, " %.0f.", v[i]);
    else fprintf(f, " %.16e", v[i]);
    if (i % 3 == 2 && i + 1 < rows * cols) fprintf(f, "\n   ");
  }
  fprintf(f, "</data></%s%d>\n", name, idx);
}
}  // namespace

int WriteOutputs(const rsba_problem& p, const char* camera_transform_xml, const char* extrinsics_dir, const char* point3d_txt) {
  if (camera_transform_xml) {
    FILE* f = fopen(camera_transform_xml, "w");
    if (!f) return RSBA_ERR_IO;
    fprintf(f, "<?xml version=\"1.0\"?>\n<opencv_storage>\n");
    for (int i = 0; i < p.num_cameras; ++i) {
      const double* cam = &p.parameters[6 * i];
      if (p.model == RSBA_MODEL_MARKER_CHAIN_TEST2) {
        WriteMatNode(f, "R", i, 3, 1, cam);  // Test2 main.cpp:128 stores the rvec itself
      } else {
        double R[9]; Rodrigues(cam, R);      // bundle_adjustment_manager.cpp:118-121,130
        WriteMatNode(f, "R", i, 3, 3, R);
      }
      WriteMatNode(f, "t", i, 3, 1, cam + 3);
    }
    fprintf(f, "</opencv_storage>\n");
    fclose(f);
  }
  if (extrinsics_dir) {
    for (int i = 0; i < p.num_cameras; ++i) {
      const double* cam = &p.parameters[6 * i];
      double R[9]; Rodrigues(cam, R);
      const std::string path = std::string(extrinsics_dir) + "/mat" + std::to_string(i) + ".txt";
      FILE* f = fopen(path.c_str(), "w");
      if (!f) return RSBA_ERR_IO;
      for (int row = 0; row < 3; ++row) {  // [R^T | -R^T t], one value per line, default ostream precision
        const double rt[3] = {R[0 + row], R[3 + row], R[6 + row]};
        const double ti = -(rt[0] * cam[3] + rt[1] * cam[4] + rt[2] * cam[5]);
        fprintf(f, "%g\n%g\n%g\n%g\n", rt[0], rt[1], rt[2], ti);
      }
      fclose(f);
    }
  }
  if (point3d_txt) {
    if (!p.is_marker_chain()) return RSBA_ERR_UNSUPPORTED;
    std::vector<double> pts(12 * p.num_observations);
    MarkerCorners3d(p, pts.data());
    FILE* f = fopen(point3d_txt, "w");
    if (!f) return RSBA_ERR_IO;
    fprintf(f, "%lld %d %d\n", (long long)(4 * p.num_observations), p.num_times, p.num_cameras);
    for (int t = 0; t < p.num_times; ++t) {
      fprintf(f, "%d", t);
      for (int c = 0; c < p.num_cameras; ++c) fprintf(f, " %d", p.obs_per_time_camera[t * p.num_cameras + c] * 4);
      fprintf(f, "\n");
    }
    for (int64_t i = 0; i < 4 * p.num_observations; ++i) fprintf(f, "%g %g %g\n", pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    fclose(f);
  }
  return RSBA_OK;
}

}  // namespace rsba
