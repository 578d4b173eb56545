#include "convert.h"

#include <cmath>

using namespace svo_host;

Mat44f convert::R_t_to_Tcw(const Mat33f& R, const Vec3f& t) {
  Mat44f T = eye4();
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.at(r, c) = R.at(r, c); T.at(r, 3) = t.at(r); }
  return T;
}
Mat44f convert::R_t_to_Twc(const Mat33f& Rcw, const Vec3f& tcw) {
  Mat44f T = eye4();
  for (int r = 0; r < 3; ++r) {
    double acc = 0;
    for (int c = 0; c < 3; ++c) { T.at(r, c) = Rcw.at(c, r); acc += (double)Rcw.at(c, r) * (double)tcw.at(c); }
    T.at(r, 3) = (float)(-acc);
  }
  return T;
}
std::vector<float> convert::toQuaternion(const Mat33f& M) {
  double m[9];
  for (int i = 0; i < 9; ++i) m[i] = M.m[i];
  double q[4];
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
  return {(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
}
