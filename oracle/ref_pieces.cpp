// ref_pieces.cpp - our own extern "C" driver around the reference's OWN objects, built
// only in the container that has /root/reference (oracle/Makefile target `ref`).
// Test infrastructure: used by tests/test_oracle_vs_ref.py to pin the oracle's warp
// model, blob rasteriser and small helpers, and by tests/golden/make_golden.py.
#include "model_class.hpp"    // /root/reference/model_class.hpp
#include "parameters.hpp"     // /root/reference/parameters.hpp
#include "polygon_class.h"    // /root/reference/polygon_class.h

#include <cstring>
#include <vector>

extern "C" {

// ModelClass_*::compute_model on n samples. model: fittingModelEnum value.
// def_xy[2n], dTxydp[2*P*n] laid out as the reference does (model_class.cpp:150-202).
int ref_compute_model(int model, int n, const float *xy, const float *p, float cx, float cy,
                      float *def_xy, float *dTxydp) {
  ModelClass *m = ModelClass::new_ModelClass((fittingModelEnum)model, 1);
  if (!m)
    return -1;
  std::vector<float> xyc(xy, xy + 2 * n), pc(p, p + 6);
  m[0].set_points(n, xyc.data(), def_xy, pc.data(), cx, cy, dTxydp);
  m[0].compute_model();
  int np = ModelClass::get_number_of_model_parameters((fittingModelEnum)model);
  delete[] m;
  return np;
}

// polygonBlob_class: returns the number of inside points, -1 on error_bad_domain.
long ref_blob_points(const float *contour, int nv, float *xy, long cap) {
  v_points c(nv);
  for (int i = 0; i < nv; ++i)
    c[i] = std::make_pair(contour[2 * i], contour[2 * i + 1]);
  polygonBlob_class poly(c);
  if (poly.getError())
    return -1;
  v_points pts = poly.getInsidePoints();
  long n = (long)pts.size();
  for (long i = 0; i < n && i < cap; ++i) {
    xy[2 * i] = pts[i].first;
    xy[2 * i + 1] = pts[i].second;
  }
  return n;
}

float ref_best_rotation(const float *p) {
  float q[6];
  std::memcpy(q, p, sizeof(q));
  return best_rotation_UVUxUyVxVy(q);
}

float ref_blob_center(const float *contour, int nv, int axis) {
  v_points c(nv);
  for (int i = 0; i < nv; ++i)
    c[i] = std::make_pair(contour[2 * i], contour[2 * i + 1]);
  return axis == 0 ? makeblobXc(c) : makeblobYc(c);
}
}
