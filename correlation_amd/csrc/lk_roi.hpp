// lk_roi.hpp - ROI -> level-0 sample lists on the host (one-off per sequence, frame 0).
//
// The sample ORDER fixes the float summation order of everything downstream, so each
// generator keeps the CPU engine's order and predicates (not the CUDA path's, which uses
// different ones - SURVEY.md section 8a row a13):
//   rectangular  manager_class.cpp:276-310 (sector geometry), :1596-1614 (x outer, y inner)
//   annular      manager_class.cpp:816-940, single OpenMP thread
//   blob         polygon_class.cpp:224-429 (ear clipping + per-triangle scan fill)
// plus the per-level decimation and centres of pyramid_class.cpp:289-362.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace lkroi {

struct RectGrid {
  int xdim, ydim;          // half sizes
  std::vector<int> cx, cy; // hs and vs centre coordinates
};

// manager_class.cpp:276-310
inline RectGrid rect_grid(float x_begin, float y_begin, float x_end, float y_end, int hs, int vs) {
  RectGrid g;
  int x1 = (int)x_end, x0 = (int)x_begin, y1 = (int)y_end, y0 = (int)y_begin;
  g.xdim = (std::abs(x1 - x0) / hs - 1) / 2;
  g.ydim = (std::abs(y1 - y0) / vs - 1) / 2;
  float fxdim = (std::fabs(x_end - x_begin) / (float)hs - 1.f) / 2.f;
  float fydim = (std::fabs(y_end - y_begin) / (float)vs - 1.f) / 2.f;
  g.cx.resize(hs);
  g.cy.resize(vs);
  for (int i = 0; i < hs; ++i)
    g.cx[i] = (int)(0.5f + x_begin + fxdim + (2.f * fxdim + 1.f) * (float)i);
  for (int j = 0; j < vs; ++j)
    g.cy[j] = (int)(0.5f + y_begin + fydim + (2.f * fydim + 1.f) * (float)j);
  return g;
}

// manager_class.cpp:1596-1614: appends (x1-x0+1)*(y1-y0+1) samples
inline void rect_points(int x0, int y0, int x1, int y1, std::vector<float> &xy) {
  for (int ix = x0; ix <= x1; ++ix)
    for (int iy = y0; iy <= y1; ++iy) {
      xy.push_back((float)ix);
      xy.push_back((float)iy);
    }
}

// manager_class.cpp:816-940.  The geometry of one annular sector as the scan uses it: bounding box
// from the four corners (outer ones x 1.2, the reference's "cheap sag" margin) and the arc midpoint,
// truncated to int (:862-895), squared radii, corners for the wedge test (:907-918).  Shared by the
// host scan below and the device mask (lk_roi_tile_kernel) so that both test the same floats.
struct AnnularGeometry {
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  float cx = 0, cy = 0, ri2 = 0, ro2 = 0, ro = 0; // (ro: outer radius - every kept sample is closer to the centre)
  float q00x = 0, q01x = 0, q10x = 0, q11x = 0, q00y = 0, q01y = 0, q10y = 0, q11y = 0;
  int as = 1;
};
inline bool annular_geometry(float r, float dr, float a, float da, float cx, float cy, int as, AnnularGeometry &g) {
  if (as <= 0)
    return false;
  g = AnnularGeometry();
  g.cx = cx, g.cy = cy, g.as = as;
  const float ro = r + dr;
  if (as == 1) {
    g.x0 = (int)(cx - ro);
    g.x1 = (int)(cx + ro);
    g.y0 = (int)(cy - ro);
    g.y1 = (int)(cy + ro);
  } else {
    float s0 = (float)std::sin((double)a), c0 = (float)std::cos((double)a);
    float s1 = (float)std::sin((double)(a + da)), c1 = (float)std::cos((double)(a + da));
    float s2 = (float)std::sin((double)(a + da / 2.f)), c2 = (float)std::cos((double)(a + da / 2.f));
    g.q00x = cx + r * c0;
    g.q01x = cx + r * c1;
    g.q10x = cx + ro * c0 * 1.2f; // the reference's "cheap sag" margin
    g.q11x = cx + ro * c1 * 1.2f;
    g.q00y = cy + r * s0;
    g.q01y = cy + r * s1;
    g.q10y = cy + ro * s0 * 1.2f;
    g.q11y = cy + ro * s1 * 1.2f;
    float ax = cx + ro * c2, ay = cy + ro * s2;
    auto mn = [](float u, float v) { return v < u ? v : u; };
    auto mx = [](float u, float v) { return u < v ? v : u; };
    g.x0 = (int)mn(ax, mn(mn(g.q00x, g.q01x), mn(g.q10x, g.q11x)));
    g.x1 = (int)mx(ax, mx(mx(g.q00x, g.q01x), mx(g.q10x, g.q11x)));
    g.y0 = (int)mn(ay, mn(mn(g.q00y, g.q01y), mn(g.q10y, g.q11y)));
    g.y1 = (int)mx(ay, mx(mx(g.q00y, g.q01y), mx(g.q10y, g.q11y)));
  }
  g.ro = ro;
  g.ro2 = ro * ro;
  g.ri2 = r * r;
  return true;
}

inline void annular_points(const AnnularGeometry &g, std::vector<float> &xy) {
  for (float fx = (float)g.x0; fx < g.x1; ++fx) {
    for (int j = g.y0; j < g.y1; ++j) {
      float ex = fx - g.cx, ey = j - g.cy;
      float r2 = ex * ex + ey * ey;
      if (r2 > g.ri2 && r2 < g.ro2) {
        float w1 = (g.q11x - fx) * (g.q01y - g.q11y) - (g.q11y - j) * (g.q01x - g.q11x);
        float w2 = (g.q00x - fx) * (g.q10y - g.q00y) - (g.q00y - j) * (g.q10x - g.q00x);
        if (w1 * w2 > 0 || g.as == 1) {
          xy.push_back(fx);
          xy.push_back((float)j);
        }
      }
    }
  }
}
inline bool annular_points(float r, float dr, float a, float da, float cx, float cy, int as,
                           std::vector<float> &xy) {
  AnnularGeometry g;
  if (!annular_geometry(r, dr, a, da, cx, cy, as, g))
    return false;
  annular_points(g, xy);
  return true;
}

// ---- blob: simple-polygon test, ear clipping, scan fill (polygon_class.cpp) ----------
class BlobPolygon {
  struct V {
    float x, y;
    bool ear;
    int nx, pv;
  };
  std::vector<V> v_;
  int head_ = 0, live_ = 0;

  float twice_area(int a, int b, int c) const {
    return (v_[b].x - v_[a].x) * (v_[c].y - v_[a].y) - (v_[c].x - v_[a].x) * (v_[b].y - v_[a].y);
  }
  bool left(int a, int b, int c) const { return twice_area(a, b, c) > 0.f; }
  bool left_on(int a, int b, int c) const { return twice_area(a, b, c) >= 0.f; }
  bool collinear(int a, int b, int c) const { return twice_area(a, b, c) == 0.f; }
  bool between(int a, int b, int c) const {
    if (!collinear(a, b, c))
      return false;
    if (v_[a].x != v_[b].x)
      return (v_[a].x <= v_[c].x && v_[c].x <= v_[b].x) || (v_[a].x >= v_[c].x && v_[c].x >= v_[b].x);
    return (v_[a].y <= v_[c].y && v_[c].y <= v_[b].y) || (v_[a].y >= v_[c].y && v_[c].y >= v_[b].y);
  }
  bool crosses_properly(int a, int b, int c, int d) const {
    if (collinear(a, b, c) || collinear(a, b, d) || collinear(b, d, a) || collinear(c, d, b))
      return false;
    return (!left(a, b, c) ^ !left(a, b, d)) && (!left(c, d, a) ^ !left(c, d, b));
  }
  bool intersects(int a, int b, int c, int d) const {
    return crosses_properly(a, b, c, d) || between(a, b, c) || between(a, b, d) ||
           between(c, d, a) || between(c, d, b);
  }
  bool clear_of_edges(int a, int b) const {
    int c = head_;
    do {
      int c1 = v_[c].nx;
      if (c != a && c1 != a && c != b && c1 != b && intersects(a, b, c, c1))
        return false;
      c = c1;
    } while (c != head_);
    return true;
  }
  bool in_cone(int a, int b) const {
    int a1 = v_[a].nx, a0 = v_[a].pv;
    if (left_on(a, a1, a0))
      return left(a, b, a0) && left(b, a, a1);
    return !(left_on(a, b, a1) && left_on(b, a, a0));
  }
  bool diagonal(int a, int b) const { return in_cone(a, b) && in_cone(b, a) && clear_of_edges(a, b); }
  float signed_area2() const {
    float sum = 0.f;
    int a = v_[head_].nx;
    do {
      sum += twice_area(head_, a, v_[a].nx);
      a = v_[a].nx;
    } while (v_[a].nx != head_);
    return sum;
  }
  bool simple() const { // polygon_class.cpp:198-222
    if (live_ < 4)
      return true;
    int ol = head_;
    do {
      int orr = v_[ol].nx, il = v_[orr].nx;
      do {
        int ir = v_[il].nx;
        if (intersects(ol, orr, il, ir))
          return false;
        il = ir;
      } while (il != head_ && il != v_[ol].pv);
      ol = orr;
    } while (ol != v_[v_[head_].pv].pv);
    return true;
  }

  static bool edge_line(float xa, float ya, float xb, float yb, float &slope, float &icpt) {
    float den = yb - ya;
    if (den == 0)
      return false;
    slope = (xb - xa) / den;
    icpt = xa - slope * ya;
    return true;
  }
public:
  // one flat-sided half triangle as the scan fill walks it: rows j0 .. j1-1, pixels
  // ceil(ls*j + li) .. ceil(rs*j + ri) - 1 of each (polygon_class.cpp:349-403)
  struct Flat {
    float ls = 0, li = 0, rs = 0, ri = 0;
    int j0 = 0, j1 = 0;
  };

private:
  // flat-sided triangle: (x1,y1) and (x2,y2) share y; false: nothing to fill (:356-361)
  static bool flat_of(float x1, float y1, float x2, float y2, float x3, float y3, Flat &f) {
    int dy = (int)(std::floor((double)y3) - std::floor((double)y1));
    int dx = (int)(std::floor((double)x2) - std::floor((double)x1));
    if (dx == 0 || dy == 0)
      return false;
    float lx = dx > 0 ? x1 : x2, ly = dx > 0 ? y1 : y2;
    float rx = dx > 0 ? x2 : x1, ry = dx > 0 ? y2 : y1;
    f = Flat();
    edge_line(lx, ly, x3, y3, f.ls, f.li);
    edge_line(rx, ry, x3, y3, f.rs, f.ri);
    f.j0 = dy > 0 ? (int)std::ceil((double)y1) : (int)std::ceil((double)y3);
    f.j1 = dy > 0 ? (int)std::ceil((double)y3) : (int)std::ceil((double)y1);
    return f.j1 > f.j0;
  }
  static void fill_flat(const Flat &f, std::vector<float> &xy) {
    for (int j = f.j0; j < f.j1; ++j) {
      int i0 = (int)std::ceil(f.ls * (float)j + f.li), i1 = (int)std::ceil(f.rs * (float)j + f.ri);
      for (int i = i0; i < i1; ++i) {
        xy.push_back((float)i);
        xy.push_back((float)j);
      }
    }
  }
  // the (at most two) half triangles of one triangle, in fill order (:283-347)
  static int flats_of_triangle(const float *t, Flat out[2]) {
    const float x[3] = {t[0], t[2], t[4]}, y[3] = {t[1], t[3], t[5]};
    int hi, mid, lo;
    if (y[0] > y[1]) {
      if (y[1] > y[2]) { hi = 0; mid = 1; lo = 2; }
      else if (y[2] > y[0]) { hi = 2; mid = 0; lo = 1; }
      else { hi = 0; mid = 2; lo = 1; }
    } else {
      if (y[0] > y[2]) { hi = 1; mid = 0; lo = 2; }
      else if (y[2] > y[1]) { hi = 2; mid = 1; lo = 0; }
      else { hi = 1; mid = 2; lo = 0; }
    }
    float slope, icpt;
    if (!edge_line(x[lo], y[lo], x[hi], y[hi], slope, icpt))
      return 0;
    float sy = y[mid], sx = slope * sy + icpt; // split point on the long edge
    int n = 0;
    if (flat_of(x[mid], y[mid], sx, sy, x[hi], y[hi], out[n]))
      ++n;
    if (flat_of(x[mid], y[mid], sx, sy, x[lo], y[lo], out[n]))
      ++n;
    return n;
  }
  static void fill_triangle(const float *t, std::vector<float> &xy) {
    Flat f[2];
    const int n = flats_of_triangle(t, f);
    for (int k = 0; k < n; ++k)
      fill_flat(f[k], xy);
  }
  // simple-polygon check, orientation, ear clipping: triangles (6 floats each) in clipping order;
  // false for a self-intersecting or degenerate contour
  static bool triangulate(const float *contour, int nv, std::vector<float> &tris) {
    if (nv < 3)
      return false;
    BlobPolygon P;
    P.v_.resize(nv);
    for (int i = 0; i < nv; ++i)
      P.v_[i] = V{contour[2 * i], contour[2 * i + 1], false, (i + 1) % nv, (i + nv - 1) % nv};
    P.live_ = nv;
    if (!P.simple())
      return false;
    if (P.signed_area2() < 0)
      for (auto &q : P.v_)
        std::swap(q.nx, q.pv);
    int w = P.head_;
    do {
      P.v_[w].ear = P.diagonal(P.v_[w].pv, P.v_[w].nx);
      w = P.v_[w].nx;
    } while (w != P.head_);
    auto emit = [&](int a, int b, int c) {
      const int id[3] = {a, b, c};
      for (int k = 0; k < 3; ++k) {
        tris.push_back(P.v_[id[k]].x);
        tris.push_back(P.v_[id[k]].y);
      }
    };
    while (P.live_ > 3) {
      int v2 = P.head_;
      bool clipped = false;
      do {
        if (P.v_[v2].ear) {
          int v3 = P.v_[v2].nx, v4 = P.v_[v3].nx, v1 = P.v_[v2].pv, v0 = P.v_[v1].pv;
          emit(v1, v2, v3);
          P.v_[v1].ear = P.diagonal(v0, v3);
          P.v_[v3].ear = P.diagonal(v1, v4);
          P.v_[v1].nx = v3;
          P.v_[v3].pv = v1;
          P.head_ = v3;
          --P.live_;
          clipped = true;
          break;
        }
        v2 = P.v_[v2].nx;
      } while (v2 != P.head_);
      if (!clipped)
        return false; // degenerate input (the reference would not terminate)
    }
    emit(P.v_[P.head_].pv, P.head_, P.v_[P.head_].nx);
    return true;
  }

public:
  // the half triangles of the whole blob in fill order (what the device mask walks);
  // false for a self-intersecting contour (error_bad_domain)
  static bool flat_triangles(const float *contour, int nv, std::vector<Flat> &out) {
    std::vector<float> tris;
    if (!triangulate(contour, nv, tris))
      return false;
    for (size_t t = 0; t < tris.size() / 6; ++t) {
      Flat f[2];
      const int n = flats_of_triangle(&tris[6 * t], f);
      out.insert(out.end(), f, f + n);
    }
    return true;
  }

  // the scan fill of those half triangles, in order (the host form of the device mask)
  static void fill(const std::vector<Flat> &flats, std::vector<float> &xy) {
    for (const Flat &f : flats)
      fill_flat(f, xy);
  }

  // returns false for a self-intersecting contour (error_bad_domain)
  static bool inside_points(const float *contour, int nv, std::vector<float> &xy) {
    std::vector<float> tris;
    if (!triangulate(contour, nv, tris))
      return false;
    // Scan fill, triangle by triangle in clipping order.  Large blobs: the triangles are filled
    // by a few threads into lists of their own and joined in that order (same samples, same order).
    const size_t n_tri = tris.size() / 6;
    double area2 = 0.0;
    for (size_t t = 0; t < n_tri; ++t) {
      const float *q = &tris[6 * t];
      area2 += std::fabs((double)(q[2] - q[0]) * (q[5] - q[1]) - (double)(q[4] - q[0]) * (q[3] - q[1]));
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t workers = std::min<size_t>({n_tri, 16, hw ? hw : 1});
    if (area2 < 2.0 * 262144.0 || workers < 2) { // fewer than ~2.6e5 samples: not worth the threads
      for (size_t t = 0; t < n_tri; ++t)
        fill_triangle(&tris[6 * t], xy);
      return true;
    }
    std::vector<std::vector<float>> part(n_tri);
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (size_t t = next.fetch_add(1); t < n_tri; t = next.fetch_add(1))
        fill_triangle(&tris[6 * t], part[t]);
    };
    std::vector<std::thread> th;
    for (size_t w = 1; w < workers; ++w)
      th.emplace_back(work);
    work();
    for (std::thread &x : th)
      x.join();
    size_t total = xy.size();
    for (const std::vector<float> &v : part)
      total += v.size();
    xy.reserve(total);
    for (const std::vector<float> &v : part)
      xy.insert(xy.end(), v.begin(), v.end());
    return true;
  }
};

// pyramid_class.cpp:301-322: keep samples whose rounded coordinates are divisible by
// 2^delta, scaled by 1/2^delta; appends to out, returns the number kept
inline int decimate(const float *xy, int n, int delta, std::vector<float> &out) {
  const int mag = 1 << delta;
  const float inv = 1.f / (float)mag;
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    int ix = (int)(xy[2 * i] + 0.5f), iy = (int)(xy[2 * i + 1] + 0.5f);
    if (ix % mag == 0 && iy % mag == 0) {
      out.push_back(xy[2 * i] * inv);
      out.push_back(xy[2 * i + 1] * inv);
      ++kept;
    }
  }
  return kept;
}

// pyramid_class.cpp:325-340: sequential float mean
inline void mean_center(const float *xy, int n, float &cx, float &cy) {
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < n; ++i) {
    sx += xy[2 * i];
    sy += xy[2 * i + 1];
  }
  cx = sx / (float)n;
  cy = sy / (float)n;
}

// the warp of one sample (ModelClass_*::compute_model, model_class.cpp:48-202), host side:
// same operation order as the device code, no contraction (-ffp-contract=off)
inline void warp_point(int model, float x, float y, float cx, float cy, const float *p, float &xd, float &yd) {
  const float dx = x - cx, dy = y - cy;
  switch (model) {
  case 0: // fm_U
    xd = x + p[0];
    yd = y;
    break;
  case 1: // fm_UV
    xd = x + p[0];
    yd = y + p[1];
    break;
  case 2: // fm_UVQ
    xd = x + p[0] - p[2] * dy;
    yd = y + p[1] + p[2] * dx;
    break;
  default: // fm_UVUxUyVxVy
    xd = x + p[0] + p[2] * dx + p[3] * dy;
    yd = y + p[1] + p[4] * dx + p[5] * dy;
    break;
  }
}

// the functor the manager moves Lagrangian sample lists with (manager_class.cpp:38-47)
inline float add_pair_round(float offset, float v) { return (float)(int)(offset + v + 0.5f); }

} // namespace lkroi
