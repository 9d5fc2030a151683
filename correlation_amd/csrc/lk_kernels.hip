// lk_kernels.hip - hand-written CDNA4 (gfx950) kernels of the Lucas-Kanade engine.
//
//   lk_solve_kernel    one lane group owns one sector for its whole coarse-to-fine
//                      Levenberg-Marquardt solve: warp -> bicubic sample of the deformed image +
//                      gradient -> residual -> per-lane accumulation of the 21+6+1 sums -> DPP
//                      reduction -> 6x6 solve in registers -> accept/reject on the device.
//                      Flavours by template: GROUP = 16 / 32 / 64 lanes, 256 / 512 threads
//                      (+ teams of 512-thread workgroups for giant sectors); GROUP = 1, one lane per
//                      sector in the reference's summation order with the restated Eigen QR for
//                      starved pyramid levels; SAFE 16-lane = the finisher of that kernel's
//                      stragglers (ordered sums by row_newbcast, QR spread over the row).
//                      Reference-order mode (lk_set_reference_order): the SAFE 16- / 64-lane instances
//                      solve every level with evaluate_ordered (the CPU engine's summation order for any
//                      number_of_threads) + that QR - records byte-identical to the CPU class.
//                      Frame-pipelined instances (template parameter SEQ, lk_correlate_sequence_async): the same state
//                      machine over (frame, sector) tickets of a window of resident frames, drawn frame-major; a sector's
//                      parameters travel from frame to frame through a chain of 8-byte granules, its guess
//                      (manager_class.cpp:2677-2699) is formed in the kernel, a group whose sector's previous frame is
//                      not in yet waits without holding up the other sectors of its wavefront (PH_WAIT).
//                      Scheduling inside a wavefront: level alignment, solo (32 lanes) and adaptive
//                      width (16 lanes) - idle lanes join the sectors still being solved.  Kept sums:
//                      a rejected LM trip continues from the sums of the last accepted evaluation.
//                      Replaces kCorrelation + k_global_reduction + k_build_LS_problem_in_GPU0 +
//                      cuSOLVER potrf/potrs + kScale + kUpdateParameters and the host loop of
//                      CudaClass::correlate (cuda_class.cu:104-473, correlationKernel.cu,
//                      kernels.cu:12-103, cuda_solver.cu), following the CPU engine's semantics
//                      (correlation_class.cpp:349-640).
//   lk_pyramid2_kernel upload copy + pyramid levels 1 and 2 of one or two frames in one launch;
//   lk_pyramid_kernel  one further level; both with the CPU engine's arithmetic
//                      (pyramid_class.cpp:83-122), replacing k_pyramid_bw (kernels.cu:761).
//   lk_roi_*_kernel    annular / blob ROI masks with the CPU engine's predicates and sample order (tiles,
//                      count -> offsets -> fill), replacing cudaPolygon's thrust mask + remove_if
//                      (cuda_polygon.cuh:180-292); lk_mean_*: the sequential float mean centre of such
//                      lists evaluated in parallel, bit for bit (parity maps per binade).
//   lk_decimate_*, lk_rewarp_kernel   coarser-level lists and moved lists on the device (pyramid_class.cpp:289-323).
//   small utilities    level table upload, initial-guess policy, sample warping, stale iteration counts
//                      (reference-order mode), stand-alone evaluation / sampling / solve (known-answer entry points).
//
// This translation unit is compiled with -ffp-contract=off: the reference's x86-64 builds
// have no FMA, and per-sample values (warp, bicubic value/gradient, residual, H) are kept
// bit-identical to that arithmetic.  FMAs appear only where they are provably exact
// (the bicubic coefficient build, see bicubic_sample) or where only the summation order
// already differs from the reference (the A/b/chi accumulators).
#include "lk_device.hpp"
#include "lk_internal.hpp"

#include <float.h>
#include <stdlib.h>

#include <atomic>

// Smallest pivot ratio d_j / A_jj the fast flavour factors through (below it the sector goes to
// the SAFE kernel and the reference's QR).  1e-3 sent 0.4 % of config 4's solves there and, before
// that pass existed, cost its 1 % tail a factor of ten against the reference; root-free Cholesky
// in float32 is fine down to 1e-6.
#ifndef LK_FAST_PIVOT
#define LK_FAST_PIVOT 1e-6f
#endif

namespace {

constexpr int kWave = 64;

// pointers that are known to be global memory (loaded from a struct they would be
// generic and cost flat_load instead of global_load)
template <class T> using gptr = const __attribute__((address_space(1))) T *;
typedef float f32x2 __attribute__((ext_vector_type(2))); // same layout as float2

__host__ __device__ constexpr int n_params(int model) {
  return model == LK_FM_U ? 1 : model == LK_FM_UV ? 2 : model == LK_FM_UVQ ? 3 : 6;
}

// ------------------------------------------------------------------------------------
// wavefront reduction: 4 DPP steps inside each row of 16 lanes, then 4 readlanes.
// Every lane ends with the same bits (the tree is identical for all lanes).
// ------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(t);
}

// lane I of the own 16-lane row (row_newbcast, gfx90a and later)
template <int I> __device__ __forceinline__ float dpp_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xF, 0xF, false));
}

__device__ __forceinline__ uint32_t load_u32_unaligned(gptr<uint8_t> p) {
  uint32_t v;
  typedef uint32_t __attribute__((aligned(1))) u32_u;
  v = *(const __attribute__((address_space(1))) u32_u *)p;
  return v;
}

__device__ __forceinline__ float ub0(uint32_t v) { return (float)(v & 0xffu); }
__device__ __forceinline__ float ub1(uint32_t v) { return (float)((v >> 8) & 0xffu); }
__device__ __forceinline__ float ub2(uint32_t v) { return (float)((v >> 16) & 0xffu); }
__device__ __forceinline__ float ub3(uint32_t v) { return (float)(v >> 24); }

// ------------------------------------------------------------------------------------
// bicubic (interpolation_class.cpp:79-138, :243-336)
// ------------------------------------------------------------------------------------
// The reference builds, per deformed pixel, a 16-vector of values and central differences
// and multiplies it by a 16x16 integer matrix (:296-333).  That map factors as
//     a[jk][ik] = sum_r sum_c Cm[jk][r] * Cm[ik][c] * Pix[r][c]
// with Pix the 4x4 u8 window (rows iy-1..iy+2, cols ix-1..ix+2) and Cm the 1-D
// "4 pixels -> monomial coefficients on [1,2]" matrix below.  Every product and every
// partial sum is a multiple of 1/4 below 2^24/4 in magnitude, hence exactly representable
// in float32: the factored form, in any order and with FMAs, yields the reference's
// coefficients bit for bit (tests/test_parity_gpu.py::test_bicubic_coefficients_exact).
// c = Cm * p with Cm = [2 -3 3 -1; -4 9.5 -8 2.5; 2.5 -7 6.5 -2; -0.5 1.5 -1.5 0.5] in 11 operations
// instead of 16, through the identities the cubic itself provides (value and slope at dx = 1):
//   e = (p3 - p0) + 3 (p1 - p2),  c3 = e / 2,  c0 = p0 - e,
//   c1 + c2 = p1 - c0 - c3,       c1 + 2 c2 + 3 c3 = (p2 - p0) / 2.
// Exact like the matrix form: every intermediate is a multiple of 1/4 below 2^22 in magnitude.
__device__ __forceinline__ void cubic_1d(float p0, float p1, float p2, float p3, float &c0,
                                         float &c1, float &c2, float &c3) {
  const float e = __builtin_fmaf(3.0f, p1 - p2, p3 - p0);
  c3 = 0.5f * e;
  c0 = p0 - e;
  const float u = (p1 - c0) - c3;
  c2 = __builtin_fmaf(-3.0f, c3, __builtin_fmaf(0.5f, p2 - p0, -u));
  c1 = u - c2;
}

// the 4x4 window as four (unaligned) dwords, rows iy-1..iy+2, columns ix-1..ix+2
struct Window4 {
  uint32_t r0, r1, r2, r3;
};
__device__ __forceinline__ Window4 load_window(gptr<uint8_t> def, int cols, int ix, int iy) {
  gptr<uint8_t> base = def + (size_t)(iy - 1) * (size_t)cols + (size_t)(ix - 1);
  Window4 w;
  w.r0 = load_u32_unaligned(base);
  w.r1 = load_u32_unaligned(base + cols);
  w.r2 = load_u32_unaligned(base + 2 * (size_t)cols);
  w.r3 = load_u32_unaligned(base + 3 * (size_t)cols);
  return w;
}

// Value and gradient of the bicubic at (ix + dx - 1, iy + dy - 1), dx,dy in [1,2).
// The 16 coefficients are produced exactly (see above) by 4 + 4 one-dimensional transforms, and
// then W, dW/dx, dW/dy are accumulated in the reference's order (:94-126): three running sums,
// jk outer / ik inner, each term built left to right.
__device__ __forceinline__ void bicubic_window(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, float dx, float dy,
                                               float &W, float &Wx, float &Wy) {
  // t[r][k]: x-direction transform of image row r
  float t0[4], t1[4], t2[4], t3[4];
  cubic_1d(ub0(r0), ub1(r0), ub2(r0), ub3(r0), t0[0], t0[1], t0[2], t0[3]);
  cubic_1d(ub0(r1), ub1(r1), ub2(r1), ub3(r1), t1[0], t1[1], t1[2], t1[3]);
  cubic_1d(ub0(r2), ub1(r2), ub2(r2), ub3(r2), t2[0], t2[1], t2[2], t2[3]);
  cubic_1d(ub0(r3), ub1(r3), ub2(r3), ub3(r3), t3[0], t3[1], t3[2], t3[3]);
  // y-direction transform, column by column: a[jk][ik] (exact in any order)
  float a0[4], a1[4], a2[4], a3[4];
#pragma unroll
  for (int ik = 0; ik < 4; ++ik)
    cubic_1d(t0[ik], t1[ik], t2[ik], t3[ik], a0[ik], a1[ik], a2[ik], a3[ik]);
  const float px[4] = {1.f, dx, dx * dx, dx * dx * dx};
  const float py[4] = {1.f, dy, dy * dy, dy * dy * dy};
  W = 0.f;
  Wx = 0.f;
  Wy = 0.f;
#pragma unroll
  for (int jk = 0; jk < 4; ++jk) {
#pragma unroll
    for (int ik = 0; ik < 4; ++ik) {
      const float c = jk == 0 ? a0[ik] : jk == 1 ? a1[ik] : jk == 2 ? a2[ik] : a3[ik];
      W += c * py[jk] * px[ik];
      if (ik > 0)
        Wx += (float)ik * c * py[jk] * px[ik - 1];
      if (jk > 0)
        Wy += (float)jk * c * py[jk - 1] * px[ik];
    }
  }
}

__device__ __forceinline__ void bicubic_sample(gptr<uint8_t> def, int cols, int ix, int iy, float dx,
                                               float dy, float &W, float &Wx, float &Wy) {
  const Window4 w = load_window(def, cols, ix, iy);
  bicubic_window(w.r0, w.r1, w.r2, w.r3, dx, dy, W, Wx, Wy);
}

// Catmull-Rom weights of the four samples at -1, 0, 1, 2 for position t in [0, 1) and their
// derivatives (Horner form)
__device__ __forceinline__ void catmull_rom(float t, float (&w)[4], float (&g)[4]) {
  w[0] = t * __builtin_fmaf(t, __builtin_fmaf(t, -0.5f, 1.0f), -0.5f);
  w[1] = __builtin_fmaf(t * t, __builtin_fmaf(t, 1.5f, -2.5f), 1.0f);
  w[2] = t * __builtin_fmaf(t, __builtin_fmaf(t, -1.5f, 2.0f), 0.5f);
  w[3] = t * t * __builtin_fmaf(t, 0.5f, -0.5f);
  g[0] = __builtin_fmaf(t, __builtin_fmaf(t, -1.5f, 2.0f), -0.5f);
  g[1] = t * __builtin_fmaf(t, 4.5f, -5.0f);
  g[2] = __builtin_fmaf(t, __builtin_fmaf(t, -4.5f, 4.0f), 0.5f);
  g[3] = t * __builtin_fmaf(t, 1.5f, -1.0f);
}

// returns false when the sample leaves the image (error_interpolation_out_of_image)
template <int INTERP>
__device__ __forceinline__ bool sample_def(gptr<uint8_t> def, int rows, int cols, float xd,
                                           float yd, float &W, float &Wx, float &Wy) {
  if constexpr (INTERP == LK_IM_BICUBIC) {
    if (!(xd > 1.f && yd > 1.f && xd < (float)cols - 2.f && yd < (float)rows - 2.f))
      return false;
    int ix = (int)xd, iy = (int)yd;
    float dx = xd - (float)ix + 1.f, dy = yd - (float)iy + 1.f;
    bicubic_sample(def, cols, ix, iy, dx, dy, W, Wx, Wy);
    return true;
  } else if constexpr (INTERP == LK_IM_BICUBIC_SEPARABLE) {
    // The reference's bicubic patch (values + central differences on the cell's corners) is
    // the Catmull-Rom spline; evaluated here as two 1-D kernels instead of 16 coefficients and
    // monomials.  Same validity rule, same window; results equal to rounding, not bit for bit.
    if (!(xd > 1.f && yd > 1.f && xd < (float)cols - 2.f && yd < (float)rows - 2.f))
      return false;
    const int ix = (int)xd, iy = (int)yd;
    const float tx = xd - (float)ix, ty = yd - (float)iy;
    const Window4 w = load_window(def, cols, ix, iy);
    float wx[4], gx[4], wy[4], gy[4];
    catmull_rom(tx, wx, gx);
    catmull_rom(ty, wy, gy);
    const uint32_t r[4] = {w.r0, w.r1, w.r2, w.r3};
    float row_v[4], row_g[4]; // per image row: value and x-derivative along x
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p0 = ub0(r[j]), p1 = ub1(r[j]), p2 = ub2(r[j]), p3 = ub3(r[j]);
      row_v[j] = __builtin_fmaf(wx[3], p3, __builtin_fmaf(wx[2], p2, __builtin_fmaf(wx[1], p1, wx[0] * p0)));
      row_g[j] = __builtin_fmaf(gx[3], p3, __builtin_fmaf(gx[2], p2, __builtin_fmaf(gx[1], p1, gx[0] * p0)));
    }
    W = __builtin_fmaf(wy[3], row_v[3], __builtin_fmaf(wy[2], row_v[2], __builtin_fmaf(wy[1], row_v[1], wy[0] * row_v[0])));
    Wx = __builtin_fmaf(wy[3], row_g[3], __builtin_fmaf(wy[2], row_g[2], __builtin_fmaf(wy[1], row_g[1], wy[0] * row_g[0])));
    Wy = __builtin_fmaf(gy[3], row_v[3], __builtin_fmaf(gy[2], row_v[2], __builtin_fmaf(gy[1], row_v[1], gy[0] * row_v[0])));
    return true;
  } else if constexpr (INTERP == LK_IM_BILINEAR) { // :140-195, :338-374
    if (!(xd > 0.f && yd > 0.f && xd < (float)(cols - 1) && yd < (float)(rows - 1)))
      return false;
    int ix = (int)xd, iy = (int)yd;
    gptr<uint8_t> q = def + (size_t)iy * (size_t)cols + (size_t)ix;
    float w00 = (float)q[0], w10 = (float)q[1], w01 = (float)q[cols], w11 = (float)q[cols + 1];
    float a0 = w00, a1 = w10 - w00, a2 = w01 - w00, a3 = w11 - w10 - w01 + w00;
    float dx = xd - (float)ix, dy = yd - (float)iy;
    // jk outer / ik inner with px = {1,dx}, py = {1,dy}
    W = 0.f + a0;
    W += a1 * dx;
    Wx = 0.f + a1;
    W += a2 * dy;
    Wy = 0.f + a2;
    W += a3 * dy * dx;
    Wx += a3 * dy;
    Wy += a3 * dx;
    return true;
  } else { // nearest :197-226, :376-406
    if (!(xd > 0.f && yd > 0.f && xd < (float)(cols - 1) && yd < (float)(rows - 1)))
      return false;
    int ix = (int)(xd + 0.5f), iy = (int)(yd + 0.5f);
    gptr<uint8_t> q = def + (size_t)iy * (size_t)cols + (size_t)ix;
    float w00 = (float)q[0], w10 = (float)q[1], w01 = (float)q[cols];
    W = w00;
    Wx = w10 - w00;
    Wy = w01 - w00;
    return true;
  }
}

// ------------------------------------------------------------------------------------
// per-sample body: warp (model_class.cpp:48-202), sample, residual, H
// (interpolation_class.cpp:701-739).  H = dW/dx * dTx/dp + dW/dy * dTy/dp with the zero
// entries of dT/dp dropped (x*1 + y*0 == x exactly for finite y).
// ------------------------------------------------------------------------------------
template <int MODEL> struct Warp;
template <> struct Warp<LK_FM_U> {
  static __device__ __forceinline__ void apply(float x, float y, float, float, const float *p,
                                               float &xd, float &yd, float &, float &) {
    xd = x + p[0];
    yd = y;
  }
  static __device__ __forceinline__ void jac(float Wx, float, float, float, float *H) { H[0] = Wx; }
};
template <> struct Warp<LK_FM_UV> {
  static __device__ __forceinline__ void apply(float x, float y, float, float, const float *p,
                                               float &xd, float &yd, float &, float &) {
    xd = x + p[0];
    yd = y + p[1];
  }
  static __device__ __forceinline__ void jac(float Wx, float Wy, float, float, float *H) {
    H[0] = Wx;
    H[1] = Wy;
  }
};
template <> struct Warp<LK_FM_UVQ> {
  static __device__ __forceinline__ void apply(float x, float y, float cx, float cy,
                                               const float *p, float &xd, float &yd, float &dx,
                                               float &dy) {
    dx = x - cx;
    dy = y - cy;
    xd = x + p[0] - p[2] * dy;
    yd = y + p[1] + p[2] * dx;
  }
  static __device__ __forceinline__ void jac(float Wx, float Wy, float dx, float dy, float *H) {
    H[0] = Wx;
    H[1] = Wy;
    H[2] = Wx * (-dy) + Wy * dx;
  }
};
template <> struct Warp<LK_FM_UVUXUYVXVY> {
  static __device__ __forceinline__ void apply(float x, float y, float cx, float cy,
                                               const float *p, float &xd, float &yd, float &dx,
                                               float &dy) {
    dx = x - cx;
    dy = y - cy;
    xd = x + p[0] + p[2] * dx + p[3] * dy;
    yd = y + p[1] + p[4] * dx + p[5] * dy;
  }
  static __device__ __forceinline__ void jac(float Wx, float Wy, float dx, float dy, float *H) {
    H[0] = Wx;
    H[1] = Wy;
    H[2] = Wx * dx;
    H[3] = Wx * dy;
    H[4] = Wy * dx;
    H[5] = Wy * dy;
  }
};

template <int P> struct Sums { // upper triangle row-major, then b, then chi
  static constexpr int NA = P * (P + 1) / 2;
  static constexpr int N = NA + P + 1;
  float v[N];
};

// One evaluation of one sector at one level by its lane group
// (apply_model_and_interpolate, correlation_class.cpp:131-300).  A "group" is the set of
// lanes that owns one sector: a 16-lane DPP row (4 sectors per wavefront), a whole
// wavefront, or a whole workgroup.  On return every lane of the group holds the same
// totals; returns the error flag (some sample of the group's sector left the image).
struct LevelCtx { // what a lane needs to know about its sector at the current level
  gptr<uint8_t> und, def;
  gptr<f32x2> xy;  // explicit list, already offset to the sector's first sample
  int rx, ry, rw;  // implicit rectangle: first x, first y, width (rw == 0: explicit list)
  int n;
  int urows, ucols, drows, dcols;
  float cx, cy, scaling;
  float inv_w; // 1 / rw (implicit rectangles: sample index -> row), once per level instead of once per evaluation
};

// sum inside each 16-lane row: every lane of the row ends with the same bits
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v); // row_half_mirror
  v = dpp_add<0x140>(v); // row_mirror
  return v;
}

__device__ __forceinline__ float rows_sum(float v) { // after row16_sum: add the 4 rows
  float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

// A "team": several 512-thread workgroups share ONE giant sector (millions of samples).
// Every workgroup runs the identical state machine on identical totals, so the only
// exchange is one all-to-all of the 28 sums per evaluation through global memory:
// plain stores of the workgroup's sums -> every wave's vmcnt(0) -> workgroup barrier ->
// agent-scope release -> arrival counter; then one lane polls the counter (relaxed,
// s_sleep, bounded), agent-scope acquire, workgroup barrier, and 29 lanes add up the
// team's partial sums in a fixed order (deterministic).  Buffers alternate with the step
// parity; a workgroup cannot be two steps ahead of its team.  The host caps the grid at what is
// resident, and every other kernel that may hold CU slots drains without waiting on anybody, so
// the wait ends; should a workgroup still be missing after ~1 s (another process's kernels), the
// team is marked broken and rank 0 solves the sector alone (see the kernel).
struct TeamCtx {
  int w = 1, rank = 0, slot = 0, stride = 1; // stride: workgroups per slot in the partial-sum buffer
  uint32_t step = 0;
  float *partials = nullptr;
  uint32_t *arrivals = nullptr; // [n_sectors] arrival counters, then [n_sectors] "team is broken" flags
  int n_slots = 0;
  int fault = 0;                // test hook (LK_TEAM_FAULT): rank 1 never arrives at step `fault`
  bool timed_out = false;
};

// The all-to-all of a team (see TeamCtx): every workgroup publishes its totals (or, in reference-order mode, the sums of
// its thread chunk), waits for the others and adds all of them up in rank order, starting from zero - every workgroup
// of the team ends with the same bits.  Returns the team's error flag.
template <class SumsT>
__device__ __forceinline__ bool team_all_to_all(SumsT &S, bool any_bad, float *lds, TeamCtx *team) {
  const int team_w = team->w, team_rank = team->rank;
  ++team->step;
  float *mine = team->partials + (((size_t)team->slot * 2 + (team->step & 1u)) * (size_t)team->stride) * 32;
  __syncthreads(); // everybody has its totals out of lds
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < SumsT::N; ++i)
      mine[(size_t)team_rank * 32 + i] = S.v[i];
    mine[(size_t)team_rank * 32 + SumsT::N] = any_bad ? 1.f : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(team->fault && team->rank == 1 && (int)team->step == team->fault)) // (test hook: a workgroup goes missing)
      __hip_atomic_fetch_add(team->arrivals + team->slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t target = team->step * (uint32_t)team_w;
    uint32_t *broken = team->arrivals + team->n_slots + team->slot;
    bool ok = false;
    // normally a few microseconds; ~1 s bound.  A workgroup that gives up marks the team
    // broken for everybody (see the kernel: rank 0 then solves the sector on its own).
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (__hip_atomic_load(team->arrivals + team->slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
        ok = true;
        break;
      }
      if ((spin & 255) == 255 && __hip_atomic_load(broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok)
      __hip_atomic_store(broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      ok = __hip_atomic_load(broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds[0] = ok ? 0.f : 1.f;
  }
  __syncthreads();
  const bool timeout = lds[0] != 0.f;
  __syncthreads();
  // The partials were written on other XCDs: a dependent chain of team_w remote loads per lane (the first
  // version) cost 14-24 us per evaluation.  All 512 threads copy them into LDS first - every load in flight at
  // once, one round trip - and then one lane per value adds them up in the same fixed order as before (config
  // 3's blob, a team of 128: 0.75 -> 0.36 ms).
  __shared__ float team_stage[kLkMaxTeam * 32];
  for (int idx = (int)threadIdx.x; idx < team_w * 32; idx += 512)
    team_stage[idx] = mine[idx];
  __syncthreads();
  if ((int)threadIdx.x <= SumsT::N) { // one lane per value: fixed summation order
    float t = 0.f;
    for (int w = 0; w < team_w; ++w)
      t += team_stage[w * 32 + (int)threadIdx.x];
    lds[threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SumsT::N; ++i)
    S.v[i] = lds[i];
  any_bad = lds[SumsT::N] != 0.f;
  team->timed_out = team->timed_out || timeout;
  return any_bad;
}

template <int MODEL, int INTERP, int GROUP, int THREADS>
__device__ __forceinline__ bool evaluate(const LevelCtx &c, const float (&p)[6],
                                         Sums<n_params(MODEL)> &S, float *lds, TeamCtx *team = nullptr,
                                         bool wide = false, bool ordered = false, int width = 0) {
  constexpr int P = n_params(MODEL);
  using SumsT = Sums<P>;
  const int team_w = (GROUP == 512 && team) ? team->w : 1;
  const int team_rank = (GROUP == 512 && team) ? team->rank : 0;
  // GROUP == 32, `wide` (wavefront-uniform): the partner half-wavefront has run out of work
  // and both halves evaluate the same sector with all 64 lanes (see "solo" in the kernel)
  // GROUP == 16, `width` (wavefront-uniform, 16 / 32 / 64): the rows of the wavefront that ran
  // out of work have joined the sectors still being solved (see "adaptive width" in the kernel)
  const int lane0 = (GROUP == 32 && wide)    ? ((int)threadIdx.x & 63)
                    : (GROUP == 16 && width) ? ((int)threadIdx.x & (width - 1))
                                             : team_rank * GROUP + (int)threadIdx.x % GROUP;
  const int stride = (GROUP == 32 && wide) ? 64 : (GROUP == 16 && width) ? width : GROUP * team_w;
#pragma unroll
  for (int i = 0; i < SumsT::N; ++i)
    S.v[i] = 0.f;
  bool bad = false;
  const int umaxr = c.urows - 1, umaxc = c.ucols - 1;
  if constexpr (GROUP == 16) {
    if (ordered) {
      // Finisher of a starved level (at most 2P <= 12 samples): lane i of the row takes sample i
      // in the REFERENCE's order, forms the rounded products there, and every lane then adds
      // them up in sample order (row_newbcast reads lane j of the own row) - the same
      // sequence of roundings as interpolation_class.cpp:722-749 and the one-lane kernel.
      const int i = (int)threadIdx.x & 15;
      const bool has = i < c.n;
      float t[SumsT::N];
#pragma unroll
      for (int v = 0; v < SumsT::N; ++v)
        t[v] = 0.f;
      if (has) {
        f32x2 q;
        if (c.rw > 0) { // x outer, y inner (manager_class.cpp:1607-1611)
          const int h = c.n / c.rw, col = i / h;
          q.x = (float)(c.rx + col);
          q.y = (float)(c.ry + (i - col * h));
        } else {
          q = c.xy[i];
        }
        float xd, yd, dx = 0.f, dy = 0.f;
        Warp<MODEL>::apply(q.x, q.y, c.cx, c.cy, p, xd, yd, dx, dy);
        int uix = (int)(q.x + 0.5f), uiy = (int)(q.y + 0.5f);
        uix = min(max(uix, 0), umaxc);
        uiy = min(max(uiy, 0), umaxr);
        const float und_w = (float)c.und[(size_t)uiy * (size_t)c.ucols + (size_t)uix];
        float W, Wx, Wy;
        if (!sample_def<INTERP>(c.def, c.drows, c.dcols, xd, yd, W, Wx, Wy)) {
          bad = true;
        } else {
          const float V = und_w - W;
          float H[P];
          Warp<MODEL>::jac(Wx, Wy, dx, dy, H);
          int idx = 0;
#pragma unroll
          for (int p1 = 0; p1 < P; ++p1)
#pragma unroll
            for (int p2 = p1; p2 < P; ++p2)
              t[idx++] = H[p1] * H[p2];
#pragma unroll
          for (int p1 = 0; p1 < P; ++p1)
            t[SumsT::NA + p1] = H[p1] * V;
          t[SumsT::N - 1] = V * V;
        }
      }
      const unsigned long long badmask = __ballot(bad);
      // Lanes without a sample (j >= n, or a sample that left the image) hold t = +0, and a sum
      // that started at +0 is never -0, so adding their +0 changes no bit: no select needed.
      // (Stopping at the longest list of the wavefront's rows - config 4's level 2 has 1-4 samples - was tried: the
      // early exit keeps the compiler from scheduling the 2P x 28 broadcast-adds as one block, config 4 1.9 -> 3.7 ms.)
#pragma unroll
      for (int j = 0; j < 2 * P; ++j) {
#pragma unroll
        for (int v = 0; v < SumsT::N; ++v) {
          float tj;
          switch (j) { // row_newbcast:j (DPP control 0x150 + j)
          case 0: tj = dpp_bcast<0>(t[v]); break;
          case 1: tj = dpp_bcast<1>(t[v]); break;
          case 2: tj = dpp_bcast<2>(t[v]); break;
          case 3: tj = dpp_bcast<3>(t[v]); break;
          case 4: tj = dpp_bcast<4>(t[v]); break;
          case 5: tj = dpp_bcast<5>(t[v]); break;
          case 6: tj = dpp_bcast<6>(t[v]); break;
          case 7: tj = dpp_bcast<7>(t[v]); break;
          case 8: tj = dpp_bcast<8>(t[v]); break;
          case 9: tj = dpp_bcast<9>(t[v]); break;
          case 10: tj = dpp_bcast<10>(t[v]); break;
          default: tj = dpp_bcast<11>(t[v]); break;
          }
          S.v[v] = S.v[v] + tj;
        }
      }
      const int row = ((int)threadIdx.x & 63) >> 4;
      return ((badmask >> (16 * row)) & 0xffffull) != 0ull;
    }
  }
  // Lane groups walk implicit rectangles x-fastest so that neighbouring lanes read
  // neighbouring pixels of one image row (a 16-lane group touches 1-2 cache lines per load
  // instead of 16).  The reference enumerates y-fastest (manager_class.cpp:1607-1611); only
  // the float summation order depends on that, and a parallel reduction does not keep it
  // anyway.  GROUP == 1 (one lane = one sector, the starved-level kernel) is different: the
  // lane walks the samples in the REFERENCE's order and accumulates with a separate
  // multiply and add, exactly like interpolation_class.cpp:722-749, so A, b and chi are
  // bit-identical to the reference's.
  // (16-lane rows divide here: one more live register per row would cost that kernel its fourth wavefront per SIMD)
  const float inv_w = GROUP == 16 ? (c.rw > 0 ? 1.f / (float)c.rw : 0.f) : c.inv_w;
  const int rh = (GROUP == 1 && c.rw > 0) ? c.n / c.rw : 1; // height of the implicit rectangle
#ifndef LK_PIPE_MIN_GROUP // lane groups at least this wide prefetch the next sample (tuning hook)
#define LK_PIPE_MIN_GROUP 256
#endif
  // (levels smaller than the 4 x 4 window - deep pyramids of small frames - take the plain loop: the branch-free prefetch
  // below clamps its window INTO the image, which needs an image that holds one)
  if (GROUP >= LK_PIPE_MIN_GROUP && INTERP == LK_IM_BICUBIC && c.rw == 0 && c.dcols >= 4 && c.drows >= 4) { // (explicit lists; implicit rectangles - config 1 - lose 6 % to it)
    // Software pipeline of the workgroup-wide groups (sectors of tens of thousands to millions of samples, explicit
    // lists, two wavefronts per SIMD): the list entry, the coordinates and the five image loads of sample k + stride
    // are issued before sample k's ~250 arithmetic instructions, so the two dependent load latencies of a trip
    // (list -> window) overlap the previous trip's arithmetic.  The prefetch is branch-free - index and window position
    // are clamped into range, the loads always happen - so that the compiler can wait for exactly the loads it needs
    // (a conditional prefetch ends in `s_waitcnt vmcnt(0)` at the join and hides nothing).  Same operations on the same
    // operands in the same order for every real sample: the results do not change.
    struct Fetch {
      Window4 w;
      float und_w, fx, fy, dx, dy;
      bool ok;
    };
    const int last = c.n - 1;
    // (only explicit lists come here: a list entry that merges with computed coordinates inside the loop has to have
    // arrived at the merge, and the pipeline is gone)
    auto coords = [&](int k_in) -> f32x2 { return c.xy[min(k_in, last)]; }; // sample k of the sector (index clamped into the list)
    auto windows = [&](f32x2 q, Fetch &f) { // warp + the image loads of one sample
      float xd, yd;
      f.dx = 0.f, f.dy = 0.f;
      Warp<MODEL>::apply(q.x, q.y, c.cx, c.cy, p, xd, yd, f.dx, f.dy);
      int uix = (int)(q.x + 0.5f), uiy = (int)(q.y + 0.5f);
      uix = min(max(uix, 0), umaxc);
      uiy = min(max(uiy, 0), umaxr);
      f.und_w = (float)c.und[(size_t)uiy * (size_t)c.ucols + (size_t)uix];
      f.ok = xd > 1.f && yd > 1.f && xd < (float)c.dcols - 2.f && yd < (float)c.drows - 2.f; // sample_def's rule
      const int ix = (int)xd, iy = (int)yd;
      f.fx = xd - (float)ix + 1.f, f.fy = yd - (float)iy + 1.f;
      f.w = load_window(c.def, c.dcols, min(max(ix, 1), c.dcols - 3), min(max(iy, 1), c.drows - 3)); // (ok: unclamped)
    };
    // two stages: the list entry of sample k + 2 stride is in flight while the window of k + stride is, and sample
    // k is being computed
    Fetch cur{};
    f32x2 q1{};
    if (c.n > 0) { // (same order of the loads as in the loop: list entry first, then the window of the sample before it)
      const f32x2 q0 = coords(lane0);
      q1 = coords(lane0 + stride);
      windows(q0, cur);
    }
    for (int k = lane0; k < c.n; k += stride) {
      const f32x2 q2 = coords(k + 2 * stride);
      Fetch nxt;
      windows(q1, nxt);
      q1 = q2;
      if (!cur.ok) {
        bad = true; // the sums of an evaluation that hit the error are never used
      } else {
        float W, Wx, Wy;
        bicubic_window(cur.w.r0, cur.w.r1, cur.w.r2, cur.w.r3, cur.fx, cur.fy, W, Wx, Wy);
        const float V = cur.und_w - W;
        float H[P];
        Warp<MODEL>::jac(Wx, Wy, cur.dx, cur.dy, H);
        int idx = 0;
#pragma unroll
        for (int p1 = 0; p1 < P; ++p1)
#pragma unroll
          for (int p2 = p1; p2 < P; ++p2)
            S.v[idx] = __builtin_fmaf(H[p1], H[p2], S.v[idx]), ++idx;
#pragma unroll
        for (int p1 = 0; p1 < P; ++p1)
          S.v[SumsT::NA + p1] = __builtin_fmaf(H[p1], V, S.v[SumsT::NA + p1]);
        S.v[SumsT::N - 1] = __builtin_fmaf(V, V, S.v[SumsT::N - 1]);
      }
      cur = nxt;
    }
  } else
  for (int k = lane0; k < c.n; k += stride) {
    f32x2 q;
    if (GROUP == 1 && c.rw > 0) { // reference order: x outer, y inner
      const int col = k / rh;
      q.x = (float)(c.rx + col);
      q.y = (float)(c.ry + (k - col * rh));
    } else if (c.rw > 0) { // k -> (row, column) of the rectangle
      int row = (int)((float)k * inv_w);
      int col = k - row * c.rw;
      if (col < 0) {
        col += c.rw;
        --row;
      } else if (col >= c.rw) {
        col -= c.rw;
        ++row;
      }
      q.x = (float)(c.rx + col);
      q.y = (float)(c.ry + row);
    } else {
      q = c.xy[k];
    }
    float xd, yd, dx = 0.f, dy = 0.f;
    Warp<MODEL>::apply(q.x, q.y, c.cx, c.cy, p, xd, yd, dx, dy);
    int uix = (int)(q.x + 0.5f), uiy = (int)(q.y + 0.5f);
    uix = min(max(uix, 0), umaxc); // memory safety only; valid sample lists never clamp
    uiy = min(max(uiy, 0), umaxr);
    float und_w = (float)c.und[(size_t)uiy * (size_t)c.ucols + (size_t)uix];
    float W, Wx, Wy;
    if (!sample_def<INTERP>(c.def, c.drows, c.dcols, xd, yd, W, Wx, Wy)) {
      bad = true;
      continue; // the sums of an evaluation that hit the error are never used
    }
    float V = und_w - W;
    float H[P];
    Warp<MODEL>::jac(Wx, Wy, dx, dy, H);
    int idx = 0;
    if constexpr (GROUP == 1) { // rounded product, then rounded add (no FMA: contraction is off)
      S.v[SumsT::N - 1] += V * V;
#pragma unroll
      for (int p1 = 0; p1 < P; ++p1) {
        S.v[SumsT::NA + p1] += H[p1] * V;
#pragma unroll
        for (int p2 = p1; p2 < P; ++p2)
          S.v[idx] += H[p1] * H[p2], ++idx;
      }
    } else {
#pragma unroll
      for (int p1 = 0; p1 < P; ++p1)
#pragma unroll
        for (int p2 = p1; p2 < P; ++p2)
          S.v[idx] = __builtin_fmaf(H[p1], H[p2], S.v[idx]), ++idx;
#pragma unroll
      for (int p1 = 0; p1 < P; ++p1)
        S.v[SumsT::NA + p1] = __builtin_fmaf(H[p1], V, S.v[SumsT::NA + p1]);
      S.v[SumsT::N - 1] = __builtin_fmaf(V, V, S.v[SumsT::N - 1]);
    }
  }
  if constexpr (GROUP == 1)
    return bad; // nothing to reduce: the lane owns the whole sector
  // reconverged: all 64 lanes of every wave are active from here on
  const unsigned long long badmask = __ballot(bad);
  // stage by stage over all sums (the same four additions per sum as row16_sum, in the same
  // order): 28 independent DPP adds per stage instead of 28 chains of four dependent ones with
  // the DPP hazard nops in between
#pragma unroll
  for (int i = 0; i < SumsT::N; ++i)
    S.v[i] = dpp_add<0xB1>(S.v[i]); // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < SumsT::N; ++i)
    S.v[i] = dpp_add<0x4E>(S.v[i]); // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < SumsT::N; ++i)
    S.v[i] = dpp_add<0x141>(S.v[i]); // row_half_mirror
#pragma unroll
  for (int i = 0; i < SumsT::N; ++i)
    S.v[i] = dpp_add<0x140>(S.v[i]); // row_mirror
  if constexpr (GROUP == 16) {
    if (width >= 32) { // a + b == b + a: every lane of the widened group ends with the same bits
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i)
        S.v[i] += __shfl_xor(S.v[i], 16, 64);
      if (width == 64) {
#pragma unroll
        for (int i = 0; i < SumsT::N; ++i)
          S.v[i] += __shfl_xor(S.v[i], 32, 64);
        return badmask != 0ull;
      }
      const int half = ((int)threadIdx.x & 63) >> 5;
      return ((badmask >> (32 * half)) & 0xffffffffull) != 0ull;
    }
    const int row = ((int)threadIdx.x & 63) >> 4;
    return ((badmask >> (16 * row)) & 0xffffull) != 0ull;
  } else if constexpr (GROUP == 32) { // two sectors per wavefront: add the partner row
#pragma unroll
    for (int i = 0; i < SumsT::N; ++i)
      S.v[i] += __shfl_xor(S.v[i], 16, 64);
    if (wide) { // a + b == b + a: both halves end with the same bits
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i)
        S.v[i] += __shfl_xor(S.v[i], 32, 64);
      return badmask != 0ull;
    }
    const int half = ((int)threadIdx.x & 63) >> 5;
    return ((badmask >> (32 * half)) & 0xffffffffull) != 0ull;
  } else {
#pragma unroll
    for (int i = 0; i < SumsT::N; ++i)
      S.v[i] = rows_sum(S.v[i]);
    bool any_bad = badmask != 0ull;
    if constexpr (GROUP > kWave) {
      constexpr int WAVES = THREADS / kWave;
      const int wave = (int)threadIdx.x / kWave, lane = (int)threadIdx.x % kWave;
      constexpr int STRIDE = SumsT::N + 1;
      __syncthreads(); // previous readers of lds are done
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < SumsT::N; ++i)
          lds[wave * STRIDE + i] = S.v[i];
        lds[wave * STRIDE + SumsT::N] = any_bad ? 1.f : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i) {
        float t = lds[i];
        for (int w = 1; w < WAVES; ++w)
          t += lds[w * STRIDE + i];
        S.v[i] = t;
      }
      float eb = 0.f;
      for (int w = 0; w < WAVES; ++w)
        eb += lds[w * STRIDE + SumsT::N];
      any_bad = eb != 0.f;
      if constexpr (GROUP == 512) {
        if (team_w > 1) // all-to-all of the workgroup totals inside the team
          any_bad = team_all_to_all(S, any_bad, lds, team);
      }
    }
    return any_bad;
  }
}

// ------------------------------------------------------------------------------------
// Reference-order evaluation (lk_set_reference_order): A, b and chi with the SAME sequence of
// roundings as the CPU engine, at any sector size.
//
// The reference adds the rounded products of sample 0, 1, 2, ... (x outer / y inner for
// rectangles, manager_class.cpp:1607-1611) into one running float per sum
// (interpolation_class.cpp:722-749); with number_of_threads = T it does so per contiguous chunk
// of n/T samples (the first n%T chunks one longer, correlation_class.cpp:169-186) and adds the
// chunk sums in thread order into zeroed totals (:253-275).  The additions of one sum are a
// chain - but the 28 sums are independent chains, and everything before the additions is
// independent per sample.  So: the GROUP lanes of a lane group take GROUP consecutive samples,
// form the rounded products (warp, bicubic, residual, H - the expensive 95 %), and transpose
// them through LDS; then lane v owns sum v and walks the GROUP products of its sum in sample
// order (16-lane rows: lanes own sums v and v + 16).  Per GROUP samples that is N stores, GROUP/4
// 128-bit loads and GROUP dependent adds per lane - against ~390 instructions for the samples
// themselves.  Products of lanes beyond the list are +0: a sum that started at +0 is never -0,
// so adding +0 changes no bit.  The totals go back to every lane through LDS.
// ord: this wavefront's staging area, kOrdFloats<N, GROUP> floats.
// ------------------------------------------------------------------------------------
template <int GROUP> constexpr int ord_stride() { return GROUP + 4; } // +4: the 128-bit reads of neighbouring lanes hit different banks
template <int N, int GROUP> constexpr int ord_floats() { return (kWave / GROUP) * N * ord_stride<GROUP>(); }

// CHUNKED: number_of_threads > 1 (the chunk boundaries cost a test per addition; T = 1 has none)
template <int MODEL, int INTERP, int GROUP, bool CHUNKED>
__device__ __forceinline__ bool evaluate_ordered(const LevelCtx &c, const float (&p)[6],
                                                 Sums<n_params(MODEL)> &S, float *ord, int threads) {
  static_assert(GROUP == 16 || GROUP == 64, "one 16-lane row or one wavefront per sector");
  constexpr int P = n_params(MODEL);
  using SumsT = Sums<P>;
  constexpr int N = SumsT::N, STR = ord_stride<GROUP>();
  constexpr bool TWO = GROUP == 16 && N > 16; // a 16-lane row owns up to 32 sums: v and v + 16
  const int lane = (int)threadIdx.x & (kWave - 1), g = lane & (GROUP - 1);
  float *mine = ord + (lane / GROUP) * (N * STR); // the group's [N][STR] staging block
  const int v0 = g < N ? g : 0, v1 = (TWO && g + 16 < N) ? g + 16 : 0; // (idle lanes shadow sum 0)
  float acc0 = 0.f, acc1 = 0.f, tot0 = 0.f, tot1 = 0.f;
  // thread chunks of the reference: chunk t has n/T + (t < n%T) samples
  const int T = CHUNKED ? (threads < 1 ? 1 : threads) : 1;
  const int cq = c.n / T, cr = c.n - cq * T;
  int t_idx = 0, next_b = cq + (cr > 0 ? 1 : 0);
  bool bad = false;
  const int umaxr = c.urows - 1, umaxc = c.ucols - 1;
  const int rh = c.rw > 0 ? c.n / c.rw : 1; // height of the implicit rectangle
  const float inv_rh = 1.f / (float)rh;
  for (int base = 0; __any(base < c.n); base += GROUP) {
    const int k = base + g;
    float t[N];
#pragma unroll
    for (int v = 0; v < N; ++v)
      t[v] = 0.f;
    if (k < c.n) {
      f32x2 q;
      if (c.rw > 0) { // x outer, y inner: k -> (column, row) of the rectangle
        int col = (int)((float)k * inv_rh); // k / rh to within one unit below 2^23 samples, two below 2^25
        int row = k - col * rh;
#pragma unroll
        for (int fix = 0; fix < 2; ++fix) {
          const int lo = row < 0 ? 1 : 0, hi = row >= rh ? 1 : 0;
          row += (lo - hi) * rh;
          col += hi - lo;
        }
        q.x = (float)(c.rx + col);
        q.y = (float)(c.ry + row);
      } else {
        q = c.xy[k];
      }
      float xd, yd, dx = 0.f, dy = 0.f;
      Warp<MODEL>::apply(q.x, q.y, c.cx, c.cy, p, xd, yd, dx, dy);
      int uix = (int)(q.x + 0.5f), uiy = (int)(q.y + 0.5f);
      uix = min(max(uix, 0), umaxc);
      uiy = min(max(uiy, 0), umaxr);
      const float und_w = (float)c.und[(size_t)uiy * (size_t)c.ucols + (size_t)uix];
      float W, Wx, Wy;
      if (!sample_def<INTERP>(c.def, c.drows, c.dcols, xd, yd, W, Wx, Wy)) {
        bad = true; // (the sums of an evaluation that hit the error are never used)
      } else {
        const float V = und_w - W;
        float H[P];
        Warp<MODEL>::jac(Wx, Wy, dx, dy, H);
        int idx = 0;
#pragma unroll
        for (int p1 = 0; p1 < P; ++p1)
#pragma unroll
          for (int p2 = p1; p2 < P; ++p2)
            t[idx++] = H[p1] * H[p2]; // rounded product; the rounded add follows below
#pragma unroll
        for (int p1 = 0; p1 < P; ++p1)
          t[SumsT::NA + p1] = H[p1] * V;
        t[N - 1] = V * V;
      }
    }
    // transpose: lane g's product of sum v -> mine[v][g]  (all lanes of the wavefront take part)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the previous pass's reads are done
#pragma unroll
    for (int v = 0; v < N; ++v)
      mine[v * STR + g] = t[v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float *r0 = mine + v0 * STR, *r1 = mine + v1 * STR;
    // (a pass with no chunk boundary in it - all of them for T = 1, all but T of them on long lists - adds without looking)
    if (CHUNKED && next_b < base + GROUP) {
#pragma unroll
      for (int j4 = 0; j4 < GROUP / 4; ++j4) {
        const float4 a4 = *reinterpret_cast<const float4 *>(r0 + 4 * j4);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (TWO)
          b4 = *reinterpret_cast<const float4 *>(r1 + 4 * j4);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (base + 4 * j4 + j == next_b) { // a thread chunk ends before this sample (:253-275)
            tot0 += acc0;
            acc0 = 0.f;
            if constexpr (TWO) {
              tot1 += acc1;
              acc1 = 0.f;
            }
            ++t_idx;
            next_b += cq + (t_idx < cr ? 1 : 0);
          }
          acc0 += av[j];
          if constexpr (TWO)
            acc1 += bv[j];
        }
      }
    } else {
#pragma unroll
      for (int j4 = 0; j4 < GROUP / 4; ++j4) {
        const float4 a4 = *reinterpret_cast<const float4 *>(r0 + 4 * j4);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (TWO)
          b4 = *reinterpret_cast<const float4 *>(r1 + 4 * j4);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 += av[j];
          if constexpr (TWO)
            acc1 += bv[j];
        }
      }
    }
  }
  tot0 += acc0; // the last chunk (T == 1: 0 + acc)
  if constexpr (TWO)
    tot1 += acc1;
  // totals back to every lane of the group
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (g < N)
    mine[g] = tot0;
  if constexpr (TWO)
    if (g + 16 < N)
      mine[g + 16] = tot1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int v = 0; v < N; ++v)
    S.v[v] = mine[v];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the next evaluation overwrites mine[])
  const unsigned long long badmask = __ballot(bad);
  if constexpr (GROUP == 16)
    return ((badmask >> (16 * (lane >> 4))) & 0xffffull) != 0ull;
  return badmask != 0ull;
}

// ------------------------------------------------------------------------------------
// Reference-order evaluation of ONE big sector by a 512-thread workgroup (tens of thousands to millions of samples).
//
// One wavefront per sector (evaluate_ordered<GROUP = 64>) forms 64 products, then 28 of its lanes add them, then the
// next 64: config 3's 4.2 M-sample blob took 35 ms per evaluation that way.  Here seven wavefronts FORM the products of
// 448 consecutive samples per trip while the eighth ADDS the previous trip's - lane v owns sum v and walks the 448
// products of its sum in sample order, with the thread-chunk boundaries of correlation_class.cpp:169-186,253-275 -
// through two LDS tiles [sum][448] that swap roles at one workgroup barrier per trip.  The additions of one sum are a
// single chain whoever forms the products (448 dependent additions per trip, about what seven wavefronts need for the
// products), so this is the floor of the reference's order on one sector: ~10 ms per evaluation of the blob.
// ord: 2 * N * kWgOrdStride floats.
// ------------------------------------------------------------------------------------
constexpr int kWgOrdSamples = 7 * kWave;          // samples per trip: seven producing wavefronts
constexpr int kWgOrdStride = kWgOrdSamples + 4;   // (+4: the 128-bit reads of neighbouring owner lanes hit different banks)
template <int N> constexpr int wg_ord_floats() { return 2 * N * kWgOrdStride; }

// first, count: the samples [first, first + count) of the list only (one thread chunk of the reference, solved by one
// workgroup of a team: see the kernel) - CHUNKED is then false, the chunk is one chain.
template <int MODEL, int INTERP, bool CHUNKED>
__device__ __forceinline__ bool evaluate_ordered_wg(const LevelCtx &c, const float (&p)[6], Sums<n_params(MODEL)> &S,
                                                    float *ord, int threads, int first, int count) {
  constexpr int P = n_params(MODEL);
  using SumsT = Sums<P>;
  constexpr int N = SumsT::N, STR = kWgOrdStride;
  static_assert(N <= kWave, "one owner lane per sum");
  const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & (kWave - 1);
  const bool consumer = wave == 0;
  const int v = lane < N ? lane : 0; // (idle lanes of the adding wavefront shadow sum 0)
  float acc = 0.f, tot = 0.f;
  const int T = CHUNKED ? (threads < 1 ? 1 : threads) : 1;
  const int cq = count / T, cr = count - cq * T; // thread chunks of the reference: chunk t has n/T + (t < n%T) samples
  int t_idx = 0, next_b = cq + (cr > 0 ? 1 : 0); // (positions relative to `first`)
  bool bad = false;
  const int umaxr = c.urows - 1, umaxc = c.ucols - 1;
  const int rh = c.rw > 0 ? c.n / c.rw : 1; // height of the implicit rectangle
  const float inv_rh = 1.f / (float)rh;
  const int n_trips = (count + kWgOrdSamples - 1) / kWgOrdSamples;
  __syncthreads(); // (the previous evaluation's readers of the tiles are done)
  for (int trip = 0; trip <= n_trips; ++trip) {
    if (!consumer && trip < n_trips) { // products of this trip's samples -> tile[trip & 1]
      float *tile = ord + (trip & 1) * (N * STR);
      const int slot = (wave - 1) * kWave + lane, rel = trip * kWgOrdSamples + slot, k = first + rel;
      float t[N];
#pragma unroll
      for (int i = 0; i < N; ++i)
        t[i] = 0.f;
      if (rel < count) {
        f32x2 q;
        if (c.rw > 0) { // x outer, y inner: k -> (column, row) of the rectangle
          int col = (int)((float)k * inv_rh); // k / rh to within one unit below 2^23 samples, two below 2^25
          int row = k - col * rh;
#pragma unroll
          for (int fix = 0; fix < 2; ++fix) {
            const int lo = row < 0 ? 1 : 0, hi = row >= rh ? 1 : 0;
            row += (lo - hi) * rh;
            col += hi - lo;
          }
          q.x = (float)(c.rx + col);
          q.y = (float)(c.ry + row);
        } else {
          q = c.xy[k];
        }
        float xd, yd, dx = 0.f, dy = 0.f;
        Warp<MODEL>::apply(q.x, q.y, c.cx, c.cy, p, xd, yd, dx, dy);
        int uix = (int)(q.x + 0.5f), uiy = (int)(q.y + 0.5f);
        uix = min(max(uix, 0), umaxc);
        uiy = min(max(uiy, 0), umaxr);
        const float und_w = (float)c.und[(size_t)uiy * (size_t)c.ucols + (size_t)uix];
        float W, Wx, Wy;
        if (!sample_def<INTERP>(c.def, c.drows, c.dcols, xd, yd, W, Wx, Wy)) {
          bad = true; // (the sums of an evaluation that hit the error are never used)
        } else {
          const float V = und_w - W;
          float H[P];
          Warp<MODEL>::jac(Wx, Wy, dx, dy, H);
          int idx = 0;
#pragma unroll
          for (int p1 = 0; p1 < P; ++p1)
#pragma unroll
            for (int p2 = p1; p2 < P; ++p2)
              t[idx++] = H[p1] * H[p2]; // rounded product; the rounded add is the adding wavefront's
#pragma unroll
          for (int p1 = 0; p1 < P; ++p1)
            t[SumsT::NA + p1] = H[p1] * V;
          t[N - 1] = V * V;
        }
      }
      // (samples beyond the list leave +0: a sum that started at +0 is never -0, so adding +0 changes no bit)
#pragma unroll
      for (int i = 0; i < N; ++i)
        tile[i * STR + slot] = t[i];
    }
    if (consumer && trip > 0) { // the previous trip's products, in sample order
      const float *row = ord + ((trip - 1) & 1) * (N * STR) + v * STR;
      const int base = (trip - 1) * kWgOrdSamples;
      if (CHUNKED && next_b < base + kWgOrdSamples) { // a thread chunk ends inside this trip (:253-275): T trips per evaluation
        for (int j4 = 0; j4 < kWgOrdSamples / 4; ++j4) {
          const float4 a4 = *reinterpret_cast<const float4 *>(row + 4 * j4);
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (base + 4 * j4 + j == next_b) {
              tot += acc;
              acc = 0.f;
              ++t_idx;
              next_b += cq + (t_idx < cr ? 1 : 0);
            }
            acc += av[j];
          }
        }
      } else {
#pragma unroll 4
        for (int j4 = 0; j4 < kWgOrdSamples / 4; ++j4) {
          const float4 a4 = *reinterpret_cast<const float4 *>(row + 4 * j4);
          acc += a4.x;
          acc += a4.y;
          acc += a4.z;
          acc += a4.w;
        }
      }
    }
    __syncthreads(); // the tiles swap roles
  }
  tot += acc; // the last chunk (T == 1: 0 + acc)
  // totals and the error flag to every thread
  if (consumer && lane < N)
    ord[lane] = tot;
  const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0; // (also: the totals are visible)
#pragma unroll
  for (int i = 0; i < N; ++i)
    S.v[i] = ord[i];
  return any_bad;
}

// ------------------------------------------------------------------------------------
// Reference-order evaluation of the FOUR sectors of a wavefront at once, lanes dealt by need.
//
// With one 16-lane row per sector (evaluate_ordered<GROUP = 16>) a step of the wavefront costs as many
// trips as its largest sector needs, whatever the other rows are doing: a row whose sector is finished, or
// sits at a coarse pyramid level with a handful of samples, idles through the trips of a row at level 0 -
// and the stragglers of a launch (config 2: up to 29 evaluations where the median is 12, nearly all of them
// at levels 2 and 1) keep three idle rows waiting.  Measured (profiles/r03_reforder_before_*): the launch
// lasts 573 us while the median SIMD is done after 308.
//
// Here the unit of work is a SLOT = 16 consecutive samples of one sector, and every trip deals its four
// 16-lane tile rows to the sectors that still have slots left, most remaining first (one row each while four
// sectors are busy; two or four rows for a sector whose neighbours have run out).  Who FORMS a product does
// not matter to the bits; who ADDS it does not either, as long as every sum of a sector sees its products in
// sample order - so the owner lanes stay where they were (the sector's home row: lane v owns sums v and
// v + 16) and walk the tile rows their sector got in this trip, in ascending order = sample order.  A trip
// is: slot table (scalar) -> the lanes of a tile row fetch that sector's context from LDS if it changed ->
// products -> LDS tile [sum][64 samples] -> the home rows add.  A step now costs
// ceil(sum over the busy sectors of ceil(n / 16) / 4) trips, so the rows need no level alignment and no
// common fetch any more: every sector runs at its own pace.
// ------------------------------------------------------------------------------------
struct OrdCtx { // one sector's evaluation context, published by its home row (24 words)
  float p[6], cx, cy;
  int rx, ry, rw, n;
  int rh, urows, ucols, drows;
  int dcols;
  uint32_t und_lo, und_hi, def_lo;
  uint32_t def_hi, xy_lo, xy_hi;
  float inv_rh;
};
static_assert(sizeof(OrdCtx) == 96, "six 128-bit words");
constexpr int kFlatStride = kWave + 4; // tile row: 64 samples + 4 floats (the 128-bit reads of neighbouring owner lanes hit different banks)
template <int N> constexpr int flat_tile_floats() { return N * kFlatStride; }

template <int MODEL, int INTERP, bool CHUNKED>
__device__ __forceinline__ bool evaluate_ordered_flat(const LevelCtx &c, const float (&p)[6], Sums<n_params(MODEL)> &S,
                                                      float *tile, OrdCtx *ctx, int threads) {
  constexpr int P = n_params(MODEL);
  using SumsT = Sums<P>;
  constexpr int N = SumsT::N, STR = kFlatStride;
  constexpr bool TWO = N > 16; // a home row owns up to 32 sums: v and v + 16
  const int lane = (int)threadIdx.x & (kWave - 1), g = lane & 15, row = lane >> 4;
  // publish this row's sector (every lane of the row holds the same values)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the previous evaluation's reads of ctx / tile are done)
  if (g == 0) {
    OrdCtx o;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      o.p[i] = p[i];
    o.cx = c.cx, o.cy = c.cy;
    o.rx = c.rx, o.ry = c.ry, o.rw = c.rw, o.n = c.n;
    o.rh = c.rw > 0 ? c.n / c.rw : 1;
    o.urows = c.urows, o.ucols = c.ucols, o.drows = c.drows, o.dcols = c.dcols;
    const unsigned long long ub = (unsigned long long)(uintptr_t)c.und, db = (unsigned long long)(uintptr_t)c.def,
                             xb = (unsigned long long)(uintptr_t)c.xy;
    o.und_lo = (uint32_t)ub, o.und_hi = (uint32_t)(ub >> 32), o.def_lo = (uint32_t)db, o.def_hi = (uint32_t)(db >> 32);
    o.xy_lo = (uint32_t)xb, o.xy_hi = (uint32_t)(xb >> 32);
    o.inv_rh = 1.f / (float)o.rh;
    const float4 *src = reinterpret_cast<const float4 *>(&o);
    float4 *dst = reinterpret_cast<float4 *>(ctx + row);
#pragma unroll
    for (int i = 0; i < 6; ++i)
      dst[i] = src[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // slots left per sector (wavefront-uniform: scalar registers)
  int rem[4], used[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    rem[r] = (__builtin_amdgcn_readlane(c.n, 16 * r) + 15) >> 4;
    used[r] = 0;
  }
  // owner state of my sector (home row): running chunk sums and totals of sums v0 (and v1)
  const int v0 = g < N ? g : 0, v1 = (TWO && g + 16 < N) ? g + 16 : 0; // (idle lanes shadow sum 0)
  float acc0 = 0.f, acc1 = 0.f, tot0 = 0.f, tot1 = 0.f;
  const int T = CHUNKED ? (threads < 1 ? 1 : threads) : 1;
  const int cq = c.n / T, cr = c.n - cq * T; // thread chunks of the reference: chunk t has n/T + (t < n%T) samples
  int t_idx = 0, next_b = cq + (cr > 0 ? 1 : 0), kpos = 0;
  unsigned bad_sectors = 0u; // bit r: a sample of sector r left the image
  OrdCtx e{};                // the context of the sector my tile row works for (reloaded only when the deal changes)
  int cur = -1;
  // A DEAL says which sector each tile row works for: every busy sector gets one row, the rows left over go,
  // one by one, to the busy sector with the most slots left per row; then the sectors take their rows as consecutive
  // runs of tile rows, in sector order (nobody needs its "own" row: every lane fetches its sector's context from
  // LDS).  A sector's rows in ascending order are consecutive slots, i.e. its samples in order.  A deal lasts for
  // as many trips as every busy sector can fill all its rows (the inner loop: no scalar work besides the trip
  // count); then the next deal - a handful per evaluation.  All of this is wavefront-uniform scalar work.
  while ((rem[0] | rem[1] | rem[2] | rem[3]) != 0) {
    int cnt[4], spare = 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cnt[r] = rem[r] > 0 ? 1 : 0;
      spare -= cnt[r];
    }
    for (; spare > 0; --spare) {
      int best = -1, best_rem = 0, best_cnt = 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) // (a sector that is not busy has rem = cnt = 0 and never qualifies)
        if (rem[r] > cnt[r] && (best < 0 || rem[r] * best_cnt > best_rem * cnt[r]))
          best = r, best_rem = rem[r], best_cnt = cnt[r];
      if (best < 0)
        break;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cnt[r] += best == r ? 1 : 0;
    }
    const int off1 = cnt[0], off2 = off1 + cnt[1], off3 = off2 + cnt[2], off4 = off3 + cnt[3]; // first tile row of sectors 1, 2, 3; rows in use
    int trips = 0x7fffffff; // every busy sector fills all its rows in each of them
#pragma unroll
    for (int r = 0; r < 4; ++r) // rem / cnt for cnt = 1 .. 4 without a division (at most 32 slots: 512 samples)
      if (cnt[r] > 0)
        trips = min(trips, cnt[r] == 1 ? rem[r] : cnt[r] == 2 ? rem[r] >> 1 : cnt[r] == 3 ? (int)(((unsigned)rem[r] * 0xAAABu) >> 17) : rem[r] >> 2);
    // my tile row: its sector and its rank among the sector's rows; my home row: my own sector's run of tile rows
    const int my_sec = row >= off4 ? -1 : (row >= off1 ? 1 : 0) + (row >= off2 ? 1 : 0) + (row >= off3 ? 1 : 0);
    const int sec_off = my_sec == 1 ? off1 : my_sec == 2 ? off2 : my_sec == 3 ? off3 : 0;
    const int my_rank = row - sec_off;
    const int sec_used = my_sec == 0 ? used[0] : my_sec == 1 ? used[1] : my_sec == 2 ? used[2] : used[3];
    const int sec_cnt = my_sec == 0 ? cnt[0] : my_sec == 1 ? cnt[1] : my_sec == 2 ? cnt[2] : cnt[3];
    const int my_first = row == 0 ? 0 : row == 1 ? off1 : row == 2 ? off2 : off3;
    const int my_cnt = row == 0 ? cnt[0] : row == 1 ? cnt[1] : row == 2 ? cnt[2] : cnt[3];
    const int most = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
    if (__any(my_sec >= 0 && my_sec != cur)) {
      const int want = my_sec >= 0 ? my_sec : (cur >= 0 ? cur : 0);
      const float4 *src = reinterpret_cast<const float4 *>(ctx + want);
      float4 *dst = reinterpret_cast<float4 *>(&e);
#pragma unroll
      for (int i = 0; i < 6; ++i)
        dst[i] = src[i];
      cur = want;
    }
    gptr<uint8_t> und = (gptr<uint8_t>)(uintptr_t)(((unsigned long long)e.und_hi << 32) | e.und_lo);
    gptr<uint8_t> def = (gptr<uint8_t>)(uintptr_t)(((unsigned long long)e.def_hi << 32) | e.def_lo);
    gptr<f32x2> xyl = (gptr<f32x2>)(uintptr_t)(((unsigned long long)e.xy_hi << 32) | e.xy_lo);
    const int n_mine = my_sec >= 0 ? e.n : 0; // (a row without a sector forms no products)
    int k = ((sec_used + my_rank) << 4) + g;  // my sample in the first trip of this deal; + 16 * sec_cnt per trip
    const int k_step = sec_cnt << 4;
    for (int trip = 0; trip < trips; ++trip, k += k_step) {
      float t[N];
#pragma unroll
      for (int v = 0; v < N; ++v)
        t[v] = 0.f;
      bool bad = false;
      if (k < n_mine) {
        f32x2 q;
        if (e.rw > 0) { // x outer, y inner: k -> (column, row) of the rectangle
          const int rh = e.rh;
          int col = (int)((float)k * e.inv_rh); // k / rh to within one unit below 2^23 samples, two below 2^25
          int rr = k - col * rh;
#pragma unroll
          for (int fix = 0; fix < 2; ++fix) {
            const int lo = rr < 0 ? 1 : 0, hi = rr >= rh ? 1 : 0;
            rr += (lo - hi) * rh;
            col += hi - lo;
          }
          q.x = (float)(e.rx + col);
          q.y = (float)(e.ry + rr);
        } else {
          q = xyl[k];
        }
        float xd, yd, dx = 0.f, dy = 0.f;
        Warp<MODEL>::apply(q.x, q.y, e.cx, e.cy, e.p, xd, yd, dx, dy);
        int uix = (int)(q.x + 0.5f), uiy = (int)(q.y + 0.5f);
        uix = min(max(uix, 0), e.ucols - 1);
        uiy = min(max(uiy, 0), e.urows - 1);
        const float und_w = (float)und[(size_t)uiy * (size_t)e.ucols + (size_t)uix];
        float W, Wx, Wy;
        if (!sample_def<INTERP>(def, e.drows, e.dcols, xd, yd, W, Wx, Wy)) {
          bad = true; // (the sums of an evaluation that hit the error are never used)
        } else {
          const float V = und_w - W;
          float H[P];
          Warp<MODEL>::jac(Wx, Wy, dx, dy, H);
          int idx = 0;
#pragma unroll
          for (int p1 = 0; p1 < P; ++p1)
#pragma unroll
            for (int p2 = p1; p2 < P; ++p2)
              t[idx++] = H[p1] * H[p2]; // rounded product; the rounded add follows below
#pragma unroll
          for (int p1 = 0; p1 < P; ++p1)
            t[SumsT::NA + p1] = H[p1] * V;
          t[N - 1] = V * V;
        }
      }
      {
        const unsigned long long bm = __ballot(bad);
        if (bm != 0ull) {
#pragma unroll
          for (int tr = 0; tr < 4; ++tr)
            if (tr < off4 && ((bm >> (16 * tr)) & 0xffffull) != 0ull)
              bad_sectors |= 1u << ((tr >= off1 ? 1 : 0) + (tr >= off2 ? 1 : 0) + (tr >= off3 ? 1 : 0));
        }
      }
      // transpose: my product of sum v -> tile[v][lane]
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the previous trip's reads are done
#pragma unroll
      for (int v = 0; v < N; ++v)
        tile[v * STR + lane] = t[v];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // the home rows add what their sectors got in this trip, tile rows in ascending order = sample order
      for (int j = 0; j < most; ++j) {
        if (j < my_cnt) {
          const int tr = my_first + j;
          const float *r0 = tile + v0 * STR + 16 * tr, *r1 = tile + v1 * STR + 16 * tr;
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) {
            const float4 a4 = *reinterpret_cast<const float4 *>(r0 + 4 * j4);
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (TWO)
              b4 = *reinterpret_cast<const float4 *>(r1 + 4 * j4);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              if (CHUNKED && kpos + 4 * j4 + jj == next_b) { // a thread chunk ends before this sample (:253-275)
                tot0 += acc0;
                acc0 = 0.f;
                if constexpr (TWO) {
                  tot1 += acc1;
                  acc1 = 0.f;
                }
                ++t_idx;
                next_b += cq + (t_idx < cr ? 1 : 0);
              }
              acc0 += av[jj];
              if constexpr (TWO)
                acc1 += bv[jj];
            }
          }
          kpos += 16;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      used[r] += trips * cnt[r];
      rem[r] -= trips * cnt[r];
    }
  }
  tot0 += acc0; // the last chunk (T == 1: 0 + acc)
  if constexpr (TWO)
    tot1 += acc1;
  // totals back to every lane of the home row
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  float *back = tile + row * 32;
  if (g < N)
    back[g] = tot0;
  if constexpr (TWO)
    if (g + 16 < N)
      back[g + 16] = tot1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int v = 0; v < N; ++v)
    S.v[v] = back[v];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the next evaluation overwrites the tile)
  return ((bad_sectors >> row) & 1u) != 0u;
}

// ------------------------------------------------------------------------------------
// the damped normal-equation solve (compute_model_parameters + solve,
// correlation_class.cpp:642-768)
// ------------------------------------------------------------------------------------
// The reference hands the symmetric, LM-damped matrix to Eigen's ColPivHouseholderQR
// (correlation_class.cpp:742-747); its CUDA path uses cuSOLVER's Cholesky instead
// (cuda_solver.cu:120-149).  Two solvers live here:
//
//  * fast path - A = sum(H H^T)/n with the diagonal scaled by (1+lambda) is symmetric
//    positive (semi-)definite; it is factored as U^T D U (root-free Cholesky) entirely in
//    registers, ~200 instructions.  Used whenever every pivot is at least 1e-4 of the
//    largest diagonal entry, i.e. the system is well conditioned and any backward-stable
//    solver returns the same step to rounding (tests/test_parity_gpu.py docstring).
//  * reference path - Eigen 3.4.0 ColPivHouseholderQR restated (column norms with
//    LAPACK-style down-dating, first-largest remaining column as pivot, Householder
//    reflectors, rank decision, back-substitution), ~3000 instructions.  Used when a pivot
//    of the fast path is small or non-positive: few samples (fewer than parameters at a
//    coarse pyramid level - BASELINE config 5's level 3 has 4-9 samples for 6 parameters),
//    flat texture.  There the answer is DEFINED by the rank-revealing pivoting (dropped
//    pivots get a zero step), so the engine follows it operation by operation.
//    All indices are compile-time after unrolling; the pivot choice is applied with
//    compare-and-swap so nothing is dynamically indexed (no scratch).  M is column-major.
template <int N>
__device__ __forceinline__ void colpiv_qr_solve(float (&M)[N * N], const float (&bin)[N],
                                                float (&x)[N]) {
#define QR(r, c) M[(c)*N + (r)]
  float hc[N], normU[N], normD[N], cv[N];
  int trans[N];
  const float eps = FLT_EPSILON;
  float maxn = 0.f;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < N; ++r)
      s += QR(r, k) * QR(r, k);
    normD[k] = __builtin_sqrtf(s);
    normU[k] = normD[k];
    if (normU[k] > maxn)
      maxn = normU[k];
  }
  const float threshold_helper = (maxn * eps) * (maxn * eps) / (float)N;
  const float norm_downdate_threshold = __builtin_sqrtf(eps);
  int nonzero_pivots = N;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int big = k;
    float bigv = normU[k];
#pragma unroll
    for (int j = k + 1; j < N; ++j)
      if (normU[j] > bigv) {
        bigv = normU[j];
        big = j;
      }
    float big_sq = bigv * bigv;
    if (nonzero_pivots == N && big_sq < threshold_helper * (float)(N - k))
      nonzero_pivots = k;
    trans[k] = big;
#pragma unroll
    for (int j = k + 1; j < N; ++j) {
      if (big == j) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
          float t = QR(r, k);
          QR(r, k) = QR(r, j);
          QR(r, j) = t;
        }
        float t = normU[k];
        normU[k] = normU[j];
        normU[j] = t;
        t = normD[k];
        normD[k] = normD[j];
        normD[j] = t;
      }
    }
    float tailSq = 0.f;
#pragma unroll
    for (int r = k + 1; r < N; ++r)
      tailSq += QR(r, k) * QR(r, k);
    float c0 = QR(k, k), beta, tau;
    if (tailSq <= FLT_MIN) {
      tau = 0.f;
      beta = c0;
#pragma unroll
      for (int r = k + 1; r < N; ++r)
        QR(r, k) = 0.f;
    } else {
      beta = __builtin_sqrtf(c0 * c0 + tailSq);
      if (c0 >= 0.f)
        beta = -beta;
      float den = c0 - beta;
#pragma unroll
      for (int r = k + 1; r < N; ++r)
        QR(r, k) = QR(r, k) / den;
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    QR(k, k) = beta;
    if (N - k > 1 && tau != 0.f) {
#pragma unroll
      for (int j = k + 1; j < N; ++j) {
        float tmp = 0.f;
#pragma unroll
        for (int r = k + 1; r < N; ++r)
          tmp += QR(r, k) * QR(r, j);
        tmp += QR(k, j);
        QR(k, j) -= tau * tmp;
#pragma unroll
        for (int r = k + 1; r < N; ++r)
          QR(r, j) -= tmp * (tau * QR(r, k));
      }
    }
#pragma unroll
    for (int j = k + 1; j < N; ++j) {
      if (normU[j] != 0.f) {
        float temp = __builtin_fabsf(QR(k, j)) / normU[j];
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        float ratio = normU[j] / normD[j];
        float temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          float s = 0.f;
#pragma unroll
          for (int r = k + 1; r < N; ++r)
            s += QR(r, j) * QR(r, j);
          normD[j] = __builtin_sqrtf(s);
          normU[j] = normD[j];
        } else {
          normU[j] *= __builtin_sqrtf(temp);
        }
      }
    }
  }
  if (nonzero_pivots == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      x[i] = 0.f;
    return;
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
    cv[i] = bin[i];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (k < nonzero_pivots) {
      if (N - k == 1) {
        cv[k] *= 1.f - hc[k];
      } else if (hc[k] != 0.f) {
        float tmp = 0.f;
#pragma unroll
        for (int r = k + 1; r < N; ++r)
          tmp += QR(r, k) * cv[r];
        tmp += cv[k];
        cv[k] -= hc[k] * tmp;
#pragma unroll
        for (int r = k + 1; r < N; ++r)
          cv[r] -= tmp * (hc[k] * QR(r, k));
      }
    }
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    if (i < nonzero_pivots) {
      cv[i] = cv[i] / QR(i, i);
#pragma unroll
      for (int r = 0; r < i; ++r)
        cv[r] -= cv[i] * QR(r, i);
    } else {
      cv[i] = 0.f; // rank-deficient tail: dst rows of the dropped pivots are zero
    }
  }
  // x[perm[i]] = cv[i], perm = product of the transpositions (k, trans[k]), applied on the
  // right in ascending k.  Equivalent, without a dynamically indexed perm[]: start from
  // y = cv and undo the column swaps in descending k.
#pragma unroll
  for (int i = 0; i < N; ++i)
    x[i] = cv[i];
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int j = k + 1; j < N; ++j)
      if (trans[k] == j) {
        float t = x[k];
        x[k] = x[j];
        x[j] = t;
      }
  }
#undef QR
}


// The same factorisation and solve, operation for operation, spread over the 16 lanes of a row
// that all hold the same inputs (the finisher of starved levels): lane j owns COLUMN j - its
// six entries, its norms and its current position in the pivot order - instead of every lane
// carrying the whole matrix through ~3000 instructions.  Column swaps become exchanges of
// positions (no data moves); the pivot scan runs on the norms gathered by position, in the
// reference's order (first strictly larger wins, NaN never wins); the Householder vector, tau
// and the triangular factor travel by ds_bpermute from the lane that owns them.  Every
// arithmetic operation is the one colpiv_qr_solve performs, on the same operands, in the same
// order, so the step is bit-identical (tests: lk_damped_solve with reference_solver = 2 against 1
// on random, rank-deficient and non-finite systems; the finisher's end-to-end bit identity).
template <int N>
__device__ __forceinline__ void colpiv_qr_solve_row16(const float (&M)[N * N], const float (&bin)[N],
                                                      float (&x)[N]) {
  const int lane = (int)threadIdx.x & 63, me = lane & 15, row_base = lane & ~15;
  auto from = [&](float v, int src) { return __shfl(v, row_base | src, 64); }; // lane `src` of the own row
  // my column (lanes >= N carry a dummy column that is never selected)
  float col[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c)
      v = me == c ? M[c * N + r] : v;
    col[r] = v;
  }
  int pos = me;      // position of my column in the pivot order
  int owner[N];      // replicated: lane that owns the column at each position
#pragma unroll
  for (int q = 0; q < N; ++q)
    owner[q] = q;
  const float eps = FLT_EPSILON;
  float s0 = 0.f;
#pragma unroll
  for (int r = 0; r < N; ++r)
    s0 += col[r] * col[r];
  float normD = __builtin_sqrtf(s0), normU = normD;
  float maxn = 0.f;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float nk = from(normU, k);
    if (nk > maxn)
      maxn = nk;
  }
  const float threshold_helper = (maxn * eps) * (maxn * eps) / (float)N;
  const float norm_downdate_threshold = __builtin_sqrtf(eps);
  int nonzero_pivots = N;
  float hc[N], vk[N][N]; // replicated: tau of step k and its Householder vector (rows r > k)
  int trans[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    // pivot scan over positions k..N-1, in order
    int big = k;
    float bigv = from(normU, owner[k]);
#pragma unroll
    for (int j = k + 1; j < N; ++j) {
      const float nj = from(normU, owner[j]);
      if (nj > bigv) {
        bigv = nj;
        big = j;
      }
    }
    const float big_sq = bigv * bigv;
    if (nonzero_pivots == N && big_sq < threshold_helper * (float)(N - k))
      nonzero_pivots = k;
    trans[k] = big;
    // exchange positions k and big
    int owner_big = owner[k];
#pragma unroll
    for (int j = k + 1; j < N; ++j)
      owner_big = big == j ? owner[j] : owner_big;
    const int owner_k = owner[k];
#pragma unroll
    for (int j = k + 1; j < N; ++j)
      owner[j] = big == j ? owner_k : owner[j];
    owner[k] = owner_big;
    if (big != k) {
      if (me == owner_big)
        pos = k;
      else if (me == owner_k)
        pos = big;
    }
    // Householder of the column at position k - every lane works on its own column, the
    // pivot's result is what gets used
    float tailSq = 0.f;
#pragma unroll
    for (int r = k + 1; r < N; ++r)
      tailSq += col[r] * col[r];
    const float c0 = col[k];
    // (both sides of every data-dependent branch of this step are computed and the result selected: the row is
    // bound by the LENGTH of this chain of dependent operations, not by their number, and straight-line code lets the
    // square roots and divisions of the Householder vector, of tau and of the norm down-dating overlap)
    float beta, tau, ess[N];
    {
      const bool tiny = tailSq <= FLT_MIN;
      float b = __builtin_sqrtf(c0 * c0 + tailSq);
      b = c0 >= 0.f ? -b : b;
      const float den = c0 - b;
#pragma unroll
      for (int r = k + 1; r < N; ++r) {
        const float e = col[r] / den;
        ess[r] = tiny ? 0.f : e;
      }
      const float t = (b - c0) / b;
      tau = tiny ? 0.f : t;
      beta = tiny ? c0 : b;
    }
    const int pl = owner[k];
    const bool i_am_pivot = me == pl;
    hc[k] = from(tau, pl);
#pragma unroll
    for (int r = k + 1; r < N; ++r)
      vk[k][r] = from(ess[r], pl);
    if (i_am_pivot) {
      col[k] = beta;
#pragma unroll
      for (int r = k + 1; r < N; ++r)
        col[r] = ess[r];
    }
    const bool later = pos > k && me < N; // my column is still to the right of the pivot
    if (N - k > 1) {
      float tmp = 0.f;
#pragma unroll
      for (int r = k + 1; r < N; ++r)
        tmp += vk[k][r] * col[r];
      tmp += col[k];
      const float ck = col[k] - hc[k] * tmp;
      const bool apply = later && hc[k] != 0.f;
      col[k] = apply ? ck : col[k];
#pragma unroll
      for (int r = k + 1; r < N; ++r) {
        const float cr = col[r] - tmp * (hc[k] * vk[k][r]);
        col[r] = apply ? cr : col[r];
      }
    }
    { // norm down-dating of my column
      float temp = __builtin_fabsf(col[k]) / normU;
      temp = (1.f + temp) * (1.f - temp);
      temp = temp < 0.f ? 0.f : temp;
      const float ratio = normU / normD;
      const float temp2 = temp * (ratio * ratio);
      float s = 0.f;
#pragma unroll
      for (int r = k + 1; r < N; ++r)
        s += col[r] * col[r];
      const float fresh = __builtin_sqrtf(s), scaled = normU * __builtin_sqrtf(temp);
      const bool live = later && normU != 0.f, recompute = temp2 <= norm_downdate_threshold;
      normD = live && recompute ? fresh : normD;
      normU = live ? (recompute ? fresh : scaled) : normU;
    }
  }
  if (nonzero_pivots == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      x[i] = 0.f;
    return;
  }
  float cv[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
    cv[i] = bin[i];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (k < nonzero_pivots) {
      if (N - k == 1) {
        cv[k] *= 1.f - hc[k];
      } else if (hc[k] != 0.f) {
        float tmp = 0.f;
#pragma unroll
        for (int r = k + 1; r < N; ++r)
          tmp += vk[k][r] * cv[r];
        tmp += cv[k];
        cv[k] -= hc[k] * tmp;
#pragma unroll
        for (int r = k + 1; r < N; ++r)
          cv[r] -= tmp * (hc[k] * vk[k][r]);
      }
    }
  }
  // the triangular factor, gathered from the lanes that own its columns before the (sequential) back-substitution
  // needs it: 21 independent permutes in flight at once instead of one round trip per use
  float R[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int r = 0; r <= i; ++r)
      R[i][r] = from(col[r], owner[i]);
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    if (i < nonzero_pivots) {
      cv[i] = cv[i] / R[i][i];
#pragma unroll
      for (int r = 0; r < i; ++r)
        cv[r] -= cv[i] * R[i][r];
    } else {
      cv[i] = 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
    x[i] = cv[i];
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int j = k + 1; j < N; ++j)
      if (trans[k] == j) {
        const float t = x[k];
        x[k] = x[j];
        x[j] = t;
      }
  }
}


// S holds the raw sums (upper triangle row-major, b, chi); p += dp.
template <int P, bool SAFE>
__device__ __forceinline__ bool damped_step(const Sums<P> &S, float lambda, float scaling,
                                            float (&p)[6], bool starved, float *dp_out = nullptr,
                                            bool row16 = false) {
  // U[i][j], i <= j: starts as the scaled, damped upper triangle of A
  float U[P][P], d[P], inv_d[P], y[P], x[P];
  int idx = 0;
  float dmax = 0.f;
#pragma unroll
  for (int i = 0; i < P; ++i) {
#pragma unroll
    for (int j = i; j < P; ++j) {
      float a = S.v[idx++] * scaling; // :647-651
      if (i == j) {
        a *= (1.f + lambda); // :664
        dmax = fmaxf(dmax, a);
      }
      U[i][j] = a;
    }
    y[i] = S.v[Sums<P>::NA + i] * scaling;
  }
  // A pivot d_j is what remains of the diagonal entry after eliminating the earlier
  // parameters; d_j / A_jj is scale free.  Healthy speckle gives >= 0.1; a damped singular
  // system gives ~lambda.  The SAFE flavour switches to the reference's QR below 1e-3; the fast
  // flavour factors through down to LK_FAST_PIVOT and hands the sector to the SAFE kernel below.
  // A starved level (see evaluate<>) always takes the reference's solver: its sums are
  // bit-identical to the reference's, so the whole trajectory is.
  bool well_conditioned = !starved;
  if (!(SAFE && starved)) // (a starved level goes straight to the reference's solver)
#pragma unroll
  for (int j = 0; j < P; ++j) {
    float w[P > 1 ? P : 1];
    const float ajj = U[j][j];
    float dj = ajj;
#pragma unroll
    for (int k = 0; k < j; ++k) {
      w[k] = U[k][j] * d[k];
      dj = __builtin_fmaf(-U[k][j], w[k], dj);
    }
    const bool ok = dj > ajj * (SAFE ? 1e-3f : LK_FAST_PIVOT) && dj > dmax * 1e-7f; // false for NaN as well
    well_conditioned = well_conditioned && ok;
    d[j] = (SAFE || ok) ? dj : 0.f;
    // v_rcp_f32 (1 ulp) instead of a correctly rounded division: this factorisation is not
    // bit-matched to anything, and it saves ~55 instructions per solve
    inv_d[j] = (SAFE || ok) ? __builtin_amdgcn_rcpf(dj) : 0.f; // fast flavour: a bad pivot zeroes that parameter's step
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      float t = U[j][i];
#pragma unroll
      for (int k = 0; k < j; ++k)
        t = __builtin_fmaf(-w[k], U[k][i], t);
      U[j][i] = t * inv_d[j];
    }
  }
  if (!SAFE || well_conditioned) {
    // U^T y' = b (forward), z = y'/d, U x = z (backward)
#pragma unroll
    for (int j = 0; j < P; ++j) {
#pragma unroll
      for (int k = 0; k < j; ++k)
        y[j] = __builtin_fmaf(-U[k][j], y[k], y[j]);
    }
#pragma unroll
    for (int j = P - 1; j >= 0; --j) {
      float t = y[j] * inv_d[j];
#pragma unroll
      for (int i = j + 1; i < P; ++i)
        t = __builtin_fmaf(-U[j][i], x[i], t);
      x[j] = t;
    }
  } else if constexpr (SAFE) {
    // rebuild the damped symmetric matrix exactly as the reference does (:647-665) and
    // solve it the reference's way
    float M[P * P], b[P];
    int q = 0;
#pragma unroll
    for (int p1 = 0; p1 < P; ++p1) {
      b[p1] = S.v[Sums<P>::NA + p1] * scaling;
#pragma unroll
      for (int p2 = p1; p2 < P; ++p2) {
        float a = S.v[q++] * scaling;
        if (p1 == p2)
          a *= (1.f + lambda);
        M[p1 * P + p2] = a;
        M[p2 * P + p1] = a;
      }
    }
    if (row16) // every lane of the 16-lane row holds the same system: spread the QR over them
      colpiv_qr_solve_row16<P>(M, b, x);
    else
      colpiv_qr_solve<P>(M, b, x);
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    p[i] += x[i]; // :687-688
    if (dp_out)
      dp_out[i] = x[i];
  }
  return well_conditioned; // false: a pivot was bad (fast flavour: that parameter's step is zero; SAFE: the QR ran)
}

// translate_model_parameters (pyramid_class.cpp:260-287)
template <int P> __device__ __forceinline__ void translate(float (&p)[6], int src, int dst) {
  float mag = (dst - src > 0) ? 1.f / (float)(1 << (dst - src)) : (float)(1 << (src - dst));
  p[0] *= mag;
  if (P > 1)
    p[1] *= mag;
}

// ------------------------------------------------------------------------------------
// the solve kernel: CorrelationClass::Newton_Raphson (correlation_class.cpp:349-640)
// ------------------------------------------------------------------------------------
// The reference's nested loops (levels x LM trips, one or two evaluations per trip) are
// flattened into a per-sector state machine that performs exactly ONE evaluation + ONE
// damped solve per step, so that the lane groups of a wavefront (GROUP = 16: four sectors
// per wavefront, each on its own DPP row) can be at different points of their own
// trajectories while sharing the instruction stream:
//   EVAL0   evaluation #0 of a level (:410-437)
//   TENT    evaluation of the tentative parameters + look-ahead solve (:503-529)
//   REEVAL  reject path: re-evaluate at last_good with the larger lambda (:475-499)
// GROUP == THREADS (64, 256, 512): one sector per workgroup, control flow is uniform.
// Work distribution.  When there are many more sectors than resident lane groups the launch
// is PERSISTENT: each group pulls the next sector from a device-wide queue (one returning
// atomic per sector) when it has finished its current one, so wavefronts stay full until
// the queue is empty whatever the per-sector iteration counts are.  A group that finds the
// queue empty idles (n = 0) until its wavefront's other groups are done; every wave reaches
// the exit test after each step, so the grid drains.  With fewer than ~2 sectors per
// resident group a queue cannot balance anything (it only leaves half-empty wavefronts
// behind); then one wavefront per 64/GROUP sectors is launched and the hardware dispatcher
// fills freed slots with whole wavefronts (measured on C2: 0.33 vs 0.37 ms).
enum : int { PH_EVAL0 = 0, PH_TENT = 1, PH_REEVAL = 2, PH_FETCH = 3, PH_EXIT = 4, PH_WAIT = 5 }; // PH_WAIT: SEQ instances, see below
constexpr int kStaleIterations = (int)0x80000000; // marker in lk_result.iterations, see lk_stale_iterations_kernel

// Per-sector state that is only touched between evaluations ("cold": last-good parameters,
// damping, counters, ...).  For the small lane groups it lives in LDS - every lane of a group
// reads and writes the same words, in program order within one wavefront - which keeps ~20
// registers out of the sample loop and the kernel at 4 wavefronts per SIMD.  Workgroup-wide
// groups have registers to spare and keep it there.
struct Cold {
  float lg_p[6];
  float lambda, lg_chi, c0x, c0y;
  int iteration, reached, error, level, level_old, s, use_saved;
  uint32_t n_evals, n_sample_evals, n_point_iters;
  uint32_t n_ill; // damped solves that met a bad pivot (well-conditioned speckle: 0)
  int sums_kept;  // the sums of the evaluation at lg_p are in the sector's cache (kernel instances with KEEP_SUMS)
};
constexpr int kColdWords = sizeof(Cold) / 4;
static_assert(kColdWords % 4 == 2 && kColdWords + 7 <= kLkMidWords, "slot = 128-bit words + one 64-bit; the parked-sector record holds it");

// VEC: the slot moves as five 128-bit LDS accesses and a 64-bit one instead of 22 dwords, twice per step (32-lane groups: 32 instructions
// fewer per step; the 16-lane kernel would pay for it with its fourth wavefront per SIMD - 135 VGPRs - and keeps dwords)
template <bool IN_LDS, bool VEC = false> struct ColdStore {
  Cold reg;
  __device__ __forceinline__ Cold load(const uint32_t *slot) const {
    if constexpr (IN_LDS && VEC) {
      Cold k;
      uint4 *w = reinterpret_cast<uint4 *>(&k);
      const uint4 *src = reinterpret_cast<const uint4 *>(slot); // (slots are 16-byte aligned)
#pragma unroll
      for (int i = 0; i < kColdWords / 4; ++i)
        w[i] = src[i];
      *reinterpret_cast<uint2 *>(w + kColdWords / 4) = *reinterpret_cast<const uint2 *>(src + kColdWords / 4);
      return k;
    } else if constexpr (IN_LDS) {
      Cold k;
      uint32_t *w = reinterpret_cast<uint32_t *>(&k);
#pragma unroll
      for (int i = 0; i < kColdWords; ++i)
        w[i] = slot[i];
      return k;
    } else {
      return reg;
    }
  }
  __device__ __forceinline__ void store(uint32_t *slot, const Cold &k) {
    if constexpr (IN_LDS && VEC) {
      const uint4 *w = reinterpret_cast<const uint4 *>(&k);
      uint4 *dst = reinterpret_cast<uint4 *>(slot);
#pragma unroll
      for (int i = 0; i < kColdWords / 4; ++i)
        dst[i] = w[i];
      *reinterpret_cast<uint2 *>(dst + kColdWords / 4) = *reinterpret_cast<const uint2 *>(w + kColdWords / 4);
    } else if constexpr (IN_LDS) {
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&k);
#pragma unroll
      for (int i = 0; i < kColdWords; ++i)
        slot[i] = w[i];
    } else {
      reg = k;
    }
  }
};

#ifndef LK_MIN_WAVES
#define LK_MIN_WAVES 1
#endif
#ifndef LK_MIN_WAVES_512 // wavefronts per SIMD the 512-thread instances must fit (4: two workgroups per CU, 128 VGPRs)
#define LK_MIN_WAVES_512 1
#endif
#ifdef LK_TRACE
// tuning builds only (scripts/tune_build.sh NAME -DLK_TRACE): per-wavefront timeline of a solve launch,
// 8 words per workgroup: start, end, cycles inside evaluate<>, steps, HW_ID, XCC_ID, first sector, -
__device__ unsigned long long g_lk_trace[8 * 16384];
#endif
// REF: the reference-order instances (lk_set_reference_order) - SAFE, one wavefront per workgroup, a 16-lane row
// (four sectors per wavefront, lanes dealt by need: evaluate_ordered_flat) or a wavefront per sector.
// SEQ: the frame-pipelined instances (lk_correlate_sequence*; LkSolveArgs::seq_*).  A sequence with the Eulerian description
// solves the SAME sectors on frame after frame, and the guess of frame f + 1 of a sector depends on that sector's own results
// alone (manager_class.cpp:2677-2686) - nothing in it waits for any other sector of frame f.  One launch per pair therefore
// pays the straggler tail of every pair (a launch lasts as long as its slowest sector while half of the chip idles).  Here
// the tickets of the persistent queue are (frame, sector) pairs of a whole window of resident frames, drawn frame-major, and
// a per-sector chain of 8-byte granules hands the returned parameters from frame to frame (PH_WAIT polls it once per step
// of the wavefront, without holding the other sectors of the wavefront up).  Arithmetic per (frame, sector) is exactly the
// one-pair instances': SAFE flavour (the QR inside the kernel, no second pass), fixed lane groups (no solo / adaptive
// width), and - 16-lane groups, !REF - the starved levels of a sector solved in the same kernel by the finisher's
// arithmetic (reference-order sums by row_newbcast + the QR, bit-identical to the one-lane kernel), chosen per row.
template <int MODEL, int INTERP, int GROUP, int THREADS, bool SAFE, bool REF = false, bool SEQ = false>
__global__ void __launch_bounds__(THREADS, (THREADS == 64 ? (GROUP == 16 && !SAFE && !REF ? 4 : LK_MIN_WAVES) : THREADS == 512 ? LK_MIN_WAVES_512 : 1)) lk_solve_kernel(LkSolveArgs a) {
  static_assert(!REF || (SAFE && ((THREADS == kWave && (GROUP == 16 || GROUP == kWave)) || (THREADS == 512 && GROUP == 512))), "reference-order instances");
  static_assert(!SEQ || (THREADS == kWave && GROUP >= 16 && GROUP <= kWave), "frame-pipelined instances: one wavefront per workgroup");
  constexpr int P = n_params(MODEL);
  using SumsT = Sums<P>;
  constexpr bool COLD_IN_LDS = GROUP > 1 && GROUP <= kWave;
  constexpr bool STARVED = GROUP == 1; // one lane per sector: solves only the starved top levels
  // the 16-lane SAFE instance doubles as the finisher of that kernel's stragglers (a.finisher)
  const bool finisher = !REF && SAFE && GROUP == 16 && a.finisher != 0;
  // reference-order mode (lk_set_reference_order): the REF instances solve every level with the ordered sums
  // + the restated QR - sums, steps and records bit-identical to the CPU engine with
  // number_of_threads = a.reference_order
  constexpr bool ORD = REF;
  constexpr bool ordered_all = REF;
  const bool starved = STARVED || finisher || ordered_all;
  constexpr int RED_WORDS = GROUP > kWave ? (THREADS / kWave) * (SumsT::N + 1) + 1 : 1;
  __shared__ float lds[RED_WORDS];
  constexpr bool FLAT = REF && GROUP == 16; // four sectors per wavefront, lanes dealt by need
  constexpr bool ORD_WG = REF && GROUP == 512; // one big sector per 512-thread workgroup: seven wavefronts form the products, one adds
  __shared__ __attribute__((aligned(16))) float ord_lds[FLAT ? flat_tile_floats<SumsT::N>() : ORD_WG ? wg_ord_floats<SumsT::N>() : ORD ? ord_floats<SumsT::N, (ORD && !ORD_WG) ? GROUP : 16>() : 4];
  __shared__ __attribute__((aligned(16))) OrdCtx ord_ctx[FLAT ? 4 : 1];
  // The sums of the last ACCEPTED evaluation stay with the sector (small groups: behind its cold state in LDS; one
  // lane per sector: in registers).  A rejected trip continues from the last good parameters with a larger lambda;
  // the reference evaluates there once more (correlation_class.cpp:441-499) and gets, of course, the sums it had
  // before.  With the sums kept the rejected trip is just another solve - from THOSE sums - inside the step that
  // rejected: one evaluation fewer per rejection (config 4: 5.0 of 31.1 evaluations per sector, config 5: 6.1 of
  // 35.7), same bits.  (After a sector changed hands in the middle of a level - parked and resumed by another
  // launch - the cache is empty and the first rejection takes the re-evaluation, as the large groups always do.)
  constexpr bool KEEP_SUMS = GROUP == 1 || GROUP == 16;
  constexpr int kSlotWordsRaw = kColdWords + (KEEP_SUMS && COLD_IN_LDS ? SumsT::N : 0);
  constexpr int kSlotWords = GROUP == 32 ? (kSlotWordsRaw + 3) & ~3 : kSlotWordsRaw; // (128-bit accesses: 16-byte slots)
  __shared__ __attribute__((aligned(16))) uint32_t cold_lds[COLD_IN_LDS ? (THREADS / GROUP) * kSlotWords : 4];
  uint32_t *cold_slot = cold_lds + (COLD_IN_LDS ? ((int)threadIdx.x / GROUP) * kSlotWords : 0);
  ColdStore<COLD_IN_LDS, GROUP == 32> cold;
  // (one lane per sector: a column of LDS per lane - 28 more registers would cost the kernel a wavefront per SIMD,
  // and config 5's 3100 wavefronts of it a second round)
  __shared__ float kept_lds[KEEP_SUMS && !COLD_IN_LDS ? SumsT::N * THREADS : 1];
  auto keep_sums = [&](const SumsT &v) {
    if constexpr (KEEP_SUMS && COLD_IN_LDS) {
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i)
        cold_slot[kColdWords + i] = __float_as_uint(v.v[i]);
    } else if constexpr (KEEP_SUMS) {
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i)
        kept_lds[i * THREADS + (int)threadIdx.x] = v.v[i];
    }
  };
  auto kept_sums = [&](SumsT &v) {
    if constexpr (KEEP_SUMS && COLD_IN_LDS) {
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i)
        v.v[i] = __uint_as_float(cold_slot[kColdWords + i]);
    } else if constexpr (KEEP_SUMS) {
#pragma unroll
      for (int i = 0; i < SumsT::N; ++i)
        v.v[i] = kept_lds[i * THREADS + (int)threadIdx.x];
    }
  };

  // the per-level pointer table in LDS: a level change then costs ONE global round trip (the sector's rectangle /
  // list offsets) instead of two (the table entry of its level first)
  __shared__ LkLevelView lv_lds[LK_MAX_LEVELS];
  {
    constexpr int WORDS = (int)(sizeof(LkLevelView) / 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.lv);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lv_lds);
    for (int i = (int)threadIdx.x; i < WORDS * (a.py_stop + 1); i += THREADS)
      dst[i] = src[i];
    __syncthreads();
  }
  float p[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
    p[i] = 0.f;
  const float min_lambda = 1e-9f, max_lambda = 1e9f;
  bool first_fetch = true;
  int phase = PH_FETCH;
  int cur_level = 0; // register copy of the cold state's level (alignment of the groups of a wavefront)
  LevelCtx c{};
  SumsT S;
  bool wide = false; // GROUP == 32 only: both halves of the wavefront work on one sector
  int width = 16;    // GROUP == 16 only: lanes per sector right now (16 / 32 / 64), wavefront-uniform
  int cur_frame = 0; // SEQ: the frame (of the window) this group's sector is being solved on
  uint32_t wait_spins = 0; // SEQ: rounds in which every group of the wavefront was waiting for its sector's previous frame
  // SEQ, 16-lane rows, not the reference-order instance: a level with at most starved_max samples is solved with the
  // finisher's arithmetic, row by row (the one-pair chain: one-lane kernel -> finisher -> lane groups, in one kernel)
  constexpr bool UNIFIED = SEQ && !REF && SAFE && GROUP == 16;
  TeamCtx team;
  if constexpr (GROUP == 512) {
    if (a.team_w > 1) {
      team.w = a.team_w;
      team.rank = (int)blockIdx.x % a.team_w;
      team.slot = (int)blockIdx.x / a.team_w;
      team.partials = a.team_partials;
      team.arrivals = a.team_arrivals;
      team.n_slots = a.n_sectors;
      team.fault = a.team_fault;
      team.stride = a.team_w;
      // the team of a sector is as wide as its sample count is worth; the other workgroups
      // of its slot leave at once (uniform over the workgroup, before any barrier)
      if (a.team_min_samples > 0 && team.slot < a.n_sectors) {
        const int sector = a.order ? (int)a.order[team.slot] : team.slot;
        const LkLevelView lv0 = lv_lds[0];
        const int4 rc = lv0.rect[sector];
        const int n0 = rc.z > 0 ? rc.w : (int)(lv0.off[sector + 1] - lv0.off[sector]);
        int w = (n0 + a.team_min_samples - 1) / a.team_min_samples;
        w = w < 1 ? 1 : (w > a.team_w ? a.team_w : w);
        if (team.rank >= w)
          return;
        team.w = w;
      }
    }
  }

  auto level_count = [&](int level, int sector) -> int { // samples of a sector at a level
    const LkLevelView lv = lv_lds[level];
    const int4 rc = lv.rect[sector];
    return rc.z > 0 ? rc.w : (int)(lv.off[sector + 1] - lv.off[sector]);
  };

  auto hand_over = [&](const Cold &k) { // STARVED kernel: leave the rest to the lane groups
    LkHandoff h;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      h.p[i] = i < P ? p[i] : 0.f;
    h.level = k.level;
    h.level_old = k.level_old;
    h.reached = k.reached;
    h.n_evals = k.n_evals;
    h.n_sample_evals = k.n_sample_evals;
    h.n_point_iters = k.n_point_iters;
    a.handoff[k.s] = h;
    phase = PH_FETCH;
  };

  auto level_context = [&](const Cold &k) { // what the lanes need to know about the sector at k.level
    const LkLevelView lv = lv_lds[k.level];
    cur_level = k.level;
    const uint32_t off = lv.off[k.s];
    const int4 rc = lv.rect[k.s];
    c.rx = rc.x;
    c.ry = rc.y;
    c.rw = rc.z;
    c.n = rc.z > 0 ? rc.w : (int)(lv.off[k.s + 1] - off);
    if constexpr (SEQ) { // the images of this sector's frame (same geometry, same lists)
      c.und = (gptr<uint8_t>)a.seq_img[cur_frame].und[k.level];
      c.def = (gptr<uint8_t>)a.seq_img[cur_frame].def[k.level];
    } else {
      c.und = (gptr<uint8_t>)lv.und;
      c.def = (gptr<uint8_t>)lv.def;
    }
    const bool level_ordered = starved || (UNIFIED && c.n <= a.starved_max); // sums in the reference's order: its list
    c.xy = (gptr<f32x2>)((!level_ordered && lv.xy_eval ? lv.xy_eval : lv.xy) + off); // (unordered sums: the row-major copy)
    c.urows = lv.urows;
    c.ucols = lv.ucols;
    c.drows = lv.drows;
    c.dcols = lv.dcols;
    c.scaling = 1.f / ((float)c.n);
    if constexpr (GROUP != 16)
      c.inv_w = c.rw > 0 ? 1.f / (float)c.rw : 0.f;
    const float inv = 1.f / (float)(1 << k.level); // pyramid_class.cpp:357-361
    c.cx = k.level == 0 ? k.c0x : k.c0x * inv;
    c.cy = k.level == 0 ? k.c0y : k.c0y * inv;
  };

  auto enter_level = [&](Cold &k) { // top of the level loop (:373-408)
    translate<P>(p, k.level_old, k.level);
    k.error = LK_ERROR_NONE;
    k.lambda = 0.0001f;
    k.lg_chi = FLT_MAX;
    k.sums_kept = 0;
    level_context(k);
#pragma unroll
    for (int i = 0; i < P; ++i)
      k.lg_p[i] = p[i];
    phase = PH_EVAL0;
  };

  // STARVED kernel, a.eval_cap > 0: a lane that has used up its evaluations parks the sector
  // in the middle of its level for the 16-lane finisher (the 63 other lanes of the wavefront
  // are not kept waiting for one sector that runs into max_iters)
  auto park = [&](const Cold &k, uint32_t *list, uint32_t *count) {
    uint32_t *m = a.mid_state + (size_t)k.s * kLkMidWords;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&k);
#pragma unroll
    for (int i = 0; i < kColdWords; ++i)
      m[i] = w[i];
#pragma unroll
    for (int i = 0; i < 6; ++i)
      m[kColdWords + i] = __float_as_uint(p[i]);
    m[kColdWords + 6] = (uint32_t)phase;
    // one list entry per sector: the first lane of the (possibly widened) group, of team rank 0
    const int lanes = GROUP == 32 ? (wide ? 64 : 32) : GROUP == 16 ? width : GROUP;
    if (((int)threadIdx.x & (lanes - 1)) == 0 && team.rank == 0)
      list[atomicAdd(count, 1u)] = (uint32_t)k.s;
    phase = PH_FETCH;
  };
  int steps = 0; // evaluations of the current sector in this kernel (STARVED: eval_cap)

  auto finish_sector = [&](const Cold &k, const float (&evaluated)[6]) { // results of Newton_Raphson (:638-639, :848-870)
    if ((int)threadIdx.x % GROUP == 0 && team.rank == 0) {
      lk_result r;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        r.resultingParameters[i] = i < P ? p[i] : 0.f;
      r.chi = k.lg_chi;
      const int4 rc0 = lv_lds[0].rect[k.s];
      r.numberOfPoints = rc0.z > 0 ? rc0.w : (int)(lv_lds[0].off[k.s + 1] - lv_lds[0].off[k.s]);
      // reference-order mode: the very first evaluation of the sector failed, so no LM trip ever ran -
      // the reference then reports whatever its reached_iterations member still holds from the sector
      // before (correlation_class.cpp:413-419, :870); lk_stale_iterations_kernel fills that in
      // (mark_stale < 0: a default-mode launch that borrows the reference-order instance - it reports 0 like the default instances)
      r.iterations = (((ordered_all && a.mark_stale == 0) || a.mark_stale > 0) && k.n_evals == 1u && k.error != LK_ERROR_NONE && k.lg_chi == FLT_MAX && k.reached == 0)
                         ? kStaleIterations
                         : k.reached;
      r.errorCode = k.error;
      r.undCenterX = k.c0x;
      r.undCenterY = k.c0y;
      // SEQ: one record and one stats row per frame; the sequence state (last_p, last_eval_p) is written by the
      // window's LAST frame only - the frames of a sector run on whatever CU drew them, and two XCDs' write-back
      // copies of one address have no order
      const size_t ridx = SEQ ? (size_t)cur_frame * (size_t)a.seq_stride + (size_t)k.s : (size_t)k.s;
      const bool state_out = !SEQ || cur_frame == a.seq_frames - 1;
      a.result[ridx] = r;
      if (a.last_p && state_out) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
          a.last_p[(size_t)k.s * 6 + i] = r.resultingParameters[i];
      }
      if (a.last_eval_p && state_out) { // def_xy_positions of the last level-0 evaluation (getDefXY0, :884-896);
                                        // with py_start > 0 the reference re-applies the returned parameters
#pragma unroll
        for (int i = 0; i < 6; ++i)
          a.last_eval_p[(size_t)k.s * 6 + i] = a.py_start == 0 ? evaluated[i] : r.resultingParameters[i];
      }
      if (a.stats) {
        a.stats[ridx * 4 + 0] = k.n_evals;
        a.stats[ridx * 4 + 1] = k.n_sample_evals;
        a.stats[ridx * 4 + 2] = k.n_point_iters;
        a.stats[ridx * 4 + 3] = k.n_ill;
      }
      if constexpr (SEQ) {
        // hand the returned parameters to the sector's next frame: granules {frame + 1, bits}, write-through, the tag
        // IS the flag (lk_device.hpp: kLkSeqChainWords)
        unsigned long long *ch = a.seq_chain + (size_t)k.s * kLkSeqChainWords + (size_t)(cur_frame & 1) * 8;
        const unsigned long long tag = (unsigned long long)(uint32_t)(cur_frame + 1) << 32;
        const bool lost = a.seq_fault != 0 && cur_frame == a.seq_fault - 1 && k.s == (a.order ? (int)a.order[0] : 0); // (test hook)
        if (!lost)
#pragma unroll
        for (int i = 0; i < P; ++i)
          __hip_atomic_store(ch + i, tag | (unsigned long long)__float_as_uint(r.resultingParameters[i]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      if (starved && a.handoff) {
        LkHandoff h{};
        h.level = a.py_start - 1; // finished
        a.handoff[k.s] = h;
      }
    }
    phase = PH_FETCH;
  };

#ifdef LK_TRACE
#ifndef LK_TRACE_PICK // which instance leaves the trace, e.g. -D'LK_TRACE_PICK(G,S)=(G==1)'
#define LK_TRACE_PICK(G, S) ((G) > 1 && !(S))
#endif
#ifndef LK_TRACE_WHEN // ... and which of its launches, e.g. -D'LK_TRACE_WHEN(a)=((a).finisher!=0)'
#define LK_TRACE_WHEN(a) true
#endif
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz, one clock for the whole device
  const unsigned long long tr_c0 = __builtin_amdgcn_s_memtime();
  unsigned long long tr_eval = 0, tr_steps = 0, tr_solve = 0, tr_fetch = 0, tr_post = 0;
  unsigned long long tr_idle = 0, tr_sleeps = 0; // SEQ: lane groups without work summed over the steps; all-waiting rounds
#endif
  for (;;) {
    bool may_fetch = true;
    if constexpr (GROUP > 1 && GROUP < kWave) {
      // aligned wavefronts fetch together: a group that finished early waits for its
      // neighbours, so that the next batch of sectors starts the pyramid in step
      if (a.align && a.persistent && !FLAT) // (FLAT: lanes are dealt by need, every sector runs at its own pace)
        may_fetch = __ballot(phase < PH_FETCH) == 0ull;
    }
    if constexpr (GROUP == 16) {
      // a batch of an aligned persistent wavefront is over: back to one sector per row
      if (width != 16 && a.align && a.persistent && may_fetch) {
        width = 16;
        cold_slot = cold_lds + ((int)threadIdx.x / GROUP) * kSlotWords;
      }
    }
#ifdef LK_TRACE_FINE
    const unsigned long long tr_f0 = __builtin_amdgcn_s_memtime();
#endif
    if (phase == PH_FETCH && may_fetch) // take the next sector
     for (;;) { // (drawn sectors that need nothing - finished by an earlier launch of the chain - are skipped here, see the end of the block)
      int slot = 0;
      // (lists of parked sectors: their length is read BEFORE the ticket is drawn - see the rewind below)
      const int n_parked = (finisher || a.resume) ? (int)*(const volatile uint32_t *)a.finish_count : 0;
      if (!a.persistent) { // one sector per group, handed out by position (see launch_solve_g)
        // workgroups are dealt round-robin over the 8 XCDs: give each XCD one contiguous run
        // of sectors (neighbouring sectors share image rows in its L2)
        const int wg_slot = ((int)blockIdx.x & 7) * a.chunk + ((int)blockIdx.x >> 3);
        slot = first_fetch ? wg_slot * (THREADS / GROUP) + (int)threadIdx.x / GROUP : a.n_sectors;
        if constexpr (FLAT) { // (fewer sectors per wavefront: the rows without one lend their lanes from the start)
          if (a.rows_used > 0 && a.rows_used < 4)
            slot = (first_fetch && (int)threadIdx.x / GROUP < a.rows_used) ? wg_slot * a.rows_used + (int)threadIdx.x / GROUP : a.n_sectors;
        }
        if (GROUP == 512 && a.team_w > 1) // a team's workgroups all take the team's sector
          slot = first_fetch ? team.slot : a.n_sectors;
        first_fetch = false;
      } else if constexpr (GROUP > kWave) {
        __syncthreads(); // everybody is done reading lds from the last evaluation
        if (threadIdx.x == 0)
          reinterpret_cast<int *>(lds)[RED_WORDS - 1] = (int)atomicAdd(a.queue, 1u);
        __syncthreads();
        slot = reinterpret_cast<int *>(lds)[RED_WORDS - 1];
      } else {
        if ((int)threadIdx.x % GROUP == 0)
          slot = (int)atomicAdd(a.queue, 1u);
        slot = __shfl(slot, ((int)threadIdx.x & 63) & ~(GROUP - 1), 64);
      }
      steps = 0;
      if constexpr (SEQ) { // ticket -> (frame, sector), frame-major; the solve starts when the sector's previous frame is in
        if (slot < a.n_sectors * a.seq_frames) {
          Cold k{};
          cur_frame = slot / a.n_sectors; // (once per sector and frame)
          const int pos = slot - cur_frame * a.n_sectors;
          k.s = a.order ? (int)a.order[pos] : pos;
          const float2 c0 = a.center[k.s];
          k.c0x = c0.x;
          k.c0y = c0.y;
          k.use_saved = 1;
          k.level = a.py_stop;
          k.level_old = 0;
          cold.store(cold_slot, k);
          phase = PH_WAIT;
        } else {
          phase = PH_EXIT;
        }
      } else if (finisher || a.resume) { // resume a sector an earlier launch parked in the middle of a level
        if (a.resume && !finisher && (int)threadIdx.x % GROUP == 0 &&
            slot == n_parked + (int)gridDim.x * (THREADS / GROUP) - 1) {
          // every group draws exactly one ticket past the end of the list; whoever draws the last
          // of those rewinds list and queue for the next solve (no memsets between the launches)
          *a.finish_count = 0u;
          *a.queue = 0u;
        }
        if (slot < n_parked) {
          Cold k;
          const int s_idx = (int)a.finish_list[slot];
          const uint32_t *m = a.mid_state + (size_t)s_idx * kLkMidWords;
          uint32_t *w = reinterpret_cast<uint32_t *>(&k);
#pragma unroll
          for (int i = 0; i < kColdWords; ++i)
            w[i] = m[i];
#pragma unroll
          for (int i = 0; i < 6; ++i)
            p[i] = __uint_as_float(m[kColdWords + i]);
          phase = (int)m[kColdWords + 6];
          k.sums_kept = 0; // (the sums did not travel with the parked sector)
          level_context(k);
          cold.store(cold_slot, k);
        } else {
          phase = PH_EXIT;
        }
      } else if (slot < a.n_sectors) {
        Cold k{};
        k.s = a.order ? (int)a.order[slot] : slot;
        const float2 c0 = a.center[k.s];
        k.c0x = c0.x;
        k.c0y = c0.y;
        k.use_saved = 1;
        if (!STARVED && a.handoff) { // continue where the starved-level kernel stopped
          const LkHandoff h = a.handoff[k.s];
#pragma unroll
          for (int i = 0; i < 6; ++i)
            p[i] = i < P ? h.p[i] : 0.f;
          k.level = h.level;
          k.level_old = h.level_old;
          k.reached = h.reached;
          k.n_evals = h.n_evals;
          k.n_sample_evals = h.n_sample_evals;
          k.n_point_iters = h.n_point_iters;
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i)
            p[i] = i < P ? a.guess[(size_t)k.s * 6 + i] : 0.f;
          k.level = a.py_stop;
          k.level_old = 0;
        }
        if (k.level < a.py_start) {
          // finished by the starved-level kernel (result already written): take another (below)
          if (!a.persistent || STARVED)
            phase = PH_EXIT; // (one sector per group by position: nothing else to take)
        } else if (STARVED && level_count(k.level, k.s) > a.starved_max) {
          hand_over(k); // nothing starved here: the lane-group kernel does it all
        } else {
          enter_level(k);
          cold.store(cold_slot, k);
        }
      } else {
        phase = PH_EXIT;
      }
      // Every group that fetched here drew a sector that needs nothing: they draw again, TOGETHER - the groups of a wavefront
      // keep drawing consecutive tickets, so that which sectors share a wavefront (adaptive width: the bits of the default
      // mode) does not depend on timing.  A group that drew nothing while a neighbour got work sits this batch out.  No
      // wavefront leaves while tickets remain: it leaves only once one of its groups has seen the end of the queue.
      const bool all_drew_nothing = __ballot(phase != PH_FETCH) == 0ull; // (a statement of its own: every fetching lane takes part)
      if (phase == PH_FETCH && all_drew_nothing)
        continue;
      break;
     }
#if defined(LK_TRACE_FINE) && !defined(LK_TRACE_TRANS)
    tr_fetch += __builtin_amdgcn_s_memtime() - tr_f0;
#endif
    if constexpr (SEQ) {
      if (phase == PH_WAIT) {
        // Has frame cur_frame - 1 of this sector published its parameters?  One look per step of the wavefront: the
        // other groups of the wavefront go on solving meanwhile.  r = p(f-1), q = p(f-2) (frame 1 of the window: the
        // previous_resulting_parameters the host-side guess of frame 0 left); the guess is lk_guess_kernel's,
        // operation for operation (manager_class.cpp:2677-2699).
        Cold k = cold.load(cold_slot);
        bool ready = true;
        float g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
          g[i] = 0.f;
        if (cur_frame == 0) {
#pragma unroll
          for (int i = 0; i < P; ++i)
            g[i] = a.guess[(size_t)k.s * 6 + i];
        } else {
          const unsigned long long *ch = a.seq_chain + (size_t)k.s * kLkSeqChainWords;
          const unsigned long long *chr = ch + (size_t)((cur_frame - 1) & 1) * 8, *chq = ch + (size_t)(cur_frame & 1) * 8;
          float r[P], q[P];
#pragma unroll
          for (int i = 0; i < P; ++i) {
            const unsigned long long x = __hip_atomic_load(chr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ready = ready && (uint32_t)(x >> 32) == (uint32_t)cur_frame;
            r[i] = __uint_as_float((uint32_t)x);
            q[i] = r[i];
          }
          if (a.seq_velocity) {
            if (cur_frame == 1) {
#pragma unroll
              for (int i = 0; i < P; ++i)
                q[i] = a.seq_prev_p[(size_t)k.s * 6 + i];
            } else {
#pragma unroll
              for (int i = 0; i < P; ++i) {
                const unsigned long long x = __hip_atomic_load(chq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = ready && (uint32_t)(x >> 32) == (uint32_t)(cur_frame - 1);
                q[i] = __uint_as_float((uint32_t)x);
              }
            }
          }
#pragma unroll
          for (int i = 0; i < P; ++i)
            g[i] = a.seq_velocity ? r[i] + (r[i] - q[i]) : r[i];
          // previous_resulting_parameters after the window = p of its last frame but one (what the next guess reads)
          if (ready && cur_frame == a.seq_frames - 1 && a.seq_prev_p_out && (int)threadIdx.x % GROUP == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
              a.seq_prev_p_out[(size_t)k.s * 6 + i] = i < P ? r[i] : 0.f;
          }
        }
        if (ready) {
#pragma unroll
          for (int i = 0; i < 6; ++i)
            p[i] = g[i];
          if (a.seq_guess_out && (int)threadIdx.x % GROUP == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
              a.seq_guess_out[((size_t)cur_frame * (size_t)a.seq_stride + (size_t)k.s) * 6 + i] = g[i];
          }
          enter_level(k);
          cold.store(cold_slot, k);
        }
      }
    }
    bool active = phase < PH_FETCH;
    if constexpr (SEQ) {
      // Nobody in this wavefront can go on: every group waits for a frame another wavefront is solving.  Sleep and look
      // again; bounded - a wait that never ends (it cannot: every awaited ticket was drawn by a running wavefront) or
      // another wavefront's give-up voids the launch (seq_flags[0]) and lets the grid drain.
      const unsigned long long waiting = __ballot(phase == PH_WAIT);
      if (__ballot(active) == 0ull && waiting != 0ull) {
        ++wait_spins;
        const bool give_up = wait_spins > (1u << 20);
        if (give_up || (wait_spins & 255u) == 0u) {
          if (give_up || __hip_atomic_load(a.seq_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (((int)threadIdx.x & 63) == 0)
              __hip_atomic_store(a.seq_flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
          }
        }
#ifdef LK_TRACE
        ++tr_sleeps;
#endif
        __builtin_amdgcn_s_sleep(16);
        continue;
      }
      if (waiting != ~0ull)
        wait_spins = 0;
#ifdef LK_TRACE
      tr_idle += (unsigned long long)(__builtin_popcountll(~__ballot(active)) / GROUP);
#endif
    }
    if constexpr (GROUP >= kWave) {
      if (!active)
        break; // uniform over the workgroup: nothing left to do
    } else {
      const unsigned long long act = __ballot(active);
      if (act == 0ull)
        break; // every group of this wavefront is out of work (a group that drew a sector needing nothing drew again at once:
               // a wavefront must not leave while tickets remain - when most sectors of a launch need nothing the others
               // would not come to draw them)
      if constexpr (GROUP == 32) {
        // Solo: one half-wavefront is out of work for good (PH_EXIT) while its partner is still
        // solving - typically a sector that needs many more evaluations than its neighbour.
        // The idle half adopts the partner's sector: it copies the warm state (parameters,
        // level context, phase), shares the partner's cold slot, and from here on both halves
        // run the identical state machine on identical sums while every evaluation uses all
        // 64 lanes.  Duplicate stores (cold slot, final record) write identical values.
        const bool lo = (uint32_t)act != 0u, hi = (uint32_t)(act >> 32) != 0u;
        if (a.solo && !wide && lo != hi && __ballot(phase == PH_EXIT) == ~act) {
          auto take = [&](auto &v) {
            const auto other = __shfl_xor(v, 32, 64);
            if (!active)
              v = other;
          };
#pragma unroll
          for (int i = 0; i < 6; ++i)
            take(p[i]);
          take(c.rx), take(c.ry), take(c.rw), take(c.n);
          take(c.urows), take(c.ucols), take(c.drows), take(c.dcols);
          take(c.scaling), take(c.cx), take(c.cy), take(c.inv_w);
          auto take_ptr = [&](auto &ptr) {
            unsigned long long bits = (unsigned long long)(uintptr_t)ptr;
            uint32_t lo32 = (uint32_t)bits, hi32 = (uint32_t)(bits >> 32);
            take(lo32), take(hi32);
            ptr = (decltype(+ptr))(uintptr_t)(((unsigned long long)hi32 << 32) | lo32);
          };
          take_ptr(c.und), take_ptr(c.def), take_ptr(c.xy);
          take(phase);
          take(cur_level);
          if (!active)
            cold_slot = cold_lds + (((int)threadIdx.x / GROUP) ^ 1) * kSlotWords;
          active = true;
          wide = true;
        }
      }
    }
    if constexpr (GROUP == 16) {
      // Adaptive width: rows whose sector is finished (out of work for good, or waiting for the
      // common fetch of an aligned persistent wavefront) join the sectors still being solved -
      // 4 x 16 lanes become 2 x 32 or 1 x 64.  The joining lanes copy the warm state of the
      // sector's home row and share its cold slot; all lanes of a widened group then run the
      // identical state machine on identical sums.  It shortens exactly the part of a launch
      // that is otherwise spent waiting for the slowest sectors.
      const unsigned long long act = __ballot(active);
      const bool idle_ok = (a.align && a.persistent) || __ballot(phase == PH_EXIT) == ~act;
      if (a.solo && width < 64 && idle_ok && !finisher && !ordered_all) {
        // home rows of the sectors in progress (a widened group shows up in all its rows)
        const unsigned rows = ((act & 0xffffull) ? 1u : 0u) | ((act & 0xffff0000ull) ? 2u : 0u) |
                              ((act & 0xffff00000000ull) ? 4u : 0u) | ((act & 0xffff000000000000ull) ? 8u : 0u);
        const unsigned sect = width == 16 ? rows : ((rows & 3u) ? 1u : 0u) | ((rows & 12u) ? 4u : 0u);
        const int n_sect = __builtin_popcount(sect);
        const int new_width = n_sect == 1 ? 64 : (n_sect == 2 && width == 16) ? 32 : width;
        if (new_width != width) {
          const int first = __builtin_ctz(sect), last = 31 - __builtin_clz(sect);
          const int lane = (int)threadIdx.x & 63;
          const int src_row = (new_width == 64 || lane < 32) ? first : last;
          const int src_lane = src_row * 16 + (lane & 15);
          auto take = [&](auto &v) { v = __shfl(v, src_lane, 64); };
#pragma unroll
          for (int i = 0; i < 6; ++i)
            take(p[i]);
          take(c.rx), take(c.ry), take(c.rw), take(c.n);
          take(c.urows), take(c.ucols), take(c.drows), take(c.dcols);
          take(c.scaling), take(c.cx), take(c.cy); // (c.inv_w: not used by 16-lane groups)
          auto take_ptr = [&](auto &ptr) {
            unsigned long long bits = (unsigned long long)(uintptr_t)ptr;
            uint32_t lo32 = (uint32_t)bits, hi32 = (uint32_t)(bits >> 32);
            take(lo32), take(hi32);
            ptr = (decltype(+ptr))(uintptr_t)(((unsigned long long)hi32 << 32) | lo32);
          };
          take_ptr(c.und), take_ptr(c.def), take_ptr(c.xy);
          take(phase);
          take(cur_level);
          // the sector keeps the cold slot it was fetched into
          int slot_idx = (int)(cold_slot - cold_lds);
          take(slot_idx);
          cold_slot = cold_lds + slot_idx;
          active = phase < PH_FETCH;
          width = new_width;
        }
      }
    }
    if constexpr (GROUP > 1 && GROUP < kWave) {
      // Level alignment: the trip count of an evaluation is set by the group with the most
      // samples, so a wavefront whose groups sit at different pyramid levels pays the finest
      // level's price for every step.  A group that is ahead (at a finer level than the
      // coarsest one still being solved in its wavefront) skips steps until the others have
      // caught up; then all of them walk the expensive fine level together.
      if (a.align && !FLAT) {
        int wave_level = -1;
        for (int L = a.py_stop; L >= a.py_start && wave_level < 0; L -= a.py_step)
          if (__ballot(active && cur_level == L) != 0ull)
            wave_level = L;
        if (active && cur_level < wave_level)
          active = false; // hold: no evaluation, no state change in this step
      }
    }
    LevelCtx ce = c;
    if (!active)
      ce.n = 0;
    // UNIFIED: rows at a starved level take the finisher's evaluation (ordered sums), the others the lane-parallel one;
    // rows without work follow whatever the working rows do (both paths cost their instructions once per wavefront)
    bool row_ordered = finisher;
    if constexpr (UNIFIED) {
      const bool ord = active && c.n <= a.starved_max;
      const bool any_plain = __ballot(active && !ord) != 0ull;
      row_ordered = active ? ord : !any_plain;
    }
#ifdef LK_TRACE
    const unsigned long long tr_e0 = __builtin_amdgcn_s_memtime();
#endif
    bool err;
    if constexpr (FLAT) {
      err = a.reference_order > 1 ? evaluate_ordered_flat<MODEL, INTERP, true>(ce, p, S, ord_lds, ord_ctx, a.reference_order)
                                  : evaluate_ordered_flat<MODEL, INTERP, false>(ce, p, S, ord_lds, ord_ctx, 1);
    } else if constexpr (ORD_WG) {
      if (team.w > 1) {
        // A team in reference-order mode: its T = number_of_threads workgroups each solve ONE thread chunk of the
        // reference (chunk t: n/T + (t < n%T) samples, correlation_class.cpp:169-186) as one ordered chain, and the
        // all-to-all joins the chunk sums in thread order into zeroed totals (:253-275) - the reference's own
        // parallelism, 20 chains side by side by default.  (A broken team: rank 0 alone walks all chunks, below.)
        const int T = team.w, cq = ce.n / T, cr = ce.n - cq * T;
        const int first = team.rank * cq + (team.rank < cr ? team.rank : cr), count = cq + (team.rank < cr ? 1 : 0);
        err = evaluate_ordered_wg<MODEL, INTERP, false>(ce, p, S, ord_lds, 1, first, count);
        err = team_all_to_all(S, err, lds, &team);
      } else {
        err = a.reference_order > 1 ? evaluate_ordered_wg<MODEL, INTERP, true>(ce, p, S, ord_lds, a.reference_order, 0, ce.n)
                                    : evaluate_ordered_wg<MODEL, INTERP, false>(ce, p, S, ord_lds, 1, 0, ce.n);
      }
    } else if constexpr (ORD) {
      err = a.reference_order > 1 ? evaluate_ordered<MODEL, INTERP, GROUP, true>(ce, p, S, ord_lds, a.reference_order)
                                  : evaluate_ordered<MODEL, INTERP, GROUP, false>(ce, p, S, ord_lds, 1);
    } else {
      err = evaluate<MODEL, INTERP, GROUP, THREADS>(ce, p, S, lds, &team, wide, row_ordered, GROUP == 16 ? width : 0);
    }
#ifdef LK_TRACE
    tr_eval += __builtin_amdgcn_s_memtime() - tr_e0;
    ++tr_steps;
#endif
#ifdef LK_TRACE_FINE
    const unsigned long long tr_p0 = __builtin_amdgcn_s_memtime(); // (tr_post includes the solve)
#endif
    if constexpr (GROUP == 512) {
      // A team whose workgroups did not all show up (a workgroup not resident for ~1 s: foreign
      // kernels holding the GPU) is broken for good: its sums are incomplete.  Everybody but
      // rank 0 leaves; rank 0 starts the sector again from its guess as a lone workgroup - the
      // record is then the single-workgroup solve's, late instead of an error.
      if (team.timed_out && team.w > 1) {
        if (team.rank != 0)
          return;
        team.w = 1;
        team.timed_out = false;
        phase = PH_FETCH;
        first_fetch = true;
        continue;
      }
    }
    if (active) {
      Cold k = cold.load(cold_slot);
      float evaluated[6]; // the parameters this evaluation ran at (rescaled to level 0 only if the sector ends here)
#pragma unroll
      for (int i = 0; i < 6; ++i)
        evaluated[i] = p[i];
      const int evaluated_level = k.level;
      ++k.n_evals;
      k.n_sample_evals += (uint32_t)c.n;
      bool level_end = false, iter_start = false, finished = false, handed = false;
      if (err) { // :413-419 (evaluation #0: return at once), :484-489, :511-516 (break)
        k.error = LK_ERROR_INTERPOLATION_OUT_OF_IMAGE;
        if (phase == PH_EVAL0) {
          ++k.n_point_iters;
          translate<P>(p, k.level, 0);
          finished = true;
        } else {
          level_end = true;
        }
      } else {
        const float chi = S.v[SumsT::N - 1] * c.scaling;
        float lam_use = phase == PH_TENT ? fmaxf(k.lambda * 0.4f, min_lambda) : k.lambda;
        // KEEP_SUMS: a tentative evaluation that is rejected, has not converged and may go on (the tests of the PH_TENT
        // branch and of iter_start below, taken before the solve: they need chi only) continues at once from the
        // kept sums of the last good parameters with the larger lambda - the bookkeeping of the rejection, of the next
        // trip's start and of the reference's second evaluation there, whose sums are the kept ones, in one step.
        bool redo = false;
        if constexpr (KEEP_SUMS) {
          if (phase == PH_TENT && k.sums_kept != 0 && !(chi <= k.lg_chi)) {
            const float delta_chi = __builtin_fabsf((k.lg_chi - chi) / (fmaxf(k.lg_chi, chi) + a.precision));
            const float lambda_new = fminf(k.lambda * 10.0f, max_lambda);
            redo = !(delta_chi < a.precision) && !(k.iteration + 1 > a.max_iters || lambda_new >= max_lambda);
            if constexpr (!SAFE && !STARVED) {
              // Fast flavour: the look-ahead solve on the rejected sums is useless for the trajectory, but a bad pivot
              // in it hands the sector to the SAFE kernel - which then solves with its own pivot rule to the end.  WHO
              // solves a sector must not depend on the cache, so the factorisation still runs here (rejections are rare
              // on the levels this flavour sees), and on a bad pivot the step takes the ordinary path below.
              if (redo && a.ill_list) {
                float q[6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
                  q[i] = p[i];
                redo = damped_step<P, false>(S, lam_use, c.scaling, q, false);
              }
            }
            if (redo) {
              k.lambda = lambda_new;
              k.use_saved = 0;
              ++k.iteration;
              k.reached = k.iteration;
              ++k.n_point_iters;
              lam_use = lambda_new;
#pragma unroll
              for (int i = 0; i < P; ++i)
                p[i] = k.lg_p[i];
            }
          }
          // (the cache is written here, before the solve, so that the sums are dead after it: accepted evaluations -
          // known from chi - and the first evaluation of a level / a re-evaluation, which ran at lg_p)
          if (phase != PH_TENT || chi <= k.lg_chi)
            keep_sums(S);
          if (redo) // (those lanes solve from their kept sums; chi of this evaluation is already taken)
            kept_sums(S);
        }
        float tent[P]; // the parameters this solve starts from: where the evaluation ran, or (redo) the last good ones
#pragma unroll
        for (int i = 0; i < P; ++i)
          tent[i] = p[i];
        // (SAFE kernels up to one wavefront per sector: every lane of a 16-lane row holds the same system - the QR
        // runs spread over the row)
#ifdef LK_TRACE
        const unsigned long long tr_s0 = __builtin_amdgcn_s_memtime();
#endif
        const bool step_starved = starved || (UNIFIED && row_ordered); // (an active row's own flag: see row_ordered)
        const bool wc = damped_step<P, SAFE || STARVED>(S, lam_use, c.scaling, p, step_starved, nullptr, SAFE && GROUP >= 16 && GROUP <= kWave); // p += dp
#ifdef LK_TRACE
        tr_solve += __builtin_amdgcn_s_memtime() - tr_s0;
#endif
        if (!wc && !step_starved)
          ++k.n_ill;
        if constexpr (SEQ && !SAFE) {
          // fast flavour inside a window: there is no second pass to hand the sector to (its next frame waits for this
          // one).  The step of the bad parameter is zero and the host is told: it solves the window again with the SAFE
          // instances (never seen on textured images: lk_stats.ill_conditioned_solves is 0 on configs 2 and 5).
          if (!wc && (int)threadIdx.x % GROUP == 0)
            __hip_atomic_store(a.seq_flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ill_parked = false;
        if constexpr (!SAFE && !STARVED) {
          if (!wc && a.ill_list) {
            // a bad pivot: the reference's rank-revealing QR decides this step.  Undo it and hand
            // the sector, as it was before this evaluation, to the SAFE kernel.
#pragma unroll
            for (int i = 0; i < P; ++i)
              p[i] = tent[i];
            if (redo) { // (the evaluation of this step stands; what the SAFE kernel repeats is the one at the last good parameters)
              phase = PH_REEVAL;
            } else {
              --k.n_evals;
              k.n_sample_evals -= (uint32_t)c.n;
            }
            k.sums_kept = 0;
            park(k, a.ill_list, a.ill_count);
            ill_parked = true;
          }
        }
        if (ill_parked) {
          // nothing else happens to this sector in this launch
        } else if (redo) {
          phase = PH_TENT; // p now holds the tentative parameters of the next trip
        } else if (phase == PH_EVAL0) {
          ++k.n_point_iters;
          k.lg_chi = chi;
          k.use_saved = 1;
          k.iteration = 1;
          iter_start = true;
          if constexpr (KEEP_SUMS) { // (lg_p = where this evaluation ran, enter_level)
            k.sums_kept = a.keep_sums;
          }
        } else if (phase == PH_REEVAL) {
          phase = PH_TENT; // p now holds the tentative parameters
          if constexpr (KEEP_SUMS) {
            k.sums_kept = a.keep_sums;
          }
        } else {           // PH_TENT: p now holds the look-ahead ("saved") parameters
          const float delta_chi =
              __builtin_fabsf((k.lg_chi - chi) / (fmaxf(k.lg_chi, chi) + a.precision));
          if (chi <= k.lg_chi) {
            k.lg_chi = chi;
            k.lambda = fmaxf(k.lambda * 0.4f, min_lambda);
#pragma unroll
            for (int i = 0; i < P; ++i)
              k.lg_p[i] = tent[i];
            k.use_saved = 1;
            if constexpr (KEEP_SUMS) {
              k.sums_kept = a.keep_sums;
            }
          } else {
            k.lambda = fminf(k.lambda * 10.0f, max_lambda);
            k.use_saved = 0;
          }
          if (delta_chi < a.precision) {
            level_end = true;
          } else {
            ++k.iteration;
            iter_start = true;
          }
        }
      }
#ifdef LK_TRACE_TRANS // (one-off split of the time after the solve: in place of the fetch block's cycles)
      const unsigned long long tr_x0 = __builtin_amdgcn_s_memtime();
#endif
      if (iter_start) { // top of the iteration loop (:441-499)
        if (k.iteration > a.max_iters || k.lambda >= max_lambda) {
          k.error = LK_ERROR_CORRELATION_MAX_ITERS_REACHED;
          level_end = true;
        } else {
          k.reached = k.iteration;
          ++k.n_point_iters;
          if (k.use_saved) {
            phase = PH_TENT; // tentative = saved = p
          } else {
#pragma unroll
            for (int i = 0; i < P; ++i)
              p[i] = k.lg_p[i];
            phase = PH_REEVAL;
          }
        }
      }
      if (level_end) { // :589-591, :638
        k.level_old = k.level;
        k.level -= a.py_step;
        if (k.level < a.py_start) {
          translate<P>(p, k.level_old, 0);
          finished = true;
        } else if (starved && !ordered_all && level_count(k.level, k.s) > a.starved_max) {
          handed = true;
        } else {
          enter_level(k);
        }
      }
      if (finished) {
        translate<P>(evaluated, evaluated_level, 0);
        finish_sector(k, evaluated);
      }
      else if (handed)
        hand_over(k);
      else if (STARVED && a.eval_cap > 0 && ++steps >= a.eval_cap)
        park(k, a.finish_list, a.finish_count);
      else
        cold.store(cold_slot, k);
#ifdef LK_TRACE_TRANS
      tr_fetch += __builtin_amdgcn_s_memtime() - tr_x0;
#endif
    }
#ifdef LK_TRACE_FINE
    tr_post += __builtin_amdgcn_s_memtime() - tr_p0;
#endif
  }
#ifdef LK_TRACE
  if constexpr (LK_TRACE_PICK(GROUP, SAFE)) {
    if (((int)threadIdx.x & 63) == 0 && blockIdx.x < 16384u && LK_TRACE_WHEN(a)) {
      unsigned long long *w = g_lk_trace + 8 * ((size_t)blockIdx.x);
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      w[0] = tr_t0;
      w[1] = __builtin_amdgcn_s_memrealtime();
      w[2] = tr_eval;
      w[3] = tr_steps | ((unsigned long long)gridDim.x << 32);
#ifdef LK_TRACE_FINE // (instead of the placement: cycles of the fetch block and of everything after the evaluation)
      w[4] = tr_fetch;
      w[5] = tr_post;
#else
      w[4] = SEQ ? tr_idle : hw;
      w[5] = SEQ ? tr_sleeps : xcc;
#endif
      w[6] = __builtin_amdgcn_s_memtime() - tr_c0; // shader cycles of the whole wavefront
      w[7] = tr_solve; // cycles inside damped_step
    }
  }
#endif
}

// stand-alone evaluation of one sector/level (known-answer tests): same evaluate<>()
template <int MODEL, int INTERP, int GROUP>
__global__ void __launch_bounds__(256) lk_eval_kernel(LkEvalArgs a) {
  constexpr int P = n_params(MODEL);
  using SumsT = Sums<P>;
  __shared__ float lds[4 * (SumsT::N + 1)];
  constexpr int OG = GROUP == 16 ? 16 : kWave; // lane group of the reference-order evaluation
  __shared__ __attribute__((aligned(16))) float ord_lds[OG == 16 ? flat_tile_floats<SumsT::N>() : ord_floats<SumsT::N, OG>()];
  __shared__ __attribute__((aligned(16))) OrdCtx ord_ctx[4];
  if (a.ref_threads > 0 && (int)threadIdx.x >= kWave)
    return; // reference order: one wavefront (its first row or all of it) owns the sector
  const LkLevelView lv = a.lv[a.level];
  const uint32_t off = lv.off[a.sector];
  const float2 c0 = a.center[a.sector];
  const float inv = 1.f / (float)(1 << a.level);
  LevelCtx c{};
  const int4 rc = lv.rect[a.sector];
  c.rx = rc.x;
  c.ry = rc.y;
  c.rw = rc.z;
  c.n = rc.z > 0 ? rc.w : (int)(lv.off[a.sector + 1] - off);
  if (GROUP < kWave && (int)threadIdx.x >= GROUP)
    c.n = 0; // only the first group owns the sector; the others idle
  c.und = (gptr<uint8_t>)lv.und;
  c.def = (gptr<uint8_t>)lv.def;
  c.xy = (gptr<f32x2>)(lv.xy + off);
  c.urows = lv.urows;
  c.ucols = lv.ucols;
  c.drows = lv.drows;
  c.dcols = lv.dcols;
  c.cx = a.level == 0 ? c0.x : c0.x * inv;
  c.cy = a.level == 0 ? c0.y : c0.y * inv;
  c.inv_w = c.rw > 0 ? 1.f / (float)c.rw : 0.f;
  float p[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
    p[i] = a.p[i];
  SumsT S;
  bool err;
  if (a.ref_threads > 0) {
    if constexpr (OG == 16) { // the solve kernel's small-sector path: the sector in row 0, the other rows' lanes lent to it
      if ((int)threadIdx.x >= 16)
        c.n = 0;
      err = a.ref_threads > 1 ? evaluate_ordered_flat<MODEL, INTERP, true>(c, p, S, ord_lds, ord_ctx, a.ref_threads)
                              : evaluate_ordered_flat<MODEL, INTERP, false>(c, p, S, ord_lds, ord_ctx, 1);
    } else {
      err = a.ref_threads > 1 ? evaluate_ordered<MODEL, INTERP, OG, true>(c, p, S, ord_lds, a.ref_threads)
                              : evaluate_ordered<MODEL, INTERP, OG, false>(c, p, S, ord_lds, 1);
    }
  } else {
    err = evaluate<MODEL, INTERP, GROUP, 256>(c, p, S, lds);
  }
  if (threadIdx.x == 0) {
    for (int i = 0; i < 44; ++i)
      a.out[i] = 0.f;
    int idx = 0;
    for (int p1 = 0; p1 < P; ++p1)
      for (int p2 = p1; p2 < P; ++p2)
        a.out[p1 * 6 + p2] = S.v[idx++];
    for (int p1 = 0; p1 < P; ++p1)
      a.out[36 + p1] = S.v[SumsT::NA + p1];
    a.out[42] = S.v[SumsT::N - 1];
    a.out[43] = err ? 1.f : 0.f;
  }
}

// stand-alone sampling of the deformed image at arbitrary points (known-answer tests):
// out[k] = {W, dW/dx, dW/dy, error}
template <int INTERP>
__global__ void lk_sample_kernel(const uint8_t *def, int rows, int cols, const float2 *pts, int n,
                                 float4 *out) {
  int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n)
    return;
  float W = 0.f, Wx = 0.f, Wy = 0.f;
  bool ok = sample_def<INTERP>((gptr<uint8_t>)def, rows, cols, pts[k].x, pts[k].y, W, Wx, Wy);
  out[k] = ok ? make_float4(W, Wx, Wy, 0.f) : make_float4(0.f, 0.f, 0.f, 1.f);
}

// stand-alone damped solve: in = [A row-major 6x6 upper (36), b (6), lambda, scaling]
template <int P> __global__ void lk_solve_only_kernel(const float *in, float *dp_out) {
  Sums<P> S;
  int idx = 0;
  for (int p1 = 0; p1 < P; ++p1)
    for (int p2 = p1; p2 < P; ++p2)
      S.v[idx++] = in[p1 * 6 + p2];
  for (int p1 = 0; p1 < P; ++p1)
    S.v[Sums<P>::NA + p1] = in[36 + p1];
  S.v[Sums<P>::N - 1] = 0.f;
  float p[6] = {0, 0, 0, 0, 0, 0}, dp[6] = {0, 0, 0, 0, 0, 0};
  damped_step<P, true>(S, in[42], in[43], p, in[44] != 0.f, dp, in[44] == 2.f);
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (int i = 0; i < 6; ++i)
      dp_out[i] = i < P ? dp[i] : 0.f;
}

// ------------------------------------------------------------------------------------
// pyramid: 5x5 Gaussian, /2 decimation, CPU-engine arithmetic (pyramid_class.cpp:83-122):
// weights are float products km[i]*km[j], the 25 terms are accumulated in (dj outer, di
// inner) order with separate multiply and add, the result is truncated to u8, and the
// one-pixel border of the target stays 0.
// Each thread produces 4 horizontally adjacent target pixels (one dword store) from a
// 5-row x 11-byte source window.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lk_pyramid_kernel(const uint8_t *__restrict__ src,
                                                         int srows, int scols,
                                                         uint8_t *__restrict__ dst) {
  const int tcols = scols / 2, trows = srows / 2;
  const int tj = (int)(blockIdx.y * blockDim.y + threadIdx.y);
  const int ti0 = (int)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (tj >= trows || ti0 >= tcols)
    return;
  const float km[5] = {0.05f, 0.25f, 0.4f, 0.25f, 0.05f};
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool row_inside = tj >= 1 && tj < trows - 1;
  // interior threads fetch their 11-byte window as four (unaligned) dwords per source row
  const bool fast = 2 * ti0 - 4 >= 0 && 2 * ti0 + 12 <= scols;
  if (row_inside) {
#pragma unroll
    for (int dj = -2; dj <= 2; ++dj) {
      gptr<uint8_t> row = (gptr<uint8_t>)src + (size_t)(2 * tj + dj) * (size_t)scols;
      float px[11]; // source columns 2*ti0-2 .. 2*ti0+8
      if (fast) {
        const uint32_t w0 = load_u32_unaligned(row + 2 * ti0 - 4), w1 = load_u32_unaligned(row + 2 * ti0);
        const uint32_t w2 = load_u32_unaligned(row + 2 * ti0 + 4), w3 = load_u32_unaligned(row + 2 * ti0 + 8);
        px[0] = ub2(w0);
        px[1] = ub3(w0);
        px[2] = ub0(w1);
        px[3] = ub1(w1);
        px[4] = ub2(w1);
        px[5] = ub3(w1);
        px[6] = ub0(w2);
        px[7] = ub1(w2);
        px[8] = ub2(w2);
        px[9] = ub3(w2);
        px[10] = ub0(w3);
      } else { // border threads: clamped byte loads (their border outputs are zeroed below)
#pragma unroll
        for (int c = 0; c < 11; ++c) {
          int sc = 2 * ti0 - 2 + c;
          sc = min(max(sc, 0), scols - 1);
          px[c] = (float)row[sc];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int di = 0; di < 5; ++di)
          acc[t] += px[2 * t + di] * (km[di] * km[dj + 2]);
    }
  }
  uint8_t o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ti = ti0 + t;
    const bool inside = row_inside && ti >= 1 && ti < tcols - 1;
    o[t] = inside ? (uint8_t)acc[t] : (uint8_t)0;
  }
  uint8_t *out = dst + (size_t)tj * (size_t)tcols + (size_t)ti0;
  if (ti0 + 4 <= tcols && (tcols & 3) == 0) { // aligned row pitch: one dword store
    *reinterpret_cast<uint32_t *>(out) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (ti0 + t < tcols)
        out[t] = o[t];
  }
}

// the per-level table of image / sample-list pointers, delivered through kernel arguments
struct LkLevelTable {
  LkLevelView v[LK_MAX_LEVELS];
};
__global__ void lk_set_views_kernel(LkLevelTable t, LkLevelView *out) {
  if (threadIdx.x < LK_MAX_LEVELS)
    out[threadIdx.x] = t.v[threadIdx.x];
}

// ------------------------------------------------------------------------------------
// fused upload + two pyramid levels: one launch per image instead of copy + 2 kernels.
// A workgroup owns a 64x64 tile of level 0, the 32x32 tile of level 1 and the 16x16 tile of
// level 2 below it.  It stages the 73x73 level-0 region those need in LDS (and copies its
// own tile to the engine's level-0 buffer), computes the 35x35 level-1 values the level-2
// tile reads (the halo ring is recomputed by the neighbours too: +20 % MACs, no exchange),
// keeps them in LDS as u8, and finishes with its level-2 tile.  Arithmetic, order and
// border rule are lk_pyramid_kernel's (pyramid_class.cpp:83-122), so every level is
// bit-identical to the one-level kernel.
// ------------------------------------------------------------------------------------
constexpr int kPyrT0 = 64, kPyrR0 = 73, kPyrPitch0 = 80, kPyrR1 = 35, kPyrPitch1 = 36;

__device__ __forceinline__ uint8_t pyr_tap(const uint8_t *win, int pitch) {
  // 5x5 taps around win[2][2]; (dj outer, di inner), separate multiply and add
  const float km[5] = {0.05f, 0.25f, 0.4f, 0.25f, 0.05f};
  float acc = 0.f;
#pragma unroll
  for (int dj = 0; dj < 5; ++dj)
#pragma unroll
    for (int di = 0; di < 5; ++di)
      acc += (float)win[dj * pitch + di] * (km[di] * km[dj]);
  return (uint8_t)acc;
}

struct LkPyrJobs { // up to two images of the same size per launch (blockIdx.z picks one)
  const uint8_t *src[2];
  int step[2];
  uint8_t *l0[2], *l1[2], *l2[2];
};

__global__ void __launch_bounds__(256) lk_pyramid2_kernel(LkPyrJobs jobs, int rows, int cols) {
  const uint8_t *src = jobs.src[blockIdx.z];
  const int step = jobs.step[blockIdx.z];
  uint8_t *l0 = jobs.l0[blockIdx.z], *l1 = jobs.l1[blockIdx.z], *l2 = jobs.l2[blockIdx.z];
  __shared__ __attribute__((aligned(16))) uint8_t s0[kPyrR0 * kPyrPitch0];
  __shared__ __attribute__((aligned(16))) uint8_t s1[kPyrR1 * kPyrPitch1];
  const int rows1 = rows / 2, cols1 = cols / 2, rows2 = rows1 / 2, cols2 = cols1 / 2;
  const int x2 = (int)blockIdx.x * 16, y2 = (int)blockIdx.y * 16; // level-2 tile origin
  const int X0 = 4 * x2 - 8, Y0 = 4 * y2 - 6;                       // level-0 region origin (X0 % 4 == 0)
  const int tid = (int)threadIdx.x;
  const bool copy0 = src != l0;
  // ---- stage level 0 (zero outside the image: only border outputs would read it, and those are 0)
  const bool dwords = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)step) & 3u) == 0 && (cols & 3) == 0;
  for (int i = tid; i < kPyrR0 * (kPyrPitch0 / 4); i += 256) {
    const int r = i / (kPyrPitch0 / 4), cw = i % (kPyrPitch0 / 4);
    const int gy = Y0 + r, gx = X0 + 4 * cw;
    uint32_t v = 0;
    if (gy >= 0 && gy < rows) {
      const uint8_t *row = src + (size_t)gy * (size_t)step;
      if (dwords && gx >= 0 && gx + 4 <= cols) {
        v = *reinterpret_cast<const uint32_t *>(row + gx);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (gx + b >= 0 && gx + b < cols)
            v |= (uint32_t)row[gx + b] << (8 * b);
      }
      // this workgroup's own 64x64 tile goes to the engine's level-0 image (pitch = cols)
      if (copy0 && r >= 6 && r < 6 + kPyrT0 && cw >= 2 && cw < 2 + kPyrT0 / 4) {
        uint8_t *out = l0 + (size_t)gy * (size_t)cols + (size_t)gx;
        if ((cols & 3) == 0 && gx + 4 <= cols) {
          *reinterpret_cast<uint32_t *>(out) = v;
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (gx + b < cols)
              out[b] = (uint8_t)(v >> (8 * b));
        }
      }
    }
    reinterpret_cast<uint32_t *>(s0)[i] = v;
  }
  __syncthreads();
  // ---- level 1: local (j, i) <-> global (2*y2 - 2 + j, 2*x2 - 2 + i); taps start at local
  //      level-0 row 2*j, column 2*i + 2.  One task = 4 adjacent outputs of one row from a
  //      5-row x 11-byte window (two 8-byte LDS reads per row), like lk_pyramid_kernel.
  for (int t = tid; t < kPyrR1 * (kPyrPitch1 / 4); t += 256) {
    const int j = t / (kPyrPitch1 / 4), i0 = 4 * (t % (kPyrPitch1 / 4));
    const float km[5] = {0.05f, 0.25f, 0.4f, 0.25f, 0.05f};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dj = 0; dj < 5; ++dj) {
      const uint2 *w = reinterpret_cast<const uint2 *>(s0 + (2 * j + dj) * kPyrPitch0 + 2 * i0); // 8-byte aligned
      const uint2 a = w[0], b = w[1];
      const float px[11] = {ub2(a.x), ub3(a.x), ub0(a.y), ub1(a.y), ub2(a.y), ub3(a.y),
                            ub0(b.x), ub1(b.x), ub2(b.x), ub3(b.x), ub0(b.y)};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int di = 0; di < 5; ++di)
          acc[q] += px[2 * q + di] * (km[di] * km[dj]);
    }
    const int gj = 2 * y2 - 2 + j;
    uint32_t packed = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gi = 2 * x2 - 2 + i0 + q;
      const bool inside = gj >= 1 && gj < rows1 - 1 && gi >= 1 && gi < cols1 - 1;
      const uint32_t v = inside ? (uint32_t)(uint8_t)acc[q] : 0u;
      packed |= v << (8 * q);
      if (j >= 2 && j < 34 && i0 + q >= 2 && i0 + q < 34 && gj < rows1 && gi < cols1)
        l1[(size_t)gj * (size_t)cols1 + (size_t)gi] = (uint8_t)v;
    }
    *reinterpret_cast<uint32_t *>(s1 + j * kPyrPitch1 + i0) = packed;
  }
  __syncthreads();
  // ---- level 2: one output per thread; taps start at local level-1 (2*j, 2*i)
  {
    const int j = tid / 16, i = tid % 16;
    const int gj = y2 + j, gi = x2 + i;
    if (gj < rows2 && gi < cols2) {
      const bool inside = gj >= 1 && gj < rows2 - 1 && gi >= 1 && gi < cols2 - 1;
      l2[(size_t)gj * (size_t)cols2 + (size_t)gi] =
          inside ? pyr_tap(s1 + (2 * j) * kPyrPitch1 + 2 * i, kPyrPitch1) : (uint8_t)0;
    }
  }
}

// ------------------------------------------------------------------------------------
// initial-guess policy for every sector (managerClass::adjust_initial_guess,
// manager_class.cpp:2602-2707)
// ------------------------------------------------------------------------------------
struct LkGuessArgs {
  const float2 *center;
  const float *last_p; // results of the previous frame [S][6]
  float *prev_p;       // previous_resulting_parameters [S][6]
  float *guess;        // out [S][6]
  float global_guess[6];
  float gcx, gcy;
  int n_sectors, model, frame, constant_velocity;
};

__global__ void lk_guess_kernel(LkGuessArgs a) {
  int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= a.n_sectors)
    return;
  const int P = n_params(a.model);
  float g[6] = {0, 0, 0, 0, 0, 0};
  if (a.frame == 0) {
    for (int i = 0; i < P; ++i)
      g[i] = a.global_guess[i];
    float2 c = a.center[s];
    float dx = c.x - a.gcx, dy = c.y - a.gcy;
    if (a.model == LK_FM_UVUXUYVXVY) {
      g[0] += dx * a.global_guess[2] + dy * a.global_guess[3];
      g[1] += dx * a.global_guess[4] + dy * a.global_guess[5];
    } else {
      float Vx = a.global_guess[2];
      g[0] += -dy * Vx;
      g[1] += dx * Vx;
    }
    for (int i = 0; i < 6; ++i)
      a.prev_p[(size_t)s * 6 + i] = i < P ? g[i] : 0.f;
  } else {
    for (int i = 0; i < P; ++i) {
      float r = a.last_p[(size_t)s * 6 + i], q = a.prev_p[(size_t)s * 6 + i];
      g[i] = a.constant_velocity ? r + (r - q) : r;
      a.prev_p[(size_t)s * 6 + i] = r;
    }
  }
  for (int i = 0; i < 6; ++i)
    a.guess[(size_t)s * 6 + i] = i < P ? g[i] : 0.f;
}

// warp a sample list by p (kModel_inPlace, correlationKernel.cu:56-110; getDefXY0)
__global__ void lk_warp_points_kernel(const float2 *xy, int n, float cx, float cy, int model,
                                      const float *pp, float2 *out) {
  int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n)
    return;
  float p[6];
  for (int i = 0; i < 6; ++i)
    p[i] = pp[i];
  float2 q = xy[k];
  float xd, yd, dx, dy;
  switch (model) {
  case LK_FM_U: Warp<LK_FM_U>::apply(q.x, q.y, cx, cy, p, xd, yd, dx, dy); break;
  case LK_FM_UV: Warp<LK_FM_UV>::apply(q.x, q.y, cx, cy, p, xd, yd, dx, dy); break;
  case LK_FM_UVQ: Warp<LK_FM_UVQ>::apply(q.x, q.y, cx, cy, p, xd, yd, dx, dy); break;
  default: Warp<LK_FM_UVUXUYVXVY>::apply(q.x, q.y, cx, cy, p, xd, yd, dx, dy); break;
  }
  out[k] = make_float2(xd, yd);
}


// ---- moved sample lists, rebuilt on the device (strict Lagrangian description) ---------------
// manager_class.cpp:369-380 makes the deformed positions of the last solve the undeformed samples
// of the next frame; pyramid_class.cpp:289-323 then decimates them level by level.  Sectors are
// concatenated in sector order, so ONE order-preserving compaction of the whole level is the
// per-sector compaction of every sector at once.
// (int)(v + 0.5f) as the host code computes it: out-of-range and NaN give INT_MIN (cvttss2si)
__device__ __forceinline__ int round_like_host(float v) {
  const float t = v + 0.5f;
  return fabsf(t) < 2147483648.f ? (int)t : (int)0x80000000;
}

template <int MODEL> __global__ void lk_rewarp_kernel(LkRewarpArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.total)
    return;
  int lo = 0, hi = a.n_sectors; // dst_off[lo] <= i < dst_off[hi]: the last such lo owns sample i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.dst_off[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  const int s = lo;
  const uint32_t k = i - a.dst_off[s];
  const int4 rc = a.src_rect[s];
  float2 q;
  if (rc.z > 0 && a.rows) { // the evaluation copy: y outer, x inner (LkLevelView::xy_eval)
    const int row = (int)k / rc.z;
    q.x = (float)(rc.x + ((int)k - row * rc.z));
    q.y = (float)(rc.y + row);
  } else if (rc.z > 0) { // x outer, y inner (manager_class.cpp:1607-1611)
    const int h = rc.w / rc.z, col = (int)k / h;
    q.x = (float)(rc.x + col);
    q.y = (float)(rc.y + ((int)k - col * h));
  } else {
    q = a.src_xy[a.src_off[s] + k];
  }
  if (a.offset) { // Lagrangian description: add_pair(offset), manager_class.cpp:38-47
    const float2 o = a.offset[s];
    a.dst_xy[i] = make_float2((float)round_like_host(o.x + q.x), (float)round_like_host(o.y + q.y));
    return;
  }
  float p[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
    p[j] = a.p[(size_t)s * 6 + j];
  const float2 c = a.center[s];
  float xd, yd, dx, dy;
  Warp<MODEL>::apply(q.x, q.y, c.x, c.y, p, xd, yd, dx, dy);
  a.dst_xy[i] = make_float2(xd, yd);
}

__device__ __forceinline__ bool decimate_keeps(float2 q, int mag) {
  return round_like_host(q.x) % mag == 0 && round_like_host(q.y) % mag == 0;
}

constexpr int kScanThreads = 256, kScanItems = 4, kScanTile = kScanThreads * kScanItems;

// exclusive sum over the workgroup; *total receives the sum of all
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t *lds, uint32_t &total) {
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t up = __shfl_up(inc, d, kWave);
    if ((int)(threadIdx.x & (kWave - 1)) >= d)
      inc += up;
  }
  const int wave = (int)threadIdx.x / kWave, n_waves = (int)blockDim.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == kWave - 1)
    lds[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (int w = 0; w < n_waves; ++w) {
    const uint32_t t = lds[w];
    before += w < wave ? t : 0u;
    all += t;
  }
  __syncthreads();
  total = all;
  return before + inc - v;
}

// pass 1: keep flag (bit 31) and position among the kept samples of the own 1024-sample tile
__global__ void __launch_bounds__(kScanThreads) lk_decimate_flag_kernel(const float2 *xy, const uint32_t *n_ptr, int mag,
                                                                        uint32_t *pos, uint32_t *tile_count) {
  __shared__ uint32_t lds[kScanThreads / kWave];
  const uint32_t n = *n_ptr, base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  bool keep[kScanItems];
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    keep[j] = base + j < n && decimate_keeps(xy[base + j], mag);
    cnt += keep[j];
  }
  uint32_t total;
  uint32_t at = block_exclusive_sum(cnt, lds, total);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) {
      pos[base + j] = at | (keep[j] ? 0x80000000u : 0u);
      at += keep[j];
    }
  if (threadIdx.x == 0)
    tile_count[blockIdx.x] = total;
}

// pass 2 (one workgroup): tile counts -> exclusive prefix; [n_tiles] and *n_out get the total
__global__ void __launch_bounds__(1024) lk_scan_tiles_kernel(uint32_t *tile_count, int n_tiles, uint32_t *n_out) {
  __shared__ uint32_t lds[1024 / kWave];
  uint32_t carry = 0;
  for (int base = 0; base < n_tiles; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const uint32_t v = i < n_tiles ? tile_count[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_sum(v, lds, total);
    if (i < n_tiles)
      tile_count[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) {
    tile_count[n_tiles] = carry;
    *n_out = carry;
  }
}

// pass 3: kept samples move to their place, in the coordinates of the coarser level
__global__ void __launch_bounds__(kScanThreads) lk_decimate_scatter_kernel(const float2 *xy, const uint32_t *n_ptr, float inv,
                                                                           const uint32_t *pos, const uint32_t *tile_first,
                                                                           float2 *out) {
  const uint32_t n = *n_ptr, i = blockIdx.x * kScanThreads + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t w = pos[i];
  if (w & 0x80000000u) {
    const float2 q = xy[i];
    out[tile_first[i / kScanTile] + (w & 0x7fffffffu)] = make_float2(q.x * inv, q.y * inv);
  }
}

// pass 4: where every sector's list starts after the compaction
__global__ void lk_decimate_offsets_kernel(const uint32_t *off_prev, const uint32_t *n_ptr, const uint32_t *pos,
                                           const uint32_t *tile_first, int n_tiles, int n_sectors, uint32_t *off_new) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s > n_sectors)
    return;
  const uint32_t n = *n_ptr, o = off_prev[s];
  off_new[s] = o < n ? tile_first[o / kScanTile] + (pos[o] & 0x7fffffffu) : tile_first[n_tiles];
}

// ---- ROI masks on the device (see LkRoiSector in lk_device.hpp) ---------------------------------
// Which sector a tile belongs to: the last s with tile_begin[s] <= t (wavefront-uniform search).
__device__ __forceinline__ int roi_sector_of_tile(const uint32_t *tile_begin, int n_sectors, uint32_t t) {
  int lo = 0, hi = n_sectors;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_begin[mid] <= t)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// get_inside_points_annularDomain's test for candidate (fx, j) - manager_class.cpp:907-918, the same
// float operations in the same order as lkroi::annular_points (contraction is off in both builds)
__device__ __forceinline__ bool roi_annular_keeps(const LkRoiSector &q, float fx, int j) {
  const float ex = fx - q.cx, ey = (float)j - q.cy;
  const float r2 = ex * ex + ey * ey;
  if (!(r2 > q.ri2 && r2 < q.ro2))
    return false;
  const float w1 = (q.q11x - fx) * (q.q01y - q.q11y) - (q.q11y - (float)j) * (q.q01x - q.q11x);
  const float w2 = (q.q00x - fx) * (q.q10y - q.q00y) - (q.q00y - (float)j) * (q.q10x - q.q00x);
  return w1 * w2 > 0.f || q.as == 1;
}

__device__ __forceinline__ void roi_blob_row(const LkRoiSector &q, const LkRoiFlat *flats, uint32_t row, int &j, int &i0,
                                             int &i1) {
  int lo = q.flat_begin, hi = q.flat_begin + q.flat_count; // last flat with row_begin <= row
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((uint32_t)flats[mid].row_begin <= row)
      lo = mid;
    else
      hi = mid;
  }
  const LkRoiFlat f = flats[lo];
  j = f.j0 + (int)(row - (uint32_t)f.row_begin);
  i0 = (int)ceilf(f.ls * (float)j + f.li); // polygon_class.cpp:389-391
  i1 = (int)ceilf(f.rs * (float)j + f.ri);
}

// FILL = false: tile_count[t] = samples the tile keeps.  FILL = true: the samples, at tile_first[t].
// ROWS: the candidates of an annular sector's bounding box are walked row by row (y outer, x inner) instead of the
// reference's column by column - the evaluation copy of the lists (LkLevelView::xy_eval): same tiles, same samples.
template <bool FILL, bool ROWS = false>
__global__ void __launch_bounds__(kScanThreads) lk_roi_tile_kernel(const LkRoiSector *sectors, const LkRoiFlat *flats,
                                                                   const uint32_t *tile_begin, int n_sectors,
                                                                   uint32_t *tile_count, const uint32_t *tile_first,
                                                                   float2 *out) {
  __shared__ uint32_t lds[kScanThreads / kWave];
  const uint32_t t = blockIdx.x;
  const int s = roi_sector_of_tile(tile_begin, n_sectors, t);
  const LkRoiSector q = sectors[s];
  const uint32_t local = t - tile_begin[s];
  if (q.kind == 1) { // one scan line of a blob
    int j, i0, i1;
    roi_blob_row(q, flats, local, j, i0, i1);
    if constexpr (!FILL) {
      if (threadIdx.x == 0)
        tile_count[t] = i1 > i0 ? (uint32_t)(i1 - i0) : 0u;
    } else {
      float2 *dst = out + tile_first[t];
      for (int i = i0 + (int)threadIdx.x; i < i1; i += kScanThreads)
        dst[i - i0] = make_float2((float)i, (float)j);
    }
    return;
  }
  const int h = q.y1 - q.y0, w = q.x1 - q.x0;
  const uint32_t n_cand = (uint32_t)w * (uint32_t)h, base = local * (uint32_t)kLkRoiTile + threadIdx.x * kScanItems;
  bool keep[kScanItems];
  float fx[kScanItems];
  int jj[kScanItems];
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const uint32_t idx = base + k;
    keep[k] = false;
    fx[k] = 0.f;
    jj[k] = 0;
    if (idx < n_cand) {
      if constexpr (ROWS) {
        const uint32_t row = idx / (uint32_t)w; // y outer, x inner
        fx[k] = (float)(q.x0 + (int)(idx - row * (uint32_t)w));
        jj[k] = q.y0 + (int)row;
      } else {
        const uint32_t col = idx / (uint32_t)h; // x outer, y inner
        fx[k] = (float)(q.x0 + (int)col);
        jj[k] = q.y0 + (int)(idx - col * (uint32_t)h);
      }
      keep[k] = roi_annular_keeps(q, fx[k], jj[k]);
    }
    cnt += keep[k];
  }
  uint32_t total;
  uint32_t at = block_exclusive_sum(cnt, lds, total);
  if constexpr (!FILL) {
    if (threadIdx.x == 0)
      tile_count[t] = total;
  } else {
    float2 *dst = out + tile_first[t];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
      if (keep[k])
        dst[at++] = make_float2(fx[k], (float)jj[k]);
  }
}

// where every sector's list starts: the exclusive prefix at its first tile ([S] = the total)
__global__ void lk_roi_offsets_kernel(const uint32_t *tile_begin, const uint32_t *tile_first, int n_sectors, uint32_t *off) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s <= n_sectors)
    off[s] = tile_first[tile_begin[s]];
}

// Newton_Raphson(p, n, xy) solves about the float mean of the samples, summed in list order
// (pyramid_class.cpp:325-340).  The order is the reference's, so the additions are one chain
// per coordinate; what is parallel is the memory side: a wavefront per sector fetches 64
// consecutive samples with one coalesced load and feeds them to the chain lane by lane
// (v_readlane), instead of one lane waiting for one cache line per sample.
__global__ void __launch_bounds__(kWave) lk_mean_center_kernel(const float2 *xy, const uint32_t *off, int n_sectors,
                                                                float2 *center) {
  const int s = (int)blockIdx.x;
  if (s >= n_sectors)
    return;
  const uint32_t b = off[s], e = off[s + 1];
  const int lane = (int)threadIdx.x;
  float sx = 0.f, sy = 0.f;
  for (uint32_t base = b; base < e; base += kWave) {
    const uint32_t i = base + (uint32_t)lane;
    const float2 q = i < e ? xy[i] : make_float2(0.f, 0.f);
    const int qx = __float_as_int(q.x), qy = __float_as_int(q.y);
    if (e - base >= (uint32_t)kWave) {
#pragma unroll
      for (int l = 0; l < kWave; ++l) {
        sx += __int_as_float(__builtin_amdgcn_readlane(qx, l));
        sy += __int_as_float(__builtin_amdgcn_readlane(qy, l));
      }
    } else {
      const int n = (int)(e - base);
      for (int l = 0; l < n; ++l) {
        sx += __int_as_float(__builtin_amdgcn_readlane(qx, l));
        sy += __int_as_float(__builtin_amdgcn_readlane(qy, l));
      }
    }
  }
  if (lane == 0) {
    const float n = (float)(e - b);
    center[s] = make_float2(__fdiv_rn(sx, n), __fdiv_rn(sy, n));
  }
}

// Reference-order mode.  CorrelationClass::reached_iterations is a member that only an LM trip
// writes (correlation_class.cpp:452); a sector whose very first evaluation fails returns before any
// trip (:413-419), and get_iterations() (:870) reports the value the sector BEFORE it left behind -
// the manager solves sectors one after the other in index order (manager_class.cpp:304-307).  The
// solve kernel marks such records; here every marked record takes the count of the nearest earlier
// unmarked sector (or `carry_in`: what the last sector of the previous solve left; 0 at first - the
// reference's own first value is indeterminate).  Marked records are rare and their runs short.
__global__ void lk_stale_iterations_kernel(lk_result *r, int n, const int *carry_in, int *carry_out) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= n)
    return;
  int *it = &r[s].iterations;
  int v = __hip_atomic_load(it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v == kStaleIterations) {
    int t = s - 1;
    // (a marked neighbour may already hold its resolved value: the same value the walk would reach)
    while (t >= 0 && (v = __hip_atomic_load(&r[t].iterations, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kStaleIterations)
      --t;
    if (t < 0)
      v = *carry_in;
    __hip_atomic_store(it, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (s == n - 1)
    *carry_out = v;
}

// The same over the gathered records of a group (lk_group.cpp): n_ranks padded blocks of `cap` records, block q
// holding the sectors [q*S/G, (q+1)*S/G) - resolved in GLOBAL sector order, so a shard's leading marked records
// take the count the shard before it left behind.
__device__ __forceinline__ size_t stale_block_pos(int s, int n, int n_ranks, int cap) {
  int q = (int)(((long long)s * n_ranks + n_ranks - 1) / n); // first guess, then the exact owner: first_q <= s < first_{q+1}
  q = q < 0 ? 0 : (q >= n_ranks ? n_ranks - 1 : q);
  while (q > 0 && (int)((long long)n * q / n_ranks) > s)
    --q;
  while (q + 1 < n_ranks && (int)((long long)n * (q + 1) / n_ranks) <= s)
    ++q;
  return (size_t)q * (size_t)cap + (size_t)(s - (int)((long long)n * q / n_ranks));
}
__global__ void lk_stale_iterations_blocks_kernel(lk_result *r, int n, int n_ranks, int cap, const int *carry_in, int *carry_out) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= n)
    return;
  int *it = &r[stale_block_pos(s, n, n_ranks, cap)].iterations;
  int v = __hip_atomic_load(it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v == kStaleIterations) {
    int t = s - 1;
    while (t >= 0 && (v = __hip_atomic_load(&r[stale_block_pos(t, n, n_ranks, cap)].iterations, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT)) == kStaleIterations)
      --t;
    if (t < 0)
      v = *carry_in;
    __hip_atomic_store(it, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (s == n - 1)
    *carry_out = v;
}

// ... and over the gathered records of a frame-pipelined WINDOW of a group: n_ranks blocks of [frames][cap] records, in the
// order the reference solves them - frame 0's sectors 0 .. S-1, then frame 1's, ...
__device__ __forceinline__ size_t stale_window_pos(long long t, int n, int n_ranks, int cap, int frames) {
  const int f = (int)(t / n), s = (int)(t - (long long)f * n);
  const size_t in_frame = stale_block_pos(s, n, n_ranks, cap); // q * cap + (s - first_q)
  const size_t q = in_frame / (size_t)cap, k = in_frame - q * (size_t)cap;
  return (q * (size_t)frames + (size_t)f) * (size_t)cap + k;
}
__global__ void lk_stale_iterations_window_kernel(lk_result *r, int n, int n_ranks, int cap, int frames, const int *carry_in, int *carry_out) {
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, total = (long long)n * frames;
  if (t0 >= total)
    return;
  int *it = &r[stale_window_pos(t0, n, n_ranks, cap, frames)].iterations;
  int v = __hip_atomic_load(it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v == kStaleIterations) {
    long long t = t0 - 1;
    while (t >= 0 && (v = __hip_atomic_load(&r[stale_window_pos(t, n, n_ranks, cap, frames)].iterations, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT)) == kStaleIterations)
      --t;
    if (t < 0)
      v = *carry_in;
    __hip_atomic_store(it, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t0 == total - 1)
    *carry_out = v;
}

// ---- the sequential float mean of an INTEGER sample list, evaluated in parallel, bit for bit ----
// pyramid_class.cpp:325-340 adds the coordinates into one running float per axis, in list order.
// For a 4.2 M-sample blob that chain is 4.2 M dependent additions (tens of ms for one wavefront).
// For device-masked sectors every coordinate is a non-negative integer, and then the chain has
// structure: the running sum s is an integer-valued float; while s stays in one binade
// [2^e, 2^(e+1)) its ulp u = 2^(e-23) is fixed and  fl(s + x) = s + u * g  with
//     g = floor(x / u) + (rem > u/2) + (rem == u/2 and (s/u + floor(x/u)) odd)      (round to nearest even)
// which depends on s only through the PARITY of s/u.  So a run of samples is a map
// {parity in} -> {g total, parity out}, and such maps compose associatively: a workgroup scans
// 32 768 samples per pass (32 per thread, a composition scan over the 1024 threads).  The sum is
// monotone, so the binade changes at most ~40 times per list; the thread whose run crosses the
// boundary walks its 32 samples with the exact rounding rule, and the pass restarts behind the
// crossing with the new ulp.  One workgroup per sector; x first, then y.
__device__ __forceinline__ long long rne24(long long t) { // the float32 nearest to the integer t >= 0, ties to even
  if (t < (1ll << 24))
    return t;
  const int sh = (63 - __clzll(t)) - 23;
  const long long u = 1ll << sh, half = u >> 1, r = t & (u - 1);
  long long q = t >> sh;
  q += (r > half || (r == half && (q & 1ll))) ? 1 : 0;
  return q << sh;
}

struct ParityMap { // {0,1} -> (g, parity): g[p] = units of u added when the run starts at parity p
  int g0, g1, pn; // pn: bit 0 = parity out for parity in 0, bit 1 = for parity in 1
};
__device__ __forceinline__ ParityMap compose(const ParityMap &a, const ParityMap &b) { // a, then b
  const int pa0 = a.pn & 1, pa1 = (a.pn >> 1) & 1;
  ParityMap c;
  c.g0 = a.g0 + (pa0 ? b.g1 : b.g0);
  c.g1 = a.g1 + (pa1 ? b.g1 : b.g0);
  c.pn = ((b.pn >> pa0) & 1) | (((b.pn >> pa1) & 1) << 1);
  return c;
}

constexpr int kMeanThreads = 1024, kMeanItems = 32; // 32 768 samples per pass; both axes share the loads and barriers

struct AxisPass { // one axis of one pass
  int sh, p_start;
  long long s, limit;
  ParityMap m; // my run
  __device__ __forceinline__ void begin(long long s_in) {
    s = s_in;
    const int ex = s < (1ll << 24) ? 23 : 63 - __clzll(s);
    sh = ex - 23;                 // ulp = 2^sh
    limit = 1ll << (ex + 1);      // leaving the binade (or exactness) at this value
    p_start = (int)((s >> sh) & 1ll);
    m = ParityMap{0, 0, 2};       // identity
  }
  __device__ __forceinline__ void add(int x) {
    const int shc = sh < 30 ? sh : 30; // (coordinates are below 2^22: beyond that the quotient is 0 anyway)
    const int qq = x >> shc, r = sh < 30 ? (x & ((1 << shc) - 1)) : x, half = sh < 30 ? ((1 << shc) >> 1) : (1 << 29);
    const int up = r > half ? 1 : 0, tie = (sh > 0 && r == half) ? 1 : 0;
    ParityMap one;
    one.g0 = qq + up + (tie & (qq & 1));       // parity in 0: (0 + qq) odd -> round up
    one.g1 = qq + up + (tie & ((qq + 1) & 1)); // parity in 1
    one.pn = (one.g0 & 1) | (((1 + one.g1) & 1) << 1);
    m = compose(m, one);
  }
};

// exclusive composition prefix of `m` over the workgroup's THREADS threads; `all` = everybody's composition
template <int THREADS = kMeanThreads>
__device__ __forceinline__ ParityMap workgroup_prefix(const ParityMap &m, int *lds3, ParityMap &all) {
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  constexpr int WAVES = THREADS / kWave;
  ParityMap inc = m;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    ParityMap o;
    o.g0 = __shfl_up(inc.g0, d, kWave);
    o.g1 = __shfl_up(inc.g1, d, kWave);
    o.pn = __shfl_up(inc.pn, d, kWave);
    if (lane >= d)
      inc = compose(o, inc);
  }
  if (lane == kWave - 1) {
    lds3[3 * wave] = inc.g0;
    lds3[3 * wave + 1] = inc.g1;
    lds3[3 * wave + 2] = inc.pn;
  }
  __syncthreads();
  ParityMap before{0, 0, 2};
  all = ParityMap{0, 0, 2};
  for (int w = 0; w < WAVES; ++w) {
    const ParityMap t{lds3[3 * w], lds3[3 * w + 1], lds3[3 * w + 2]};
    if (w < wave)
      before = compose(before, t);
    all = compose(all, t);
  }
  ParityMap o;
  o.g0 = __shfl_up(inc.g0, 1, kWave);
  o.g1 = __shfl_up(inc.g1, 1, kWave);
  o.pn = __shfl_up(inc.pn, 1, kWave);
  return lane == 0 ? before : compose(before, o);
}

// A 4.2 M-sample list on ONE workgroup is 129 passes of 32 768 samples through one CU (4.4 ms).  The passes are
// independent except for the state they hand on - the running sum, which a pass needs for its binade (the ulp)
// and its parity.  The parity is what the maps are for; the binade can be PREDICTED: the float chain never strays
// far from the exact integer prefix sum, so away from the powers of two the binade of the exact sum is the
// binade of the chain.  Hence three launches:
//   lk_mean_chunk_sums_kernel   exact integer sums of every chunk of 8192 samples (all chunks of all sectors at once);
//   lk_mean_chunk_maps_kernel   per chunk and axis: the binade of the exact prefix sum at its start; if the exact
//                               sum is still in that binade at its end, the chunk's parity map for that ulp;
//   lk_mean_center_int_kernel   one workgroup per sector walks its chunks in order with the TRUE chain value: a map
//                               applies iff the value is in the map's binade and stays there (checked, not
//                               assumed) - one table lookup per chunk; any other chunk (a crossing inside, a
//                               prediction off by the chain's drift) is walked by the exact passes above.
// Nothing depends on the prediction being right; it only decides how many chunks take the slow path (C3's blob:
// ~20 of 517).
constexpr int kChunkThreads = 256, kChunkSamples = kChunkThreads * kMeanItems; // 8192 samples per chunk

struct MeanChunk {
  int gx0, gx1, pnx, shx; // shx < 0: no map (the exact sum leaves the binade inside the chunk)
  int gy0, gy1, pny, shy;
};

// chunk_begin[s] = number of chunks of the sectors before s (one workgroup)
__global__ void __launch_bounds__(1024) lk_mean_chunk_table_kernel(const uint32_t *off, int n_sectors, uint32_t *chunk_begin) {
  __shared__ uint32_t lds[1024 / kWave];
  uint32_t carry = 0;
  for (int base = 0; base < n_sectors; base += 1024) {
    const int sct = base + (int)threadIdx.x;
    const uint32_t v = sct < n_sectors ? (off[sct + 1] - off[sct] + (uint32_t)kChunkSamples - 1u) / (uint32_t)kChunkSamples : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_sum(v, lds, total);
    if (sct < n_sectors)
      chunk_begin[sct] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0)
    chunk_begin[n_sectors] = carry;
}

// the sector of global chunk c (chunk_begin is ascending, sectors without samples own no chunk)
__device__ __forceinline__ int mean_sector_of_chunk(const uint32_t *chunk_begin, int n_sectors, uint32_t c) {
  int lo = 0, hi = n_sectors; // chunk_begin[lo] <= c < chunk_begin[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_begin[mid] <= c)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(kChunkThreads) lk_mean_chunk_sums_kernel(const float2 *xy, const uint32_t *off, const uint32_t *chunk_begin,
                                                                          int n_sectors, long long *sums) {
  __shared__ long long lds[2 * (kChunkThreads / kWave)];
  const uint32_t c = blockIdx.x;
  if (c >= chunk_begin[n_sectors])
    return;
  const int sct = mean_sector_of_chunk(chunk_begin, n_sectors, c);
  const uint32_t b = off[sct] + (c - chunk_begin[sct]) * (uint32_t)kChunkSamples, e = min(off[sct + 1], b + (uint32_t)kChunkSamples);
  long long sx = 0, sy = 0;
  for (uint32_t i = b + threadIdx.x; i < e; i += kChunkThreads) { // (coalesced; the order of an exact sum is free)
    const float2 q = xy[i];
    sx += (long long)(int)q.x;
    sy += (long long)(int)q.y;
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) {
    sx += __shfl_xor(sx, d, kWave);
    sy += __shfl_xor(sy, d, kWave);
  }
  const int wave = (int)threadIdx.x / kWave;
  if (((int)threadIdx.x & (kWave - 1)) == 0) {
    lds[2 * wave] = sx;
    lds[2 * wave + 1] = sy;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long tx = 0, ty = 0;
    for (int w = 0; w < kChunkThreads / kWave; ++w) {
      tx += lds[2 * w];
      ty += lds[2 * w + 1];
    }
    sums[2 * (size_t)c] = tx;
    sums[2 * (size_t)c + 1] = ty;
  }
}

__global__ void __launch_bounds__(kChunkThreads) lk_mean_chunk_maps_kernel(const float2 *xy, const uint32_t *off, const uint32_t *chunk_begin,
                                                                          int n_sectors, const long long *sums, MeanChunk *chunks) {
  constexpr int WAVES = kChunkThreads / kWave;
  __shared__ int lds_x[3 * WAVES], lds_y[3 * WAVES];
  __shared__ long long lds_p[2 * WAVES];
  const uint32_t c = blockIdx.x;
  if (c >= chunk_begin[n_sectors])
    return;
  const int sct = mean_sector_of_chunk(chunk_begin, n_sectors, c);
  const uint32_t c0 = chunk_begin[sct];
  const uint32_t b = off[sct] + (c - c0) * (uint32_t)kChunkSamples, e = min(off[sct + 1], b + (uint32_t)kChunkSamples);
  // exact prefix sums of the sector's earlier chunks
  long long px = 0, py = 0;
  for (uint32_t k = c0 + threadIdx.x; k < c; k += kChunkThreads) {
    px += sums[2 * (size_t)k];
    py += sums[2 * (size_t)k + 1];
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) {
    px += __shfl_xor(px, d, kWave);
    py += __shfl_xor(py, d, kWave);
  }
  const int tid = (int)threadIdx.x, wave = tid / kWave;
  if ((tid & (kWave - 1)) == 0) {
    lds_p[2 * wave] = px;
    lds_p[2 * wave + 1] = py;
  }
  __syncthreads();
  px = py = 0;
  for (int w = 0; w < WAVES; ++w) {
    px += lds_p[2 * w];
    py += lds_p[2 * w + 1];
  }
  AxisPass ax, ay;
  ax.begin(px);
  ay.begin(py);
  const bool map_x = px + sums[2 * (size_t)c] < ax.limit, map_y = py + sums[2 * (size_t)c + 1] < ay.limit;
  if (map_x || map_y) { // (uniform over the workgroup)
    const uint32_t i0 = b + (uint32_t)tid * kMeanItems;
#pragma unroll
    for (int j = 0; j < kMeanItems; ++j) {
      const uint32_t i = i0 + j;
      if (i < e) {
        const float2 q = xy[i];
        ax.add((int)q.x);
        ay.add((int)q.y);
      }
    }
  }
  ParityMap all_x, all_y;
  (void)workgroup_prefix<kChunkThreads>(ax.m, lds_x, all_x);
  (void)workgroup_prefix<kChunkThreads>(ay.m, lds_y, all_y);
  if (tid == 0) {
    MeanChunk m;
    m.gx0 = all_x.g0, m.gx1 = all_x.g1, m.pnx = all_x.pn, m.shx = map_x ? ax.sh : -1;
    m.gy0 = all_y.g0, m.gy1 = all_y.g1, m.pny = all_y.pn, m.shy = map_y ? ay.sh : -1;
    chunks[c] = m;
  }
}

__global__ void __launch_bounds__(kMeanThreads) lk_mean_center_int_kernel(const float2 *xy, const uint32_t *off, int n_sectors,
                                                                          const uint32_t *chunk_begin, const MeanChunk *chunks,
                                                                          float2 *center) {
  constexpr int WAVES = kMeanThreads / kWave;
  __shared__ int lds_x[3 * WAVES], lds_y[3 * WAVES], lds_first[WAVES];
  __shared__ long long lds_l[3];
  __shared__ MeanChunk lds_chunks[kMeanThreads];
  __shared__ long long lds_walk[3]; // sx, sy, chunks taken
  const int sec = (int)blockIdx.x;
  if (sec >= n_sectors)
    return;
  const uint32_t b = off[sec], e = off[sec + 1];
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  long long sx = 0, sy = 0;

  // the exact passes over [k, k_end): workgroup-uniform
  auto walk = [&](uint32_t k, const uint32_t k_end) {
    while (k < k_end) {
      AxisPass ax, ay;
      ax.begin(sx);
      ay.begin(sy);
      const uint32_t i0 = k + (uint32_t)tid * kMeanItems;
      short x[kMeanItems], y[kMeanItems]; // (kept for the walk; coordinates are below 2^15 in any image this engine takes)
      int n_mine = 0;
#pragma unroll
      for (int j = 0; j < kMeanItems; ++j) {
        const uint32_t i = i0 + j;
        x[j] = y[j] = 0;
        if (i < k_end) {
          const float2 q = xy[i];
          x[j] = (short)(int)q.x;
          y[j] = (short)(int)q.y;
          ax.add((int)q.x);
          ay.add((int)q.y);
          ++n_mine;
        }
      }
      __syncthreads(); // (the previous pass is done with lds)
      ParityMap all_x, all_y;
      const ParityMap ex = workgroup_prefix(ax.m, lds_x, all_x);
      const ParityMap ey = workgroup_prefix(ay.m, lds_y, all_y);
      const int gx_before = ax.p_start ? ex.g1 : ex.g0, px_mine = (ex.pn >> ax.p_start) & 1;
      const int gy_before = ay.p_start ? ey.g1 : ey.g0, py_mine = (ey.pn >> ay.p_start) & 1;
      const bool crosses = sx + ((long long)(gx_before + (px_mine ? ax.m.g1 : ax.m.g0)) << ax.sh) >= ax.limit ||
                           sy + ((long long)(gy_before + (py_mine ? ay.m.g1 : ay.m.g0)) << ay.sh) >= ay.limit;
      // the first run whose end leaves a binade, on either axis
      const unsigned long long cb = __ballot(crosses);
      if (lane == 0)
        lds_first[wave] = cb ? wave * kWave + (int)__builtin_ctzll(cb) : kMeanThreads;
      __syncthreads();
      int first = kMeanThreads;
      for (int w = 0; w < WAVES; ++w)
        first = min(first, lds_first[w]);
      if (first == kMeanThreads) { // the whole pass stays inside both binades
        sx += (long long)(ax.p_start ? all_x.g1 : all_x.g0) << ax.sh;
        sy += (long long)(ay.p_start ? all_y.g1 : all_y.g0) << ay.sh;
        k += kMeanThreads * kMeanItems;
        continue;
      }
      if (tid == first) { // my start is still inside both: walk to the first crossing with the exact rule
        long long vx = sx + ((long long)gx_before << ax.sh), vy = sy + ((long long)gy_before << ay.sh);
        int j = 0;
        for (; j < n_mine; ++j) {
          vx = rne24(vx + (long long)x[j]);
          vy = rne24(vy + (long long)y[j]);
          if (vx >= ax.limit || vy >= ay.limit) {
            ++j;
            break;
          }
        }
        lds_l[0] = vx;
        lds_l[1] = vy;
        lds_l[2] = (long long)(i0 + (uint32_t)j);
      }
      __syncthreads();
      sx = lds_l[0];
      sy = lds_l[1];
      k = (uint32_t)lds_l[2];
    }
  };

  const uint32_t c0 = chunk_begin[sec], n_chunks = chunk_begin[sec + 1] - c0;
  uint32_t done = 0; // chunks of this sector behind the chain
  while (done < n_chunks) {
    // a tile of chunk records into LDS, then one thread takes as many of them as apply
    const uint32_t tile = min(n_chunks - done, (uint32_t)kMeanThreads);
    __syncthreads();
    if ((uint32_t)tid < tile)
      lds_chunks[tid] = chunks[c0 + done + (uint32_t)tid];
    __syncthreads();
    uint32_t at = 0;
    while (at < tile) {
      if (tid == 0) {
        long long vx = sx, vy = sy;
        uint32_t t = at;
        for (; t < tile; ++t) {
          const MeanChunk m = lds_chunks[t];
          AxisPass ax, ay;
          ax.begin(vx);
          ay.begin(vy);
          if (m.shx != ax.sh || m.shy != ay.sh)
            break;
          const long long nx = vx + ((long long)(ax.p_start ? m.gx1 : m.gx0) << ax.sh);
          const long long ny = vy + ((long long)(ay.p_start ? m.gy1 : m.gy0) << ay.sh);
          if (nx >= ax.limit || ny >= ay.limit)
            break;
          vx = nx;
          vy = ny;
        }
        lds_walk[0] = vx;
        lds_walk[1] = vy;
        lds_walk[2] = (long long)t;
      }
      __syncthreads();
      sx = lds_walk[0];
      sy = lds_walk[1];
      at = (uint32_t)lds_walk[2];
      __syncthreads();
      if (at < tile) { // this chunk needs the exact passes
        const uint32_t kb = b + (done + at) * (uint32_t)kChunkSamples;
        walk(kb, min(e, kb + (uint32_t)kChunkSamples));
        ++at;
      }
    }
    done += tile;
  }
  if (threadIdx.x == 0) {
    const float n = (float)(e - b);
    center[sec] = make_float2(__fdiv_rn((float)sx, n), __fdiv_rn((float)sy, n)); // (both sums are exact floats)
  }
}

} // namespace

#ifdef LK_TRACE
extern "C" int lk_debug_trace(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lk_trace), (size_t)n_words * 8);
}
#endif

hipError_t lk_launch_stale_iterations(lk_result *r, int n, const int *carry_in, int *carry_out, hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(lk_stale_iterations_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, n, carry_in,
                     carry_out);
  return hipGetLastError();
}

__global__ void lk_append_sector_kernel(LkAppendArgs a) {
  const int t = (int)threadIdx.x, s = a.sector;
  if (t < a.n_levels) {
    a.d_rect[t][s] = a.rect[t];
    if (!a.keep_state)
      a.d_off[t][s + 1] = a.off_end[t];
  }
  if (t == 0)
    a.d_center[s] = a.center;
  if (a.keep_state)
    return;
  if (t < 6) {
    a.d_guess[(size_t)s * 6 + t] = 0.f;
    a.d_last_p[(size_t)s * 6 + t] = 0.f;
    a.d_prev_p[(size_t)s * 6 + t] = 0.f;
    a.d_last_eval_p[(size_t)s * 6 + t] = 0.f;
  }
  if (t < 4)
    a.d_stats[(size_t)s * 4 + t] = 0u;
  if (t < (int)(sizeof(lk_result) / 4))
    reinterpret_cast<uint32_t *>(a.d_result + s)[t] = 0u;
}
hipError_t lk_launch_append_sector(const LkAppendArgs &a, hipStream_t st) {
  hipLaunchKernelGGL(lk_append_sector_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t lk_launch_stale_iterations_blocks(lk_result *all, int n, int n_ranks, int cap, const int *carry_in, int *carry_out,
                                             hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(lk_stale_iterations_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, all, n, n_ranks,
                     cap, carry_in, carry_out);
  return hipGetLastError();
}

hipError_t lk_launch_stale_iterations_window(lk_result *all, int n, int n_ranks, int cap, int frames, const int *carry_in,
                                             int *carry_out, hipStream_t st) {
  if (n <= 0 || frames <= 0)
    return hipSuccess;
  const long long total = (long long)n * frames;
  hipLaunchKernelGGL(lk_stale_iterations_window_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, all, n, n_ranks, cap,
                     frames, carry_in, carry_out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// launch wrappers (called from lk_engine.cpp)
// ------------------------------------------------------------------------------------
// Persistent launch: as many workgroups as the device keeps resident (or fewer when there
// is less work).  Nothing waits on another workgroup, so an over-estimate is harmless.
template <class K> static int resident_workgroups(K kernel, int threads) {
  int dev = 0, cus = 256, per_cu = 1;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1)
    per_cu = 1;
  return cus * per_cu;
}

// The team class's share of `all` workgroup slots (LkSolveArgs::slots_permille), in whole multiples of the 8 XCDs: workgroups
// are dealt round-robin over the XCDs, one 512-thread workgroup fits a CU, and an XCD that got 33 workgroups for its 32 CUs
// keeps one waiting - if that one belongs to the team, the team waits for a PERSISTENT workgroup of the other launch, i.e.
// for that launch's end (measured on config 3: 0.61 ms with 96 + 160 workgroups, 0.98 with 108 + 147).
static int split_slots(int all, int team_permille) { return (int)((long long)all * team_permille / 1000) & ~7; }

template <int MODEL, int INTERP, int GROUP, int THREADS, bool SAFE, bool REF = false>
static hipError_t launch_solve_gs(const LkSolveArgs &a, hipStream_t st) {
  static std::atomic<int> resident_cache{0}; // per template instance (one device type per process); engines launch from several threads
  int resident = resident_cache.load(std::memory_order_relaxed);
  if (resident == 0) {
    resident = resident_workgroups(lk_solve_kernel<MODEL, INTERP, GROUP, THREADS, SAFE, REF>, THREADS);
    resident_cache.store(resident, std::memory_order_relaxed);
  }
  LkSolveArgs b = a;
  if (REF && GROUP == 16) {
    static const int rows_env = [] { const char *f = getenv("LK_FLAT_ROWS"); return f ? atoi(f) : 0; }(); // tuning hook
    b.rows_used = rows_env >= 1 && rows_env <= 4 ? rows_env : 0;
  }
  const int per_wg = (REF && GROUP == 16 && b.rows_used > 0) ? b.rows_used : THREADS / GROUP;
  const int want = (a.n_sectors + per_wg - 1) / per_wg;
  static const int force_persistent = [] { // tuning / test hook
    const char *f = getenv("LK_FORCE_PERSISTENT");
    return f ? atoi(f) : -1;
  }();
  b.persistent = force_persistent >= 0 ? force_persistent : (want > 2 * resident ? 1 : 0);
  if (GROUP == 1)
    b.persistent = 0; // every lane takes exactly one sector
  if (GROUP == 16 && SAFE && !REF && a.resume && !a.finisher) {
    // sectors parked with a bad pivot: normally none - a small grid that retires at once and
    // rewinds its own list and queue (they start at zero: lk_commit_sectors)
    b.persistent = 1;
    b.chunk = 0;
    const int grid = want < 128 ? want : 128;
    hipLaunchKernelGGL((lk_solve_kernel<MODEL, INTERP, GROUP, THREADS, SAFE>), dim3((unsigned)grid), dim3(THREADS), 0,
                       st, b);
    return hipGetLastError();
  }
  if (GROUP == 16 && SAFE && !REF && a.finisher) {
    // the finisher pulls parked sectors from finish_list until *finish_count (known only on
    // the device) is used up: as many wavefronts as could be needed, capped by what is resident
    b.persistent = 1;
    b.chunk = 0;
    hipError_t qe = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (qe != hipSuccess)
      return qe;
    hipLaunchKernelGGL((lk_solve_kernel<MODEL, INTERP, GROUP, THREADS, SAFE>),
                       dim3((unsigned)(want < resident ? want : resident)), dim3(THREADS), 0, st, b);
    return hipGetLastError();
  }
  if (GROUP == 512 && a.team_w > 1) {
    // teams wait on each other: every workgroup of the launch must be resident at once - also
    // when other engines' team launches hold their share of the GPU (lk_set_pairs_in_flight)
    int share = resident / (a.gpu_share > 1 ? a.gpu_share : 1);
    if (a.slots_permille > 0) // (the rest of the slots belongs to the one-workgroup class's launch on a sibling stream)
      share = split_slots(share, a.slots_permille);
    b.team_w = a.team_w < share / a.n_sectors ? a.team_w : share / a.n_sectors;
    if (REF && b.team_w != a.team_w) // (a reference-order team IS the reference's thread count: all of it or one workgroup)
      b.team_w = 0;
    if (b.team_w > 1) {
      hipError_t te = hipMemsetAsync(a.team_arrivals, 0, 2 * (size_t)a.n_sectors * sizeof(uint32_t), st); // counters + broken flags
      if (te != hipSuccess)
        return te;
      b.persistent = 0;
      b.chunk = 0;
      hipLaunchKernelGGL((lk_solve_kernel<MODEL, INTERP, GROUP, THREADS, SAFE, REF>),
                         dim3((unsigned)(a.n_sectors * b.team_w)), dim3(THREADS), 0, st, b);
      return hipGetLastError();
    }
  }
  b.team_w = 0;
  b.chunk = (want + 7) / 8;
  int grid_persistent = resident;
  if (GROUP == 512 && a.slots_permille > 0) {
    // the one-workgroup class beside a team launch: a persistent grid on its share of the slots (largest sectors first in
    // the queue, lk_engine.cpp), so that the team's workgroups are all resident next to it from the start
    b.persistent = 1;
    const int all = resident / (a.gpu_share > 1 ? a.gpu_share : 1);
    const int share = (all & ~7) - split_slots(all, 1000 - a.slots_permille); // (what the team launch leaves, see split_slots)
    grid_persistent = share < 1 ? 1 : (share < want ? share : want);
  }
  if (b.persistent) { // rewind the sector queue
    hipError_t qe = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (qe != hipSuccess)
      return qe;
  }
  dim3 grid((unsigned)(b.persistent ? grid_persistent : 8 * b.chunk));
  hipLaunchKernelGGL((lk_solve_kernel<MODEL, INTERP, GROUP, THREADS, SAFE, REF>), grid, dim3(THREADS), 0, st, b);
  return hipGetLastError();
}

// SAFE flavour: ill-conditioned systems go through the reference's pivoted QR inside the
// lane-group kernel too (LK_FORCE_SAFE=1; costs a wavefront of occupancy); the default fast
// flavour zeroes the step of a parameter whose pivot is bad.  Starved levels never get here:
// the GROUP == 1 kernel solves them first.
template <int MODEL, int INTERP, int GROUP, int THREADS>
static hipError_t launch_solve_g(const LkSolveArgs &a, hipStream_t st) {
  if constexpr ((THREADS == kWave && (GROUP == 16 || GROUP == kWave)) || (THREADS == 512 && GROUP == 512)) {
    if (a.reference_order > 0) // the reference-order instances
      return launch_solve_gs<MODEL, INTERP, GROUP, THREADS, true, true>(a, st);
  }
  return a.safe ? launch_solve_gs<MODEL, INTERP, GROUP, THREADS, true>(a, st)
                : launch_solve_gs<MODEL, INTERP, GROUP, THREADS, false>(a, st);
}

template <int MODEL, int INTERP>
static hipError_t launch_solve_mi(const LkSolveArgs &a, int group, hipStream_t st) {
  switch (group) {
  case 1: return launch_solve_gs<MODEL, INTERP, 1, 64, false>(a, st); // starved levels, one lane per sector
  case 16: return launch_solve_g<MODEL, INTERP, 16, 64>(a, st);
  case 32: return launch_solve_g<MODEL, INTERP, 32, 64>(a, st);
  case 64: return launch_solve_g<MODEL, INTERP, 64, 64>(a, st);
  case 256: return launch_solve_g<MODEL, INTERP, 256, 256>(a, st);
  default: return launch_solve_g<MODEL, INTERP, 512, 512>(a, st);
  }
}

// LK_TUNE_ONLY_AFFINE_BICUBIC (kernel tuning builds, scripts/tune_build.sh): instantiate the solve
// kernel for the affine model with the reference bicubic only - a 15 s build instead of 90 s
template <int MODEL>
static hipError_t launch_solve_m(const LkSolveArgs &a, int interp, int group, hipStream_t st) {
  switch (interp) {
#ifndef LK_TUNE_ONLY_AFFINE_BICUBIC
  case LK_IM_NEAREST: return launch_solve_mi<MODEL, LK_IM_NEAREST>(a, group, st);
  case LK_IM_BILINEAR: return launch_solve_mi<MODEL, LK_IM_BILINEAR>(a, group, st);
  case LK_IM_BICUBIC_SEPARABLE: return launch_solve_mi<MODEL, LK_IM_BICUBIC_SEPARABLE>(a, group, st);
#endif
  default: return launch_solve_mi<MODEL, LK_IM_BICUBIC>(a, group, st);
  }
}

hipError_t lk_launch_solve(const LkSolveArgs &a, int model, int interp, int group, hipStream_t st) {
  if (a.n_sectors <= 0)
    return hipSuccess;
  switch (model) {
#ifndef LK_TUNE_ONLY_AFFINE_BICUBIC
  case LK_FM_U: return launch_solve_m<LK_FM_U>(a, interp, group, st);
  case LK_FM_UV: return launch_solve_m<LK_FM_UV>(a, interp, group, st);
  case LK_FM_UVQ: return launch_solve_m<LK_FM_UVQ>(a, interp, group, st);
#endif
  default: return launch_solve_m<LK_FM_UVUXUYVXVY>(a, interp, group, st);
  }
}

// ---- frame-pipelined launches (the SEQ instances) --------------------------------------------------------------
// One persistent grid per size class for a whole window of frames: a share of the class's sector count (below), and no more
// wavefronts than are resident (more would not hurt - a wavefront that is not running holds no ticket - but would queue
// behind the grid for nothing).
template <int MODEL, int INTERP, int GROUP, bool SAFE, bool REF>
static hipError_t launch_solve_seq_gs(const LkSolveArgs &a, hipStream_t st) {
  static std::atomic<int> resident_cache{0};
  int resident = resident_cache.load(std::memory_order_relaxed);
  if (resident == 0) {
    resident = resident_workgroups(lk_solve_kernel<MODEL, INTERP, GROUP, kWave, SAFE, REF, true>, kWave);
    resident_cache.store(resident, std::memory_order_relaxed);
  }
  const int per_wg = kWave / GROUP;
  LkSolveArgs b = a;
  b.persistent = 1;
  b.chunk = 0;
  b.team_w = 0;
  b.solo = 0;
  // A sector has at most one frame in work at a time, so groups beyond the sectors could only wait; and with nearly as many
  // groups as sectors a group that is done draws the next frame of a sector whose current frame is still being solved by a
  // slower group, and waits.  Measured on config 2 (10 000 sectors, 32-lane groups, 4096 resident wavefronts): 8192 groups
  // 0.166 ms per pair, 6144 groups 0.160, 4096 groups 0.201; config 4 (50 176 sectors, 12 288 resident groups) wants them all.
  // The reference-order rows are different: their lanes are dealt by need, a waiting row's lanes work for the wavefront's
  // other sectors, and the kernel is bound by the latency of its QR chains - it wants every resident wavefront it can get
  // (config 2: 1875 / 2500 / 3072 wavefronts 0.361 / 0.342 / 0.332 ms per pair).
  static const int fill_env = [] { // tuning hook: groups per 1000 sectors at most
    const char *f = getenv("LK_SEQ_FILL");
    return f ? atoi(f) : 0;
  }();
  const int fill_permille = fill_env > 0 ? fill_env : (REF && GROUP == 16 ? 1250 : 640);
  long long groups = (long long)a.n_sectors * fill_permille / 1000;
  // Few sectors (one rank's block of a sharded sequence): the launch is bound by the chains of its sectors, not by throughput,
  // and a sector without a group of its own only waits its turn - every sector gets one while that fills no more than half
  // of the resident groups.  Measured on blocks of config 2's grid, windows of 16 pairs (scripts/experiments/shard_latency*.sh):
  // 5000 sectors 640 / 820 / 1000 per mille -> 0.134 / 0.121 / 0.132 ms per pair, 2500: 0.121 / 0.104 / 0.101, 1250: 0.088 / - / 0.072.
  if (fill_env <= 0 && fill_permille < 1000) {
    const long long own = std::min<long long>(a.n_sectors, (long long)resident * per_wg / 2);
    groups = std::max(groups, own);
  }
  const int fill = (int)((groups + per_wg - 1) / per_wg);
  int grid = fill < resident ? fill : resident;
  static const int grid_permille = [] { // tuning hook: the grid as a share of that
    const char *f = getenv("LK_SEQ_GRID");
    return f ? atoi(f) : 1000;
  }();
  grid = (int)((long long)grid * grid_permille / 1000);
  grid = grid < 1 ? 1 : grid;
  hipError_t qe = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
  if (qe != hipSuccess)
    return qe;
  hipLaunchKernelGGL((lk_solve_kernel<MODEL, INTERP, GROUP, kWave, SAFE, REF, true>), dim3((unsigned)grid), dim3(kWave), 0, st, b);
  return hipGetLastError();
}

// flavour: 0 fast (root-free Cholesky; a bad pivot raises seq_flags[1]), 1 SAFE (the QR inside the kernel; 16-lane rows:
// starved levels with the finisher's arithmetic), 2 reference order (a.reference_order = T)
template <int MODEL, int INTERP>
static hipError_t launch_solve_seq_mi(const LkSolveArgs &a, int group, int flavour, hipStream_t st) {
  if (flavour == 2) {
    switch (group) {
    case 16: return launch_solve_seq_gs<MODEL, INTERP, 16, true, true>(a, st);
    case 64: return launch_solve_seq_gs<MODEL, INTERP, 64, true, true>(a, st);
    default: return hipErrorInvalidValue;
    }
  }
  if (flavour == 1) {
    switch (group) {
    case 16: return launch_solve_seq_gs<MODEL, INTERP, 16, true, false>(a, st);
    case 32: return launch_solve_seq_gs<MODEL, INTERP, 32, true, false>(a, st);
    case 64: return launch_solve_seq_gs<MODEL, INTERP, 64, true, false>(a, st);
    default: return hipErrorInvalidValue;
    }
  }
  switch (group) {
  case 16: return launch_solve_seq_gs<MODEL, INTERP, 16, false, false>(a, st);
  case 32: return launch_solve_seq_gs<MODEL, INTERP, 32, false, false>(a, st);
  case 64: return launch_solve_seq_gs<MODEL, INTERP, 64, false, false>(a, st);
  default: return hipErrorInvalidValue;
  }
}

template <int MODEL>
static hipError_t launch_solve_seq_m(const LkSolveArgs &a, int interp, int group, int flavour, hipStream_t st) {
  switch (interp) {
#ifndef LK_TUNE_ONLY_AFFINE_BICUBIC
  case LK_IM_NEAREST: return launch_solve_seq_mi<MODEL, LK_IM_NEAREST>(a, group, flavour, st);
  case LK_IM_BILINEAR: return launch_solve_seq_mi<MODEL, LK_IM_BILINEAR>(a, group, flavour, st);
  case LK_IM_BICUBIC_SEPARABLE: return launch_solve_seq_mi<MODEL, LK_IM_BICUBIC_SEPARABLE>(a, group, flavour, st);
#endif
  default: return launch_solve_seq_mi<MODEL, LK_IM_BICUBIC>(a, group, flavour, st);
  }
}

hipError_t lk_launch_solve_seq(const LkSolveArgs &a, int model, int interp, int group, int flavour, hipStream_t st) {
  if (a.n_sectors <= 0 || a.seq_frames <= 0)
    return hipSuccess;
  if ((long long)a.n_sectors * a.seq_frames >= (1ll << 31) - (1ll << 20)) // (tickets are ints; the grid draws a few past the end)
    return hipErrorInvalidValue;
  switch (model) {
#ifndef LK_TUNE_ONLY_AFFINE_BICUBIC
  case LK_FM_U: return launch_solve_seq_m<LK_FM_U>(a, interp, group, flavour, st);
  case LK_FM_UV: return launch_solve_seq_m<LK_FM_UV>(a, interp, group, flavour, st);
  case LK_FM_UVQ: return launch_solve_seq_m<LK_FM_UVQ>(a, interp, group, flavour, st);
#endif
  default: return launch_solve_seq_m<LK_FM_UVUXUYVXVY>(a, interp, group, flavour, st);
  }
}

template <int MODEL, int GROUP>
static hipError_t launch_eval_mg(const LkEvalArgs &a, int interp, hipStream_t st) {
  switch (interp) {
  case LK_IM_NEAREST: hipLaunchKernelGGL((lk_eval_kernel<MODEL, LK_IM_NEAREST, GROUP>), dim3(1), dim3(256), 0, st, a); break;
  case LK_IM_BILINEAR: hipLaunchKernelGGL((lk_eval_kernel<MODEL, LK_IM_BILINEAR, GROUP>), dim3(1), dim3(256), 0, st, a); break;
  case LK_IM_BICUBIC_SEPARABLE: hipLaunchKernelGGL((lk_eval_kernel<MODEL, LK_IM_BICUBIC_SEPARABLE, GROUP>), dim3(1), dim3(256), 0, st, a); break;
  default: hipLaunchKernelGGL((lk_eval_kernel<MODEL, LK_IM_BICUBIC, GROUP>), dim3(1), dim3(256), 0, st, a); break;
  }
  return hipGetLastError();
}

template <int MODEL>
static hipError_t launch_eval_m(const LkEvalArgs &a, int interp, int group, hipStream_t st) {
  switch (group) {
  case 16: return launch_eval_mg<MODEL, 16>(a, interp, st);
  case 32: return launch_eval_mg<MODEL, 32>(a, interp, st);
  case 64: return launch_eval_mg<MODEL, 64>(a, interp, st);
  default: return launch_eval_mg<MODEL, 256>(a, interp, st);
  }
}

hipError_t lk_launch_eval(const LkEvalArgs &a, int model, int interp, int group, hipStream_t st) {
  switch (model) {
  case LK_FM_U: return launch_eval_m<LK_FM_U>(a, interp, group, st);
  case LK_FM_UV: return launch_eval_m<LK_FM_UV>(a, interp, group, st);
  case LK_FM_UVQ: return launch_eval_m<LK_FM_UVQ>(a, interp, group, st);
  default: return launch_eval_m<LK_FM_UVUXUYVXVY>(a, interp, group, st);
  }
}

hipError_t lk_launch_solve_only(int n, const float *d_in, float *d_out, hipStream_t st) {
  switch (n) {
  case 1: hipLaunchKernelGGL((lk_solve_only_kernel<1>), dim3(1), dim3(64), 0, st, d_in, d_out); break;
  case 2: hipLaunchKernelGGL((lk_solve_only_kernel<2>), dim3(1), dim3(64), 0, st, d_in, d_out); break;
  case 3: hipLaunchKernelGGL((lk_solve_only_kernel<3>), dim3(1), dim3(64), 0, st, d_in, d_out); break;
  case 6: hipLaunchKernelGGL((lk_solve_only_kernel<6>), dim3(1), dim3(64), 0, st, d_in, d_out); break;
  default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t lk_launch_sample(int interp, const uint8_t *def, int rows, int cols, const float2 *pts, int n,
                            float4 *out, hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (interp) {
  case LK_IM_NEAREST: hipLaunchKernelGGL((lk_sample_kernel<LK_IM_NEAREST>), grid, block, 0, st, def, rows, cols, pts, n, out); break;
  case LK_IM_BILINEAR: hipLaunchKernelGGL((lk_sample_kernel<LK_IM_BILINEAR>), grid, block, 0, st, def, rows, cols, pts, n, out); break;
  case LK_IM_BICUBIC_SEPARABLE: hipLaunchKernelGGL((lk_sample_kernel<LK_IM_BICUBIC_SEPARABLE>), grid, block, 0, st, def, rows, cols, pts, n, out); break;
  default: hipLaunchKernelGGL((lk_sample_kernel<LK_IM_BICUBIC>), grid, block, 0, st, def, rows, cols, pts, n, out); break;
  }
  return hipGetLastError();
}

hipError_t lk_launch_pyramid(const uint8_t *src, int srows, int scols, uint8_t *dst, hipStream_t st) {
  int tcols = scols / 2, trows = srows / 2;
  if (tcols <= 0 || trows <= 0)
    return hipSuccess;
  dim3 block(64, 4);
  dim3 grid((unsigned)((tcols + 4 * 64 - 1) / (4 * 64)), (unsigned)((trows + 3) / 4));
  hipLaunchKernelGGL(lk_pyramid_kernel, grid, block, 0, st, src, srows, scols, dst);
  return hipGetLastError();
}

hipError_t lk_launch_set_views(const LkLevelView *h_views, LkLevelView *d_views, hipStream_t st) {
  LkLevelTable t;
  for (int l = 0; l < LK_MAX_LEVELS; ++l)
    t.v[l] = h_views[l];
  hipLaunchKernelGGL(lk_set_views_kernel, dim3(1), dim3(64), 0, st, t, d_views);
  return hipGetLastError();
}

// copy (src -> l0, skipped when src == l0) + levels 1 and 2 in one launch
hipError_t lk_launch_pyramid2(int n_images, const uint8_t *const *src, const int *step, int rows, int cols,
                              uint8_t *const *l0, uint8_t *const *l1, uint8_t *const *l2, hipStream_t st) {
  if (rows / 4 <= 0 || cols / 4 <= 0 || n_images < 1 || n_images > 2)
    return hipErrorInvalidValue;
  LkPyrJobs jobs{};
  for (int i = 0; i < n_images; ++i) {
    jobs.src[i] = src[i];
    jobs.step[i] = step[i];
    jobs.l0[i] = l0[i];
    jobs.l1[i] = l1[i];
    jobs.l2[i] = l2[i];
  }
  dim3 grid((unsigned)((cols + kPyrT0 - 1) / kPyrT0), (unsigned)((rows + kPyrT0 - 1) / kPyrT0), (unsigned)n_images);
  hipLaunchKernelGGL(lk_pyramid2_kernel, grid, dim3(256), 0, st, jobs, rows, cols);
  return hipGetLastError();
}

hipError_t lk_launch_guess(const float2 *center, const float *last_p, float *prev_p, float *guess,
                           const float *global_guess, float gcx, float gcy, int n_sectors, int model,
                           int frame, int constant_velocity, hipStream_t st) {
  LkGuessArgs a;
  a.center = center;
  a.last_p = last_p;
  a.prev_p = prev_p;
  a.guess = guess;
  for (int i = 0; i < 6; ++i)
    a.global_guess[i] = global_guess[i];
  a.gcx = gcx;
  a.gcy = gcy;
  a.n_sectors = n_sectors;
  a.model = model;
  a.frame = frame;
  a.constant_velocity = constant_velocity;
  if (n_sectors <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(lk_guess_kernel, dim3((unsigned)((n_sectors + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t lk_launch_warp_points(const float2 *xy, int n, float cx, float cy, int model, const float *d_p,
                                 float2 *out, hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(lk_warp_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xy, n, cx,
                     cy, model, d_p, out);
  return hipGetLastError();
}

hipError_t lk_launch_rewarp(const LkRewarpArgs &a, int model, hipStream_t st) {
  if (a.total == 0)
    return hipSuccess;
  const dim3 grid((a.total + 255) / 256), block(256);
  switch (model) {
  case LK_FM_U: hipLaunchKernelGGL(lk_rewarp_kernel<LK_FM_U>, grid, block, 0, st, a); break;
  case LK_FM_UV: hipLaunchKernelGGL(lk_rewarp_kernel<LK_FM_UV>, grid, block, 0, st, a); break;
  case LK_FM_UVQ: hipLaunchKernelGGL(lk_rewarp_kernel<LK_FM_UVQ>, grid, block, 0, st, a); break;
  default: hipLaunchKernelGGL(lk_rewarp_kernel<LK_FM_UVUXUYVXVY>, grid, block, 0, st, a); break;
  }
  return hipGetLastError();
}

int lk_decimate_tiles(uint32_t n_max) { return (int)((n_max + kScanTile - 1) / kScanTile); }

// One level of pyramid_class.cpp:289-323 for all sectors: xy_prev[*n_prev] (at most n_max) ->
// xy_out, off_prev[S+1] -> off_out[S+1], *n_out = samples kept.  pos: n_max words,
// tiles: lk_decimate_tiles(n_max) + 1 words.
hipError_t lk_launch_decimate(const float2 *xy_prev, const uint32_t *off_prev, const uint32_t *n_prev, uint32_t n_max,
                              int level_delta, int n_sectors, uint32_t *pos, uint32_t *tiles, float2 *xy_out,
                              uint32_t *off_out, uint32_t *n_out, hipStream_t st) {
  const int mag = 1 << level_delta, n_tiles = lk_decimate_tiles(n_max);
  if (n_tiles == 0)
    return hipSuccess;
  hipLaunchKernelGGL(lk_decimate_flag_kernel, dim3((unsigned)n_tiles), dim3(kScanThreads), 0, st, xy_prev, n_prev, mag,
                     pos, tiles);
  hipLaunchKernelGGL(lk_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, tiles, n_tiles, n_out);
  hipLaunchKernelGGL(lk_decimate_scatter_kernel, dim3((n_max + kScanThreads - 1) / kScanThreads), dim3(kScanThreads), 0,
                     st, xy_prev, n_prev, 1.f / (float)mag, pos, tiles, xy_out);
  hipLaunchKernelGGL(lk_decimate_offsets_kernel, dim3((unsigned)(n_sectors + 256) / 256), dim3(256), 0, st, off_prev,
                     n_prev, pos, tiles, n_tiles, n_sectors, off_out);
  return hipGetLastError();
}

static_assert(kLkRoiTile == kScanTile, "one ROI tile = one pass of the block scan");

// pass 1: per-tile counts -> exclusive prefix in place (tiles[n_tiles] = total, also *n_out)
hipError_t lk_launch_roi_count(const LkRoiSector *sectors, const LkRoiFlat *flats, const uint32_t *tile_begin, int n_sectors,
                               uint32_t n_tiles, uint32_t *tiles, uint32_t *n_out, hipStream_t st, int rows) {
  if (n_tiles == 0)
    return hipSuccess;
  if (rows)
    hipLaunchKernelGGL((lk_roi_tile_kernel<false, true>), dim3(n_tiles), dim3(kScanThreads), 0, st, sectors, flats, tile_begin,
                       n_sectors, tiles, (const uint32_t *)nullptr, (float2 *)nullptr);
  else
  hipLaunchKernelGGL(lk_roi_tile_kernel<false>, dim3(n_tiles), dim3(kScanThreads), 0, st, sectors, flats, tile_begin, n_sectors,
                     tiles, (const uint32_t *)nullptr, (float2 *)nullptr);
  hipLaunchKernelGGL(lk_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, tiles, (int)n_tiles, n_out);
  return hipGetLastError();
}

// pass 2: the samples and the per-sector offsets
hipError_t lk_launch_roi_fill(const LkRoiSector *sectors, const LkRoiFlat *flats, const uint32_t *tile_begin, int n_sectors,
                              uint32_t n_tiles, const uint32_t *tiles, float2 *xy, uint32_t *off, hipStream_t st, int rows) {
  if (n_tiles == 0)
    return hipSuccess;
  if (rows)
    hipLaunchKernelGGL((lk_roi_tile_kernel<true, true>), dim3(n_tiles), dim3(kScanThreads), 0, st, sectors, flats, tile_begin,
                       n_sectors, (uint32_t *)nullptr, tiles, xy);
  else
  hipLaunchKernelGGL(lk_roi_tile_kernel<true>, dim3(n_tiles), dim3(kScanThreads), 0, st, sectors, flats, tile_begin, n_sectors,
                     (uint32_t *)nullptr, tiles, xy);
  hipLaunchKernelGGL(lk_roi_offsets_kernel, dim3((unsigned)(n_sectors + 256) / 256), dim3(256), 0, st, tile_begin, tiles, n_sectors,
                     off);
  return hipGetLastError();
}

// integer, non-negative sample lists (device-masked annular / blob sectors): the same mean, evaluated in parallel
// scratch: lk_mean_center_int_scratch_bytes(total, n_sectors) bytes (chunk table, exact chunk sums, chunk maps)
size_t lk_mean_center_int_scratch_bytes(uint32_t n_samples, int n_sectors) {
  const size_t max_chunks = (size_t)n_samples / kChunkSamples + (size_t)n_sectors + 1;
  return ((size_t)n_sectors + 2) * sizeof(uint32_t) + 8 + max_chunks * (2 * sizeof(long long) + sizeof(MeanChunk));
}
hipError_t lk_launch_mean_center_int(const float2 *xy, const uint32_t *off, uint32_t n_samples, int n_sectors, void *scratch,
                                     float2 *center, hipStream_t st) {
  if (n_sectors <= 0)
    return hipSuccess;
  const size_t max_chunks = (size_t)n_samples / kChunkSamples + (size_t)n_sectors + 1;
  uint32_t *chunk_begin = static_cast<uint32_t *>(scratch);
  const size_t table = (((size_t)n_sectors + 2) * sizeof(uint32_t) + 7) & ~(size_t)7;
  long long *sums = reinterpret_cast<long long *>(static_cast<char *>(scratch) + table);
  MeanChunk *chunks = reinterpret_cast<MeanChunk *>(sums + 2 * max_chunks);
  hipLaunchKernelGGL(lk_mean_chunk_table_kernel, dim3(1), dim3(1024), 0, st, off, n_sectors, chunk_begin);
  hipLaunchKernelGGL(lk_mean_chunk_sums_kernel, dim3((unsigned)max_chunks), dim3(kChunkThreads), 0, st, xy, off, chunk_begin, n_sectors, sums);
  hipLaunchKernelGGL(lk_mean_chunk_maps_kernel, dim3((unsigned)max_chunks), dim3(kChunkThreads), 0, st, xy, off, chunk_begin, n_sectors,
                     sums, chunks);
  hipLaunchKernelGGL(lk_mean_center_int_kernel, dim3((unsigned)n_sectors), dim3(kMeanThreads), 0, st, xy, off, n_sectors, chunk_begin,
                     chunks, center);
  return hipGetLastError();
}

hipError_t lk_launch_mean_center(const float2 *xy, const uint32_t *off, int n_sectors, float2 *center, hipStream_t st) {
  if (n_sectors <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(lk_mean_center_kernel, dim3((unsigned)n_sectors), dim3(kWave), 0, st, xy, off, n_sectors, center);
  return hipGetLastError();
}
