// reference_types_min.hpp - TEST INFRASTRUCTURE: the handful of reference declarations that
// include/lk_cuda_class_adapter.hpp uses (enums.hpp:10-35,73-78; domains.hpp:10,59,110-118),
// restated so that the adapter can be compiled and RUN on a box without /root/reference (the
// GPU box).  The CPU test suite compiles the adapter against the reference's real headers
// (tests/test_ffi_host.py) - that is what pins the real layout; this file only has to agree
// with the adapter's static_assert on the 48-byte record.
#pragma once
#include <utility>
#include <vector>

enum interpolationModelEnum { im_nearest, im_bilinear, im_bicubic, im_NUMBER_OF_ITEMS };
enum fittingModelEnum { fm_U, fm_UV, fm_UVQ, fm_UVUxUyVxVy, fm_NUMBER_OF_ITEMS };
enum errorEnum {
  error_none, error_model_out_of_image, error_interpolation_out_of_image,
  error_correlation_max_iters_reached, error_bad_domain, error_cuSolver, error_cuda,
  error_multiThread, error_NUMBER_OF_ITEMS
};
enum colorEnum { color_monochrome, color_color, color_NUMBER_OF_ITEMS };
enum deformationDescriptionEnum { def_strict_Lagrangian, def_Lagrangian, def_Eulerian, def_NUMBER_OF_ITEMS };

typedef std::vector<std::pair<float, float>> v_points;

struct frame_results { bool empty; }; // the adapter only passes it through by reference

struct CorrelationResult {
  float resultingParameters[6];
  float chi;
  int numberOfPoints;
  int iterations;
  errorEnum errorCode{error_none};
  float undCenterX;
  float undCenterY;
};
