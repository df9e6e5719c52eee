// The parameter keys of the six "entire data" operators, in one place.
//
// A driver of the reference talks to an operator through a bag of string -> pointer pairs (keys read at
// cuda_operation_solve.cpp:89-138, _registration.cpp:83-98, _resample.cpp:84-88, _add.cpp:77-79, _median.cpp:81-87,
// _convolution.cpp:146-150).  The key strings ARE the interface and are kept verbatim; everything else about a call -- which of the
// driver's variables plays which role -- is the driver's own business.  Each function below takes the roles by name and fills the bag
// with the operator's keys, so the three drivers of this repository (resident, out-of-core resident levels, statistics) state a call in
// one line and the keys are written once.  The bag owns nothing: every pointer must outlive the Execute() it is handed to.
#ifndef F3D_HOST_OPERATOR_CALLS_H_
#define F3D_HOST_OPERATOR_CALLS_H_

#include "common_utils.h"
#include "data_types.h"
#include "hip_utils.h"

namespace calls {

struct Spacing {  // grid spacing of a pyramid level in original voxels
  float *hx, *hy, *hz;
};
struct Flow {  // three device containers: (u, v, w) or (du, dv, dw)
  DevicePtr *u, *v, *w;
};
struct SolverSettings {
  size_t *outer_iterations, *inner_iterations;
  float *alpha, *smoothness, *data;
};

inline OperationParameters& Convolution(OperationParameters& bag, DevicePtr* input, DevicePtr* output, DevicePtr* temp, DataSize4* size,
                                        float* sigma)
{
  return FillBag(bag, {{"dev_input", input}, {"dev_output", output}, {"dev_temp", temp}, {"data_size", size}, {"gaussian_sigma", sigma}});
}

inline OperationParameters& Resample(OperationParameters& bag, DevicePtr* input, DevicePtr* output, DevicePtr* temp, DataSize4* from,
                                     DataSize4* to)
{
  return FillBag(bag, {{"dev_input", input}, {"dev_output", output}, {"dev_temp", temp}, {"data_size", from}, {"resample_size", to}});
}

inline OperationParameters& Registration(OperationParameters& bag, DevicePtr* frame_0, DevicePtr* frame_1, const Flow& flow,
                                         DevicePtr* output, DataSize4* size, const Spacing& h)
{
  return FillBag(bag, {{"dev_frame_0", frame_0}, {"dev_frame_1", frame_1}, {"dev_flow_u", flow.u}, {"dev_flow_v", flow.v},
                       {"dev_flow_w", flow.w}, {"dev_output", output}, {"data_size", size}, {"hx", h.hx}, {"hy", h.hy}, {"hz", h.hz}});
}

// `partner`: the ping-pong partners of the increments (the operator swaps the six pointers through the bag)
inline OperationParameters& Solve(OperationParameters& bag, DevicePtr* frame_0, DevicePtr* frame_1_registered, const Flow& flow,
                                  const Flow& increment, const Flow& partner, DevicePtr* phi, DevicePtr* ksi, const SolverSettings& s,
                                  DataSize4* size, const Spacing& h)
{
  return FillBag(bag, {{"dev_frame_0", frame_0}, {"dev_frame_1", frame_1_registered}, {"dev_flow_u", flow.u}, {"dev_flow_v", flow.v},
                       {"dev_flow_w", flow.w}, {"dev_flow_du", increment.u}, {"dev_flow_dv", increment.v}, {"dev_flow_dw", increment.w},
                       {"dev_phi", phi}, {"dev_ksi", ksi}, {"dev_temp_du", partner.u}, {"dev_temp_dv", partner.v},
                       {"dev_temp_dw", partner.w}, {"outer_iterations_count", s.outer_iterations},
                       {"inner_iterations_count", s.inner_iterations}, {"equation_alpha", s.alpha}, {"equation_smoothness", s.smoothness},
                       {"equation_data", s.data}, {"data_size", size}, {"hx", h.hx}, {"hy", h.hy}, {"hz", h.hz}});
}

inline OperationParameters& Add(OperationParameters& bag, DevicePtr* accumulator, DevicePtr* addend, DataSize4* size)
{
  return FillBag(bag, {{"operand_0", accumulator}, {"operand_1", addend}, {"data_size", size}});
}

inline OperationParameters& Median(OperationParameters& bag, DevicePtr* input, DevicePtr* output, DataSize4* size, size_t* window)
{
  return FillBag(bag, {{"dev_input", input}, {"dev_output", output}, {"data_size", size}, {"radius", window}});
}

inline OperationParameters& Statistics(OperationParameters& bag, const Flow& flow, DataSize4* size, Stat3* result)
{
  return FillBag(bag, {{"dev_flow_u", flow.u}, {"dev_flow_v", flow.v}, {"dev_flow_w", flow.w}, {"data_size", size}, {"stat", result}});
}

}  // namespace calls

#endif
