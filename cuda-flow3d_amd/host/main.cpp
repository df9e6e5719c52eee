// flow3d -- command-line application of the MI355X-native 3-D optical-flow solver.
//
// Mirrors what src/main.cpp:53-239 of the reference does (load a pair of RAW volumes, run OpticalFlowE with
// the nine-key parameter bag, write flow-u/v/w as RAW float32), with the compile-time constants turned into
// flags:  flow3d --dims W H D --frames f0.raw f1.raw [--f32] [--out prefix] [--levels N] [--scale s]
//                [--outer N] [--inner N] [--alpha a] [--eps-smooth e] [--eps-data e] [--median r] [--sigma s]
//                [--synthetic] [--vtk] [--silent]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "hip_utils.h"
#include "optical_flow.h"
#include "synth.h"

static void Usage()
{
  std::printf("usage: flow3d --dims W H D (--frames f0.raw f1.raw [--f32] | --synthetic) [--out prefix]\n"
              "              [--levels N] [--scale s] [--outer N] [--inner N] [--alpha a] [--eps-smooth e]\n"
              "              [--eps-data e] [--median r] [--sigma s] [--vtk] [--silent]\n");
}

int main(int argc, char** argv)
{
  size_t width = 0, height = 0, depth = 0;
  std::string file_0, file_1, prefix = "flow3d";
  bool f32_input = false, synthetic = false, write_vtk = false, silent_mode = false;

  // defaults of src/main.cpp:77-85
  size_t warp_levels_count = 40;
  float warp_scale_factor = 0.95f;
  size_t outer_iterations_count = 40;
  size_t inner_iterations_count = 5;
  float equation_alpha = 7.5f;
  float equation_smoothness = 0.001f;
  float equation_data = 0.001f;
  size_t median_radius = 5;
  float gaussian_sigma = 2.0f;

  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto need = [&](int n) {
      if (i + n >= argc) {
        Usage();
        std::exit(64);
      }
    };
    if (a == "--dims") { need(3); width = std::strtoull(argv[++i], nullptr, 10); height = std::strtoull(argv[++i], nullptr, 10); depth = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--frames") { need(2); file_0 = argv[++i]; file_1 = argv[++i]; }
    else if (a == "--out") { need(1); prefix = argv[++i]; }
    else if (a == "--levels") { need(1); warp_levels_count = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--scale") { need(1); warp_scale_factor = std::strtof(argv[++i], nullptr); }
    else if (a == "--outer") { need(1); outer_iterations_count = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--inner") { need(1); inner_iterations_count = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--alpha") { need(1); equation_alpha = std::strtof(argv[++i], nullptr); }
    else if (a == "--eps-smooth") { need(1); equation_smoothness = std::strtof(argv[++i], nullptr); }
    else if (a == "--eps-data") { need(1); equation_data = std::strtof(argv[++i], nullptr); }
    else if (a == "--median") { need(1); median_radius = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--sigma") { need(1); gaussian_sigma = std::strtof(argv[++i], nullptr); }
    else if (a == "--f32") f32_input = true;
    else if (a == "--synthetic") synthetic = true;
    else if (a == "--vtk") write_vtk = true;
    else if (a == "--silent") silent_mode = true;
    else { Usage(); return 64; }
  }
  if (width == 0 || height == 0 || depth == 0 || (!synthetic && file_0.empty())) {
    Usage();
    return 64;
  }

  std::printf("//----------------------------------------------------------------------//\n");
  std::printf("//        3D optical flow, MI355X-native (HIP / CDNA4) implementation     //\n");
  std::printf("//----------------------------------------------------------------------//\n");

  if (!InitDeviceContextWithFirstAvailableDevice()) return 1;

  Data3D frame_0, frame_1;
  if (synthetic) {
    if (!frame_0.Allocate(width, height, depth) || !frame_1.Allocate(width, height, depth)) return 2;
    f3d_synth::TranslatedGaussianPair(width, height, depth, frame_0.DataPtr(), frame_1.DataPtr());
  } else {
    const bool ok = f32_input ? (frame_0.ReadRAWFromFileF32(file_0.c_str(), width, height, depth) &&
                                 frame_1.ReadRAWFromFileF32(file_1.c_str(), width, height, depth))
                              : (frame_0.ReadRAWFromFileU8(file_0.c_str(), width, height, depth) &&
                                 frame_1.ReadRAWFromFileU8(file_1.c_str(), width, height, depth));
    if (!ok) return 2;
  }

  DataSize4 data_size = {width, height, depth, 0};
  OpticalFlowE optical_flow_e;
  if (!optical_flow_e.Initialize(data_size)) return 3;

  Data3D flow_u(width, height, depth), flow_v(width, height, depth), flow_w(width, height, depth);
  OperationParameters params;
  params.PushValuePtr("warp_levels_count", &warp_levels_count);
  params.PushValuePtr("warp_scale_factor", &warp_scale_factor);
  params.PushValuePtr("outer_iterations_count", &outer_iterations_count);
  params.PushValuePtr("inner_iterations_count", &inner_iterations_count);
  params.PushValuePtr("equation_alpha", &equation_alpha);
  params.PushValuePtr("equation_smoothness", &equation_smoothness);
  params.PushValuePtr("equation_data", &equation_data);
  params.PushValuePtr("median_radius", &median_radius);
  params.PushValuePtr("gaussian_sigma", &gaussian_sigma);

  std::printf("Mode: Full GPU mode \n");
  optical_flow_e.silent = silent_mode;
  optical_flow_e.ComputeFlow(frame_0, frame_1, flow_u, flow_v, flow_w, params);

  const std::string suffix =
      "-" + std::to_string(width) + "-" + std::to_string(height) + "-" + std::to_string(depth) + ".raw";
  flow_u.WriteRAWToFileF32((prefix + "_flow-u" + suffix).c_str());
  flow_v.WriteRAWToFileF32((prefix + "_flow-v" + suffix).c_str());
  flow_w.WriteRAWToFileF32((prefix + "_flow-w" + suffix).c_str());
  if (write_vtk) Data3D::WriteFlowToFileVTK((prefix + "_flow.vtk").c_str(), flow_u, flow_v, flow_w);

  optical_flow_e.Destroy();
  f3d_shutdown();
  return 0;
}
