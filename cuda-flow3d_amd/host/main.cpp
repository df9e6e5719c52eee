// flow3d -- command-line application of the MI355X-native 3-D optical-flow solver.
//
// Mirrors what src/main.cpp:53-239 of the reference does (load a pair of RAW volumes, run OpticalFlowE with
// the nine-key parameter bag, write flow-u/v/w as RAW float32), with the compile-time constants turned into
// flags:  flow3d --dims W H D --frames f0.raw f1.raw [f2.raw ...] [--f32] [--out prefix] [--levels N] [--scale s]
//                [--outer N] [--inner N] [--alpha a] [--eps-smooth e] [--eps-data e] [--median r] [--sigma s]
//                [--synthetic] [--vtk] [--stats] [--silent] [--partial [--full] [--budget-mb N]] [--concurrent N]
// More than two frames make a sequence: the driver, its containers and operators are set up once (the reference does
// Initialize / Destroy per pair, src/main.cpp:150,184) and the flow of every consecutive pair is written as
// <prefix>_<k>_flow-{u,v,w}-W-H-D.raw.  --partial runs the out-of-core driver (the reference's use_partial_gpu branch,
// src/main.cpp:187-220): volumes stay in host memory, output files end in "-partial.raw"; --full adds the pre-blur and the
// median the reference's piecemeal driver leaves out, which makes the result equal the resident mode's.
// --concurrent N solves N pairs of a sequence AT ONCE: N host threads, each with a driver and a lane of its own (f3d_lane_*: its
// own stream and container geometry), pair k going to thread k mod N.  Pair k+1 does not depend on pair k, and a small volume
// (up to ~128^3) is a chain of dependent launches of 10-20 us that leaves most of the chip idle: two such chains side by side
// nearly double the pairs per second.  Large volumes fill the chip by themselves and gain nothing.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "f3d_host.h"
#include "hip_utils.h"
#include "optical_flow.h"
#include "optical_flow_p.h"
#include "synth.h"

static void Usage()
{
  std::printf("usage: flow3d --dims W H D (--frames f0.raw f1.raw [f2.raw ...] [--f32] | --synthetic) [--out prefix]\n"
              "              [--levels N] [--scale s] [--outer N] [--inner N] [--alpha a] [--eps-smooth e]\n"
              "              [--eps-data e] [--median r] [--sigma s] [--vtk] [--stats] [--silent] [--partial [--full] [--budget-mb N]]\n"
              "              [--concurrent N]\n");
}

int main(int argc, char** argv)
{
  size_t width = 0, height = 0, depth = 0;
  std::vector<std::string> files;
  std::string prefix = "flow3d";
  bool f32_input = false, synthetic = false, write_vtk = false, silent_mode = false, print_stats = false;
  bool use_partial_gpu = false, partial_full = false;
  size_t concurrent = 1;

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
    else if (a == "--frames") {
      need(2);
      while (i + 1 < argc && std::strncmp(argv[i + 1], "--", 2) != 0) files.push_back(argv[++i]);
    }
    else if (a == "--stats") print_stats = true;
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
    else if (a == "--partial") use_partial_gpu = true;
    else if (a == "--full") partial_full = true;
    else if (a == "--budget-mb") { need(1); setenv("F3D_P_BUDGET_MB", argv[++i], 1); }
    else if (a == "--concurrent") { need(1); concurrent = std::strtoull(argv[++i], nullptr, 10); }
    else { Usage(); return 64; }
  }
  if (width == 0 || height == 0 || depth == 0 || (!synthetic && files.size() < 2)) {
    Usage();
    return 64;
  }

  std::printf("//----------------------------------------------------------------------//\n");
  std::printf("//        3D optical flow, MI355X-native (HIP / CDNA4) implementation     //\n");
  std::printf("//----------------------------------------------------------------------//\n");

  if (!InitDeviceContextWithFirstAvailableDevice()) return 1;

  DataSize4 data_size = {width, height, depth, 0};
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
  auto load = [&](Data3D& frame, const std::string& path) {
    return f32_input ? frame.ReadRAWFromFileF32(path.c_str(), width, height, depth)
                     : frame.ReadRAWFromFileU8(path.c_str(), width, height, depth);
  };
  Data3D frame_0, frame_1;
  Data3D flow_u(width, height, depth), flow_v(width, height, depth), flow_w(width, height, depth);
  const size_t pairs = synthetic ? 1 : files.size() - 1;
  if (synthetic) {
    if (!frame_0.Allocate(width, height, depth) || !frame_1.Allocate(width, height, depth)) return 2;
    f3d_synth::TranslatedGaussianPair(width, height, depth, frame_0.DataPtr(), frame_1.DataPtr());
  } else if (!load(frame_0, files[0])) {
    return 2;
  }
  const std::string suffix = "-" + std::to_string(width) + "-" + std::to_string(height) + "-" + std::to_string(depth) +
                             (use_partial_gpu ? "-partial.raw" : ".raw");

  if (use_partial_gpu) {
    OpticalFlowP optical_flow_p;
    if (!optical_flow_p.Initialize(data_size)) return 3;
    std::printf("Mode: Partial processing mode \n");
    optical_flow_p.silent = silent_mode;
    optical_flow_p.full_pipeline = partial_full;  // pre-blur and median as in the resident mode (the reference's piecemeal driver has neither)
    for (size_t k = 0; k < pairs; ++k) {
      if (!synthetic && !load(frame_1, files[k + 1])) return 2;
      optical_flow_p.ComputeFlow(frame_0, frame_1, flow_u, flow_v, flow_w, params);
      if (print_stats) {
        CudaOperationStatP stat_p;
        Stat3 stat = {0.f, 0.f, 0.f};
        OperationParameters op;
        op.PushValuePtr("flow_u", &flow_u);
        op.PushValuePtr("flow_v", &flow_v);
        op.PushValuePtr("flow_w", &flow_w);
        op.PushValuePtr("data_size", &data_size);
        op.PushValuePtr("stat", &stat);
        stat_p.silent = true;
        if (stat_p.Initialize()) stat_p.Execute(op);
        std::printf("Flow magnitude  min: %8.4f  max: %8.4f  avg: %8.4f\n", stat.min, stat.max, stat.avg);
      }
      const std::string tag = pairs > 1 ? prefix + "_" + std::to_string(k) : prefix;
      flow_u.WriteRAWToFileF32((tag + "_flow-u" + suffix).c_str());
      flow_v.WriteRAWToFileF32((tag + "_flow-v" + suffix).c_str());
      flow_w.WriteRAWToFileF32((tag + "_flow-w" + suffix).c_str());
      if (write_vtk) Data3D::WriteFlowToFileVTK((tag + "_flow.vtk").c_str(), flow_u, flow_v, flow_w);
      std::printf("pair %zu of %zu: %.3f s, %zu solver residencies, %zu levels streamed, %zu levels on the device\n", k + 1, pairs,
                  optical_flow_p.LastDeviceSeconds(), optical_flow_p.LastSolvePasses(), optical_flow_p.LastStreamedLevels(),
                  optical_flow_p.LastResidentLevels());
      if (pairs > 1) frame_0.Swap(frame_1);
    }
    optical_flow_p.Destroy();
    f3d_host_shutdown();
    return 0;
  }

  if (concurrent > 1 && pairs > 1) {
    // N pairs at once: every worker thread binds a lane of its own and drives a driver of its own on it
    const size_t workers = std::min(concurrent, pairs);
    std::printf("Mode: Full GPU mode, %zu pairs at a time\n", workers);
    std::mutex print_mutex;
    std::atomic<int> failed{0};
    const auto t_start = std::chrono::steady_clock::now();
    auto work = [&](size_t me) {
      f3d_lane lane = nullptr;
      if (CheckDeviceError(f3d_lane_create(&lane)) || CheckDeviceError(f3d_lane_make_current(lane))) {
        failed = 3;
        return;
      }
      {
        OpticalFlowE flow;
        Data3D f0, f1, u(width, height, depth), v(width, height, depth), w(width, height, depth);
        bool ok;
        {
          std::lock_guard<std::mutex> lock(print_mutex);   // the set-up lines of one driver at a time
          ok = flow.Initialize(data_size);
        }
        flow.silent = true;
        // the bag holds pointers to main's variables, which nobody writes from here on: the workers share it read-only
        for (size_t k = me; ok && k < pairs && !failed; k += workers) {
          if (!load(f0, files[k]) || !load(f1, files[k + 1])) {
            failed = 2;
            break;
          }
          flow.ComputeFlow(f0, f1, u, v, w, params);
          const std::string tag = prefix + "_" + std::to_string(k);
          u.WriteRAWToFileF32((tag + "_flow-u" + suffix).c_str());
          v.WriteRAWToFileF32((tag + "_flow-v" + suffix).c_str());
          w.WriteRAWToFileF32((tag + "_flow-w" + suffix).c_str());
          if (write_vtk) Data3D::WriteFlowToFileVTK((tag + "_flow.vtk").c_str(), u, v, w);
          std::lock_guard<std::mutex> lock(print_mutex);
          std::printf("pair %zu of %zu done by worker %zu\n", k + 1, pairs, me);
        }
        if (!ok) failed = 3;
        flow.Destroy();
      }
      f3d_lane_make_current(nullptr);
      f3d_lane_destroy(lane);
    };
    std::vector<std::thread> threads;
    for (size_t t = 0; t < workers; ++t) threads.emplace_back(work, t);
    for (std::thread& t : threads) t.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::printf("%zu pairs in %.3f s: %.2f pairs per second with %zu at a time\n", pairs, secs, pairs / secs, workers);
    f3d_host_shutdown();
    return failed;
  }

  OpticalFlowE optical_flow_e;
  if (!optical_flow_e.Initialize(data_size)) {
    std::printf("The resident driver needs 15 containers of the volume on the device; for larger volumes run with --partial\n"
                "(host-resident volumes streamed through the GPU, no pre-blur and no median like the reference's piecemeal driver).\n");
    return 3;
  }
  if (!optical_flow_e.AllocateResidentFrames()) return 3;
  std::printf("Mode: Full GPU mode \n");
  optical_flow_e.silent = silent_mode;
  optical_flow_e.collect_level_statistics = print_stats;

  auto report = [&]() {
    Stat3 stat = {0.f, 0.f, 0.f};
    if (optical_flow_e.ResultStatistics(stat))
      std::printf("Flow magnitude  min: %8.4f  max: %8.4f  avg: %8.4f\n", stat.min, stat.max, stat.avg);
    if (silent_mode)  // otherwise the driver has printed each level as it went
      for (const OpticalFlowE::LevelStatistics& st : optical_flow_e.LevelStats())
        std::printf("level %2d (%4zu x%4zu x%4zu)  residual before the solve: rms %.5f  mean |.| %.5f  max %.4f;  flow after it: "
                    "min %.4f  max %.4f  avg %.4f\n", st.level, st.size.width, st.size.height, st.size.depth, st.before.rms,
                    st.before.mean_abs, st.before.max_abs, st.flow.min, st.flow.max, st.flow.avg);
    OpticalFlowE::Residual reg, unreg;
    if (optical_flow_e.FinalResidual(reg, unreg))
      std::printf("Registration residual (frame_1 warped by the flow vs frame_0)  rms: %.5f  mean |.|: %.5f  max: %.4f   "
                  "(unregistered  rms: %.5f  mean |.|: %.5f  max: %.4f)\n", reg.rms, reg.mean_abs, reg.max_abs, unreg.rms,
                  unreg.mean_abs, unreg.max_abs);
  };
  auto write_pair = [&](size_t k, Data3D& u, Data3D& v, Data3D& w) {
    const std::string tag = pairs > 1 ? prefix + "_" + std::to_string(k) : prefix;
    u.WriteRAWToFileF32((tag + "_flow-u" + suffix).c_str());
    v.WriteRAWToFileF32((tag + "_flow-v" + suffix).c_str());
    w.WriteRAWToFileF32((tag + "_flow-w" + suffix).c_str());
    if (write_vtk) Data3D::WriteFlowToFileVTK((tag + "_flow.vtk").c_str(), u, v, w);
  };

  if (pairs == 1) {
    if (!synthetic && !load(frame_1, files[1])) return 2;
    optical_flow_e.UploadResidentFrames(frame_0, frame_1);
    optical_flow_e.ComputeFlowResident(params);
    if (print_stats) report();
    optical_flow_e.DownloadFlow(flow_u, flow_v, flow_w);
    write_pair(0, flow_u, flow_v, flow_w);
  } else {
    // Sequence: pair k solves on the device while the host reads frame k+2 and uploads it on one copy queue, and downloads and
    // writes the flow of pair k-1 on another -- from and to page-locked buffers (the reference's ALLOCATE_PINNED_MEMORY switch,
    // data3d.cpp:30,57-61), so the copies really run beside the kernels.  A frame crosses the link once although it serves two
    // pairs.  With --stats or without --silent the driver reads results back per level, which serialises the solve with the
    // host; the copies still overlap.
    if (!optical_flow_e.AllocateSequenceFrames()) return 3;
    const DataSize4& c = optical_flow_e.ContainerSize();
    Data3D host_frame[3], host_flow[2][3];
    std::vector<void*> pinned;
    // (volumes of 32 MiB and more only: smaller ones live in the allocator's shared heap, see OpticalFlowP::ComputeFlow)
    auto pin = [&](Data3D& v) {
      const size_t bytes = width * height * depth * sizeof(float);
      if (bytes >= (static_cast<size_t>(32) << 20) && f3d_host_register(v.DataPtr(), bytes) == 0) pinned.push_back(v.DataPtr());
    };
    for (Data3D& f : host_frame)
      if (!f.Allocate(width, height, depth)) return 2;
    for (auto& set : host_flow)
      for (Data3D& f : set)
        if (!f.Allocate(width, height, depth)) return 2;
    for (Data3D& f : host_frame) pin(f);
    for (auto& set : host_flow)
      for (Data3D& f : set) pin(f);
    f3d_queue up = nullptr, down = nullptr;
    f3d_event uploaded[3] = {nullptr, nullptr, nullptr};
    if (CheckDeviceError(f3d_queue_create(&up)) || CheckDeviceError(f3d_queue_create(&down))) return 3;
    for (f3d_event& e : uploaded)
      if (CheckDeviceError(f3d_event_create(&e))) return 3;
    auto upload = [&](size_t frame_index) {  // file -> page-locked buffer -> device container, slot = frame index mod 3
      const int slot = static_cast<int>(frame_index % 3);
      if (!load(host_frame[slot], files[frame_index])) return false;
      CheckDeviceError(f3d_copy_planes_h2d_on(up, optical_flow_e.SequenceFrame(slot), c.pitch, c.height, 0, host_frame[slot].DataPtr(),
                                              width, height, width, height, depth));
      CheckDeviceError(f3d_event_record_on(uploaded[slot], up));
      return true;
    };
    if (!upload(0) || !upload(1)) return 2;
    DevicePtr taken[3] = {0, 0, 0};
    bool pending_output = false;
    const bool serial_sequence = std::getenv("F3D_SEQ_SERIAL") && std::atoi(std::getenv("F3D_SEQ_SERIAL")) != 0;
    for (size_t k = 0; k < pairs; ++k) {
      // the solve of pair k waits (on the device) for the uploads of frames k and k+1
      CheckDeviceError(f3d_queue_wait_event(nullptr, uploaded[k % 3]));
      CheckDeviceError(f3d_queue_wait_event(nullptr, uploaded[(k + 1) % 3]));
      optical_flow_e.SelectResidentPair(static_cast<int>(k % 3), static_cast<int>((k + 1) % 3));
      optical_flow_e.BeginComputeFlowResident(params);
      if (serial_sequence) optical_flow_e.EndComputeFlowResident();  // A/B timing: nothing runs beside the solve
      // beside it: frame k+2 into the container pair k-1 has released, and the previous pair's flow out to its files
      if (k + 2 <= pairs && !upload(k + 2)) return 2;
      if (pending_output) {
        CheckDeviceError(f3d_queue_sync(down));
        optical_flow_e.GiveResultBack(taken);
        write_pair(k - 1, host_flow[(k - 1) & 1][0], host_flow[(k - 1) & 1][1], host_flow[(k - 1) & 1][2]);
        pending_output = false;
      }
      optical_flow_e.EndComputeFlowResident();
      if (print_stats) report();
      std::printf("pair %zu of %zu: %.3f s on the device\n", k + 1, pairs, optical_flow_e.LastDeviceSeconds());
      if (!optical_flow_e.TakeResult(taken)) return 3;
      for (int i = 0; i < 3; ++i)
        CheckDeviceError(f3d_copy_planes_d2h_on(down, host_flow[k & 1][i].DataPtr(), width, height, width, height, depth, taken[i],
                                                c.pitch, c.height, 0));
      pending_output = true;
    }
    CheckDeviceError(f3d_queue_sync(down));
    optical_flow_e.GiveResultBack(taken);
    write_pair(pairs - 1, host_flow[(pairs - 1) & 1][0], host_flow[(pairs - 1) & 1][1], host_flow[(pairs - 1) & 1][2]);
    CheckDeviceError(f3d_queue_sync(up));
    for (f3d_event e : uploaded) f3d_event_destroy(e);
    f3d_queue_destroy(up);
    f3d_queue_destroy(down);
    for (void* p : pinned) f3d_host_unregister(p);
  }

  optical_flow_e.Destroy();
  f3d_host_shutdown();
  return 0;
}
