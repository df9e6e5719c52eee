// OpticalFlowP: the coarse-to-fine loop of src/optical_flow/optical_flow_p.cpp:57-318 on host volumes -- per level
// {resample both frames from the originals, resample the flow in place, warp frame 1, solve, add} -- with the piecemeal
// operators doing the device work chunk by chunk.
#include "optical_flow_p.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "common_utils.h"
#include "hip_utils.h"

OpticalFlowP::OpticalFlowP() : OpticalFlowBase("Optical Flow Single GPU Piecemeal Processing")
{
  // initialisation order of the reference's forward_list (optical_flow_p.cpp:29-33)
  cuda_operations_ = {&cuop_add_p_, &cuop_stat_p_, &cuop_solve_p_, &cuop_resample_p_, &cuop_register_p_};
}

OpticalFlowP::~OpticalFlowP()
{
  if (initialized_) Destroy();
}

bool OpticalFlowP::Initialize(const DataSize4& data_size)
{
  initialized_ = true;
  data_size_ = data_size;
  // the two filters of the full pipeline are not part of the reference's list and stay out of its console lines
  if (!cuop_convolution_p_.Initialize() || !cuop_median_p_.Initialize()) initialized_ = false;
  std::printf("Initialization of cuda operations...\n");
  for (CudaOperationBase* cuop : cuda_operations_) {
    std::printf("%-18s: ", cuop->GetName());
    if (cuop->Initialize()) {
      std::printf("OK\n");
    } else {
      Destroy();
      initialized_ = false;
    }
  }
  return initialized_;
}

void OpticalFlowP::Destroy()
{
  for (CudaOperationBase* cuop : cuda_operations_) cuop->Destroy();
  cuop_convolution_p_.Destroy();
  cuop_median_p_.Destroy();
  PiecemealReleaseArena();
  initialized_ = false;
}

void OpticalFlowP::ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                               OperationParameters& params)
{
  if (!IsInitialized()) return;
  size_t warp_levels_count, outer_iterations_count, inner_iterations_count, median_radius;
  float warp_scale_factor, equation_alpha, equation_smoothness, equation_data, gaussian_sigma;
  GET_PARAM_OR_RETURN(params, size_t, warp_levels_count, "warp_levels_count");
  GET_PARAM_OR_RETURN(params, float, warp_scale_factor, "warp_scale_factor");
  GET_PARAM_OR_RETURN(params, size_t, outer_iterations_count, "outer_iterations_count");
  GET_PARAM_OR_RETURN(params, size_t, inner_iterations_count, "inner_iterations_count");
  GET_PARAM_OR_RETURN(params, float, equation_alpha, "equation_alpha");
  GET_PARAM_OR_RETURN(params, float, equation_smoothness, "equation_smoothness");
  GET_PARAM_OR_RETURN(params, float, equation_data, "equation_data");
  GET_PARAM_OR_RETURN(params, size_t, median_radius, "median_radius");   // read like the reference, not used by this driver
  GET_PARAM_OR_RETURN(params, float, gaussian_sigma, "gaussian_sigma");  // likewise
  const char* full_env = std::getenv("F3D_P_FULL");
  const bool full = full_pipeline || (full_env && full_env[0] == '1');
  size_t level_median = 1;  // 1 = no median, the reference's piecemeal behaviour
  if (full) {
    level_median = median_radius;
    if (level_median != 1 && level_median % 2 == 0) level_median -= 1;
    if (level_median != 1 && (level_median < 3 || level_median > 7)) {
      std::printf("Error. Wrong median raduis (%zu). Supported values: 3, 5, 7\n", level_median);
      return;
    }
  }

  float hx, hy, hz;
  DataSize4 original_data_size = {frame_0.Width(), frame_0.Height(), frame_0.Depth(), 0};
  DataSize4 current_data_size = {0, 0, 0, 0};
  DataSize4 prev_data_size = {0, 0, 0, 0};
  Stat3 flow_stat = {0.f, 0.f, 0.f};
  for (Data3D* v : {&frame_1, &flow_u, &flow_v, &flow_w})
    if (v->Width() != original_data_size.width || v->Height() != original_data_size.height || v->Depth() != original_data_size.depth) {
      std::printf("'%s': Error. Frames and flow volumes must have the same size.\n", GetName());
      return;
    }

  const size_t max_warp_level = GetMaxWarpLevel(original_data_size.width, original_data_size.height, original_data_size.depth, warp_scale_factor);
  int current_warp_level = static_cast<int>(std::min(warp_levels_count, max_warp_level)) - 1;

  const size_t W0 = original_data_size.width, H0 = original_data_size.height, D0 = original_data_size.depth;
  const size_t volume_bytes = W0 * H0 * D0 * sizeof(float);

  // Page-lock what the copies touch.  Data3D::Swap exchanges storage between volumes of this set only, so the pointers
  // registered here are the ones to release at the end.
  std::vector<void*> pinned;
  const char* pin_env = std::getenv("F3D_P_PIN");
  const bool pin = pin_host_memory && !(pin_env && pin_env[0] == '0');
  auto pin_volume = [&](Data3D* v) {
    if (!pin) return;
    if (f3d_host_register(v->DataPtr(), volume_bytes) == 0) {
      pinned.push_back(v->DataPtr());
    } else if (!silent) {
      std::printf("'%s': host memory could not be page-locked (%s); copies will be staged.\n", GetName(), f3d_last_error());
    }
  };
  for (Data3D* v : {&frame_0, &frame_1, &flow_u, &flow_v, &flow_w}) pin_volume(v);

  // Full pipeline: the pyramid reads Gaussian-blurred copies of the two frames (optical_flow_e.cpp:213-242), made here chunk
  // by chunk into two more host volumes; the caller's frames are only read.
  Data3D blur_0, blur_1;
  Data3D* src_0 = &frame_0;
  Data3D* src_1 = &frame_1;
  const bool blur = full && gaussian_sigma > 0.f;
  if (blur) {
    if (!blur_0.Allocate(W0, H0, D0) || !blur_1.Allocate(W0, H0, D0)) {
      for (void* p : pinned) f3d_host_unregister(p);
      return;
    }
    pin_volume(&blur_0);
    pin_volume(&blur_1);
    src_0 = &blur_0;
    src_1 = &blur_1;
  }

  f3d_event ev_start = nullptr, ev_stop = nullptr;
  CheckDeviceError(f3d_event_create(&ev_start));
  CheckDeviceError(f3d_event_create(&ev_stop));
  CheckDeviceError(f3d_event_record(ev_start));
  if (!silent) std::printf("\nStarting optical flow computation...\n");
  solve_passes_ = 0;
  streamed_levels_ = 0;
  resident_levels_ = 0;
  for (double& t : op_seconds_) t = 0.0;
  auto finish = [&]() {
    float elapsed_time = 0.f;
    CheckDeviceError(f3d_event_record(ev_stop));
    CheckDeviceError(f3d_event_sync(ev_stop));
    CheckDeviceError(f3d_event_elapsed_ms(&elapsed_time, ev_start, ev_stop));
    last_device_seconds_ = elapsed_time / 1000.f;
    if (!silent) {
      std::printf("Total GPU computation time: % 4.4fs\n", elapsed_time / 1000.);
      std::printf("  resident levels %.3fs | frames %.3fs  flow resample %.3fs  registration %.3fs  solve %.3fs  add %.3fs\n", op_seconds_[5],
                  op_seconds_[0], op_seconds_[1], op_seconds_[2], op_seconds_[3], op_seconds_[4]);
    }
    f3d_event_destroy(ev_start);
    f3d_event_destroy(ev_stop);
    for (void* p : pinned) f3d_host_unregister(p);
  };

  if (current_warp_level < 0) {  // no level requested: the flow is identically zero
    flow_u.ZeroData();
    flow_v.ZeroData();
    flow_w.ZeroData();
    finish();
    return;
  }

  if (blur) {
    const auto t0 = std::chrono::steady_clock::now();
    Data3D* from[2] = {&frame_0, &frame_1};
    Data3D* to[2] = {&blur_0, &blur_1};
    for (int i = 0; i < 2; ++i) {
      OperationParameters bag;
      bag.PushValuePtr("input", from[i]);
      bag.PushValuePtr("output", to[i]);
      bag.PushValuePtr("data_size", &original_data_size);
      bag.PushValuePtr("gaussian_sigma", &gaussian_sigma);
      cuop_convolution_p_.Execute(bag);
    }
    op_seconds_[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  // ---- coarse levels that fit: on the device -------------------------------------------------------------------------
  const char* res_env = std::getenv("F3D_P_RESIDENT");
  if (resident_coarse_levels && !(res_env && res_env[0] == '0') && current_warp_level >= 0) {
    const size_t budget = PiecemealBudgetBytes();
    auto container_bytes = [&](int level) {
      const DataSize4 s = GetLevel(original_data_size, warp_scale_factor, level).size;
      const size_t pitch = (s.width * sizeof(float) + 255) / 256 * 256;
      return pitch * s.height * s.depth + 66048;  // + what f3d_alloc_pitched adds for alignment and stagger
    };
    const size_t originals_bytes = 2 * (((W0 * sizeof(float) + 255) / 256 * 256) * H0 * D0 + 66048);
    auto fits = [&](int level, bool with_originals) {
      const size_t need = 14 * container_bytes(level) + (with_originals ? originals_bytes : 0);
      const size_t stream = level == 0 && current_warp_level == 0 && !with_originals ? 0 : PiecemealMinResampleBytes(W0, H0);
      return need <= static_cast<size_t>(0.97 * static_cast<double>(budget)) && need + stream <= budget;
    };
    auto last_resident = [&](bool with_originals) {
      int last = current_warp_level + 1;
      while (last > 0 && fits(last - 1, with_originals)) --last;
      return last;
    };
    // The coarsest levels leave room for device copies of the two original frames beside their working set: those levels
    // resample from the copies (phase A).  Finer levels that still fit without the copies follow with the originals streamed
    // through for each of them (phase B, 8 B per original voxel and level over the link); the flow crosses between the two
    // phases through the host, three small sub-boxes.
    const int last_plain = last_resident(false), last_with = last_resident(true);
    const int last = last_plain;
    originals_on_device_ = false;
    if (last <= current_warp_level) {
      const auto t0 = std::chrono::steady_clock::now();
      bool ok = true;
      DataSize4 carried = {0, 0, 0, 0};
      int next = current_warp_level;
      if (last_with <= next) {
        ok = RunResidentLevels(*src_0, *src_1, flow_u, flow_v, flow_w, params, next, last_with, container_bytes(last_with), true, carried,
                               level_median);
        originals_on_device_ = true;
        carried = GetLevel(original_data_size, warp_scale_factor, last_with).size;
        next = last_with - 1;
      }
      if (ok && last <= next)
        ok = RunResidentLevels(*src_0, *src_1, flow_u, flow_v, flow_w, params, next, last, container_bytes(last), false, carried,
                               level_median);
      op_seconds_[5] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (!ok) {
        std::printf("'%s': Error in the resident levels.\n", GetName());
        finish();
        return;
      }
      resident_levels_ = static_cast<size_t>(current_warp_level - last + 1);
      prev_data_size = GetLevel(original_data_size, warp_scale_factor, last).size;
      current_warp_level = last - 1;
      if (current_warp_level < 0) {
        finish();
        return;
      }
    }
  }

  // ---- the remaining levels go through the host --------------------------------------------------------------------
  // Host scratch: eight volumes of the original size (the reference keeps ten: phi and ksi stay on the device here).
  if (!silent) {
    std::printf("Allocating additional memory on the host...\n");
    std::printf("Total RAM memory usage: %.0fMB\n", (5 + 8) * volume_bytes / (1024.f * 1024.f));
  }
  Data3D frame_0_res, frame_1_res_br, flow_du, flow_dv, flow_dw, temp_0, temp_1, temp_2;
  Data3D* own[8] = {&frame_0_res, &frame_1_res_br, &flow_du, &flow_dv, &flow_dw, &temp_0, &temp_1, &temp_2};
  for (Data3D* v : own) {
    if (!v->Allocate(W0, H0, D0)) {
      finish();
      return;
    }
    pin_volume(v);
  }

  Data3D* p_frame_0 = src_0;
  Data3D* p_frame_1 = src_1;
  Data3D* p_frame_0_res = &frame_0_res;
  Data3D* p_frame_1_res_br = &frame_1_res_br;

  // every piecemeal Execute drains the stream before it returns, so host clocks around the calls time the device work
  auto timed = [this](int slot, CudaOperationBase& cuop, OperationParameters& bag) {
    const auto t0 = std::chrono::steady_clock::now();
    cuop.Execute(bag);
    op_seconds_[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  cuop_stat_p_.silent = silent;
  OperationParameters op;

  while (current_warp_level >= 0) {
    const PyramidLevel lv = GetLevel(original_data_size, warp_scale_factor, current_warp_level);
    current_data_size = lv.size;
    hx = lv.hx;
    hy = lv.hy;
    hz = lv.hz;
    if (!silent)
      std::printf("Solve level %2d (%4zu x%4zu x%4zu) \n", current_warp_level, current_data_size.width, current_data_size.height,
                  current_data_size.depth);

    /* Data resampling */
    if (current_warp_level == 0) {
      std::swap(p_frame_0, p_frame_0_res);
      std::swap(p_frame_1, p_frame_1_res_br);
    } else {
      op.Clear();
      op.PushValuePtr("input", p_frame_0);
      op.PushValuePtr("output", p_frame_0_res);
      op.PushValuePtr("data_size", &original_data_size);
      op.PushValuePtr("resample_size", &current_data_size);
      timed(0, cuop_resample_p_, op);

      op.Clear();
      op.PushValuePtr("input", p_frame_1);
      op.PushValuePtr("output", p_frame_1_res_br);
      op.PushValuePtr("data_size", &original_data_size);
      op.PushValuePtr("resample_size", &current_data_size);
      timed(0, cuop_resample_p_, op);
    }

    /* Flow field resampling (in place) */
    if (prev_data_size.width == 0) {
      flow_u.ZeroData();
      flow_v.ZeroData();
      flow_w.ZeroData();
    } else {
      for (Data3D* flow : {&flow_u, &flow_v, &flow_w}) {
        op.Clear();
        op.PushValuePtr("input", flow);
        op.PushValuePtr("output", flow);
        op.PushValuePtr("data_size", &prev_data_size);
        op.PushValuePtr("resample_size", &current_data_size);
        timed(1, cuop_resample_p_, op);
      }
    }

    /* Backward registration: *p_frame_1_res_br and temp_0 trade storage, the warped frame ends up in the former */
    {
      size_t max_magnitude = static_cast<size_t>(std::ceil(flow_stat.max / warp_scale_factor));
      op.Clear();
      op.PushValuePtr("frame_0", p_frame_0_res);
      op.PushValuePtr("frame_1", p_frame_1_res_br);
      op.PushValuePtr("flow_u", &flow_u);
      op.PushValuePtr("flow_v", &flow_v);
      op.PushValuePtr("flow_w", &flow_w);
      op.PushValuePtr("temp", &temp_0);
      op.PushValuePtr("hx", &hx);
      op.PushValuePtr("hy", &hy);
      op.PushValuePtr("hz", &hz);
      op.PushValuePtr("data_size", &current_data_size);
      op.PushValuePtr("max_mag", &max_magnitude);
      timed(2, cuop_register_p_, op);
    }

    /* Difference problem solver */
    {
      op.Clear();
      op.PushValuePtr("frame_0", p_frame_0_res);
      op.PushValuePtr("frame_1", p_frame_1_res_br);
      op.PushValuePtr("flow_u", &flow_u);
      op.PushValuePtr("flow_v", &flow_v);
      op.PushValuePtr("flow_w", &flow_w);
      op.PushValuePtr("flow_du", &flow_du);
      op.PushValuePtr("flow_dv", &flow_dv);
      op.PushValuePtr("flow_dw", &flow_dw);
      op.PushValuePtr("temp_du", &temp_1);
      op.PushValuePtr("temp_dv", &temp_2);
      // at level 0 temp_0 holds the caller's unwarped frame 1 (see below), so the third scratch is the free resample buffer
      Data3D* third = current_warp_level == 0 ? p_frame_1 : &temp_0;
      op.PushValuePtr("temp_dw", third);
      op.PushValuePtr("outer_iterations_count", &outer_iterations_count);
      op.PushValuePtr("inner_iterations_count", &inner_iterations_count);
      op.PushValuePtr("equation_alpha", &equation_alpha);
      op.PushValuePtr("equation_smoothness", &equation_smoothness);
      op.PushValuePtr("equation_data", &equation_data);
      op.PushValuePtr("data_size", &current_data_size);
      op.PushValuePtr("hx", &hx);
      op.PushValuePtr("hy", &hy);
      op.PushValuePtr("hz", &hz);
      cuop_solve_p_.silent = silent;
      timed(3, cuop_solve_p_, op);
      solve_passes_ += cuop_solve_p_.LastPasses();
      if (cuop_solve_p_.LastPlan().halo > 0) ++streamed_levels_;
    }

    /* Add the solved flow increment to the global flow */
    {
      Data3D* flows[3] = {&flow_u, &flow_v, &flow_w};
      Data3D* incs[3] = {&flow_du, &flow_dv, &flow_dw};
      for (int i = 0; i < 3; ++i) {
        op.Clear();
        op.PushValuePtr("operand_0", flows[i]);
        op.PushValuePtr("operand_1", incs[i]);
        op.PushValuePtr("data_size", &current_data_size);
        timed(4, cuop_add_p_, op);
      }
    }

    // At level 0 the registration swapped the CALLER's frame_1 with temp_0; give the caller its storage (and its data,
    // which the warp only read) back.  The reference leaves the warped frame in the caller's volume.
    if (current_warp_level == 0) src_1->Swap(temp_0);

    /* Flow field median filtering (full pipeline only; commented out in the reference, optical_flow_p.cpp:268-302) */
    if (level_median != 1) {
      for (Data3D* flow : {&flow_u, &flow_v, &flow_w}) {
        op.Clear();
        op.PushValuePtr("input", flow);
        op.PushValuePtr("output", flow);
        op.PushValuePtr("data_size", &current_data_size);
        op.PushValuePtr("radius", &level_median);
        timed(4, cuop_median_p_, op);
      }
    }

    prev_data_size = current_data_size;
    --current_warp_level;
  }

  finish();
}

bool OpticalFlowP::RunResidentLevels(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                                     OperationParameters& params, int first_level, int last_level, size_t container_bytes,
                                     bool originals_on_device, const DataSize4& carried_flow_size, size_t median_radius)
{
  size_t outer_iterations_count, inner_iterations_count;
  float warp_scale_factor, equation_alpha, equation_smoothness, equation_data;
  GET_PARAM_OR_RETURN_VALUE(params, float, warp_scale_factor, "warp_scale_factor", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, outer_iterations_count, "outer_iterations_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, inner_iterations_count, "inner_iterations_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_alpha, "equation_alpha", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_smoothness, "equation_smoothness", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_data, "equation_data", false);

  DataSize4 original = {frame_0.Width(), frame_0.Height(), frame_0.Depth(), 0};
  // one compact container geometry for all resident levels: the finest of them
  DataSize4 container = GetLevel(original, warp_scale_factor, last_level).size;
  enum { F0R, F1R, FU, FV, FW, DU, DV, DW, PHI, KSI, TDU, TDV, TDW, TMP, kBuffers };
  DevicePtr buf[kBuffers] = {0};
  bool ok = true;
  const size_t rows = container.height * container.depth;
  for (int i = 0; i < kBuffers && ok; ++i) {
    size_t pitch = 0;
    ok = !CheckDeviceError(f3d_alloc_pitched(&buf[i], &pitch, container.width * sizeof(float), rows));
    container.pitch = pitch;
  }
  // the two original frames on the device, in their own geometry, when the caller found room for them
  DevicePtr orig[2] = {0, 0};
  size_t orig_pitch = 0;
  size_t reserved = kBuffers * container_bytes;
  if (ok && originals_on_device) {
    Data3D* frames[2] = {&frame_0, &frame_1};
    for (int i = 0; i < 2 && ok; ++i) {
      ok = !CheckDeviceError(f3d_alloc_pitched(&orig[i], &orig_pitch, original.width * sizeof(float), original.height * original.depth)) &&
           !CheckDeviceError(f3d_copy_planes_h2d(orig[i], orig_pitch, original.height, 0, frames[i]->DataPtr(), original.width,
                                                 original.height, original.width, original.height, original.depth));
      reserved += orig_pitch * original.height * original.depth + 66048;
    }
  }
  PiecemealSetReservedBytes(reserved);
  auto release = [&]() {
    f3d_stream_sync();
    for (DevicePtr& p : buf) {
      if (p) CheckDeviceError(f3d_free(p));
      p = 0;
    }
    for (DevicePtr& p : orig) {
      if (p) CheckDeviceError(f3d_free(p));
      p = 0;
    }
    PiecemealSetReservedBytes(0);
  };
  if (!ok) {
    release();
    return false;
  }
  OperationParameters init;
  init.PushValuePtr("container_size", &container);
  for (CudaOperationBase* cuop : {static_cast<CudaOperationBase*>(&cuop_resample_e_), static_cast<CudaOperationBase*>(&cuop_register_e_),
                                  static_cast<CudaOperationBase*>(&cuop_solve_e_), static_cast<CudaOperationBase*>(&cuop_add_e_),
                                  static_cast<CudaOperationBase*>(&cuop_median_e_)})
    ok = cuop->Initialize(&init) && ok;

  OperationParameters op;
  DataSize4 prev = carried_flow_size;
  if (ok && prev.width != 0) {  // the flow an earlier phase left in the host volumes' sub-box
    Data3D* flows[3] = {&flow_u, &flow_v, &flow_w};
    for (int i = 0; i < 3 && ok; ++i)
      ok = !CheckDeviceError(f3d_copy_planes_h2d(buf[FU + i], container.pitch, container.height, 0, flows[i]->DataPtr(), flows[i]->Width(),
                                                 flows[i]->Height(), prev.width, prev.height, prev.depth));
  }
  for (int level = first_level; level >= last_level && ok; --level) {
    const PyramidLevel lv = GetLevel(original, warp_scale_factor, level);
    DataSize4 current = lv.size;
    float hx = lv.hx, hy = lv.hy, hz = lv.hz;
    if (!silent)
      std::printf("Solve level %2d (%4zu x%4zu x%4zu) on the device\n", level, current.width, current.height, current.depth);

    // frames of this level straight into device containers
    if (level == 0 && originals_on_device) {
      ok = !CheckDeviceError(f3d_copy_rect_d2d(buf[F0R], container.pitch, container.height, 0, orig[0], orig_pitch, original.height, 0,
                                               original.width, original.height, original.depth)) &&
           !CheckDeviceError(f3d_copy_rect_d2d(buf[F1R], container.pitch, container.height, 0, orig[1], orig_pitch, original.height, 0,
                                               original.width, original.height, original.depth));
    } else if (level == 0) {
      ok = !CheckDeviceError(f3d_copy_planes_h2d(buf[F0R], container.pitch, container.height, 0, frame_0.DataPtr(), original.width,
                                                 original.height, original.width, original.height, original.depth)) &&
           !CheckDeviceError(f3d_copy_planes_h2d(buf[F1R], container.pitch, container.height, 0, frame_1.DataPtr(), original.width,
                                                 original.height, original.width, original.height, original.depth));
    } else if (originals_on_device) {
      ok = cuop_resample_p_.ExecuteDeviceToDevice(orig[0], orig_pitch, original.height, original, current, buf[F0R], container.pitch,
                                                  container.height) &&
           cuop_resample_p_.ExecuteDeviceToDevice(orig[1], orig_pitch, original.height, original, current, buf[F1R], container.pitch,
                                                  container.height);
    } else {
      ok = cuop_resample_p_.ExecuteToDevice(frame_0, original, current, buf[F0R], container.pitch, container.height) &&
           cuop_resample_p_.ExecuteToDevice(frame_1, original, current, buf[F1R], container.pitch, container.height);
    }
    if (!ok) break;

    // flow of the previous level (values stay in original-voxel units)
    if (prev.width == 0) {
      for (int i = FU; i <= FW; ++i)
        ok = !CheckDeviceError(f3d_memset2d(buf[i], container.pitch, 0, container.width * sizeof(float), rows)) && ok;
    } else {
      for (int i = 0; i < 3; ++i) {
        op.Clear();
        op.PushValuePtr("dev_input", &buf[FU + i]);
        op.PushValuePtr("dev_output", &buf[DU + i]);
        op.PushValuePtr("dev_temp", &buf[TMP]);
        op.PushValuePtr("data_size", &prev);
        op.PushValuePtr("resample_size", &current);
        cuop_resample_e_.Execute(op);
        std::swap(buf[FU + i], buf[DU + i]);
      }
    }

    // backward registration
    op.Clear();
    op.PushValuePtr("dev_frame_0", &buf[F0R]);
    op.PushValuePtr("dev_frame_1", &buf[F1R]);
    op.PushValuePtr("dev_flow_u", &buf[FU]);
    op.PushValuePtr("dev_flow_v", &buf[FV]);
    op.PushValuePtr("dev_flow_w", &buf[FW]);
    op.PushValuePtr("dev_output", &buf[TMP]);
    op.PushValuePtr("data_size", &current);
    op.PushValuePtr("hx", &hx);
    op.PushValuePtr("hy", &hy);
    op.PushValuePtr("hz", &hz);
    cuop_register_e_.Execute(op);
    std::swap(buf[F1R], buf[TMP]);

    // difference problem
    op.Clear();
    op.PushValuePtr("dev_frame_0", &buf[F0R]);
    op.PushValuePtr("dev_frame_1", &buf[F1R]);
    op.PushValuePtr("dev_flow_u", &buf[FU]);
    op.PushValuePtr("dev_flow_v", &buf[FV]);
    op.PushValuePtr("dev_flow_w", &buf[FW]);
    op.PushValuePtr("dev_flow_du", &buf[DU]);
    op.PushValuePtr("dev_flow_dv", &buf[DV]);
    op.PushValuePtr("dev_flow_dw", &buf[DW]);
    op.PushValuePtr("dev_phi", &buf[PHI]);
    op.PushValuePtr("dev_ksi", &buf[KSI]);
    op.PushValuePtr("dev_temp_du", &buf[TDU]);
    op.PushValuePtr("dev_temp_dv", &buf[TDV]);
    op.PushValuePtr("dev_temp_dw", &buf[TDW]);
    op.PushValuePtr("outer_iterations_count", &outer_iterations_count);
    op.PushValuePtr("inner_iterations_count", &inner_iterations_count);
    op.PushValuePtr("equation_alpha", &equation_alpha);
    op.PushValuePtr("equation_smoothness", &equation_smoothness);
    op.PushValuePtr("equation_data", &equation_data);
    op.PushValuePtr("data_size", &current);
    op.PushValuePtr("hx", &hx);
    op.PushValuePtr("hy", &hy);
    op.PushValuePtr("hz", &hz);
    cuop_solve_e_.silent = true;
    cuop_solve_e_.Execute(op);
    ++solve_passes_;

    for (int i = 0; i < 3; ++i) {
      op.Clear();
      op.PushValuePtr("operand_0", &buf[FU + i]);
      op.PushValuePtr("operand_1", &buf[DU + i]);
      op.PushValuePtr("data_size", &current);
      cuop_add_e_.Execute(op);
    }
    if (median_radius != 1) {
      for (int i = 0; i < 3; ++i) {
        op.Clear();
        op.PushValuePtr("dev_input", &buf[FU + i]);
        op.PushValuePtr("dev_output", &buf[TMP]);
        op.PushValuePtr("data_size", &current);
        op.PushValuePtr("radius", &median_radius);
        cuop_median_e_.Execute(op);
        std::swap(buf[FU + i], buf[TMP]);
      }
    }
    prev = current;
  }

  // hand the flow to the host: the sub-box of the last resident level (the whole volume when that is level 0)
  if (ok) {
    Data3D* flows[3] = {&flow_u, &flow_v, &flow_w};
    for (int i = 0; i < 3 && ok; ++i)
      ok = !CheckDeviceError(f3d_copy_planes_d2h(flows[i]->DataPtr(), flows[i]->Width(), flows[i]->Height(), prev.width, prev.height, prev.depth,
                                                 buf[FU + i], container.pitch, container.height, 0));
    ok = !CheckDeviceError(f3d_stream_sync()) && ok;
  }
  for (CudaOperationBase* cuop : {static_cast<CudaOperationBase*>(&cuop_resample_e_), static_cast<CudaOperationBase*>(&cuop_register_e_),
                                  static_cast<CudaOperationBase*>(&cuop_solve_e_), static_cast<CudaOperationBase*>(&cuop_add_e_),
                                  static_cast<CudaOperationBase*>(&cuop_median_e_)})
    cuop->Destroy();
  release();
  return ok;
}
