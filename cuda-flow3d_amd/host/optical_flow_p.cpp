// OpticalFlowP: the coarse-to-fine loop of src/optical_flow/optical_flow_p.cpp:57-318 on host volumes -- per level
// {resample both frames from the originals, resample the flow in place, warp frame 1, solve, add} -- with the piecemeal
// operators doing the device work chunk by chunk.
#include "optical_flow_p.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common_utils.h"
#include "operator_calls.h"
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
  // Only volumes of 32 MiB and more: an allocator serves smaller requests from its shared heap (glibc: below its mmap threshold, which
  // adapts up to 32 MiB), where a volume shares its first and last page with its neighbours and the heap's top is unmapped and mapped
  // again as it shrinks and grows -- and a ten-minute soak of small runs with page-locked heap volumes ended in a GPU memory access
  // fault on a heap address once in ~4 000 runs, not once in 11 000 with every volume in a mapping of its own (LABBOOK, round 4).
  // Copies of small volumes are staged by the runtime, which is what they cost anyway.  F3D_P_PIN=0: never; =2: every size (the tests
  // of the overlapped schedule on small volumes; run them with MALLOC_MMAP_THRESHOLD_=131072).
  std::vector<void*> pinned;
  const char* pin_env = std::getenv("F3D_P_PIN");
  const int pin_mode = pin_env ? std::atoi(pin_env) : 1;
  const bool pin = pin_host_memory && pin_mode != 0 && (pin_mode == 2 || volume_bytes >= (static_cast<size_t>(32) << 20));
  auto pin_volume = [&](Data3D* v) {
    if (!pin) return;
    if (f3d_host_register(v->DataPtr(), volume_bytes) == 0) {
      pinned.push_back(v->DataPtr());
    } else if (!silent) {
      std::printf("'%s': host memory could not be page-locked (%s); copies will be staged.\n", GetName(), f3d_last_error());
    }
  };
  // (one after the other: four or eight ranges at once on worker threads made a 1024^3 run 0.6 / 1.0 s SLOWER, profiles/r04_piecemeal_host_side.txt)
  auto pin_volumes = [&](std::vector<Data3D*> volumes) {
    for (Data3D* v : volumes) pin_volume(v);
  };
  pin_volumes({&frame_0, &frame_1, &flow_u, &flow_v, &flow_w});

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
    pin_volumes({&blur_0, &blur_1});
    src_0 = &blur_0;
    src_1 = &blur_1;
  }

  // Host scratch of the levels that go through the host: eight volumes of the original size (the reference keeps ten: phi and ksi stay
  // on the device here), allocated and page-locked when the first such level is reached.  (Doing that on a helper thread beside the
  // resident levels was built and measured twice: page-locking stalls the submission of kernels, the resident levels lose what the
  // thread saves -- 30.57 s in line, 30.70 s beside, profiles/r04_piecemeal_huge_pages.txt -- so it is done in line.  Asking for
  // transparent huge pages before page-locking saved 0.4 s of the 1.9 s and was taken out as well: a ten-minute soak of this path died of
  // a GPU memory access fault on a host address after 3 000 runs with it in, LABBOOK.)
  Data3D scratch[8];
  bool scratch_ok = true;
  auto prepare_scratch = [&]() {
    std::vector<Data3D*> all;
    for (Data3D& v : scratch) {
      if (!v.Allocate(W0, H0, D0)) {
        scratch_ok = false;
        return;
      }
      all.push_back(&v);
    }
    pin_volumes(all);
  };

  f3d_event ev_start = nullptr, ev_stop = nullptr;
  CheckDeviceError(f3d_event_create(&ev_start));
  CheckDeviceError(f3d_event_create(&ev_stop));
  CheckDeviceError(f3d_event_record(ev_start));
  if (!silent) std::printf("\nStarting optical flow computation...\n");
  solve_passes_ = 0;
  streamed_levels_ = 0;
  levels_registered_inside_ = 0;
  levels_with_constants_ = 0;
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
  if (!silent) {
    std::printf("Allocating additional memory on the host...\n");
    std::printf("Total RAM memory usage: %.0fMB\n", (5 + 8) * volume_bytes / (1024.f * 1024.f));
  }
  prepare_scratch();
  if (!scratch_ok) {
    finish();
    return;
  }
  // Roles of the host volumes in a level (the operator keys they are bound to are the reference's, optical_flow_p.cpp:152-266):
  //   whole[2]    the two frames at the original size (the caller's, or their blurred copies) -- resampled FROM at every level
  //   level[2]    the two frames at the level's size; level[1] is replaced by its registered version
  //   flow[3]     u, v, w: the caller's volumes, resampled in place from level to level
  //   step[3]     the solver's increments du, dv, dw
  //   spare[3]    scratch: the registration's output, two (three) of the solver's ping-pong partners
  Data3D* whole[2] = {src_0, src_1};
  Data3D* level[2] = {&scratch[0], &scratch[1]};
  Data3D* flow[3] = {&flow_u, &flow_v, &flow_w};
  Data3D* step[3] = {&scratch[2], &scratch[3], &scratch[4]};
  Data3D* spare[3] = {&scratch[5], &scratch[6], &scratch[7]};

  // every piecemeal Execute drains the stream before it returns, so host clocks around the calls time the device work
  OperationParameters bag;
  auto run = [this, &bag](int clock, CudaOperationBase& cuop, std::initializer_list<BagEntry> call) {
    FillBag(bag, call);
    const auto t0 = std::chrono::steady_clock::now();
    cuop.Execute(bag);
    op_seconds_[clock] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  enum Clock { kFrames = 0, kFlowResample = 1, kRegistration = 2, kSolve = 3, kAddAndMedian = 4 };
  cuop_stat_p_.silent = silent;
  cuop_solve_p_.silent = silent;
  // "flow += increments" inside the solver's last residency (9 field transfers per level less: CudaOperationSolveP::
  // add_increments_to_flow).  The sums come back in the increments' volumes, so such a level trades the STORAGE of flow[c] and
  // step[c] (Data3D::Swap: the caller's objects stay the flow, the solver's pass-to-pass swaps keep to the driver's own volumes); an
  // even number of trades leaves the caller's objects on their own buffers, so with an odd number of host levels the first -- the
  // smallest -- keeps the separate add.  F3D_P_FUSED_ADD=0 keeps it everywhere.
  const char* fw_env = std::getenv("F3D_P_FUSED_WARP");
  const bool register_inside_allowed = !(fw_env && fw_env[0] == '0');
  const char* fa_env = std::getenv("F3D_P_FUSED_ADD");
  const bool fuse_add_allowed = !(fa_env && fa_env[0] == '0');
  const int host_levels = current_warp_level + 1;
  const int first_fused_level = fuse_add_allowed ? host_levels - (host_levels % 2) - 1 : -1;   // levels first_fused_level .. 0 fuse
  float* const callers_storage[3] = {flow[0]->DataPtr(), flow[1]->DataPtr(), flow[2]->DataPtr()};

  for (; current_warp_level >= 0; --current_warp_level) {
    const bool finest = current_warp_level == 0;
    const PyramidLevel geometry = GetLevel(original_data_size, warp_scale_factor, current_warp_level);
    current_data_size = geometry.size;
    hx = geometry.hx;
    hy = geometry.hy;
    hz = geometry.hz;
    if (!silent)
      std::printf("Solve level %2d (%4zu x%4zu x%4zu) \n", current_warp_level, current_data_size.width, current_data_size.height,
                  current_data_size.depth);

    // 1. the frames of this level: the originals themselves at the finest level (their storage changes roles), area-resampled
    //    copies of the ORIGINALS everywhere else
    for (int f = 0; f < 2; ++f) {
      if (finest)
        std::swap(whole[f], level[f]);
      else
        run(kFrames, cuop_resample_p_,
            {{"input", whole[f]}, {"output", level[f]}, {"data_size", &original_data_size}, {"resample_size", &current_data_size}});
    }

    // 2. the flow so far at this level's size, in place (zero before the first level); values stay in original-voxel units
    for (Data3D* component : flow) {
      if (prev_data_size.width == 0)
        component->ZeroData();
      else
        run(kFlowResample, cuop_resample_p_,
            {{"input", component}, {"output", component}, {"data_size", &prev_data_size}, {"resample_size", &current_data_size}});
    }

    // 3. frame 1 registered with that flow.  Inside the solver's first residency where that is possible (register_frame_1: the
    //    solver has frame 0, u, v, w on the device for the chunk anyway) -- level[1] holds the registered frame afterwards; at the
    //    finest level, where level[1] is the CALLER's frame 1, the registered frame collects in spare[0] and the caller's is only read.
    //    Otherwise by the operator, which writes into spare[0] and trades its storage with level[1] (spare[0]: the unregistered one).
    bool registration_traded_storage = false;
    auto register_separately = [&]() {
      size_t max_magnitude = static_cast<size_t>(std::ceil(flow_stat.max / warp_scale_factor));
      run(kRegistration, cuop_register_p_,
          {{"frame_0", level[0]}, {"frame_1", level[1]}, {"flow_u", flow[0]}, {"flow_v", flow[1]}, {"flow_w", flow[2]}, {"temp", spare[0]},
           {"hx", &hx}, {"hy", &hy}, {"hz", &hz}, {"data_size", &current_data_size}, {"max_mag", &max_magnitude}});
      registration_traded_storage = true;
    };
    if (!register_inside_allowed) register_separately();

    // 4. the increments.  The third ping-pong partner is spare[0] -- except at the finest level, where spare[0] holds the caller's
    //    unregistered frame 1 or the registered one (step 6) and the volume the frames were resampled from is free instead
    Data3D* third_partner = finest ? whole[1] : spare[0];
    cuop_solve_p_.add_increments_to_flow = current_warp_level <= first_fused_level;
    for (int attempt = 0; attempt < 2; ++attempt) {
      cuop_solve_p_.register_frame_1 = !registration_traded_storage;
      cuop_solve_p_.registered_frame_1 = finest ? spare[0] : nullptr;
      run(kSolve, cuop_solve_p_,
          {{"frame_0", level[0]}, {"frame_1", level[1]}, {"flow_u", flow[0]}, {"flow_v", flow[1]}, {"flow_w", flow[2]},
           {"flow_du", step[0]}, {"flow_dv", step[1]}, {"flow_dw", step[2]}, {"temp_du", spare[1]}, {"temp_dv", spare[2]},
           {"temp_dw", third_partner}, {"outer_iterations_count", &outer_iterations_count},
           {"inner_iterations_count", &inner_iterations_count}, {"equation_alpha", &equation_alpha},
           {"equation_smoothness", &equation_smoothness}, {"equation_data", &equation_data}, {"data_size", &current_data_size},
           {"hx", &hx}, {"hy", &hy}, {"hz", &hz}});
      if (registration_traded_storage || cuop_solve_p_.LastRegistered()) break;
      register_separately();   // the solver declined (a reach too deep for its buffers) and has done nothing: the classical order
    }
    if (!silent)
      std::printf("  solver of level %d: %.3f s so far in all levels, %zu residencies, chunks of %d planes + 2 x %d, %s%s\n", current_warp_level,
                  op_seconds_[kSolve], cuop_solve_p_.LastPasses(), cuop_solve_p_.LastPlan().chunk, cuop_solve_p_.LastPlan().halo,
                  cuop_solve_p_.LastPlan().overlapped ? "copies beside the kernels" : "in order",
                  cuop_solve_p_.LastPlan().constants_on_device ? ", frames and flow held on the device" : "");
    if (cuop_solve_p_.LastRegistered()) ++levels_registered_inside_;
    if (cuop_solve_p_.LastPlan().constants_on_device && cuop_solve_p_.LastPasses() > 0) ++levels_with_constants_;
    solve_passes_ += cuop_solve_p_.LastPasses();
    if (cuop_solve_p_.LastPlan().halo > 0) ++streamed_levels_;

    // 5. flow += increments: done by the solver's last residency where it was asked to (the sums are in step[]: trade the storage),
    //    by the add operator otherwise
    if (cuop_solve_p_.LastAddedToFlow()) {
      for (int c = 0; c < 3; ++c) flow[c]->Swap(*step[c]);
    } else {
      for (int c = 0; c < 3; ++c)
        run(kAddAndMedian, cuop_add_p_, {{"operand_0", flow[c]}, {"operand_1", step[c]}, {"data_size", &current_data_size}});
    }

    // 6. At the finest level step 3 traded the CALLER's frame 1 for spare[0]: the caller gets its storage (and its data, which
    //    the registration only read) back.  The reference leaves the registered frame in the caller's volume.
    if (finest && registration_traded_storage) src_1->Swap(*spare[0]);

    // 7. median of every component, in place (full pipeline only: commented out in the reference, optical_flow_p.cpp:268-302)
    if (level_median != 1)
      for (Data3D* component : flow)
        run(kAddAndMedian, cuop_median_p_,
            {{"input", component}, {"output", component}, {"data_size", &current_data_size}, {"radius", &level_median}});

    prev_data_size = current_data_size;
  }
  cuop_solve_p_.add_increments_to_flow = false;
  cuop_solve_p_.register_frame_1 = false;
  cuop_solve_p_.registered_frame_1 = nullptr;
  // The flow must end in the caller's own buffers.  A trade hands the caller's buffer to the driver's volumes, among which the
  // solver's pass-to-pass swaps and the registration move it on -- also into the volume of ANOTHER component (the registered frame's
  // volume takes part in u's trades at one level and in w's at another) -- so after the last level a result may sit anywhere.  Two
  // steps, a copy each (all cores; a fraction of what the saved transfers cost): a result that sits in another component's buffer
  // moves to a scratch buffer that is nobody's; then every caller volume takes its own buffer back from whichever scratch volume
  // holds it and the result is copied across.  Volumes that never left are left alone.
  const long long count = static_cast<long long>(W0) * static_cast<long long>(H0) * static_cast<long long>(D0);
  auto copy_volume = [&](float* to, const float* from) {
#pragma omp parallel for schedule(static)
    for (long long block = 0; block < (count + (1 << 20) - 1) / (1 << 20); ++block) {
      const long long lo = block << 20, n = std::min<long long>(1 << 20, count - lo);
      std::memcpy(to + lo, from + lo, static_cast<size_t>(n) * sizeof(float));
    }
  };
  auto callers = [&](const float* p) { return p == callers_storage[0] || p == callers_storage[1] || p == callers_storage[2]; };
  bool restored = true;
  for (int c = 0; c < 3 && restored; ++c) {
    if (flow[c]->DataPtr() == callers_storage[c] || !callers(flow[c]->DataPtr())) continue;
    Data3D* nobodys = nullptr;
    for (Data3D& v : scratch)
      if (!callers(v.DataPtr())) nobodys = &v;
    if (!nobodys) {
      restored = false;
      break;
    }
    copy_volume(nobodys->DataPtr(), flow[c]->DataPtr());
    flow[c]->Swap(*nobodys);   // flow[c]: the scratch buffer with the result; *nobodys: the other component's buffer
  }
  for (int c = 0; c < 3 && restored; ++c) {
    if (flow[c]->DataPtr() == callers_storage[c]) continue;
    Data3D* holder = nullptr;
    for (Data3D& v : scratch)
      if (v.DataPtr() == callers_storage[c]) holder = &v;
    if (!holder) {
      restored = false;
      break;
    }
    flow[c]->Swap(*holder);   // flow[c]: the caller's buffer (stale), *holder: the result
    copy_volume(flow[c]->DataPtr(), holder->DataPtr());
  }
  if (!restored) std::printf("'%s': Error. A caller volume was lost among the host scratch volumes.\n", GetName());

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

    // the flow so far at this size (zero before the first level; values stay in original-voxel units)
    for (int c = 0; c < 3; ++c) {
      if (prev.width == 0) {
        ok = !CheckDeviceError(f3d_memset2d(buf[FU + c], container.pitch, 0, container.width * sizeof(float), rows)) && ok;
      } else {
        cuop_resample_e_.Execute(calls::Resample(op, &buf[FU + c], &buf[DU + c], &buf[TMP], &prev, &current));
        std::swap(buf[FU + c], buf[DU + c]);
      }
    }
    const calls::Flow flow = {&buf[FU], &buf[FV], &buf[FW]};
    const calls::Spacing spacing = {&hx, &hy, &hz};

    // frame 1 registered with it
    cuop_register_e_.Execute(calls::Registration(op, &buf[F0R], &buf[F1R], flow, &buf[TMP], &current, spacing));
    std::swap(buf[F1R], buf[TMP]);

    // the increments
    cuop_solve_e_.silent = true;
    cuop_solve_e_.Execute(calls::Solve(op, &buf[F0R], &buf[F1R], flow, {&buf[DU], &buf[DV], &buf[DW]}, {&buf[TDU], &buf[TDV], &buf[TDW]},
                                       &buf[PHI], &buf[KSI],
                                       {&outer_iterations_count, &inner_iterations_count, &equation_alpha, &equation_smoothness, &equation_data},
                                       &current, spacing));
    ++solve_passes_;

    // flow += increments; median of every component through the scratch container
    for (int c = 0; c < 3; ++c) cuop_add_e_.Execute(calls::Add(op, &buf[FU + c], &buf[DU + c], &current));
    if (median_radius != 1)
      for (int c = 0; c < 3; ++c) {
        cuop_median_e_.Execute(calls::Median(op, &buf[FU + c], &buf[TMP], &current, &median_radius));
        std::swap(buf[FU + c], buf[TMP]);
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
