/*
 * f3d_host.h -- C ABI of libf3d_host.so: the C++ host side (driver + operator classes that mirror the
 * reference's src/optical_flow and src/cuda_operations/entire_data) exposed to other languages.
 * The Python package binds exactly these entry points with ctypes; nothing here carries torch or numpy types.
 *
 *   f3d_flow_*  : OpticalFlowE  (src/optical_flow/optical_flow_e.h:38-66; ComputeFlow optical_flow_e.cpp:132-601)
 *   f3d_op_*    : the six CudaOperation* classes through their string-keyed parameter bag
 *                 (src/cuda_operations/cuda_operation_base.h:40-46, keys in SURVEY.md 8b)
 *   host helpers: level schedule (optical_flow_base.cpp:31-56), Gaussian taps
 *                 (cuda_operation_convolution.cpp:85-108), RAW volume I/O (data3d.cpp:95-237),
 *                 the synthetic translated-Gaussian benchmark pair.
 * All functions return 0 on success unless stated otherwise.
 */
#ifndef F3D_HOST_H_
#define F3D_HOST_H_

#include "f3d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the 9-key driver bag of src/main.cpp:156-165 as a plain struct */
typedef struct f3d_flow_params {
  size_t warp_levels_count;
  float  warp_scale_factor;
  size_t outer_iterations_count;
  size_t inner_iterations_count;
  float  equation_alpha;
  float  equation_smoothness;
  float  equation_data;
  size_t median_radius;
  float  gaussian_sigma;
} f3d_flow_params;

typedef struct f3d_flow_s* f3d_flow;
typedef struct f3d_op_s* f3d_op;

/* defaults of src/main.cpp:77-85 */
void f3d_flow_default_params(f3d_flow_params* p);

/* Orderly end of device use (the reference's cuCtxDestroy at src/main.cpp:236 with what this library adds around it):
 * releases the out-of-core path's device arena, copy queues and events, destroys the RCCL communicator if one exists,
 * then f3d_shutdown().  Idempotent; every step is a no-op when there is nothing to release.  Drivers and operators that
 * are still alive must be destroyed BEFORE this call -- their containers are theirs to free.  Bindings call it from their
 * own exit hook (the Python package registers it with atexit at import) so that nothing depends on the order in which
 * the process tears libraries down. */
int f3d_host_shutdown(void);

int f3d_flow_create(f3d_flow* flow);
/* OpticalFlowE::Initialize: allocates the 15 containers and initialises the six operators */
int f3d_flow_initialize(f3d_flow flow, size_t width, size_t height, size_t depth);
/* OpticalFlowE::ComputeFlow on dense host volumes (x fastest); u, v, w receive width*height*depth floats */
int f3d_flow_compute(f3d_flow flow, const float* frame_0, const float* frame_1, const f3d_flow_params* params,
                     int silent, float* u, float* v, float* w);
/* device-resident variant: upload once, solve any number of times, download on demand */
int f3d_flow_upload(f3d_flow flow, const float* frame_0, const float* frame_1);
int f3d_flow_compute_resident(f3d_flow flow, const f3d_flow_params* params, int silent, float* device_seconds);
int f3d_flow_download(f3d_flow flow, float* u, float* v, float* w);
int f3d_flow_container(f3d_flow flow, f3d_size4* container);
/* Diagnostics (the reference's counterpart is the disabled block optical_flow_e.cpp:536-571 that registers
 * frame_1 with the final flow and dumps the volume).  With level statistics enabled every later compute records, per pyramid
 * level, the residual of frame_1 warped by the flow handed down from the coarser level against frame_0 (before the solve) and the
 * min / max / average flow magnitude after the level's median. */
typedef struct f3d_level_stat {
  int level;
  size_t width, height, depth;
  double residual_rms, residual_mean_abs;
  float residual_max_abs;
  float flow_min, flow_max, flow_avg;
} f3d_level_stat;
int f3d_flow_set_level_stats(f3d_flow flow, int enable);
int f3d_flow_level_stat_count(f3d_flow flow, size_t* count);
int f3d_flow_level_stat(f3d_flow flow, size_t index, f3d_level_stat* out);  /* index 0 = coarsest level of the last compute */
/* residual of the ORIGINAL frame_1 registered with the flow on the device (h = 1) against the original frame_0, and of the pair
 * as it stands; values: rms, mean |.|, max |.| each.  Needs f3d_flow_upload + f3d_flow_compute_resident. */
int f3d_flow_final_residual(f3d_flow flow, double registered[3], double unregistered[3]);
int f3d_flow_destroy(f3d_flow flow);

/* name: "add" | "convolution" | "median" | "registration" | "resample" | "solve" */
int f3d_op_create(f3d_op* op, const char* name);
const char* f3d_op_name(f3d_op op);
int f3d_op_initialize(f3d_op op, const f3d_size4* container_size);
/* Execute(OperationParameters&): keys[i] -> value_ptrs[i] (non-owning, like the reference's bag) */
int f3d_op_execute(f3d_op op, const char* const* keys, void* const* value_ptrs, size_t count);
/* ExecuteBatch of the add, median and resample operators: `bags` bags laid end to end, bag b holding counts[b] entries -- up to
 * three volumes of one box in one launch per kernel where the bags allow it, otherwise one Execute after the other (same results).
 * Non-zero for operators without a batch form. */
int f3d_op_execute_batch(f3d_op op, const char* const* keys, void* const* value_ptrs, const size_t* counts, size_t bags);
int f3d_op_set_slab(f3d_op op, const f3d_slab* slab);
int f3d_op_destroy(f3d_op op);

/* host-only helpers (no device needed) */
/* ---- piecemeal (out-of-core) path: host volumes streamed through the device in z-chunks --------------------
 * Operators "add_p", "resample_p", "registration_p", "solve_p", "stat_p" (f3d_op_create) mirror
 * src/cuda_operations/partial_data/cuda_operation_*_p.cpp: they take no container at initialize (pass NULL) and their
 * Data3D* keys (operand_0, input, output, frame_0, flow_u, temp, ...) take f3d_volume_object() of a wrapped volume. */
typedef struct f3d_volume_s* f3d_volume;
/* non-owning Data3D view of caller memory (src/data_types/data3d.h:22-62) */
int f3d_volume_wrap(f3d_volume* vol, float* data, size_t width, size_t height, size_t depth);
void* f3d_volume_object(f3d_volume vol);
/* the storage the view addresses NOW: registration_p and solve_p swap storage between their volumes like the reference
 * (Data3D::Swap, cuda_operation_register_p.cpp:138, cuda_operation_solve_p.cpp:167-169) */
float* f3d_volume_data(f3d_volume vol);
int f3d_volume_destroy(f3d_volume vol);
/* what the last solve_p Execute did */
int f3d_op_solve_p_last(f3d_op op, int* chunk, int* outer_per_pass, int* halo, size_t* passes, int* overlapped);
/* 1 when the last execute of the "solve_p" operator ran the last sweep of an outer iteration together with the weights of the next
 * one (a second weight pair per chunk set: 15 fields instead of 13, taken when they fit and a residency holds two outer iterations
 * or more) */
int f3d_op_solve_p_fused_weights(f3d_op op, int* fused);
/* chunk plan of solve_p for a level (pure host arithmetic) and the device budget it would use now.  overlap_mode 0 = copies
 * and kernels in order, 1 = two chunk sets with the copies beside the kernels, -1 = whichever the cost model prefers */
int f3d_plan_solve_piecemeal(size_t budget_bytes, size_t width, size_t height, int depth, int inner_iterations, int outer_iterations,
                             int forced_outer_per_pass, int overlap_mode, int* chunk, int* outer_per_pass, int* halo, int* max_planes,
                             int* overlapped);
size_t f3d_piecemeal_budget_bytes(void);

/* OpticalFlowP (src/optical_flow/optical_flow_p.h:35-57; ComputeFlow optical_flow_p.cpp:57-318): no pre-blur, no median */
typedef struct f3d_pflow_s* f3d_pflow;
int f3d_pflow_create(f3d_pflow* flow);
int f3d_pflow_initialize(f3d_pflow flow, size_t width, size_t height, size_t depth);
int f3d_pflow_compute(f3d_pflow flow, const float* frame_0, const float* frame_1, size_t width, size_t height, size_t depth,
                      const f3d_flow_params* params, int silent, float* u, float* v, float* w, float* device_seconds);
/* of the last compute: solver residencies, levels cut into chunks, coarse levels that ran wholly on the device */
int f3d_pflow_stats(f3d_pflow flow, size_t* solve_passes, size_t* streamed_levels, size_t* resident_levels);
/* of the last compute: host levels whose frame 1 was registered inside the solver's first residency instead of by the separate
 * registration operator (cuda_operation_register_p.cpp:54-139); F3D_P_FUSED_WARP=0 keeps the operator everywhere */
int f3d_pflow_levels_registered_inside(f3d_pflow flow, size_t* levels);
/* of the last compute: host levels whose solver held the two frames and u, v, w on the device for the whole level beside the chunk sets
 * (three fields up per residency instead of eight; chosen by the cost model, F3D_P_CONSTANTS=0 / 1 pins it) */
int f3d_pflow_levels_with_constants_on_device(f3d_pflow flow, size_t* levels);
/* whether the resident levels of the last compute resampled their frames from device copies of the two originals */
int f3d_pflow_originals_on_device(f3d_pflow flow, int* yes);
/* also apply the Gaussian pre-blur and the per-level median, i.e. OpticalFlowE's whole pipeline on host volumes (off by
 * default like the reference's piecemeal driver; F3D_P_FULL=1 also turns it on).  The two extra operators are
 * "convolution_p" (keys input, output, data_size, gaussian_sigma) and "median_p" (input, output, data_size, radius). */
int f3d_pflow_set_full_pipeline(f3d_pflow flow, int enabled);
/* coarse levels whose working set fits the budget stay on the device (default on; F3D_P_RESIDENT=0 also turns it off) */
int f3d_pflow_set_resident(f3d_pflow flow, int enabled);
/* wall seconds of the last compute in {frame resample, flow resample, registration, solve, add} of the levels that went
 * through the host, and [5] in the resident coarse levels */
int f3d_pflow_operator_seconds(f3d_pflow flow, double* seconds6);
int f3d_pflow_destroy(f3d_pflow flow);

size_t f3d_max_warp_level(size_t width, size_t height, size_t depth, float scale_factor);
int f3d_level_geometry(size_t width, size_t height, size_t depth, float scale_factor, int level,
                       f3d_size4* size, float* hx, float* hy, float* hz);
int f3d_gaussian_taps(float sigma, float* taps, size_t capacity, size_t* radius);
int f3d_raw_read_u8(const char* path, size_t width, size_t height, size_t depth, float* out);
int f3d_raw_read_f32(const char* path, size_t width, size_t height, size_t depth, float* out);
int f3d_raw_write_u8(const char* path, const float* in, size_t width, size_t height, size_t depth);
int f3d_raw_write_f32(const char* path, const float* in, size_t width, size_t height, size_t depth);
int f3d_vtk_write_flow(const char* path, const float* u, const float* v, const float* w, size_t width, size_t height,
                       size_t depth);
/* translated-Gaussian pair: 64 blobs, splitmix64 seed 20241003, frame_1(p) = frame_0(p - t), t = (2, -1, 0.5) */
int f3d_synth_pair(size_t width, size_t height, size_t depth, float* frame_0, float* frame_1);
/* slab form: planes [z_lo, z_hi) only, unscaled, plus the maximum of frame_0 over them (scale = 255 / global max) */
int f3d_synth_planes(size_t width, size_t height, size_t depth, size_t z_lo, size_t z_hi, float* frame_0, float* frame_1,
                     float* frame_0_max);

/* ---- multi-GPU z-slab driver (OpticalFlowSlab; no reference counterpart) ------------------------ */

typedef struct f3d_slabflow_s* f3d_slabflow;

/* n_ranks slabs in total; this process computes local_ranks[0..n_local): one rank (RCCL between processes, after
 * f3d_comm_init) or all n_ranks (one-GPU rehearsal with plane copies instead of RCCL). */
int f3d_slabflow_create(f3d_slabflow* flow, int n_ranks, const int* local_ranks, int n_local, int halo_capacity);
int f3d_slabflow_initialize(f3d_slabflow flow, size_t width, size_t height, size_t depth);
/* frames and flows are FULL volumes; only the planes of the local ranks are read / written */
int f3d_slabflow_compute(f3d_slabflow flow, const float* frame_0, const float* frame_1, const f3d_flow_params* params,
                         float* u, float* v, float* w);
int f3d_slabflow_upload(f3d_slabflow flow, const float* frame_0, const float* frame_1);
int f3d_slabflow_compute_resident(f3d_slabflow flow, const f3d_flow_params* params, float* device_seconds);
int f3d_slabflow_download(f3d_slabflow flow, float* u, float* v, float* w);
/* outer iterations of the last solve whose halo exchange ran beside the interior of the slab (diagnostics) */
int f3d_slabflow_overlapped_iterations(f3d_slabflow flow, size_t* count);
/* groups of several outer iterations the last compute ran between two exchanges (thin slabs of small levels take
 * n (K + 1) halo planes at once; F3D_SLAB_OUTER_PER_EXCHANGE=n forces n, 1 = one exchange per outer iteration) */
int f3d_slabflow_batched_exchanges(f3d_slabflow flow, size_t* count);
/* pyramid levels of the last compute whose warp reached further along z than the halo room of the local containers: frame 1 was
 * gathered into a container of its own from as many ranks as the reach spans (SURVEY.md 8e fallback) */
int f3d_slabflow_gathered_warps(f3d_slabflow flow, size_t* count);
/* exchanges of the last compute made after a solver stage (F3D_SLAB_EXCHANGE=stage: 2 / 1 / 3 planes after the fused pairs and the
 * last sweep instead of 6 planes once per outer iteration; same bits, fewer redundant planes, three times the messages) */
int f3d_slabflow_stage_exchanges(f3d_slabflow flow, size_t* count);
/* the exchange order of the solves that follow: 0 = once per outer iteration (default), 1 = after every solver stage.  Both give the
 * single-GPU bits; which is faster depends on the machine's exchange latency, so bench.py --gpus N times both in one run. */
int f3d_slabflow_set_exchange_per_stage(f3d_slabflow flow, int per_stage);
int f3d_slabflow_destroy(f3d_slabflow flow);

/* the decomposition plan (pure host arithmetic, usable without a device) */
int f3d_plan_owned(int depth, int rank, int n_ranks, int* lo, int* hi);
/* fills up to `capacity` transfers; returns their number, or -1 if capacity is too small */
int f3d_plan_exchange(int depth, int rank, int n_ranks, int need_lo, int need_hi, int* peer, int* send_lo, int* send_hi,
                      int* recv_lo, int* recv_hi, int capacity);
int f3d_plan_resample_source(int in_depth, int out_depth, int out_lo, int out_hi, int* lo, int* hi);

#ifdef __cplusplus
}
#endif
#endif /* F3D_HOST_H_ */
