/*
 * f3d.h -- C ABI of libf3d_hip.so, the MI355X (gfx950) device library behind the
 * axruff/cuda-flow3d operator surface (src/cuda_operations/entire_data/ and src/optical_flow/optical_flow_e.{h,cpp}).
 *
 * The reference's operator hosts talk to the GPU through the CUDA *driver* API: cuModuleLoad of
 * kernels/<name>.ptx, cuModuleGetFunction, cuModuleGetGlobal("container_size"), cuLaunchKernel(args[]),
 * cuMemAllocPitch, cuMemcpy3D, cuMemsetD2D8, cuEvent*.  Every one of those uses on the hot path maps to
 * one entry point below; the comment on each cites the reference call site it replaces.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, floats.  Device pointers are 64-bit integers (f3d_devptr),
 *     the same width as the CUdeviceptr values that travel through the reference's OperationParameters bag.
 *   - every function returns 0 on success, non-zero on failure; f3d_last_error() gives the message
 *     (thread local).  Nothing here throws and nothing silently falls back to a CPU path.
 *   - all device work is enqueued on one library-owned HIP stream, in order (the reference uses the NULL
 *     stream).  Launchers do not synchronise; call f3d_stream_sync().
 *   - volumes live in pitched "containers" addressed ((z - z_base) * container.height + y) * (pitch/4) + x
 *     (reference IND macro, src/kernels/solve_3d.cu:26); coarse pyramid levels occupy the corner sub-box.
 *     f3d_set_container() replaces the per-module __constant__ container_size upload.
 *   - `slab` (nullable) restricts a launcher to global planes [z_lo, z_hi) of a container whose plane 0
 *     holds global plane z_base; `depth` is always the GLOBAL depth of the level, so mirror / zero-padding
 *     rules are those of the whole volume.  NULL means the whole volume (z_base 0, planes [0, depth)).
 */
#ifndef F3D_H_
#define F3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t f3d_devptr;

/* src/data_types/data_structs.h:20-25 (DataSize4; pitch in bytes) */
typedef struct f3d_size4 {
  size_t width;
  size_t height;
  size_t depth;
  size_t pitch;
} f3d_size4;

typedef struct f3d_slab {
  int z_base; /* global z held by container plane 0 */
  int z_lo;   /* first global plane to produce       */
  int z_hi;   /* one past the last plane to produce  */
} f3d_slab;

typedef struct f3d_event_s* f3d_event;
typedef struct f3d_queue_s* f3d_queue; /* a side stream for copies; NULL always means the library stream */

/* ---- runtime: context, memory, copies, events ------------------------------------------------ */

/* cuInit + cuDeviceGet + cuCtxCreate: src/utils/cuda_utils.cpp:21-57.  device < 0 picks LOCAL_RANK or 0. */
int f3d_init(int device);
/* cuCtxDestroy: src/main.cpp:236.  Idempotent: drains the library stream, destroys the timing events and the stream.
 * Memory is the owners' to free first (f3d_free, the operators' and drivers' Destroy). */
int f3d_shutdown(void);
/* 1 between a successful f3d_init() and f3d_shutdown(), else 0 (exit-time code asks before it touches the device) */
int f3d_is_initialized(void);
/* Lanes (no reference counterpart: the reference has one context and the NULL stream, src/utils/cuda_utils.cpp:21-57).  Every call
 * below that says "library stream", and f3d_set_container / f3d_set_conv_taps, act on the LANE of the calling thread: by default the
 * one lane the library creates in f3d_init.  A driver that is to run beside another one in the same process -- two frame pairs of a
 * sequence of small volumes solved at once (bin/flow3d --concurrent) -- creates a lane of its own (a stream, a container geometry, blur
 * taps) and makes it current in the thread that drives it; launches of different lanes are not ordered with each other.  Allocation,
 * events, queues and host registration are lane-agnostic; the profiling bracket (f3d_prof_*), the communicator and the out-of-core arena
 * belong to the default lane. */
typedef struct f3d_lane_s* f3d_lane;
int f3d_lane_create(f3d_lane* lane);
int f3d_lane_make_current(f3d_lane lane);   /* for the calling thread; NULL = back to the default lane */
int f3d_lane_is_private(void);              /* 1 when the calling thread is on a lane of its own */
int f3d_lane_get_current(f3d_lane* lane);   /* the calling thread's lane (NULL = the default one): code that borrows the thread for
                                               another lane puts this one back afterwards */
int f3d_lane_destroy(f3d_lane lane);        /* waits for the lane's stream first */

/* Diagnostics (no reference counterpart): from now on a fatal signal (SIGSEGV, SIGBUS, SIGILL, SIGFPE, SIGABRT) first writes
 * the signal number, the fault address and /proc/self/maps to `path`, then hands the signal to the handler that was installed
 * before (Python's faulthandler, a profiler's, the default action), so that the frames of a native stack trace can be resolved
 * to libraries.  Needs no device; calling it again only changes the file name. */
int f3d_crash_maps_enable(const char* path);
/* cuDeviceGetCount / cuDeviceGetName: src/utils/cuda_utils.cpp:27,44 */
int f3d_device_count(int* count);
int f3d_device_name(char* name, size_t capacity);
/* cuMemGetInfo: src/optical_flow/optical_flow_e.cpp:83 */
int f3d_mem_info(size_t* free_bytes, size_t* total_bytes);
/* cuDeviceGetAttribute(MAX_SHARED_MEMORY_PER_BLOCK): src/cuda_operations/entire_data/cuda_operation_solve.cpp:147 */
int f3d_lds_per_workgroup(int* bytes);
const char* f3d_last_error(void);

/* cuMemAllocPitch: src/optical_flow/optical_flow_e.cpp:104-108 (pitch is a multiple of 256 B) */
int f3d_alloc_pitched(f3d_devptr* ptr, size_t* pitch, size_t width_bytes, size_t rows);
/* cuMemFree: src/optical_flow/optical_flow_e.cpp:612 */
int f3d_free(f3d_devptr ptr);
/* cuMemsetD2D8: src/optical_flow/optical_flow_e.cpp:305-310, cuda_operation_solve.cpp:183-188 */
int f3d_memset2d(f3d_devptr ptr, size_t pitch, int value, size_t width_bytes, size_t rows);
/* cuMemcpy3D host->device / device->host, dense host volume <-> pitched container:
 * src/utils/cuda_utils.cpp:59-101.  depth planes are copied starting at container plane dev_plane0. */
int f3d_copy3d_h2d(f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0,
                   const float* src, size_t width, size_t height, size_t depth);
int f3d_copy3d_d2h(float* dst, size_t width, size_t height, size_t depth,
                   f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0);
/* The piecemeal operators' chunk copies (cuMemcpy3D with srcZ / dstZ and host strides,
 * src/cuda_operations/partial_data/cuda_operation_solve_p.cpp:217-243, 278-296): `depth` planes of a width x height
 * region between a host volume whose rows are *_row_floats apart and whose planes are *_rows rows apart (the pointer
 * already addresses the first plane) and container planes dev_plane0...  Asynchronous on the library stream; order
 * reuse of the host memory with f3d_stream_sync(). */
int f3d_copy_planes_h2d(f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                        size_t src_row_floats, size_t src_rows, size_t width, size_t height, size_t depth);
int f3d_copy_planes_d2h(float* dst, size_t dst_row_floats, size_t dst_rows, size_t width, size_t height, size_t depth,
                        f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0);
/* Copy queues: the piecemeal solver uploads the next chunk and downloads the previous one beside the kernels of the current
 * one.  Work on a queue is ordered; order BETWEEN queues (and the library stream, queue == NULL) is expressed with events:
 * f3d_event_record_on(ev, a); f3d_queue_wait_event(b, ev) makes everything issued to b afterwards wait for what a had been
 * given before the record.  (The reference has one NULL stream and synchronous cuMemcpy3D, cuda_operation_solve_p.cpp:226.) */
int f3d_queue_create(f3d_queue* queue);
int f3d_queue_destroy(f3d_queue queue);
int f3d_queue_sync(f3d_queue queue);
int f3d_event_record_on(f3d_event ev, f3d_queue queue);
int f3d_queue_wait_event(f3d_queue queue, f3d_event ev);
int f3d_copy_planes_h2d_on(f3d_queue queue, f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                           size_t src_row_floats, size_t src_rows, size_t width, size_t height, size_t depth);
int f3d_copy_planes_d2h_on(f3d_queue queue, float* dst, size_t dst_row_floats, size_t dst_rows, size_t width, size_t height,
                           size_t depth, f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0);
/* width x height x depth floats between two containers of different geometry (pitch in bytes, rows per plane), on the
 * library stream: moves a chunk staged in one geometry into a resident container of another */
int f3d_copy_rect_d2d(f3d_devptr dst, size_t dst_pitch, size_t dst_rows, size_t dst_plane0, f3d_devptr src, size_t src_pitch,
                      size_t src_rows, size_t src_plane0, size_t width, size_t height, size_t depth);
/* Page-lock caller memory so the copies above run at full link rate and asynchronously (the reference's
 * ALLOCATE_PINNED_MEMORY switch, src/data_types/data3d.cpp:30,57-61, applied to memory the caller already owns). */
int f3d_host_register(void* ptr, size_t bytes);
int f3d_host_unregister(void* ptr);
/* *yes = 1 when ptr lies in page-locked host memory (registered here or allocated pinned by the caller) */
int f3d_host_is_pinned(const void* ptr, int* yes);
/* cuMemcpyDtoD: src/cuda_operations/entire_data/cuda_operation_median.cpp:96-98 */
int f3d_copy_d2d(f3d_devptr dst, f3d_devptr src, size_t bytes);
/* cuModuleGetGlobal("container_size") + cuMemcpyHtoD in every op's Initialize, e.g. cuda_operation_solve.cpp:59-61 */
int f3d_set_container(const f3d_size4* container);
/* the geometry last set (operators that switch it for a chunk put it back afterwards) */
int f3d_get_container(f3d_size4* container);

/* cuEventCreate/Record/Synchronize/ElapsedTime/Destroy: optical_flow_e.cpp:163-169,579-587 */
int f3d_event_create(f3d_event* ev);
int f3d_event_record(f3d_event ev);
int f3d_event_sync(f3d_event ev);
int f3d_event_elapsed_ms(float* ms, f3d_event start, f3d_event stop);
int f3d_event_destroy(f3d_event ev);
/* cuStreamSynchronize(NULL): cuda_operation_solve.cpp:257 */
int f3d_stream_sync(void);

/* ---- kernel launchers (one per reference __global__; same scalar lists as the reference arg arrays) -- */

/* compute_phi_ksi_3d, 18 args: cuda_operation_solve.cpp:195-213; kernel src/kernels/solve_3d.cu:33-262 */
int f3d_phi_ksi(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw,
                size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                float equation_smoothness, float equation_data, f3d_devptr phi, f3d_devptr ksi,
                const f3d_slab* slab);

/* compute_phi_ksi_3d on TWO disjoint windows of the same container in one launch: the two zones of a z-slab whose weights had
 * to wait for the neighbours' increments (host/optical_flow_slab.cpp: CompleteWeights) -- two launches of a few planes each
 * otherwise.  Same results as two f3d_phi_ksi calls; an empty window falls back to one. */
int f3d_phi_ksi_zones(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                      f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, size_t width, size_t height, size_t depth, float hx,
                      float hy, float hz, float equation_smoothness, float equation_data, f3d_devptr phi, f3d_devptr ksi,
                      const f3d_slab* zone_a, const f3d_slab* zone_b);

/* solve_3d, 20 args: cuda_operation_solve.cpp:224-244; kernel src/kernels/solve_3d.cu:264-508.
 * One Jacobi sweep (in-voxel Gauss-Seidel du->dv->dw) into temp_d*; the caller ping-pongs the buffers. */
int f3d_solve_sweep(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                    f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                    size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                    f3d_devptr temp_du, f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab);

/* TWO consecutive solve_3d sweeps in one launch: temp_d* receive what two f3d_solve_sweep calls with a buffer swap in
 * between would leave in flow_d* (bit for bit); the intermediate field never goes to HBM, so the pair moves the bytes of
 * one sweep.  Replaces two iterations of the inner loop of cuda_operation_solve.cpp:222-255; the caller swaps ONCE.
 * A slab window [z_lo, z_hi) needs planes z_lo-2 .. z_hi+1 of every input inside the container.
 * RESTRICTION of every fused entry (f3d_solve_sweep2, f3d_solve_sweep_phi_ksi[_edges] and their _fd forms): the three face weights
 * equation_alpha / (h * h) must be finite and not negative -- the kernels select w or +0 where solve_3d.cu:437-445 multiplies by
 * (float)(flag), which is the same float only then; other parameters are refused (status 1, f3d_last_error says so) and the
 * caller uses f3d_solve_sweep / f3d_phi_ksi, which multiply like the reference (the host operators do that by themselves). */
int f3d_solve_sweep2(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                     f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                     size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                     f3d_devptr temp_du, f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab);

/* The last solve_3d sweep of an outer iteration AND compute_phi_ksi_3d of the next one in one launch: temp_d* receive the
 * sweep (what f3d_solve_sweep would write), phi_next / ksi_next what f3d_phi_ksi would then compute from temp_d* -- bit for
 * bit; the kernel reads frame_0 .. ksi once for both.  Replaces the launch pair cuda_operation_solve.cpp:246-252 (last j) +
 * :215-221 (next i).  phi_next / ksi_next must be buffers of their own: other tiles are still reading phi / ksi while this
 * launch writes (the operator ping-pongs two pairs).  A slab window [z_lo, z_hi) needs planes z_lo-2 .. z_hi+1 of every input
 * inside the container; the container pitch must be a multiple of 256 bytes. */
int f3d_solve_sweep_phi_ksi(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                            f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                            size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                            float equation_smoothness, float equation_data, f3d_devptr temp_du, f3d_devptr temp_dv,
                            f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab);

/* The same launch for a z-slab that exchanges its increments AFTER it (host/optical_flow_slab.cpp): the sweep is computed on
 * planes z_lo-1 .. z_hi anyway (the weights of z_lo and z_hi-1 need it); keep_below / keep_above also STORE it there, so a rank
 * launches on the window [own.lo + 1, own.hi - 1) and ends up with the sweep on all of its planes and the next weights on all
 * but the two whose neighbours it does not have yet (those follow from f3d_phi_ksi once the halo planes have arrived).  Planes
 * outside the volume do not exist: keep_below at z_lo = 0 and keep_above at z_hi = depth are ignored. */
int f3d_solve_sweep_phi_ksi_edges(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                                  f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                                  size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                                  float equation_smoothness, float equation_data, f3d_devptr temp_du, f3d_devptr temp_dv,
                                  f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab,
                                  int keep_below, int keep_above);

/* 1 when f3d_solve_sweep2 / f3d_solve_sweep_phi_ksi on a level of this size (whole volume, current container) take the tile that marches
 * along y with every z plane in it -- thin volumes, BASELINE config 3 -- which exists for the entry points on FRAMES only: a caller that
 * would otherwise read frame derivatives (the _fd entries march along z) is better off on the frames there.  Pure geometry, no launch. */
int f3d_fused_launches_march_along_y(size_t width, size_t height, size_t depth);

/* THREE consecutive solve_3d sweeps in one launch (k_tri, csrc/f3d_solve_tri.h): temp_d* receive what three f3d_solve_sweep calls with
 * the buffer swaps in between would leave, bit for bit.  Replaces three iterations of the inner loop of
 * cuda_operation_solve.cpp:222-255; the caller swaps ONCE.  For small and mid-size levels, where a launch is bound by its own
 * skeleton and a third stage costs a fifth of it (profiles/r04_three_stage_probe.txt); the fused-entry restriction on alpha / h^2
 * holds.  A slab window [z_lo, z_hi) needs planes z_lo-3 .. z_hi+2 of every input inside the container; the container pitch must be a
 * multiple of 256 bytes; no output may be one of the inputs. */
int f3d_solve_sweep3(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                     f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi, size_t width,
                     size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha, f3d_devptr temp_du,
                     f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab);
/* The last TWO sweeps of an outer iteration and compute_phi_ksi_3d of the next one in one launch: temp_d* receive the second sweep,
 * phi_next / ksi_next what f3d_phi_ksi would then compute from temp_d* (cuda_operation_solve.cpp:246-252 twice + :215-221 of the next
 * i).  With the default five sweeps an outer iteration is f3d_solve_sweep3 + f3d_solve_sweep2_phi_ksi: two launches for the
 * reference's six.  Same conditions as f3d_solve_sweep3; phi_next / ksi_next must be buffers of their own. */
int f3d_solve_sweep2_phi_ksi(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                             f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi, size_t width,
                             size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                             float equation_smoothness, float equation_data, f3d_devptr temp_du, f3d_devptr temp_dv,
                             f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab);

/* The frame derivatives of a level, once: fx, fy, fz = (((F0[+1] - F0[-1]) + F1[+1]) - F1[-1]) / (4 h) and ft = F1 - F0, exactly as
 * compute_phi_ksi_3d and solve_3d form them for every voxel in every launch (src/kernels/solve_3d.cu:205-215, :438-448).  They depend
 * on the two frames of the level only; the _fd launchers below read them instead of the frames.  A slab window [z_lo, z_hi) needs planes
 * z_lo-1 .. z_hi of both frames inside the container. */
int f3d_frame_derivatives(f3d_devptr frame_0, f3d_devptr frame_1, size_t width, size_t height, size_t depth, float hx, float hy,
                          float hz, f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, const f3d_slab* slab);
/* f3d_solve_sweep2 and f3d_solve_sweep_phi_ksi on precomputed frame derivatives (same results bit for bit; the four derivative
 * volumes must cover the planes the frames would have had to: z_lo-1 .. z_hi for a window [z_lo, z_hi)). */
int f3d_solve_sweep2_fd(f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                        f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi, size_t width,
                        size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha, f3d_devptr temp_du,
                        f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab);
int f3d_solve_sweep_phi_ksi_fd(f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, f3d_devptr flow_u, f3d_devptr flow_v,
                               f3d_devptr flow_w, f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi,
                               f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                               float equation_alpha, float equation_smoothness, float equation_data, f3d_devptr temp_du,
                               f3d_devptr temp_dv, f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab);
/* ... and f3d_solve_sweep_phi_ksi_edges on frame derivatives (the z-slab driver: derivative volumes per slab, computed once per level on
 * the slab widened by the halo depth minus one) */
int f3d_solve_sweep_phi_ksi_edges_fd(f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, f3d_devptr flow_u, f3d_devptr flow_v,
                                     f3d_devptr flow_w, f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi,
                                     f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                                     float equation_alpha, float equation_smoothness, float equation_data, f3d_devptr temp_du,
                                     f3d_devptr temp_dv, f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next,
                                     const f3d_slab* slab, int keep_below, int keep_above);

/* registration_3d, 12 args: cuda_operation_registration.cpp:110-122; kernel src/kernels/registration_3d.cu:28-82 */
int f3d_warp(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
             size_t width, size_t height, size_t depth, float hx, float hy, float hz, f3d_devptr output,
             const f3d_slab* slab);

/* resample_{x,y,z}_3d, 6 args: cuda_operation_resample.cpp:115-120,138-143,161-166; src/kernels/resample_3d.cu.
 * slab_in describes the input container (z pass only reads through it), slab the output planes. */
int f3d_resample_x(f3d_devptr input, f3d_devptr output, size_t out_width, size_t out_height, size_t out_depth,
                   size_t in_width, const f3d_slab* slab);
int f3d_resample_y(f3d_devptr input, f3d_devptr output, size_t out_width, size_t out_height, size_t out_depth,
                   size_t in_height, const f3d_slab* slab);
int f3d_resample_z(f3d_devptr input, f3d_devptr output, size_t out_width, size_t out_height, size_t out_depth,
                   size_t in_depth, const f3d_slab* slab_in, const f3d_slab* slab);

/* add_3d, 5 args: cuda_operation_add.cpp:86-91; src/kernels/add_3d.cu:26-41 */
int f3d_add(f3d_devptr operand_0, f3d_devptr operand_1, size_t width, size_t height, size_t depth,
            const f3d_slab* slab);

/* median_3d, 6 args: cuda_operation_median.cpp:131-137; src/kernels/median_3d.cu:49-299.
 * radius is the window DIAMETER, one of 3, 5, 7 (the host op applies the 1 / even rules). */
int f3d_median(f3d_devptr input, size_t width, size_t height, size_t depth, size_t radius, f3d_devptr output,
               const f3d_slab* slab);

/* Up to three volumes of ONE box in one launch (extensions, no cu* site of their own): the reference runs "+=", the median and the
 * three resampling passes once per flow component and frame (optical_flow_e.cpp:274-345, 420-473); the components do not depend
 * on each other there, and on levels below ~128^3 a launch is mostly fixed cost.  Element i of every array belongs to volume i;
 * the kernels, arguments and results are those of the single-volume entries above (which are the count = 1 case of these).  No
 * input of a batch may be an output of the same batch.  count = 1 .. 3. */
int f3d_resample_x_n(const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t out_width, size_t out_height,
                     size_t out_depth, size_t in_width, const f3d_slab* slab);
int f3d_resample_y_n(const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t out_width, size_t out_height,
                     size_t out_depth, size_t in_height, const f3d_slab* slab);
int f3d_resample_z_n(const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t out_width, size_t out_height,
                     size_t out_depth, size_t in_depth, const f3d_slab* slab_in, const f3d_slab* slab);
int f3d_add_n(const f3d_devptr* operand_0, const f3d_devptr* operand_1, size_t count, size_t width, size_t height, size_t depth,
              const f3d_slab* slab);
int f3d_median_n(const f3d_devptr* inputs, size_t count, size_t width, size_t height, size_t depth, size_t radius,
                 const f3d_devptr* outputs, const f3d_slab* slab);
/* The box width x height x [slab planes] of up to three volumes of the current container set to +0.f in one launch: the
 * increments du, dv, dw at the start of a level's solve.  The reference clears every row of every plane of the container there
 * (three cuMemsetD2D8, cuda_operation_solve.cpp:183-188); nothing ever reads a row or plane outside the level's box -- every
 * kernel mirrors by address inside it -- so only the box is written. */
int f3d_clear_box_n(const f3d_devptr* volumes, size_t count, size_t width, size_t height, size_t depth, const f3d_slab* slab);

/* c_Kernel upload: cuda_operation_convolution.cpp:160-161 (at most 51 taps, MAX_KERNEL_LENGTH) */
int f3d_set_conv_taps(const float* taps, size_t count);
/* convolution{Rows,Columns,Slices}Kernel, 7 args: cuda_operation_convolution.cpp:221-228,274-281,327-334;
 * src/kernels/convolution_3d.cu:75-172,186-271,284-372.  Zero padding; taps from f3d_set_conv_taps. */
int f3d_conv_rows(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                  const f3d_slab* slab);
int f3d_conv_cols(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                  const f3d_slab* slab);
int f3d_conv_slices(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                    const f3d_slab* slab);
/* convolutionRowsKernel followed by convolutionColumnsKernel in ONE launch (the two launches of
 * cuda_operation_convolution.cpp:172-177): dst receives what f3d_conv_rows into a scratch volume and f3d_conv_cols from it
 * would leave, bit for bit; the row-convolved volume only ever exists in LDS. */
int f3d_conv_rows_cols(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                       const f3d_slab* slab);

/* Self-test (no reference counterpart): the fused sweep + phi/ksi kernel forms the weights 1 / (2 sqrt(a)) of
 * src/kernels/solve_3d.cu:203-204,259-260 by a shorter instruction sequence that has been checked against the IEEE square root and
 * division for EVERY argument it is used on.  This entry point repeats that check on the device it runs on: all bit patterns in
 * [lo_bits, hi_bits] (as floats), `excluded` = arguments the kernel sends down the IEEE road instead (outside 2^-100 .. 2^100, or
 * a root with an all-ones significand), `checked` = the rest, `mismatches` = how many of those differ from the IEEE chain (0 on
 * gfx950), first_mismatch = one such bit pattern.  tests/test_gpu_kernels.py sweeps the whole binary32 range. */
int f3d_selftest_weights(unsigned lo_bits, unsigned hi_bits, unsigned long long* checked, unsigned long long* excluded,
                         unsigned long long* mismatches, unsigned* first_mismatch);

/* ---- profiler ranges (no reference counterpart; SURVEY.md section 5 "tracing") ---------------------------------
 * roctx ranges around operators and pyramid levels so that rocprofv3 --marker-trace attributes kernels to them.
 * librocprofiler-sdk-roctx is dlopen'ed on the first push and only when F3D_ROCTX=1 is set or a rocprofiler tool library
 * is preloaded; otherwise both calls return at once.  Ranges nest; pop closes the innermost. */
int f3d_range_push(const char* name);
int f3d_range_pop(void);

/* ---- per-kernel timing (HIP events on the library stream), used by bench.py's roofline leg ----------- */

enum { F3D_K_PHI_KSI = 0, F3D_K_SWEEP = 1, F3D_K_SWEEP2 = 2, F3D_K_SWEEP_PHI_KSI = 3, F3D_K_SWEEP3 = 4, F3D_K_SWEEP2_PHI_KSI = 5,
       F3D_K_COUNT = 6 };
/* enable = 1 brackets every launch of the solver kernels (phi/ksi, one sweep, two fused sweeps) with events on the library stream */
int f3d_prof_enable(int enable);
int f3d_prof_reset(void);
/* which kernels get events while profiling is enabled: bit k = kernel id k (default all).  Two event records per launch
 * cost ~7 us of dispatch, 2 % of a 512^3 solve when all 6400 solver launches carry them */
int f3d_prof_select(unsigned kernel_mask);
/* drains the pending events; min_voxels filters launches by level size (0 = all) */
int f3d_prof_read(int kernel, size_t min_voxels, double* total_ms, uint64_t* launches, double* total_voxels);

/* ---- multi-GPU: z-slab halo exchange on RCCL over xGMI (no reference counterpart) --------- */

/* 128-byte ncclUniqueId, created on rank 0 and handed to the other ranks by the launcher (bench.py sends it
 * through torch.distributed).  librccl is loaded on the first of these calls, never for single-GPU use. */
int f3d_comm_unique_id(void* id128);
int f3d_comm_init(const void* id128, int rank, int n_ranks);
int f3d_comm_destroy(void);
int f3d_comm_rank(int* rank, int* n_ranks);
/* What the transport itself says (any pointer may be null): backend 0 = none, 1 = RCCL, 2 = the shared-memory rehearsal
 * transport; with RCCL comm_ranks / comm_rank / comm_device are the answers of ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice for the live communicator (-1 where the library lacks the query); sent_bytes / exchanges count what
 * this rank has handed to the transport since f3d_comm_init (bench.py reports them per run). */
int f3d_comm_info(int* backend, int* comm_ranks, int* comm_rank, int* comm_device, unsigned long long* sent_bytes,
                  unsigned long long* exchanges);
/* Gather `count` container planes (sub-box width x height of each) of `field`, starting at container plane
 * plane0, into the dense staging buffer at staging[offset_floats ...]; unpack is the inverse. */
int f3d_pack_planes(f3d_devptr field, int plane0, int count, size_t width, size_t height, f3d_devptr staging,
                    size_t offset_floats);
int f3d_unpack_planes(f3d_devptr field, int plane0, int count, size_t width, size_t height, f3d_devptr staging,
                      size_t offset_floats);
/* the same for up to 32 (field, plane range) segments in ONE launch: segment i covers count[i] planes of fields[i]
 * from container plane plane0[i] and sits at staging[offset_floats[i] ...] */
int f3d_pack_segments(const f3d_devptr* fields, const int* plane0, const int* count, const size_t* offset_floats,
                      int n_segments, size_t width, size_t height, f3d_devptr staging);
int f3d_unpack_segments(const f3d_devptr* fields, const int* plane0, const int* count, const size_t* offset_floats,
                        int n_segments, size_t width, size_t height, f3d_devptr staging);
/* same-device plane copy between two containers (one-GPU rehearsal of the slab decomposition) */
int f3d_copy_planes(f3d_devptr dst, int dst_plane0, f3d_devptr src, int src_plane0, int count, size_t width,
                    size_t height);
/* ... and up to any number of (destination, source, plane range) triples in as few launches as possible (32 per launch): a whole
 * in-process exchange -- every rank, peer and field -- instead of a launch per triple */
int f3d_copy_plane_segments(const f3d_devptr* dst, const int* dst_plane0, const f3d_devptr* src, const int* src_plane0, const int* count,
                            int n_segments, size_t width, size_t height);
/* One grouped exchange on the library stream: for every i, send send_count[i] floats from send_buf + send_offset[i]
 * to peers[i] and receive recv_count[i] floats into recv_buf + recv_offset[i] from peers[i] (zero counts skipped). */
int f3d_comm_sendrecv(f3d_devptr send_buf, const size_t* send_offset, const size_t* send_count, f3d_devptr recv_buf,
                      const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers);
/* The same exchange split in two: _begin starts it behind everything issued on the library stream so far (on a side
 * stream, so kernels issued afterwards run beside it); _end makes the library stream wait for the received data.  The
 * send buffer must not be rewritten, nor the receive buffer read, in between.  One exchange may be open at a time. */
int f3d_comm_sendrecv_begin(f3d_devptr send_buf, const size_t* send_offset, const size_t* send_count, f3d_devptr recv_buf,
                            const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers);
int f3d_comm_sendrecv_end(void);
/* Measured cost of an exchange (nothing in the reference to match: src/utils/cuda_utils.cpp:38,52 use one device).  While
 * f3d_comm_timing(1) is in force HIP events bracket three kinds of interval on the stream the work runs on:
 *   class 0  a whole blocking exchange -- the caller marks it: f3d_comm_mark(0, 0) before the pack launch, f3d_comm_mark(1, 0) after
 *            the unpack launch (host/optical_flow_slab.cpp: Exchange)
 *   class 1  the same marks around an exchange whose transfer runs beside kernels (f3d_comm_sendrecv_begin / _end with the
 *            interior in between): the interval includes the kernels it hides behind
 *   class 2  the grouped ncclSend / ncclRecv alone (recorded by the library on the stream it was posted to)
 * f3d_comm_timing(1) also clears the sums; f3d_comm_timing_read drains the streams and returns total / count / min / max in
 * microseconds and the bytes this rank sent inside the intervals.  Off by default; with it off f3d_comm_mark does nothing. */
int f3d_comm_timing(int enable);
int f3d_comm_mark(int what, int cls);
int f3d_comm_timing_read(int cls, double* total_us, unsigned long long* count, double* min_us, double* max_us,
                         unsigned long long* bytes);
/* max over all ranks of *value (host in/out) */
int f3d_comm_allreduce_max_f32(float* value);
/* max |field| over the slab's planes, on the device (feeds the warp halo depth) */
int f3d_abs_max(f3d_devptr field, size_t width, size_t height, size_t depth, const f3d_slab* slab, float* result);

/* Flow statistics on the device: min, max and SUM of sqrt(u^2 + v^2 + w^2) over the slab's planes (host out; the
 * caller divides by the voxel count, or adds the sums of several slabs first).  Device counterpart of the host loop
 * in src/cuda_operations/partial_data/cuda_operation_stat_p.cpp:85-104; min and max are exact, the sum is accumulated in
 * double (the reference adds floats in scan order). */
int f3d_flow_stats(f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w, size_t width, size_t height, size_t depth,
                   const f3d_slab* slab, float* min_magnitude, float* max_magnitude, double* sum_magnitude);

/* Registration residual on the device: sum of (warped - frame_0)^2, sum of |warped - frame_0| (both accumulated in double)
 * and max |warped - frame_0| over the slab's planes (host out).  The reference's counterpart is the disabled debug block of
 * src/optical_flow/optical_flow_e.cpp:536-571, which registers frame_1 with the final flow (cuop_register_, h = 1) and dumps
 * the volume for inspection; here the comparison with frame_0 is a reduction and nothing is written to disk. */
int f3d_residual_stats(f3d_devptr frame_0, f3d_devptr frame_1_warped, size_t width, size_t height, size_t depth,
                       const f3d_slab* slab, double* sum_squares, double* sum_abs, float* max_abs);

#ifdef __cplusplus
}
#endif
#endif /* F3D_H_ */
