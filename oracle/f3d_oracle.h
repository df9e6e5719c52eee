/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement (plain C, IEEE float32, no FMA contraction) of the numerics of the
 * axruff/cuda-flow3d "entire data" hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (cuda-flow3d_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - level schedule (orc_max_warp_level / orc_level_geometry): PINNED against the reference's own
 *     src/optical_flow/optical_flow_base.cpp compiled verbatim into oracle/_ref/ (tests/test_host_logic.py).
 *   - RAW U8/F32 volume I/O: PINNED against the reference's src/data_types/data3d.cpp in oracle/_ref/.
 *   - warp (orc_warp) and flow statistics (orc_flow_stats): PINNED against the reference's own host implementations
 *     (partial_data/cuda_operation_register_p.cpp:96-139, cuda_operation_stat_p.cpp:85-104) compiled in place into
 *     oracle/_ref/libf3d_ref_ops.so (tests/test_oracle.py).
 *   - kernel numerics (resample A.1, phi/ksi A.3, sweep A.4, median A.5, Gaussian A.6, and the warp A.2 once more): PINNED ON THE
 *     GPU against the reference's own kernels.  The reference ships no tests, golden vectors or fixtures for them and no nvcc
 *     exists here, but its entire_data .cu files are plain CUDA C that hipcc accepts as HIP source unmodified: oracle/Makefile
 *     compiles the six of them (solve, median, resample, registration, convolution, add) where they lie into
 *     oracle/_ref/<file>.hsaco (gfx950, contraction off, IEEE division and square root; the recipe's comment names the two
 *     include-guard definitions it needs and why nothing is written in place of any header), tests/ref_kernels.py launches them
 *     with the reference operators' block sizes, shared-memory sizes and argument lists, and
 *     tests/test_gpu_reference_kernels.py holds reference == oracle and product == reference bit for bit, kernel by kernel and
 *     for whole pyramid solves driven on the reference's kernels, the golden results of BASELINE configs 1 - 4 among them (73 cases).
 *     What that does not cover: whether nvcc would have fused multiply-adds (this repository defines the reference's numbers
 *     with contraction off), and the host-side operator code around the kernels (CUDA driver API; restated,
 *     file:line cited).
 *     Each function below cites the reference file:line it restates.
 *
 * Layout convention (reference IND macro, src/kernels/solve_3d.cu:26): a "container" is a pitched
 * array addressed ((z - z_base) * Hc + y) * pitch_f + x, Hc = container height, pitch_f = row pitch
 * in floats.  Coarse pyramid levels live in the corner sub-box of the full-size container.
 * z_base/z_lo/z_hi describe a z-slab: the container's plane 0 holds global plane z_base and the
 * function writes global planes [z_lo, z_hi); a whole volume is z_base = 0, z_lo = 0, z_hi = D.
 */
#ifndef F3D_ORACLE_H_
#define F3D_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_geom {
  int Hc;       /* container height (rows per plane)           */
  int pitch_f;  /* container row pitch in floats               */
  int z_base;   /* global z of container plane 0               */
  int z_lo;     /* first global plane written                  */
  int z_hi;     /* one past the last global plane written      */
} orc_geom;

typedef struct orc_params {
  size_t warp_levels_count;
  float  warp_scale_factor;
  size_t outer_iterations_count;
  size_t inner_iterations_count;
  float  equation_alpha;
  float  equation_smoothness;
  float  equation_data;
  size_t median_radius;
  float  gaussian_sigma;
} orc_params;

/* A.0 level schedule */
size_t orc_max_warp_level(size_t width, size_t height, size_t depth, float scale_factor);
void   orc_level_geometry(size_t w0, size_t h0, size_t d0, float scale_factor, int level,
                          size_t* w, size_t* h, size_t* d, float* hx, float* hy, float* hz);

/* A.6 Gaussian taps; returns the radius, writes 2*radius+1 taps */
int  orc_gaussian_taps(float sigma, float* taps, int max_taps);
/* axis 0 = x (rows), 1 = y (columns), 2 = z (slices); zero padding; D = global depth */
void orc_conv_axis(float* dst, const float* src, int W, int H, int D, int radius,
                   const float* taps, int axis, const orc_geom* g);

/* A.1 box/area resample along one axis; (ow,oh,od) is the extent of this pass's output,
 * in_n the input length along `axis`; od / in_n are global depths when axis == 2. */
void orc_resample_axis(const float* in, float* out, int ow, int oh, int od, int in_n,
                       int axis, const orc_geom* g_in, const orc_geom* g_out);

/* A.2 trilinear backward warp */
/* PINNED: equals the reference's own host implementation (partial_data/cuda_operation_register_p.cpp:96-139, built in place into
 * oracle/_ref/libf3d_ref_ops.so) bit for bit -- tests/test_oracle.py::test_warp_equals_the_reference_cpu_warp */
void orc_warp(const float* f0, const float* f1, const float* u, const float* v, const float* w,
              int W, int H, int D, float hx, float hy, float hz, float* out, const orc_geom* g);

/* A.3 robust-penalty weights */
void orc_phi_ksi(const float* f0, const float* f1, const float* u, const float* v, const float* w,
                 const float* du, const float* dv, const float* dw, int W, int H, int D,
                 float hx, float hy, float hz, float eps_s, float eps_d,
                 float* phi, float* ksi, const orc_geom* g);

/* A.4 one Jacobi / in-voxel Gauss-Seidel sweep */
void orc_solve_sweep(const float* f0, const float* f1, const float* u, const float* v, const float* w,
                     const float* du, const float* dv, const float* dw, const float* phi, const float* ksi,
                     int W, int H, int D, float hx, float hy, float hz, float alpha,
                     float* tdu, float* tdv, float* tdw, const orc_geom* g);

void orc_add(float* a, const float* b, int W, int H, int D, const orc_geom* g);
/* src/cuda_operations/partial_data/cuda_operation_stat_p.cpp:85-104: min, max, and the float sum in scan order (avg = sum
 * / count as the reference computes it) plus the same sum in double */
void orc_flow_stats(const float* u, const float* v, const float* w, int W, int H, int D, const orc_geom* g,
                    float* min_mag, float* max_mag, float* avg_float, double* sum_double);

/* residual of a registration: sums of (warped - frame_0)^2 and |warped - frame_0| in double, maximum of the latter
 * (diagnostic of src/optical_flow/optical_flow_e.cpp:536-571, which only dumps the registered volume) */
void orc_residual_stats(const float* f0, const float* fw, int W, int H, int D, const orc_geom* g, double* sum_sq,
                        double* sum_abs, float* max_abs);

/* A.5 median, window diameter r in {3,5,7} */
void orc_median(const float* in, float* out, int W, int H, int D, int r, const orc_geom* g);

/* Whole pipeline (OpticalFlowE::ComputeFlow) on dense host volumes (x fastest).
 * pitch_f >= W lets a test exercise a padded container; 0 means dense. Returns levels used. */
int orc_compute_flow(const float* frame0, const float* frame1, size_t W, size_t H, size_t D,
                     const orc_params* p, int pitch_f, float* u, float* v, float* w);

int orc_num_threads(void);
void orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
