/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see f3d_oracle.h for the pinning status).
 *
 * Plain-C restatement of the reference's float32 numerics.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp
 * so that every + - * / sqrt below is one correctly rounded IEEE binary32 operation, in the
 * association order of the reference source.  Threads only split independent output planes
 * (every kernel is a pure gather), so the result does not depend on the thread count.
 */
#include "f3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(g, x, y, z) \
  (((size_t)((z) - (g)->z_base) * (size_t)(g)->Hc + (size_t)(y)) * (size_t)(g)->pitch_f + (size_t)(x))

/* mirror index of the reference halo loads: src/kernels/solve_3d.cu:73-75,89-90,104-105;
 * src/kernels/median_3d.cu:70-72,79-80,87-88 */
/* small levels run serially: forking a team costs more than the loop */
#define ORC_BIG(g, W, H) ((long)(W) * (H) * ((g)->z_hi - (g)->z_lo) > 20000)

static inline int mir(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - i - 2 : i); }

void orc_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* A.0  src/optical_flow/optical_flow_base.cpp:31-56 */
size_t orc_max_warp_level(size_t width, size_t height, size_t depth, float scale_factor)
{
  size_t rw = 1, rh = 1, rd = 1, level = 1;
  while (scale_factor < 1.f) {
    float scale = powf(scale_factor, (float)level);
    rw = (size_t)ceilf((float)width * scale);
    rh = (size_t)ceilf((float)height * scale);
    rd = (size_t)ceilf((float)depth * scale);
    if (rw < 4 || rh < 4 || rd < 4) break;
    ++level;
  }
  if (rw == 1 || rh == 1 || rd == 1) --level;
  return level;
}

/* src/optical_flow/optical_flow_e.cpp:262-268 */
void orc_level_geometry(size_t w0, size_t h0, size_t d0, float scale_factor, int level,
                        size_t* w, size_t* h, size_t* d, float* hx, float* hy, float* hz)
{
  float scale = powf(scale_factor, (float)level);
  *w = (size_t)ceilf((float)w0 * scale);
  *h = (size_t)ceilf((float)h0 * scale);
  *d = (size_t)ceilf((float)d0 * scale);
  *hx = (float)w0 / (float)*w;
  *hy = (float)h0 / (float)*h;
  *hz = (float)d0 / (float)*d;
}

/* ------------------------------------------------------------------------------------------ */
/* A.6 taps: src/cuda_operations/entire_data/cuda_operation_convolution.cpp:85-108
 * (precision = 3, pixel_size = 1.0f at the only call site, :165) */
int orc_gaussian_taps(float sigma, float* taps, int max_taps)
{
  const size_t precision = 3;
  const float pixel_size = 1.0f;
  size_t radius = (size_t)((float)precision * sigma / pixel_size);
  int r = (int)radius;
  int n = 2 * r + 1;
  if (n > max_taps) return -1;
  for (int i = -r; i <= r; i++) {
    float val = 1.0 / ((double)sigma * sqrt(2.0 * 3.1415926)) *
                exp((double)(-((float)(i * i) * pixel_size * pixel_size)) / (2.0 * (double)sigma * (double)sigma));
    taps[i + r] = val;
  }
  float sum = 0.0f;
  for (int i = 0; i < n; i++) sum = sum + taps[i];
  for (int i = 0; i < n; i++) taps[i] = taps[i] / sum;
  return r;
}

/* A.6 passes: src/kernels/convolution_3d.cu:75-172 (rows), :186-271 (columns), :284-372 (slices).
 * Clean spec: zero padding outside the volume, sum = sum + k[R - j] * s[i + j],
 * j = -R..R ascending, starting from 0. */
void orc_conv_axis(float* dst, const float* src, int W, int H, int D, int radius,
                   const float* taps, int axis, const orc_geom* g)
{
  const int n[3] = { W, H, D };
#pragma omp parallel for schedule(static) if (ORC_BIG(g, W, H))
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        int c[3] = { x, y, z };
        float sum = 0;
        for (int j = -radius; j <= radius; j++) {
          int q[3] = { x, y, z };
          q[axis] = c[axis] + j;
          float s = (q[axis] >= 0 && q[axis] < n[axis]) ? src[IDX(g, q[0], q[1], q[2])] : 0.f;
          sum = sum + taps[radius - j] * s;
        }
        dst[IDX(g, x, y, z)] = sum;
      }
}

/* ------------------------------------------------------------------------------------------ */
/* A.1  src/kernels/resample_3d.cu:28-71 (x), :73-116 (y), :118-161 (z) */
void orc_resample_axis(const float* in, float* out, int ow, int oh, int od, int in_n,
                       int axis, const orc_geom* g_in, const orc_geom* g_out)
{
  const int out_n = axis == 0 ? ow : (axis == 1 ? oh : od);
  const float delta = (float)in_n / (float)out_n;
  const float normalization = (float)out_n / (float)in_n;
  (void)od;
#pragma omp parallel for schedule(static) if (ORC_BIG(g_out, ow, oh))
  for (int z = g_out->z_lo; z < g_out->z_hi; z++)
    for (int y = 0; y < oh; y++)
      for (int x = 0; x < ow; x++) {
        int c[3] = { x, y, z };
        int i = c[axis];
        float left_f = (float)i * delta;
        float right_f = (float)(i + 1) * delta;
        int left_i = (int)floorf(left_f);
        int right_i = (int)fminf((float)in_n, (float)(size_t)ceilf(right_f));
        float value = 0.f;
        int cnt = right_i - left_i;
        for (int j = 0; j < cnt; j++) {
          float frac = 1.f;
          if (j == 0) frac = (float)(left_i + 1) - left_f;
          if (j == cnt - 1) frac = right_f - (float)(left_i + j);
          if (cnt == 1) frac = delta;
          int q[3] = { x, y, z };
          q[axis] = left_i + j;
          value = value + in[IDX(g_in, q[0], q[1], q[2])] * frac;
        }
        out[IDX(g_out, x, y, z)] = value * normalization;
      }
}

/* ------------------------------------------------------------------------------------------ */
/* A.2  src/kernels/registration_3d.cu:28-82 */
void orc_warp(const float* f0, const float* f1, const float* u, const float* v, const float* w,
              int W, int H, int D, float hx, float hy, float hz, float* out, const orc_geom* g)
{
#pragma omp parallel for schedule(static) if (ORC_BIG(g, W, H))
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        size_t c = IDX(g, x, y, z);
        float x_f = (float)x + (u[c] * (1.f / hx));
        float y_f = (float)y + (v[c] * (1.f / hy));
        float z_f = (float)z + (w[c] * (1.f / hz));
        if ((x_f < 0.) || (x_f > (float)(W - 1)) || (y_f < 0.) || (y_f > (float)(H - 1)) ||
            (z_f < 0.) || (z_f > (float)(D - 1)) || isnan(x_f) || isnan(y_f) || isnan(z_f)) {
          out[c] = f0[c];
        } else {
          int xi = (int)floorf(x_f);
          int yi = (int)floorf(y_f);
          int zi = (int)floorf(z_f);
          float dx = x_f - (float)xi;
          float dy = y_f - (float)yi;
          float dz = z_f - (float)zi;
          int x1 = (int)fminf((float)(W - 1), (float)(size_t)(xi + 1));
          int y1 = (int)fminf((float)(H - 1), (float)(size_t)(yi + 1));
          int z1 = (int)fminf((float)(D - 1), (float)(size_t)(zi + 1));
          float v0 = (1.f - dx) * (1.f - dy) * f1[IDX(g, xi, yi, zi)] +
                     (dx) * (1.f - dy) * f1[IDX(g, x1, yi, zi)] +
                     (1.f - dx) * (dy) * f1[IDX(g, xi, y1, zi)] +
                     (dx) * (dy) * f1[IDX(g, x1, y1, zi)];
          float v1 = (1.f - dx) * (1.f - dy) * f1[IDX(g, xi, yi, z1)] +
                     (dx) * (1.f - dy) * f1[IDX(g, x1, yi, z1)] +
                     (1.f - dx) * (dy) * f1[IDX(g, xi, y1, z1)] +
                     (dx) * (dy) * f1[IDX(g, x1, y1, z1)];
          out[c] = (1.f - dz) * v0 + dz * v1;
        }
      }
}

/* ------------------------------------------------------------------------------------------ */
/* A.3  src/kernels/solve_3d.cu:33-262 (arithmetic :177-260) */
void orc_phi_ksi(const float* f0, const float* f1, const float* u, const float* v, const float* w,
                 const float* du, const float* dv, const float* dw, int W, int H, int D,
                 float hx, float hy, float hz, float eps_s, float eps_d,
                 float* phi, float* ksi, const orc_geom* g)
{
#pragma omp parallel for schedule(static) if (ORC_BIG(g, W, H))
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        size_t c = IDX(g, x, y, z);
        size_t xm = IDX(g, mir(x - 1, W), y, z), xp = IDX(g, mir(x + 1, W), y, z);
        size_t ym = IDX(g, x, mir(y - 1, H), z), yp = IDX(g, x, mir(y + 1, H), z);
        size_t zm = IDX(g, x, y, mir(z - 1, D)), zp = IDX(g, x, y, mir(z + 1, D));

        float dux = (u[xp] - u[xm] + du[xp] - du[xm]) / (2.f * hx);
        float duy = (u[yp] - u[ym] + du[yp] - du[ym]) / (2.f * hy);
        float duz = (u[zp] - u[zm] + du[zp] - du[zm]) / (2.f * hz);
        float dvx = (v[xp] - v[xm] + dv[xp] - dv[xm]) / (2.f * hx);
        float dvy = (v[yp] - v[ym] + dv[yp] - dv[ym]) / (2.f * hy);
        float dvz = (v[zp] - v[zm] + dv[zp] - dv[zm]) / (2.f * hz);
        float dwx = (w[xp] - w[xm] + dw[xp] - dw[xm]) / (2.f * hx);
        float dwy = (w[yp] - w[ym] + dw[yp] - dw[ym]) / (2.f * hy);
        float dwz = (w[zp] - w[zm] + dw[zp] - dw[zm]) / (2.f * hz);

        phi[c] = 1.f / (2.f * sqrtf(dux * dux + duy * duy + duz * duz + dvx * dvx + dvy * dvy + dvz * dvz +
                                    dwx * dwx + dwy * dwy + dwz * dwz + eps_s * eps_s));

        float fx = (f0[xp] - f0[xm] + f1[xp] - f1[xm]) / (4.f * hx);
        float fy = (f0[yp] - f0[ym] + f1[yp] - f1[ym]) / (4.f * hy);
        float fz = (f0[zp] - f0[zm] + f1[zp] - f1[zm]) / (4.f * hz);
        float ft = f1[c] - f0[c];

        float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
        float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
        float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft, J44 = ft * ft;

        float cu = du[c], cv = dv[c], cw = dw[c];
        float s = (J11 * cu + J12 * cv + J13 * cw + J14) * cu +
                  (J12 * cu + J22 * cv + J23 * cw + J24) * cv +
                  (J13 * cu + J23 * cv + J33 * cw + J34) * cw +
                  (J14 * cu + J24 * cv + J34 * cw + J44);
        s = (float)(s > 0) * s;
        ksi[c] = 1.f / (2.f * sqrtf(s + eps_d * eps_d));
      }
}

/* ------------------------------------------------------------------------------------------ */
/* A.4  src/kernels/solve_3d.cu:264-508 (arithmetic :425-506) */
void orc_solve_sweep(const float* f0, const float* f1, const float* u, const float* v, const float* w,
                     const float* du, const float* dv, const float* dw, const float* phi, const float* ksi,
                     int W, int H, int D, float hx, float hy, float hz, float alpha,
                     float* tdu, float* tdv, float* tdw, const orc_geom* g)
{
#pragma omp parallel for schedule(static) if (ORC_BIG(g, W, H))
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        size_t c = IDX(g, x, y, z);
        size_t ixm = IDX(g, mir(x - 1, W), y, z), ixp = IDX(g, mir(x + 1, W), y, z);
        size_t iym = IDX(g, x, mir(y - 1, H), z), iyp = IDX(g, x, mir(y + 1, H), z);
        size_t izm = IDX(g, x, y, mir(z - 1, D)), izp = IDX(g, x, y, mir(z + 1, D));

        float fx = (f0[ixp] - f0[ixm] + f1[ixp] - f1[ixm]) / (4.f * hx);
        float fy = (f0[iyp] - f0[iym] + f1[iyp] - f1[iym]) / (4.f * hy);
        float fz = (f0[izp] - f0[izm] + f1[izp] - f1[izm]) / (4.f * hz);
        float ft = f1[c] - f0[c];

        float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
        float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
        float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft;

        float hx_2 = alpha / (hx * hx);
        float hy_2 = alpha / (hy * hy);
        float hz_2 = alpha / (hz * hz);

        float xp = (float)(x < W - 1) * hx_2;
        float xm = (float)(x > 0) * hx_2;
        float yp = (float)(y < H - 1) * hy_2;
        float ym = (float)(y > 0) * hy_2;
        float zp = (float)(z < D - 1) * hz_2;
        float zm = (float)(z > 0) * hz_2;

        float phi_xp = (phi[ixp] + phi[c]) / 2.f;
        float phi_xm = (phi[ixm] + phi[c]) / 2.f;
        float phi_yp = (phi[iyp] + phi[c]) / 2.f;
        float phi_ym = (phi[iym] + phi[c]) / 2.f;
        float phi_zp = (phi[izp] + phi[c]) / 2.f;
        float phi_zm = (phi[izm] + phi[c]) / 2.f;

        float sumH = (xp * phi_xp + xm * phi_xm + yp * phi_yp + ym * phi_ym + zp * phi_zp + zm * phi_zm);
        float sumU = phi_xp * xp * (u[ixp] + du[ixp] - u[c]) + phi_xm * xm * (u[ixm] + du[ixm] - u[c]) +
                     phi_yp * yp * (u[iyp] + du[iyp] - u[c]) + phi_ym * ym * (u[iym] + du[iym] - u[c]) +
                     phi_zp * zp * (u[izp] + du[izp] - u[c]) + phi_zm * zm * (u[izm] + du[izm] - u[c]);
        float sumV = phi_xp * xp * (v[ixp] + dv[ixp] - v[c]) + phi_xm * xm * (v[ixm] + dv[ixm] - v[c]) +
                     phi_yp * yp * (v[iyp] + dv[iyp] - v[c]) + phi_ym * ym * (v[iym] + dv[iym] - v[c]) +
                     phi_zp * zp * (v[izp] + dv[izp] - v[c]) + phi_zm * zm * (v[izm] + dv[izm] - v[c]);
        float sumW = phi_xp * xp * (w[ixp] + dw[ixp] - w[c]) + phi_xm * xm * (w[ixm] + dw[ixm] - w[c]) +
                     phi_yp * yp * (w[iyp] + dw[iyp] - w[c]) + phi_ym * ym * (w[iym] + dw[iym] - w[c]) +
                     phi_zp * zp * (w[izp] + dw[izp] - w[c]) + phi_zm * zm * (w[izm] + dw[izm] - w[c]);

        float k = ksi[c];
        float r_du = (k * (-J14 - J12 * dv[c] - J13 * dw[c]) + sumU) / (k * J11 + sumH);
        float r_dv = (k * (-J24 - J12 * r_du - J23 * dw[c]) + sumV) / (k * J22 + sumH);
        float r_dw = (k * (-J34 - J13 * r_du - J23 * r_dv) + sumW) / (k * J33 + sumH);

        tdu[c] = r_du;
        tdv[c] = r_dv;
        tdw[c] = r_dw;
      }
}

/* src/kernels/add_3d.cu:26-41 */
void orc_add(float* a, const float* b, int W, int H, int D, const orc_geom* g)
{
  (void)D;
#pragma omp parallel for schedule(static) if (ORC_BIG(g, W, H))
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        size_t c = IDX(g, x, y, z);
        a[c] = a[c] + b[c];
      }
}

/* What one reads off the volume the reference's debug block dumps (optical_flow_e.cpp:536-571: frame_1 registered with the
 * final flow): sum of squares, sum of absolute values (double, scan order) and maximum of |warped - frame_0|. */
void orc_residual_stats(const float* f0, const float* fw, int W, int H, int D, const orc_geom* g, double* sum_sq,
                        double* sum_abs, float* max_abs)
{
  (void)D;
  double ssq = 0.0, sab = 0.0;
  float mx = 0.f;
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        size_t c = IDX(g, x, y, z);
        float d = fw[c] - f0[c];
        float a = fabsf(d);
        mx = fmaxf(mx, a);
        sab += (double)a;
        ssq += (double)d * (double)d;
      }
  *sum_sq = ssq;
  *sum_abs = sab;
  *max_abs = mx;
}

/* cuda_operation_stat_p.cpp:85-104 (serial on purpose: the reference's float sum depends on the scan order) */
void orc_flow_stats(const float* u, const float* v, const float* w, int W, int H, int D, const orc_geom* g,
                    float* min_mag, float* max_mag, float* avg_float, double* sum_double)
{
  (void)D;
  float mn = 3.402823466e+38f, mx = 1.175494351e-38f, avg = 0.f;   /* numeric_limits<float>::max() / ::min() */
  double sum = 0.0;
  size_t count = 0;
  for (int z = g->z_lo; z < g->z_hi; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        size_t c = IDX(g, x, y, z);
        float magnitude = sqrtf(u[c] * u[c] + v[c] * v[c] + w[c] * w[c]);
        mn = fminf(mn, magnitude);
        mx = fmaxf(mx, magnitude);
        avg += magnitude;
        sum += (double)magnitude;
        count++;
      }
  if (count) avg /= (float)count;
  *min_mag = mn;
  *max_mag = mx;
  *avg_float = avg;
  *sum_double = sum;
}

/* ------------------------------------------------------------------------------------------ */
/* A.5  src/kernels/median_3d.cu:34-45 (stable insertion sort), :282-297 (window gather, element r^3/2) */
static void insertion_sort(float* window, int size)
{
  for (int i = 0; i < size; i++) {
    float temp = window[i];
    int j;
    for (j = i - 1; j >= 0 && temp < window[j]; j--) window[j + 1] = window[j];
    window[j + 1] = temp;
  }
}

void orc_median(const float* in, float* out, int W, int H, int D, int r, const orc_geom* g)
{
  const int h = r / 2;
  const int len = r * r * r;
#pragma omp parallel for schedule(dynamic, 1) if ((long)W * H * (g->z_hi - g->z_lo) > 2000)
  for (int z = g->z_lo; z < g->z_hi; z++) {
    float buffer[343];
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        for (int iz = 0; iz < r; ++iz)
          for (int iy = 0; iy < r; ++iy)
            for (int ix = 0; ix < r; ++ix)
              buffer[(iz * r + iy) * r + ix] =
                  in[IDX(g, mir(x - ix + h, W), mir(y - iy + h, H), mir(z - iz + h, D))];
        insertion_sort(buffer, len);
        out[IDX(g, x, y, z)] = buffer[len / 2];
      }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Whole pipeline: src/optical_flow/optical_flow_e.cpp:132-601 with the operator hosts
 * src/cuda_operations/entire_data/cuda_operation_{convolution,resample,registration,solve,add,median}.cpp */

typedef struct { int w, h, d; } dims3;

static void fill_rows(float* p, const orc_geom* g, int wcur, int Dc)
{
  /* cuMemsetD2D8(ptr, pitch, 0, width*4, Hc*Dc rows): optical_flow_e.cpp:305-310, cuda_operation_solve.cpp:183-188 */
  for (size_t row = 0; row < (size_t)g->Hc * (size_t)Dc; row++)
    memset(p + row * (size_t)g->pitch_f, 0, (size_t)wcur * sizeof(float));
}

static void resample3(const float* in, float* out, float* tmp, dims3 a, dims3 b, orc_geom g)
{
  /* cuda_operation_resample.cpp:95-105: X in->out, Y out->temp, Z temp->out */
  orc_geom gi = g, go = g;
  gi.z_lo = go.z_lo = 0; gi.z_hi = go.z_hi = a.d;
  orc_resample_axis(in, out, b.w, a.h, a.d, a.w, 0, &gi, &go);
  orc_resample_axis(out, tmp, b.w, b.h, a.d, a.h, 1, &gi, &go);
  go.z_hi = b.d;
  orc_resample_axis(tmp, out, b.w, b.h, b.d, a.d, 2, &gi, &go);
}

static void swapf(float** a, float** b) { float* t = *a; *a = *b; *b = t; }

int orc_compute_flow(const float* frame0, const float* frame1, size_t W, size_t H, size_t D,
                     const orc_params* p, int pitch_f, float* u_out, float* v_out, float* w_out)
{
  if (pitch_f <= 0) pitch_f = (int)W;
  const size_t csize = (size_t)pitch_f * H * D;
  enum { NBUF = 15 };
  float* buf[NBUF];
  for (int i = 0; i < NBUF; i++) {
    buf[i] = (float*)malloc(csize * sizeof(float));
    /* poison: anything read outside what was written shows up as NaN */
    for (size_t k = 0; k < csize; k++) buf[i][k] = NAN;
  }
  float *f0 = buf[0], *f1 = buf[1], *f0r = buf[2], *f1r = buf[3], *fu = buf[4], *fv = buf[5], *fw = buf[6],
        *du = buf[7], *dv = buf[8], *dw = buf[9], *phi = buf[10], *ksi = buf[11], *tu = buf[12], *tv = buf[13],
        *tw = buf[14];
  /* tu doubles as the single dev_temp scratch of the non-solver stages (LIFO stack top) */

  orc_geom g = { (int)H, pitch_f, 0, 0, (int)D };
  dims3 orig = { (int)W, (int)H, (int)D };

  for (size_t z = 0; z < D; z++)
    for (size_t y = 0; y < H; y++) {
      memcpy(f0 + (z * H + y) * pitch_f, frame0 + (z * H + y) * W, W * sizeof(float));
      memcpy(f1 + (z * H + y) * pitch_f, frame1 + (z * H + y) * W, W * sizeof(float));
    }

  /* Gaussian pre-blur: optical_flow_e.cpp:213-242; rows in->out, columns out->temp, slices temp->out */
  if (p->gaussian_sigma > 0.0) {
    float taps[51];
    int r = orc_gaussian_taps(p->gaussian_sigma, taps, 51);
    orc_conv_axis(fu, f0, orig.w, orig.h, orig.d, r, taps, 0, &g);
    orc_conv_axis(tu, fu, orig.w, orig.h, orig.d, r, taps, 1, &g);
    orc_conv_axis(fu, tu, orig.w, orig.h, orig.d, r, taps, 2, &g);
    orc_conv_axis(fv, f1, orig.w, orig.h, orig.d, r, taps, 0, &g);
    orc_conv_axis(tu, fv, orig.w, orig.h, orig.d, r, taps, 1, &g);
    orc_conv_axis(fv, tu, orig.w, orig.h, orig.d, r, taps, 2, &g);
    swapf(&f0, &fu);
    swapf(&f1, &fv);
  }

  size_t max_level = orc_max_warp_level(W, H, D, p->warp_scale_factor);
  size_t levels = p->warp_levels_count < max_level ? p->warp_levels_count : max_level;
  int level = (int)levels - 1;
  dims3 prev = { 0, 0, 0 };

  while (level >= 0) {
    size_t cw, ch, cd;
    float hx, hy, hz;
    orc_level_geometry(W, H, D, p->warp_scale_factor, level, &cw, &ch, &cd, &hx, &hy, &hz);
    dims3 cur = { (int)cw, (int)ch, (int)cd };
    orc_geom gl = g;
    gl.z_hi = cur.d;

    /* frames: optical_flow_e.cpp:274-300 */
    if (level == 0) {
      swapf(&f0, &f0r);
      swapf(&f1, &f1r);
    } else {
      resample3(f0, f0r, tu, orig, cur, g);
      resample3(f1, f1r, tu, orig, cur, g);
    }
    /* flow: :303-345 (values are not rescaled) */
    if (prev.w == 0) {
      fill_rows(fu, &g, orig.w, orig.d);
      fill_rows(fv, &g, orig.w, orig.d);
      fill_rows(fw, &g, orig.w, orig.d);
    } else {
      resample3(fu, du, tu, prev, cur, g);
      resample3(fv, dv, tu, prev, cur, g);
      resample3(fw, dw, tu, prev, cur, g);
      swapf(&fu, &du);
      swapf(&fv, &dv);
      swapf(&fw, &dw);
    }
    /* warp: :348-369 */
    orc_warp(f0r, f1r, fu, fv, fw, cur.w, cur.h, cur.d, hx, hy, hz, tu, &gl);
    swapf(&f1r, &tu);

    /* solve: cuda_operation_solve.cpp:183-266 */
    fill_rows(du, &g, cur.w, orig.d);
    fill_rows(dv, &g, cur.w, orig.d);
    fill_rows(dw, &g, cur.w, orig.d);
    for (size_t i = 0; i < p->outer_iterations_count; i++) {
      orc_phi_ksi(f0r, f1r, fu, fv, fw, du, dv, dw, cur.w, cur.h, cur.d, hx, hy, hz,
                  p->equation_smoothness, p->equation_data, phi, ksi, &gl);
      for (size_t j = 0; j < p->inner_iterations_count; j++) {
        orc_solve_sweep(f0r, f1r, fu, fv, fw, du, dv, dw, phi, ksi, cur.w, cur.h, cur.d, hx, hy, hz,
                        p->equation_alpha, tu, tv, tw, &gl);
        swapf(&du, &tu);
        swapf(&dv, &tv);
        swapf(&dw, &tw);
      }
    }
    /* add: optical_flow_e.cpp:420-438 */
    orc_add(fu, du, cur.w, cur.h, cur.d, &gl);
    orc_add(fv, dv, cur.w, cur.h, cur.d, &gl);
    orc_add(fw, dw, cur.w, cur.h, cur.d, &gl);

    /* median: :444-473 with the radius rules of cuda_operation_median.cpp:95-106 */
    {
      int r = (int)p->median_radius;
      float** flows[3] = { &fu, &fv, &fw };
      for (int k = 0; k < 3; k++) {
        if (r == 1) {
          memcpy(tu, *flows[k], csize * sizeof(float));
        } else {
          int rr = (r % 2 == 0) ? r - 1 : r;
          if (rr >= 3 && rr <= 7) orc_median(*flows[k], tu, cur.w, cur.h, cur.d, rr, &gl);
        }
        swapf(flows[k], &tu);
      }
    }
    prev = cur;
    --level;
  }

  for (size_t z = 0; z < D; z++)
    for (size_t y = 0; y < H; y++) {
      memcpy(u_out + (z * H + y) * W, fu + (z * H + y) * pitch_f, W * sizeof(float));
      memcpy(v_out + (z * H + y) * W, fv + (z * H + y) * pitch_f, W * sizeof(float));
      memcpy(w_out + (z * H + y) * W, fw + (z * H + y) * pitch_f, W * sizeof(float));
    }
  for (int i = 0; i < NBUF; i++) free(buf[i]);
  return (int)levels;
}
