/*
 * sage_oracle.c -- CPU ORACLE (test infrastructure; see sage_oracle.h header
 * comment for scope, usage rules and the parity-pinning status).
 *
 * Structure deliberately mirrors the reference: every residual's Jacobian row
 * is materialised (J buffers), then reduced with Jt*J / Jt*r.  The reduction
 * accumulates in double (the reference uses an fp32 GEMM whose summation order
 * is library-defined; double accumulation of the same fp32 rows is the
 * order-free statement of that sum).
 *
 * Paths cited are relative to /root/reference/system/sources.
 */
#include "sage_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define R_(x) ((REAL)(x))

static inline REAL orc_floor(REAL x) { return (REAL)floor((double)x); }
static inline REAL orc_round(REAL x) { return (REAL)round((double)x); } /* C round(): half away from zero, as CUDA round() */

/* float -> int conversion used for tap coordinates.  CUDA's conversion
 * saturates; in C out-of-range is UB, so clamp first.  Any clamped value is
 * far outside every image and therefore fails WITHIN_BOUNDS exactly like the
 * saturated CUDA value (photometric_factor_kernels.cpp:16). */
static inline int orc_to_int(REAL x)
{
  if (!(x == x))
    return -1000000000;
  if (x > R_(1.0e9))
    return 1000000000;
  if (x < R_(-1.0e9))
    return -1000000000;
  return (int)x;
}

static inline int within(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

/* 4-tap zero-padded bilinear sampler.  Tap order and summation order
 * nw + se + sw + ne follow photometric_factor_kernels.cpp:106-139 and
 * geometric_factor_kernels.cpp:546-571. */
typedef struct
{
  int ok[4];
  int off[4];
  REAL w[4];
} tap4_t;

static inline void make_taps(tap4_t *tp, REAL u, REAL v, int W, int H)
{
  const int xf = orc_to_int(orc_floor(u)), yf = orc_to_int(orc_floor(v));
  const int xc = xf + 1, yc = yf + 1;
  const REAL lx = (REAL)xc - u, ly = (REAL)yc - v; /* lower weights */
  const REAL ux = 1 - lx, uy = 1 - ly;             /* upper weights */
  tp->w[0] = lx * ly; /* nw: (xf, yf) */
  tp->w[1] = ux * uy; /* se: (xc, yc) */
  tp->w[2] = lx * uy; /* sw: (xf, yc) */
  tp->w[3] = ux * ly; /* ne: (xc, yf) */
  tp->ok[0] = within(xf, yf, W, H);
  tp->ok[1] = within(xc, yc, W, H);
  tp->ok[2] = within(xf, yc, W, H);
  tp->ok[3] = within(xc, yf, W, H);
  tp->off[0] = yf * W + xf;
  tp->off[1] = yc * W + xc;
  tp->off[2] = yc * W + xf;
  tp->off[3] = yf * W + xc;
}

static inline REAL sample(const tap4_t *tp, const REAL *img)
{
  return (tp->ok[0] ? img[tp->off[0]] * tp->w[0] : 0) +
         (tp->ok[1] ? img[tp->off[1]] * tp->w[1] : 0) +
         (tp->ok[2] ? img[tp->off[2]] * tp->w[2] : 0) +
         (tp->ok[3] ? img[tp->off[3]] * tp->w[3] : 0);
}

/* strided variant for texel-major images ([H,W,C], channel c) */
static inline REAL sample_strided(const tap4_t *tp, const REAL *img, int stride, int c)
{
  return (tp->ok[0] ? img[(size_t)tp->off[0] * stride + c] * tp->w[0] : 0) +
         (tp->ok[1] ? img[(size_t)tp->off[1] * stride + c] * tp->w[1] : 0) +
         (tp->ok[2] ? img[(size_t)tp->off[2] * stride + c] * tp->w[2] : 0) +
         (tp->ok[3] ? img[(size_t)tp->off[3] * stride + c] * tp->w[3] : 0);
}

/* nearest full-resolution mask lookup with C round():
 * photometric_factor_kernels.cpp:159-166, geometric_factor_kernels.cpp:585-598 */
static inline REAL mask_lookup(const REAL *mask, REAL p, REAL q, int W, int H)
{
  const int xr = orc_to_int(orc_round(p)), yr = orc_to_int(orc_round(q));
  return within(xr, yr, W, H) ? mask[yr * W + xr] : 0;
}

void ORC(camera_pyramid)(const ORC(cam_t) * base, int levels, ORC(cam_t) * out)
{
  /* common/camera_pyramid.h:18-32: level i = level i-1 resized to
   * (size_t)(w/2),(size_t)(h/2); ResizeViewport (pinhole_camera_impl.h:120-132)
   * scales fx,u0 by new_w/w and fy,v0 by new_h/h. */
  for (int i = 0; i < levels; ++i)
  {
    out[i] = (i == 0) ? *base : out[i - 1];
    if (i != 0)
    {
      const size_t nw = (size_t)(out[i - 1].w / 2);
      const size_t nh = (size_t)(out[i - 1].h / 2);
      const REAL xr = (REAL)nw / out[i].w;
      const REAL yr = (REAL)nh / out[i].h;
      out[i].fx *= xr;
      out[i].fy *= yr;
      out[i].cx *= xr;
      out[i].cy *= yr;
      out[i].w = (REAL)nw;
      out[i].h = (REAL)nh;
    }
  }
}

/* -------------------------------------------------------------------------
 * Jt*W*J reduction of a materialised row buffer.
 *   AtA[D,D] = scale * sum_rows wrow * J[row,:]^T J[row,:]
 *   Atb[D]   = scale * sum_rows wrow * J[row,:]^T r[row]
 * rows are grouped in `groups` equal chunks with weight gw[g] (level weights);
 * follows photometric_factor_kernels.cpp:1143-1152 / geometric...:936-940.
 * ------------------------------------------------------------------------- */
static void reduce_normal_eq(REAL *AtA, REAL *Atb, const REAL *J, const REAL *r,
                             size_t rows_per_group, int groups, const REAL *gw, int D, double scale)
{
  const size_t DD = (size_t)D * D;
  double *acc = (double *)calloc(DD + D, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    double *loc = (double *)calloc(DD + D, sizeof(double));
    for (int g = 0; g < groups; ++g)
    {
      const double w = gw ? (double)gw[g] : 1.0;
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
      for (long long row = 0; row < (long long)rows_per_group; ++row)
      {
        const REAL *j = J + ((size_t)g * rows_per_group + (size_t)row) * D;
        const double rr = (double)r[(size_t)g * rows_per_group + (size_t)row];
        int nz = 0;
        for (int a = 0; a < D; ++a)
          nz |= (j[a] != 0);
        if (!nz)
          continue; /* all-zero row contributes nothing */
        for (int a = 0; a < D; ++a)
        {
          const double wa = w * (double)j[a];
          if (wa == 0.0)
            continue;
          double *rowp = loc + (size_t)a * D;
          for (int b = 0; b < D; ++b)
            rowp[b] += wa * (double)j[b];
          loc[DD + a] += wa * rr;
        }
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      for (size_t i = 0; i < DD + D; ++i)
        acc[i] += loc[i];
    }
    free(loc);
  }
  for (size_t i = 0; i < DD; ++i)
    AtA[i] = (REAL)(scale * acc[i]);
  for (int i = 0; i < D; ++i)
    Atb[i] = (REAL)(scale * acc[DD + i]);
  free(acc);
}

/* sampled depth: scale0*(bias0[i] + basis0[i,:].code0)
 * photometric_factor_kernels.cpp:1094-1095 (gather + matmul on host),
 * geometric_factor_kernels.cpp:514-521 (in-kernel loop). */
static inline REAL sampled_depth(const REAL *bias0, const REAL *basis0, const REAL *code0,
                                 long long i, int CS, REAL scale0)
{
  REAL d = bias0[i];
  for (int k = 0; k < CS; ++k)
    d += basis0[(size_t)i * CS + k] * code0[k];
  return d * scale0;
}

/* =========================================================================
 * a1  photometric_jac_error_calculate  (kernel :33-368, host :1061-1164)
 * ========================================================================= */
void ORC(photo_jac_error)(REAL *AtA, REAL *Atb, REAL *error, REAL *num_inliers,
                          const REAL *R10, const REAL *t10, const REAL *R0, const REAL *t0,
                          const REAL *R1, const REAL *t1,
                          const REAL *bias0, const REAL *basis0, const REAL *code0,
                          const REAL *mask1, const int64_t *loc1d, const REAL *homo,
                          const REAL *feat0, const REAL *feat1, const REAL *grad1,
                          const int32_t *level_offsets, REAL scale0,
                          const ORC(cam_t) * cams, int L, int N, int FS, int CS, int P,
                          REAL eps, const REAL *weights,
                          REAL *J_out, REAL *r_out, REAL *err_out, REAL *valid_out)
{
  const int D = 13 + CS;
  const size_t rows = (size_t)L * N * FS;
  REAL *J = J_out ? J_out : (REAL *)malloc(rows * D * sizeof(REAL));
  REAL *r = r_out ? r_out : (REAL *)malloc(rows * sizeof(REAL));
  REAL *serr = err_out ? err_out : (REAL *)malloc((size_t)L * N * sizeof(REAL));
  REAL *sval = valid_out ? valid_out : (REAL *)malloc((size_t)L * N * sizeof(REAL));

  const int W0 = (int)cams[0].w, H0 = (int)cams[0].h;
  const REAL fx0 = cams[0].fx, fy0 = cams[0].fy, cx0 = cams[0].cx, cy0 = cams[0].cy;

#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
  for (int level = 0; level < L; ++level)
  {
    for (int idx = 0; idx < N; ++idx)
    {
      const int Hl = (int)cams[level].h, Wl = (int)cams[level].w;
      const REAL fx = cams[level].fx, fy = cams[level].fy;
      const REAL *hm = homo + (size_t)idx * 3;
      const long long i1d = (long long)loc1d[idx];
      const REAL d = sampled_depth(bias0, basis0, code0, i1d, CS, scale0);

      REAL rh[3], X[3];
      for (int i = 0; i < 3; ++i) /* :78-93 */
        rh[i] = R10[i * 3 + 0] * hm[0] + R10[i * 3 + 1] * hm[1] + R10[i * 3 + 2] * hm[2];
      for (int i = 0; i < 3; ++i)
        X[i] = d * rh[i] + t10[i];
      const int pos = X[2] > eps; /* :96 */

      /* source coords from homo (:101-103) */
      const REAL u0 = (hm[0] * fx0 + cx0 + R_(0.5)) * fx / fx0 - R_(0.5);
      const REAL v0 = (hm[1] * fy0 + cy0 + R_(0.5)) * fy / fy0 - R_(0.5);
      tap4_t tp0, tp1;
      make_taps(&tp0, u0, v0, Wl, Hl);
      /* destination coords (:142-144) */
      const REAL p = (X[0] / X[2]) * fx0 + cx0;
      const REAL q = (X[1] / X[2]) * fy0 + cy0;
      const REAL u1 = (p + R_(0.5)) * fx / fx0 - R_(0.5);
      const REAL v1 = (q + R_(0.5)) * fy / fy0 - R_(0.5);
      make_taps(&tp1, u1, v1, Wl, Hl);
      const REAL m = mask_lookup(mask1, p, q, W0, H0); /* :159-166 */

      const size_t lo = (size_t)level_offsets[level];

      /* Jacobian geometry (:241-335) */
      const REAL inv_z = 1 / X[2];
      const REAL x_z = inv_z * X[0], y_z = inv_z * X[1];
      const REAL Jpi[2][3] = {{fx * inv_z, 0, -fx * x_z * inv_z}, {0, fy * inv_z, -fy * y_z * inv_z}};
      REAL Xw[3];
      for (int i = 0; i < 3; ++i) /* :247-255 */
        Xw[i] = d * (R0[i * 3 + 0] * hm[0] + R0[i * 3 + 1] * hm[1] + R0[i * 3 + 2] * hm[2]) + t0[i];
      REAL dX1[3][6], dX0[3][6]; /* dX/dT1 (:258-268), dX/dT0 (:283-297) */
      for (int i = 0; i < 3; ++i)
      {
        dX1[i][0] = -R1[0 * 3 + i];
        dX1[i][1] = -R1[1 * 3 + i];
        dX1[i][2] = -R1[2 * 3 + i];
        dX1[i][3] = R1[1 * 3 + i] * Xw[2] - R1[2 * 3 + i] * Xw[1];
        dX1[i][4] = -R1[0 * 3 + i] * Xw[2] + R1[2 * 3 + i] * Xw[0];
        dX1[i][5] = R1[0 * 3 + i] * Xw[1] - R1[1 * 3 + i] * Xw[0];
      }
      const REAL E[3][6] = {{1, 0, 0, 0, Xw[2], -Xw[1]}, {0, 1, 0, -Xw[2], 0, Xw[0]}, {0, 0, 1, Xw[1], -Xw[0], 0}};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j)
          dX0[i][j] = R1[0 * 3 + i] * E[0][j] + R1[1 * 3 + i] * E[1][j] + R1[2 * 3 + i] * E[2][j];
      REAL P0[2][6], P1[2][6];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j)
        {
          P1[i][j] = Jpi[i][0] * dX1[0][j] + Jpi[i][1] * dX1[1][j] + Jpi[i][2] * dX1[2][j];
          P0[i][j] = Jpi[i][0] * dX0[0][j] + Jpi[i][1] * dX0[1][j] + Jpi[i][2] * dX0[2][j];
        }
      const REAL qd[2] = {fx * (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z), /* :324-325 */
                          fy * (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z)};
      const REAL qs[2] = {qd[0] * d / scale0, qd[1] * d / scale0}; /* :335 */

      REAL ferr = 0;
      for (int c = 0; c < FS; ++c)
      {
        const REAL f0 = sample(&tp0, feat0 + (size_t)c * P + lo);
        const REAL f1 = sample(&tp1, feat1 + (size_t)c * P + lo);
        REAL g[2];
        for (int j = 0; j < 2; ++j) /* :191-222 */
          g[j] = pos ? m * sample(&tp1, grad1 + ((size_t)j * FS + c) * P + lo) : 0;
        const REAL diff = f0 - f1;
        ferr += pos ? m * (diff * diff) : 0; /* :228 */
        const size_t row = ((size_t)level * N + idx) * FS + c;
        r[row] = pos ? m * diff : 0; /* :234 */
        REAL *jr = J + row * D;
        for (int j = 0; j < 6; ++j) /* :319-320, :355-356 */
        {
          jr[j] = g[0] * P0[0][j] + g[1] * P0[1][j];
          jr[6 + j] = g[0] * P1[0][j] + g[1] * P1[1][j];
        }
        for (int k = 0; k < CS; ++k) /* :331-332, :345, :361 */
        {
          const REAL b = basis0[(size_t)i1d * CS + k];
          jr[12 + k] = g[0] * (qd[0] * scale0 * b) + g[1] * (qd[1] * scale0 * b);
        }
        jr[12 + CS] = g[0] * qs[0] + g[1] * qs[1]; /* :347, :363 */
      }
      sval[(size_t)level * N + idx] = pos ? m : 0; /* :237 */
      serr[(size_t)level * N + idx] = ferr;        /* :238 */
    }
  }

  /* host reduction (:1139-1161) */
  double n_in = 0;
  for (int idx = 0; idx < N; ++idx)
    n_in += (double)sval[idx]; /* level 0 only (:1139) */
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
  {
    reduce_normal_eq(AtA, Atb, J, r, (size_t)N * FS, L, weights, D, 1.0 / n_in);
    double e = 0;
    for (int level = 0; level < L; ++level)
    {
      double s = 0;
      for (int idx = 0; idx < N; ++idx)
        s += (double)serr[(size_t)level * N + idx];
      e += (double)weights[level] * s;
    }
    *error = (REAL)(e / n_in);
  }
  else
  {
    double ws = 0;
    for (int level = 0; level < L; ++level)
      ws += (double)weights[level];
    *error = (REAL)(ws * 10.0); /* :1158 */
    memset(AtA, 0, sizeof(REAL) * D * D);
    memset(Atb, 0, sizeof(REAL) * D);
  }
  if (!J_out)
    free(J);
  if (!r_out)
    free(r);
  if (!err_out)
    free(serr);
  if (!valid_out)
    free(sval);
}

/* =========================================================================
 * a2  photometric_error_calculate (kernel :370-522, host :990-1059)
 * source coords come from loc2d = (fmod(loc1d,W), floor(loc1d/W)) (:1012-1014, :423-424)
 * ========================================================================= */
REAL ORC(photo_error)(const REAL *R10, const REAL *t10,
                      const REAL *bias0, const REAL *basis0, const REAL *code0,
                      const REAL *mask1, const int64_t *loc1d, const REAL *homo,
                      const REAL *feat0, const REAL *feat1,
                      const int32_t *level_offsets, REAL scale0,
                      const ORC(cam_t) * cams, int L, int N, int FS, int CS, int P,
                      REAL eps, const REAL *weights, REAL *num_inliers)
{
  const int W0 = (int)cams[0].w, H0 = (int)cams[0].h;
  const REAL fx0 = cams[0].fx, fy0 = cams[0].fy, cx0 = cams[0].cx, cy0 = cams[0].cy;
  double e = 0, n_in = 0;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : e, n_in)
#endif
  for (int level = 0; level < L; ++level)
  {
    for (int idx = 0; idx < N; ++idx)
    {
      const int Hl = (int)cams[level].h, Wl = (int)cams[level].w;
      const REAL fx = cams[level].fx, fy = cams[level].fy;
      const REAL *hm = homo + (size_t)idx * 3;
      const long long i1d = (long long)loc1d[idx];
      const REAL d = sampled_depth(bias0, basis0, code0, i1d, CS, scale0);
      REAL rh[3], X[3];
      for (int i = 0; i < 3; ++i)
        rh[i] = R10[i * 3 + 0] * hm[0] + R10[i * 3 + 1] * hm[1] + R10[i * 3 + 2] * hm[2];
      for (int i = 0; i < 3; ++i)
        X[i] = d * rh[i] + t10[i];
      const int pos = X[2] > eps;
      const REAL lx = (REAL)fmod((double)(REAL)i1d, (double)cams[0].w);
      const REAL ly = orc_floor((REAL)i1d / cams[0].w);
      const REAL u0 = (lx + R_(0.5)) * fx / fx0 - R_(0.5);
      const REAL v0 = (ly + R_(0.5)) * fy / fy0 - R_(0.5);
      tap4_t tp0, tp1;
      make_taps(&tp0, u0, v0, Wl, Hl);
      const REAL p = (X[0] / X[2]) * fx0 + cx0;
      const REAL q = (X[1] / X[2]) * fy0 + cy0;
      make_taps(&tp1, (p + R_(0.5)) * fx / fx0 - R_(0.5), (q + R_(0.5)) * fy / fy0 - R_(0.5), Wl, Hl);
      const REAL m = mask_lookup(mask1, p, q, W0, H0);
      const size_t lo = (size_t)level_offsets[level];
      REAL ferr = 0;
      for (int c = 0; c < FS; ++c)
      {
        const REAL f0 = sample(&tp0, feat0 + (size_t)c * P + lo);
        const REAL f1 = sample(&tp1, feat1 + (size_t)c * P + lo);
        const REAL diff = f1 - f0;
        ferr += pos ? m * (diff * diff) : 0; /* :514 */
      }
      e += (double)weights[level] * (double)ferr;
      if (level == 0)
        n_in += pos ? (double)m : 0.0;
    }
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
    return (REAL)(e / n_in);
  double ws = 0;
  for (int level = 0; level < L; ++level)
    ws += (double)weights[level];
  return (REAL)(ws * 10.0); /* :1057 */
}

/* =========================================================================
 * a3 tracker kernels: relative pose only, pre-sampled source features
 * 6-dof :524-695, 7-dof :697-873, hosts :1166-1325
 * ========================================================================= */
void ORC(tracker_photo_jac_error)(REAL *AtA, REAL *Atb, REAL *error, REAL *num_inliers, int dof,
                                  const REAL *R, const REAL *t, const REAL *mask1,
                                  const REAL *dpts0, const REAL *homo, const REAL *feat0s,
                                  const REAL *feat1, const REAL *grad1,
                                  const int32_t *level_offsets, const ORC(cam_t) * cams,
                                  int L, int N, int FS, int P, REAL scale0, REAL eps,
                                  const REAL *weights, REAL *J_out, REAL *r_out)
{
  const int D = dof;
  const size_t rows = (size_t)L * N * FS;
  REAL *J = J_out ? J_out : (REAL *)malloc(rows * D * sizeof(REAL));
  REAL *r = r_out ? r_out : (REAL *)malloc(rows * sizeof(REAL));
  REAL *serr = (REAL *)malloc((size_t)L * N * sizeof(REAL));
  REAL *sval = (REAL *)malloc((size_t)L * N * sizeof(REAL));
  const int W0 = (int)cams[0].w, H0 = (int)cams[0].h;
  const REAL fx0 = cams[0].fx, fy0 = cams[0].fy, cx0 = cams[0].cx, cy0 = cams[0].cy;

#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
  for (int level = 0; level < L; ++level)
  {
    for (int idx = 0; idx < N; ++idx)
    {
      const int Hl = (int)cams[level].h, Wl = (int)cams[level].w;
      const REAL fx = cams[level].fx, fy = cams[level].fy;
      const REAL *hm = homo + (size_t)idx * 3;
      const REAL d = dpts0[idx];
      REAL rh[3], X[3];
      for (int i = 0; i < 3; ++i)
        rh[i] = R[i * 3 + 0] * hm[0] + R[i * 3 + 1] * hm[1] + R[i * 3 + 2] * hm[2];
      for (int i = 0; i < 3; ++i)
        X[i] = d * rh[i] + t[i]; /* :560-574 */
      const int pos = X[2] > eps;
      const REAL p = (X[0] / X[2]) * fx0 + cx0;
      const REAL q = (X[1] / X[2]) * fy0 + cy0;
      tap4_t tp1;
      make_taps(&tp1, (p + R_(0.5)) * fx / fx0 - R_(0.5), (q + R_(0.5)) * fy / fy0 - R_(0.5), Wl, Hl);
      const REAL m = mask_lookup(mask1, p, q, W0, H0);
      const size_t lo = (size_t)level_offsets[level];

      const REAL inv_z = 1 / X[2];
      const REAL x_z = inv_z * X[0], y_z = inv_z * X[1];
      /* closed-form 2x6 (:680-681) */
      const REAL Jp[2][6] = {{fx * inv_z, 0, -fx * x_z * inv_z, -fx * x_z * y_z, fx * (1 + x_z * x_z), -fx * y_z},
                             {0, fy * inv_z, -fy * y_z * inv_z, -fy * (1 + y_z * y_z), fy * x_z * y_z, fy * x_z}};
      const REAL qd[2] = {fx * (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z), /* :854-855 */
                          fy * (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z)};
      REAL qs[2] = {0, 0};
      if (dof == 7)
      {
        qs[0] = qd[0] * d / scale0; /* :856 */
        qs[1] = qd[1] * d / scale0;
      }
      REAL ferr = 0;
      for (int c = 0; c < FS; ++c)
      {
        const REAL f0 = feat0s[((size_t)level * N + idx) * FS + c];
        const REAL f1 = sample(&tp1, feat1 + (size_t)c * P + lo);
        REAL g[2];
        for (int j = 0; j < 2; ++j)
          g[j] = pos ? m * sample(&tp1, grad1 + ((size_t)j * FS + c) * P + lo) : 0;
        const REAL diff = f0 - f1;
        ferr += pos ? m * (diff * diff) : 0; /* :665 */
        const size_t row = ((size_t)level * N + idx) * FS + c;
        r[row] = pos ? m * diff : 0;
        REAL *jr = J + row * D;
        for (int j = 0; j < 6; ++j)
          jr[j] = g[0] * Jp[0][j] + g[1] * Jp[1][j]; /* :688-689 */
        if (dof == 7)
          jr[6] = g[0] * qs[0] + g[1] * qs[1]; /* :867-868 */
      }
      sval[(size_t)level * N + idx] = pos ? m : 0;
      serr[(size_t)level * N + idx] = ferr;
    }
  }
  double n_in = 0;
  for (int idx = 0; idx < N; ++idx)
    n_in += (double)sval[idx];
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
  {
    reduce_normal_eq(AtA, Atb, J, r, (size_t)N * FS, L, weights, D, 1.0 / n_in);
    double e = 0;
    for (int level = 0; level < L; ++level)
    {
      double s = 0;
      for (int idx = 0; idx < N; ++idx)
        s += (double)serr[(size_t)level * N + idx];
      e += (double)weights[level] * s;
    }
    *error = (REAL)(e / n_in);
  }
  else
  {
    double ws = 0;
    for (int level = 0; level < L; ++level)
      ws += (double)weights[level];
    *error = (REAL)(ws * 10.0); /* :1239 */
    memset(AtA, 0, sizeof(REAL) * D * D);
    memset(Atb, 0, sizeof(REAL) * D);
  }
  if (!J_out)
    free(J);
  if (!r_out)
    free(r);
  free(serr);
  free(sval);
}

REAL ORC(tracker_photo_error)(const REAL *R, const REAL *t, const REAL *mask1,
                              const REAL *dpts0, const REAL *homo, const REAL *feat0s,
                              const REAL *feat1, const int32_t *level_offsets,
                              const ORC(cam_t) * cams, int L, int N, int FS, int P,
                              REAL eps, const REAL *weights, REAL *num_inliers)
{
  const int W0 = (int)cams[0].w, H0 = (int)cams[0].h;
  const REAL fx0 = cams[0].fx, fy0 = cams[0].fy, cx0 = cams[0].cx, cy0 = cams[0].cy;
  double e = 0, n_in = 0;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : e, n_in)
#endif
  for (int level = 0; level < L; ++level)
  {
    for (int idx = 0; idx < N; ++idx)
    {
      const int Hl = (int)cams[level].h, Wl = (int)cams[level].w;
      const REAL fx = cams[level].fx, fy = cams[level].fy;
      const REAL *hm = homo + (size_t)idx * 3;
      const REAL d = dpts0[idx];
      REAL X[3];
      for (int i = 0; i < 3; ++i)
        X[i] = d * (R[i * 3 + 0] * hm[0] + R[i * 3 + 1] * hm[1] + R[i * 3 + 2] * hm[2]) + t[i];
      const int pos = X[2] > eps;
      const REAL p = (X[0] / X[2]) * fx0 + cx0;
      const REAL q = (X[1] / X[2]) * fy0 + cy0;
      tap4_t tp1;
      make_taps(&tp1, (p + R_(0.5)) * fx / fx0 - R_(0.5), (q + R_(0.5)) * fy / fy0 - R_(0.5), Wl, Hl);
      const REAL m = mask_lookup(mask1, p, q, W0, H0);
      const size_t lo = (size_t)level_offsets[level];
      REAL ferr = 0;
      for (int c = 0; c < FS; ++c)
      {
        const REAL f0 = feat0s[((size_t)level * N + idx) * FS + c];
        const REAL f1 = sample(&tp1, feat1 + (size_t)c * P + lo);
        const REAL diff = f0 - f1;
        ferr += pos ? m * (diff * diff) : 0;
      }
      e += (double)weights[level] * (double)ferr;
      if (level == 0)
        n_in += pos ? (double)m : 0.0;
    }
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
    return (REAL)(e / n_in);
  double ws = 0;
  for (int level = 0; level < L; ++level)
    ws += (double)weights[level];
  return (REAL)(ws * 10.0); /* :1382 */
}

/* =========================================================================
 * a4  geometric_jac_error_calculate (kernel :472-720, host :882-950)
 * level-0 camera only, NO half-pixel shifts.
 * ========================================================================= */
void ORC(geo_jac_error)(REAL *AtA, REAL *Atb, REAL *error, REAL *num_inliers,
                        const REAL *R10, const REAL *t10, const REAL *R0, const REAL *t0,
                        const REAL *R1, const REAL *t1,
                        const REAL *bias0, const REAL *basis0, const REAL *code0,
                        const REAL *dpt1, const REAL *dpt_grad1, const REAL *basis1,
                        const REAL *mask1, const int32_t *loc1d, const REAL *homo,
                        REAL scale0, REAL scale1, const ORC(cam_t) * cam, int N, int CS,
                        REAL eps, REAL loss_param, REAL weight, REAL *J_out, REAL *r_out)
{
  const int D = 14 + 2 * CS;
  REAL *J = J_out ? J_out : (REAL *)malloc((size_t)N * D * sizeof(REAL));
  REAL *r = r_out ? r_out : (REAL *)malloc((size_t)N * sizeof(REAL));
  REAL *serr = (REAL *)malloc((size_t)N * sizeof(REAL));
  REAL *sval = (REAL *)malloc((size_t)N * sizeof(REAL));
  const REAL fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
  const int H = (int)cam->h, W = (int)cam->w;
  const size_t HW = (size_t)H * W;

#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *hm = homo + (size_t)idx * 3;
    const long long i1d = (long long)loc1d[idx];
    const REAL d0 = sampled_depth(bias0, basis0, code0, i1d, CS, scale0); /* :514-521 */
    REAL rh[3], X[3];
    for (int i = 0; i < 3; ++i)
      rh[i] = R10[i * 3 + 0] * hm[0] + R10[i * 3 + 1] * hm[1] + R10[i * 3 + 2] * hm[2];
    for (int i = 0; i < 3; ++i)
      X[i] = d0 * rh[i] + t10[i];
    const int pos = X[2] > eps; /* :541 */
    const REAL u = (X[0] / X[2]) * fx + cx; /* :543-544 */
    const REAL v = (X[1] / X[2]) * fy + cy;
    tap4_t tp;
    make_taps(&tp, u, v, W, H);
    const REAL Ds = sample(&tp, dpt1); /* :567-571 */
    const REAL gD[2] = {sample(&tp, dpt_grad1), sample(&tp, dpt_grad1 + HW)}; /* :575-582 */
    const REAL m = mask_lookup(mask1, u, v, W, H); /* :585-598 */
    const REAL rho = Ds - X[2];
    {
      const REAL mr = m * rho;
      serr[idx] = pos ? (REAL)log(1.0 + (double)(mr * mr) / (double)loss_param) : 0; /* :600 */
    }
    sval[idx] = pos ? m : 0;

    const REAL inv_z = 1 / X[2];
    const REAL x_z = inv_z * X[0], y_z = inv_z * X[1];
    const REAL Jpi[2][3] = {{fx * inv_z, 0, -fx * x_z * inv_z}, {0, fy * inv_z, -fy * y_z * inv_z}};
    REAL Xw[3];
    for (int i = 0; i < 3; ++i)
      Xw[i] = d0 * (R0[i * 3 + 0] * hm[0] + R0[i * 3 + 1] * hm[1] + R0[i * 3 + 2] * hm[2]) + t0[i];
    REAL dX1[3][6], dX0[3][6];
    for (int i = 0; i < 3; ++i)
    {
      dX1[i][0] = -R1[0 * 3 + i];
      dX1[i][1] = -R1[1 * 3 + i];
      dX1[i][2] = -R1[2 * 3 + i];
      dX1[i][3] = R1[1 * 3 + i] * Xw[2] - R1[2 * 3 + i] * Xw[1];
      dX1[i][4] = -R1[0 * 3 + i] * Xw[2] + R1[2 * 3 + i] * Xw[0];
      dX1[i][5] = R1[0 * 3 + i] * Xw[1] - R1[1 * 3 + i] * Xw[0];
    }
    const REAL E[3][6] = {{1, 0, 0, 0, Xw[2], -Xw[1]}, {0, 1, 0, -Xw[2], 0, Xw[0]}, {0, 0, 1, Xw[1], -Xw[0], 0}};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 6; ++j)
        dX0[i][j] = R1[0 * 3 + i] * E[0][j] + R1[1 * 3 + i] * E[1][j] + R1[2 * 3 + i] * E[2][j];
    REAL a0[6], a1[6];
    for (int j = 0; j < 6; ++j) /* :671-679 */
    {
      const REAL P0x = Jpi[0][0] * dX0[0][j] + Jpi[0][1] * dX0[1][j] + Jpi[0][2] * dX0[2][j];
      const REAL P0y = Jpi[1][0] * dX0[0][j] + Jpi[1][1] * dX0[1][j] + Jpi[1][2] * dX0[2][j];
      const REAL P1x = Jpi[0][0] * dX1[0][j] + Jpi[0][1] * dX1[1][j] + Jpi[0][2] * dX1[2][j];
      const REAL P1y = Jpi[1][0] * dX1[0][j] + Jpi[1][1] * dX1[1][j] + Jpi[1][2] * dX1[2][j];
      a0[j] = dX0[2][j] - (gD[0] * P0x + gD[1] * P0y);
      a1[j] = dX1[2][j] - (gD[0] * P1x + gD[1] * P1y);
    }
    const REAL qd[2] = {fx * (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z), /* :681-682 */
                        fy * (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z)};
    const REAL dD_dd0 = gD[0] * qd[0] + gD[1] * qd[1]; /* :684 */
    const REAL kappa_s0 = (rh[2] - dD_dd0) * scale0;   /* :685 */
    const REAL a_s0 = (rh[2] - dD_dd0) * d0 / scale0;  /* :687 */
    const REAL a_s1 = -Ds / scale1;                    /* :688 */
    const REAL sw = pos ? m * (REAL)sqrt(1.0 / ((double)(rho * rho) + (double)loss_param)) : 0; /* :690 */

    r[idx] = sw * rho; /* :699 */
    REAL *jr = J + (size_t)idx * D;
    for (int j = 0; j < 6; ++j)
    {
      jr[j] = sw * a0[j];
      jr[6 + j] = sw * a1[j];
    }
    for (int k = 0; k < CS; ++k) /* :592-595, :695-696, :711-712 */
    {
      const REAL beta = sample_strided(&tp, basis1, CS, k);
      jr[12 + k] = sw * (kappa_s0 * basis0[(size_t)i1d * CS + k]);
      jr[12 + CS + k] = sw * (-scale1 * beta);
    }
    jr[12 + 2 * CS] = sw * a_s0;
    jr[13 + 2 * CS] = sw * a_s1;
  }

  double n_in = 0, e = 0;
  for (int idx = 0; idx < N; ++idx)
  {
    n_in += (double)sval[idx];
    e += (double)serr[idx];
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
  {
    *error = (REAL)((double)weight / n_in * e); /* :934 */
    reduce_normal_eq(AtA, Atb, J, r, (size_t)N, 1, NULL, D, (double)weight / n_in);
  }
  else
  {
    *error = (REAL)((double)weight * 10.0); /* :944 */
    memset(AtA, 0, sizeof(REAL) * D * D);
    memset(Atb, 0, sizeof(REAL) * D);
  }
  if (!J_out)
    free(J);
  if (!r_out)
    free(r);
  free(serr);
  free(sval);
}

/* a5 geometric_error_calculate (kernel :127-218, host :837-880) */
REAL ORC(geo_error)(const REAL *R10, const REAL *t10,
                    const REAL *bias0, const REAL *basis0, const REAL *code0,
                    const REAL *dpt1, const REAL *mask1, const int32_t *loc1d, const REAL *homo,
                    REAL scale0, const ORC(cam_t) * cam, int N, int CS,
                    REAL eps, REAL loss_param, REAL weight, REAL *num_inliers)
{
  const REAL fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
  const int H = (int)cam->h, W = (int)cam->w;
  double e = 0, n_in = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : e, n_in)
#endif
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *hm = homo + (size_t)idx * 3;
    const long long i1d = (long long)loc1d[idx];
    const REAL d0 = sampled_depth(bias0, basis0, code0, i1d, CS, scale0);
    REAL X[3];
    for (int i = 0; i < 3; ++i)
      X[i] = d0 * (R10[i * 3 + 0] * hm[0] + R10[i * 3 + 1] * hm[1] + R10[i * 3 + 2] * hm[2]) + t10[i];
    const int pos = X[2] > eps;
    const REAL u = (X[0] / X[2]) * fx + cx;
    const REAL v = (X[1] / X[2]) * fy + cy;
    tap4_t tp;
    make_taps(&tp, u, v, W, H);
    const REAL Ds = sample(&tp, dpt1);
    const REAL m = mask_lookup(mask1, u, v, W, H);
    const REAL mr = m * (Ds - X[2]);
    const REAL se = pos ? (REAL)log(1.0 + (double)(mr * mr) / (double)loss_param) : 0; /* :213 */
    e += (double)se;
    n_in += pos ? (double)m : 0.0;
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
    return (REAL)((double)weight * e / n_in); /* :874 */
  return (REAL)((double)weight * 10.0);       /* :878 */
}

/* =========================================================================
 * f1 producers
 * ========================================================================= */
void ORC(update_depth)(REAL *dpt, const REAL *bias, const REAL *basis, const REAL *code,
                       REAL scale, int HW, int CS)
{
  /* core/mapping/mapping_utils.h:215-222 */
  for (int i = 0; i < HW; ++i)
  {
    REAL s = 0;
    for (int k = 0; k < CS; ++k)
      s += basis[(size_t)i * CS + k] * code[k];
    dpt[i] = scale * (bias[i] + s);
  }
}

void ORC(spatial_grad)(REAL *grad, const REAL *img, int C, int H, int W)
{
  /* core/mapping/mapping_utils.h:236-252: replicate pad, 0.5*(next - prev), x then y */
  const size_t HW = (size_t)H * W;
  for (int c = 0; c < C; ++c)
  {
    const REAL *a = img + (size_t)c * HW;
    REAL *gx = grad + (size_t)c * HW;
    REAL *gy = grad + ((size_t)C + c) * HW;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
      {
        const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
        const int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : H - 1;
        gx[(size_t)y * W + x] = R_(0.5) * (a[(size_t)y * W + xp] - a[(size_t)y * W + xm]);
        gy[(size_t)y * W + x] = R_(0.5) * (a[(size_t)yp * W + x] - a[(size_t)ym * W + x]);
      }
  }
}

void ORC(gaussian_pyramid_with_grad)(REAL *pyr, REAL *grad, const REAL *feat, const REAL *mask,
                                     int FS, int H, int W, int L, const int32_t *level_offsets, int P)
{
  /* core/mapping/mapper.cpp:1384-1426; kernel [1 2 1;2 4 2;1 2 1]/16 (:30-37),
   * stride 2, zero pad 1; mask pyramid by nearest interpolate to (h/2, w/2)
   * (mapping_utils.cpp:321-342; torch 'nearest' picks source index 2*i). */
  static const REAL G[3][3] = {{1, 2, 1}, {2, 4, 2}, {1, 2, 1}};
  REAL *cur = (REAL *)malloc((size_t)FS * H * W * sizeof(REAL));
  REAL *curm = (REAL *)malloc((size_t)H * W * sizeof(REAL));
  memcpy(cur, feat, (size_t)FS * H * W * sizeof(REAL));
  memcpy(curm, mask, (size_t)H * W * sizeof(REAL));
  int h = H, w = W;
  for (int l = 0; l < L; ++l)
  {
    const size_t hw = (size_t)h * w;
    REAL *g = (REAL *)malloc(2 * (size_t)FS * hw * sizeof(REAL));
    ORC(spatial_grad)(g, cur, FS, h, w);
    for (int c = 0; c < FS; ++c)
    {
      memcpy(pyr + (size_t)c * P + level_offsets[l], cur + (size_t)c * hw, hw * sizeof(REAL));
      memcpy(grad + ((size_t)0 * FS + c) * P + level_offsets[l], g + (size_t)c * hw, hw * sizeof(REAL));
      memcpy(grad + ((size_t)1 * FS + c) * P + level_offsets[l], g + ((size_t)FS + c) * hw, hw * sizeof(REAL));
    }
    free(g);
    if (l == L - 1)
      break;
    const int nh = h / 2, nw = w / 2;
    REAL *nxt = (REAL *)malloc((size_t)FS * nh * nw * sizeof(REAL));
    REAL *nxtm = (REAL *)malloc((size_t)nh * nw * sizeof(REAL));
    for (int y = 0; y < nh; ++y)
      for (int x = 0; x < nw; ++x)
      {
        REAL rm = 0;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx)
          {
            const int yy = 2 * y + dy, xx = 2 * x + dx;
            if (within(xx, yy, w, h))
              rm += (G[dy + 1][dx + 1] / R_(16.0)) * curm[(size_t)yy * w + xx];
          }
        for (int c = 0; c < FS; ++c)
        {
          REAL rf = 0;
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
            {
              const int yy = 2 * y + dy, xx = 2 * x + dx;
              if (within(xx, yy, w, h))
                rf += (G[dy + 1][dx + 1] / R_(16.0)) *
                      (cur[(size_t)c * hw + (size_t)yy * w + xx] * curm[(size_t)yy * w + xx]);
            }
          nxt[((size_t)c * nh + y) * nw + x] = rf / (rm + R_(1.0e-8));
        }
        nxtm[(size_t)y * nw + x] = curm[(size_t)(2 * y) * w + 2 * x];
      }
    free(cur);
    free(curm);
    cur = nxt;
    curm = nxtm;
    h = nh;
    w = nw;
  }
  free(cur);
  free(curm);
}

/* =========================================================================
 * se3_exp: core/mapping/mapping_utils.h:316-346
 * ========================================================================= */
void ORC(se3_exp)(const REAL *omega, const REAL *v, REAL *R, REAL *t)
{
  REAL theta = (REAL)sqrt((double)(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]));
  REAL n[3] = {1, 0, 0};
  if (theta > 0)
  {
    n[0] = omega[0] / theta;
    n[1] = omega[1] / theta;
    n[2] = omega[2] / theta;
  }
  if (theta < R_(1.0e-14))
    theta = R_(1.0e-14);
  const REAL s = (REAL)sin((double)theta), c = (REAL)cos((double)theta);
  const REAL K[3][3] = {{0, -n[2], n[1]}, {n[2], 0, -n[0]}, {-n[1], n[0], 0}};
  REAL K2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
  const REAL a = (1 - c) / theta, b = (theta - s) / theta;
  for (int i = 0; i < 3; ++i)
  {
    REAL acc = 0;
    for (int j = 0; j < 3; ++j)
    {
      const REAL id = (i == j) ? R_(1.0) : R_(0.0);
      R[i * 3 + j] = id + s * K[i][j] + (1 - c) * K2[i][j];
      const REAL V = id + a * K[i][j] + b * K2[i][j];
      acc += V * v[j];
    }
    t[i] = acc;
  }
}

/* ------------------------------------------------------------------------------------------------
 * f1: valid locations (mapping_utils.h:254-287) and the seeded shuffle (mapper.cpp:1326-1333)
 * ------------------------------------------------------------------------------------------------ */
int ORC(valid_locations)(long long *loc1d, REAL *homo, const REAL *mask, const ORC(cam_t) * cam)
{
  const int W = (int)cam->w, H = (int)cam->h;
  int n = 0;
  for (int i = 0; i < W * H; ++i)
    if (mask[i] > (REAL)0.5) /* :267 */
    {
      const REAL x = (REAL)(i % W), y = (REAL)(i / W); /* fmod / floor of the flat index (:269-270) */
      loc1d[n] = i;
      homo[3 * n + 0] = (x - cam->cx) / cam->fx; /* :279-280 */
      homo[3 * n + 1] = (y - cam->cy) / cam->fy;
      homo[3 * n + 2] = (REAL)1;
      ++n;
    }
  return n;
}

typedef struct
{
  uint32_t mt[624];
  int pos;
} OrcMt19937;

static void orc_mt_seed(OrcMt19937 *g, uint32_t s)
{
  g->mt[0] = s;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->pos = 624;
}

static uint32_t orc_mt_next(OrcMt19937 *g)
{
  if (g->pos >= 624)
  {
    for (int i = 0; i < 624; ++i)
    {
      const uint32_t yv = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (yv >> 1) ^ ((yv & 1u) ? 0x9908b0dfu : 0u);
    }
    g->pos = 0;
  }
  uint32_t yv = g->mt[g->pos++];
  yv ^= yv >> 11;
  yv ^= (yv << 7) & 0x9d2c5680u;
  yv ^= (yv << 15) & 0xefc60000u;
  yv ^= yv >> 18;
  return yv;
}

/* uniform integer in [0, b] from a 32-bit engine: libstdc++ >= 11 uses Lemire's nearly divisionless method */
static uint64_t orc_uniform(OrcMt19937 *g, uint64_t b)
{
  if (b >= 0xffffffffull)
    return orc_mt_next(g); /* (ranges beyond 32 bits are never requested by the shuffle of < 2^32 elements) */
  const uint32_t range = (uint32_t)(b + 1);
  uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range)
  {
    const uint32_t threshold = (uint32_t)(0u - range) % range;
    while (low < threshold)
    {
      product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return product >> 32;
}

void ORC(shuffle_indices)(long long *idx, long long n, long long seed)
{
  for (long long i = 0; i < n; ++i)
    idx[i] = i;
  if (n <= 0)
    return;
  OrcMt19937 g;
  orc_mt_seed(&g, (uint32_t)((uint64_t)seed & 0xffffffffull)); /* mersenne_twister_engine::seed reduces mod 2^32 */
  const uint64_t urng = 0xffffffffull, un = (uint64_t)n;
#define ORC_SWAP(a, b)            \
  {                               \
    const long long t_ = idx[a];  \
    idx[a] = idx[b];              \
    idx[b] = t_;                  \
  }
  if (urng / un >= un)
  {
    long long i = 1;
    if ((un % 2) == 0)
    {
      const uint64_t pp = orc_uniform(&g, 1);
      ORC_SWAP(i, (long long)pp)
      ++i;
    }
    while (i != n)
    {
      const uint64_t sr = (uint64_t)i + 1; /* two positions from one draw: x in [0, sr*(sr+1)) */
      const uint64_t x = orc_uniform(&g, sr * (sr + 1) - 1);
      const uint64_t p1 = x / (sr + 1), p2 = x % (sr + 1);
      ORC_SWAP(i, (long long)p1)
      ++i;
      ORC_SWAP(i, (long long)p2)
      ++i;
    }
    return;
  }
  for (long long i = 1; i < n; ++i)
  {
    const uint64_t pp = orc_uniform(&g, (uint64_t)i);
    ORC_SWAP(i, (long long)pp)
  }
#undef ORC_SWAP
}

/* ------------------------------------------------------------------------------------------------
 * f3: reprojection factor (fair loss), cuda/reprojection_factor_kernels.cpp
 * ------------------------------------------------------------------------------------------------ */
/* shared per-keypoint part: projection, normalised differences, error, sqrt fair weights (:76-103 / :321-343) */
static int reproj_point(const REAL X[3], const REAL *match, const ORC(cam_t) * cam, REAL eps, REAL loss_param,
                        REAL diff[2], REAL sw[2], REAL *err)
{
  const int pos = X[2] > eps;
  const REAL px = (X[0] / X[2]) * cam->fx + cam->cx, py = (X[1] / X[2]) * cam->fy + cam->cy;
  const REAL sl = (REAL)sqrt((double)loss_param);
  diff[0] = match[0] - px;
  diff[1] = match[1] - py;
  const REAL nx = (REAL)fabs((double)diff[0]) / sl, ny = (REAL)fabs((double)diff[1]) / sl;
  sw[0] = pos ? (REAL)sqrt(1.0 / (double)(loss_param * (1 + nx))) : 0;
  sw[1] = pos ? (REAL)sqrt(1.0 / (double)(loss_param * (1 + ny))) : 0;
  *err = pos ? 2 * (nx + ny - (REAL)log(1.0 + (double)nx) - (REAL)log(1.0 + (double)ny)) : 0;
  return pos;
}

/* AtA = (weight/n) J^T J, Atb = (weight/n) J^T r over the 2N weighted rows; fallback error = 10*weight (:507-528) */
static void reproj_reduce(REAL *AtA, REAL *Atb, REAL *error, REAL *num_inliers, const REAL *J, const REAL *r,
                          const REAL *serr, const REAL *sval, int N, int D, REAL weight)
{
  double n_in = 0, se = 0;
  for (int i = 0; i < N; ++i)
  {
    n_in += sval[i];
    se += serr[i];
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  if (n_in > 0)
  {
    const double sc = (double)weight / n_in;
    *error = (REAL)(sc * se);
    for (int a = 0; a < D; ++a)
    {
      for (int b = 0; b < D; ++b)
      {
        double acc = 0;
        for (int k = 0; k < 2 * N; ++k)
          acc += (double)J[(size_t)k * D + a] * (double)J[(size_t)k * D + b];
        AtA[a * D + b] = (REAL)(sc * acc);
      }
      double accb = 0;
      for (int k = 0; k < 2 * N; ++k)
        accb += (double)J[(size_t)k * D + a] * (double)r[k];
      Atb[a] = (REAL)(sc * accb);
    }
  }
  else
  {
    *error = weight * 10;
    memset(AtA, 0, sizeof(REAL) * D * D);
    memset(Atb, 0, sizeof(REAL) * D);
  }
}

void ORC(reproj_jac_error)(REAL *AtA, REAL *Atb, REAL *error, REAL *num_inliers,
                           const REAL *R10, const REAL *t10, const REAL *R0, const REAL *t0, const REAL *R1, const REAL *t1,
                           const REAL *bias0, const REAL *basis0, const REAL *code0, const int32_t *loc1d, const REAL *homo,
                           const REAL *matched, REAL scale0, const ORC(cam_t) * cam, int N, int CS, REAL eps,
                           REAL loss_param, REAL weight, REAL *J_out, REAL *r_out, REAL *sw_out)
{
  const int D = 13 + CS;
  REAL *J = J_out ? J_out : (REAL *)malloc((size_t)(N > 0 ? N : 1) * 2 * D * sizeof(REAL));
  REAL *r = r_out ? r_out : (REAL *)malloc((size_t)(N > 0 ? N : 1) * 2 * sizeof(REAL));
  REAL *serr = (REAL *)malloc((size_t)(N > 0 ? N : 1) * sizeof(REAL));
  REAL *sval = (REAL *)malloc((size_t)(N > 0 ? N : 1) * sizeof(REAL));
  const REAL fx = cam->fx, fy = cam->fy;
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *hm = homo + (size_t)idx * 3;
    const long long i1d = (long long)loc1d[idx];
    const REAL d0 = sampled_depth(bias0, basis0, code0, i1d, CS, scale0); /* :57-64 */
    REAL rh[3], X[3];
    for (int i = 0; i < 3; ++i)
      rh[i] = R10[i * 3 + 0] * hm[0] + R10[i * 3 + 1] * hm[1] + R10[i * 3 + 2] * hm[2];
    for (int i = 0; i < 3; ++i)
      X[i] = d0 * rh[i] + t10[i];
    REAL diff[2], sw[2], e;
    const int pos = reproj_point(X, matched + (size_t)idx * 2, cam, eps, loss_param, diff, sw, &e);
    serr[idx] = e;
    sval[idx] = pos ? 1 : 0;
    const REAL inv_z = 1 / X[2];
    const REAL x_z = inv_z * X[0], y_z = inv_z * X[1];
    const REAL Jpi[2][3] = {{fx * inv_z, 0, -fx * x_z * inv_z}, {0, fy * inv_z, -fy * y_z * inv_z}}; /* :108-109 */
    REAL Xw[3];
    for (int i = 0; i < 3; ++i)
      Xw[i] = d0 * (R0[i * 3 + 0] * hm[0] + R0[i * 3 + 1] * hm[1] + R0[i * 3 + 2] * hm[2]) + t0[i];
    REAL dX1[3][6], dX0[3][6];
    for (int i = 0; i < 3; ++i) /* :124-133 */
    {
      dX1[i][0] = -R1[0 * 3 + i];
      dX1[i][1] = -R1[1 * 3 + i];
      dX1[i][2] = -R1[2 * 3 + i];
      dX1[i][3] = R1[1 * 3 + i] * Xw[2] - R1[2 * 3 + i] * Xw[1];
      dX1[i][4] = -R1[0 * 3 + i] * Xw[2] + R1[2 * 3 + i] * Xw[0];
      dX1[i][5] = R1[0 * 3 + i] * Xw[1] - R1[1 * 3 + i] * Xw[0];
    }
    {
      const REAL E[3][6] = {{1, 0, 0, 0, Xw[2], -Xw[1]}, {0, 1, 0, -Xw[2], 0, Xw[0]}, {0, 0, 1, Xw[1], -Xw[0], 0}};
      for (int i = 0; i < 3; ++i) /* :148-161 */
        for (int j = 0; j < 6; ++j)
          dX0[i][j] = R1[0 * 3 + i] * E[0][j] + R1[1 * 3 + i] * E[1][j] + R1[2 * 3 + i] * E[2][j];
    }
    const REAL jd[2] = {fx * (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z), /* :175-176 */
                        fy * (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z)};
    for (int c = 0; c < 2; ++c)
    {
      REAL *row = J + ((size_t)idx * 2 + c) * D;
      for (int j = 0; j < 6; ++j)
      {
        const REAL p0 = Jpi[c][0] * dX0[0][j] + Jpi[c][1] * dX0[1][j] + Jpi[c][2] * dX0[2][j];
        const REAL p1 = Jpi[c][0] * dX1[0][j] + Jpi[c][1] * dX1[1][j] + Jpi[c][2] * dX1[2][j];
        row[j] = sw[c] * p0;
        row[6 + j] = sw[c] * p1;
      }
      for (int i = 0; i < CS; ++i)
        row[12 + i] = sw[c] * (jd[c] * scale0 * basis0[(size_t)i1d * CS + i]); /* :182-183 */
      row[12 + CS] = sw[c] * (jd[c] * d0 / scale0);                           /* :186 */
      r[(size_t)idx * 2 + c] = sw[c] * diff[c];                               /* :188-189 */
      if (sw_out)
        sw_out[(size_t)idx * 2 + c] = sw[c];
    }
  }
  reproj_reduce(AtA, Atb, error, num_inliers, J, r, serr, sval, N, D, weight);
  if (!J_out)
    free(J);
  if (!r_out)
    free(r);
  free(serr);
  free(sval);
}

REAL ORC(reproj_error)(const REAL *R10, const REAL *t10, const REAL *bias0, const REAL *basis0, const REAL *code0,
                       const int32_t *loc1d, const REAL *homo, const REAL *matched, REAL scale0, const ORC(cam_t) * cam,
                       int N, int CS, REAL eps, REAL loss_param, REAL weight, REAL *num_inliers)
{
  double se = 0, n_in = 0;
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *hm = homo + (size_t)idx * 3;
    const REAL d0 = sampled_depth(bias0, basis0, code0, (long long)loc1d[idx], CS, scale0);
    REAL X[3];
    for (int i = 0; i < 3; ++i)
      X[i] = d0 * (R10[i * 3 + 0] * hm[0] + R10[i * 3 + 1] * hm[1] + R10[i * 3 + 2] * hm[2]) + t10[i];
    REAL diff[2], sw[2], e;
    const int pos = reproj_point(X, matched + (size_t)idx * 2, cam, eps, loss_param, diff, sw, &e);
    se += e;
    n_in += pos ? 1 : 0;
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  return n_in > 0 ? (REAL)((double)weight * se / n_in) : weight * 10; /* :457-464 */
}

void ORC(tracker_reproj_jac_error)(REAL *AtA, REAL *Atb, REAL *error, REAL *num_inliers, const REAL *R, const REAL *t,
                                   const REAL *dpts0, const REAL *homo, const REAL *matched, const ORC(cam_t) * cam, int N,
                                   REAL eps, REAL loss_param, REAL weight)
{
  const int D = 6;
  REAL *J = (REAL *)malloc((size_t)(N > 0 ? N : 1) * 2 * D * sizeof(REAL));
  REAL *r = (REAL *)malloc((size_t)(N > 0 ? N : 1) * 2 * sizeof(REAL));
  REAL *serr = (REAL *)malloc((size_t)(N > 0 ? N : 1) * sizeof(REAL));
  REAL *sval = (REAL *)malloc((size_t)(N > 0 ? N : 1) * sizeof(REAL));
  const REAL fx = cam->fx, fy = cam->fy;
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *hm = homo + (size_t)idx * 3;
    REAL X[3];
    for (int i = 0; i < 3; ++i)
      X[i] = dpts0[idx] * (R[i * 3 + 0] * hm[0] + R[i * 3 + 1] * hm[1] + R[i * 3 + 2] * hm[2]) + t[i];
    REAL diff[2], sw[2], e;
    const int pos = reproj_point(X, matched + (size_t)idx * 2, cam, eps, loss_param, diff, sw, &e);
    serr[idx] = e;
    sval[idx] = pos ? 1 : 0;
    const REAL inv_z = 1 / X[2];
    const REAL x_z = inv_z * X[0], y_z = inv_z * X[1];
    const REAL Jr[2][6] = {{fx * inv_z, 0, -fx * x_z * inv_z, -fx * x_z * y_z, fx * (1 + x_z * x_z), -fx * y_z}, /* :348-349 */
                           {0, fy * inv_z, -fy * y_z * inv_z, -fy * (1 + y_z * y_z), fy * x_z * y_z, fy * x_z}};
    for (int c = 0; c < 2; ++c)
    {
      for (int j = 0; j < 6; ++j)
        J[((size_t)idx * 2 + c) * D + j] = sw[c] * Jr[c][j];
      r[(size_t)idx * 2 + c] = sw[c] * diff[c];
    }
  }
  reproj_reduce(AtA, Atb, error, num_inliers, J, r, serr, sval, N, D, weight);
  free(J);
  free(r);
  free(serr);
  free(sval);
}

REAL ORC(tracker_reproj_error)(const REAL *R, const REAL *t, const REAL *dpts0, const REAL *homo, const REAL *matched,
                               const ORC(cam_t) * cam, int N, REAL eps, REAL loss_param, REAL weight, REAL *num_inliers)
{
  double se = 0, n_in = 0;
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *hm = homo + (size_t)idx * 3;
    REAL X[3];
    for (int i = 0; i < 3; ++i)
      X[i] = dpts0[idx] * (R[i * 3 + 0] * hm[0] + R[i * 3 + 1] * hm[1] + R[i * 3 + 2] * hm[2]) + t[i];
    REAL diff[2], sw[2], e;
    const int pos = reproj_point(X, matched + (size_t)idx * 2, cam, eps, loss_param, diff, sw, &e);
    se += e;
    n_in += pos ? 1 : 0;
  }
  if (num_inliers)
    *num_inliers = (REAL)n_in;
  return n_in > 0 ? (REAL)((double)weight * se / n_in) : weight * 10;
}

/* ------------------------------------------------------------------------------------------------
 * f3: match-geometry factors, cuda/match_geometry_factor_kernels.cpp
 * ------------------------------------------------------------------------------------------------ */
static int mg_dim(int mode, int CS) { return mode == 0 ? 14 + 2 * CS : (mode == 1 ? 14 : (mode == 2 ? 6 : 7)); }

/* depths of the two matched points and the (possibly rescaled) scale pair (:601-616, unbiased :429-447, loop :303-304) */
static void mg_depths(int mode, int loss, int idx, const REAL *bias0, const REAL *bias1, const REAL *basis0,
                      const REAL *basis1, const REAL *code0, const REAL *code1, const REAL *dpts0, const REAL *dpts1,
                      const int32_t *loc0, const int32_t *loc1, REAL scale0, REAL scale1, int CS, REAL *d0, REAL *d1)
{
  if (mode == 0)
  {
    REAL a0 = bias0[loc0[idx]], a1 = bias1[loc1[idx]];
    for (int i = 0; i < CS; ++i)
      a0 += basis0[(size_t)loc0[idx] * CS + i] * code0[i];
    for (int i = 0; i < CS; ++i)
      a1 += basis1[(size_t)loc1[idx] * CS + i] * code1[i];
    if (loss == 3)
    {
      const REAL ss = scale0 + scale1;
      *d0 = a0 * scale0 / ss;
      *d1 = a1 * scale1 / ss;
    }
    else
    {
      *d0 = a0 * scale0;
      *d1 = a1 * scale1;
    }
  }
  else if (mode == 1)
  {
    *d0 = dpts0[idx] * scale0;
    *d1 = dpts1[idx] * scale1;
  }
  else
  {
    *d0 = dpts0[idx];
    *d1 = dpts1[idx];
  }
}

/* per-component error and sqrt weights of the loss (fair :640-656, L2 :792-797, huber :935-962) */
static REAL mg_loss(int loss, const REAL diff[3], REAL loss_param, REAL sw[3])
{
  if (loss == 1)
  {
    sw[0] = sw[1] = sw[2] = 1;
    return diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
  }
  if (loss == 2)
  {
    REAL e = 0;
    for (int i = 0; i < 3; ++i)
    {
      const REAL sq = diff[i] * diff[i];
      e += (sq <= loss_param) ? sq : (REAL)(2.0 * sqrt((double)(loss_param * sq)) - (double)loss_param);
      const REAL q = (REAL)sqrt((double)(loss_param / sq));
      sw[i] = q < 1 ? q : 1; /* min(1.0f, sqrt(c / sq)); sq == 0 -> inf -> 1 */
    }
    return e;
  }
  const REAL sl = (REAL)sqrt((double)loss_param);
  REAL e = 0;
  for (int i = 0; i < 3; ++i)
  {
    const REAL n = (REAL)fabs((double)diff[i]) / sl;
    e += n - (REAL)log(1.0 + (double)n);
    sw[i] = (REAL)sqrt(1.0 / (double)(loss_param * (1 + n)));
  }
  return 2 * e;
}

void ORC(match_geom_jac_error)(REAL *AtA, REAL *Atb, REAL *error, int mode, int loss,
                               const REAL *R10, const REAL *t10, const REAL *R0, const REAL *t0, const REAL *R1,
                               const REAL *t1, const REAL *bias0, const REAL *bias1, const REAL *basis0,
                               const REAL *basis1, const REAL *code0, const REAL *code1, const REAL *dpts0,
                               const REAL *dpts1, const REAL *homo0, const REAL *homo1, const int32_t *loc0,
                               const int32_t *loc1, REAL scale0, REAL scale1, int N, int CS, REAL loss_param,
                               REAL weight, REAL *J_out, REAL *r_out, REAL *sw_out)
{
  const int D = mg_dim(mode, CS);
  REAL *J = J_out ? J_out : (REAL *)malloc((size_t)N * 3 * D * sizeof(REAL));
  REAL *r = r_out ? r_out : (REAL *)malloc((size_t)N * 3 * sizeof(REAL));
  double se = 0;
  const REAL ss = scale0 + scale1;
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *h0 = homo0 + (size_t)idx * 3, *h1 = homo1 + (size_t)idx * 3;
    REAL d0, d1;
    mg_depths(mode, loss, idx, bias0, bias1, basis0, basis1, code0, code1, dpts0, dpts1, loc0, loc1, scale0, scale1, CS,
              &d0, &d1);
    REAL rh[3], X[3], diff[3], sw[3];
    for (int i = 0; i < 3; ++i)
      rh[i] = R10[i * 3 + 0] * h0[0] + R10[i * 3 + 1] * h0[1] + R10[i * 3 + 2] * h0[2];
    for (int i = 0; i < 3; ++i)
    {
      X[i] = d0 * rh[i] + t10[i];
      diff[i] = d1 * h1[i] - X[i];
    }
    se += mg_loss(loss, diff, loss_param, sw);
    REAL dX0[3][6];
    if (mode <= 1)
    {
      REAL Xw[3];
      for (int i = 0; i < 3; ++i)
        Xw[i] = d0 * (R0[i * 3 + 0] * h0[0] + R0[i * 3 + 1] * h0[1] + R0[i * 3 + 2] * h0[2]) + t0[i];
      const REAL E[3][6] = {{1, 0, 0, 0, Xw[2], -Xw[1]}, {0, 1, 0, -Xw[2], 0, Xw[0]}, {0, 0, 1, Xw[1], -Xw[0], 0}};
      for (int i = 0; i < 3; ++i) /* :694-705 */
        for (int j = 0; j < 6; ++j)
          dX0[i][j] = R1[0 * 3 + i] * E[0][j] + R1[1 * 3 + i] * E[1][j] + R1[2 * 3 + i] * E[2][j];
    }
    else
    {
      const REAL E[3][6] = {{1, 0, 0, 0, X[2], -X[1]}, {0, 1, 0, -X[2], 0, X[0]}, {0, 0, 1, X[1], -X[0], 0}}; /* :197-199 */
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j)
          dX0[i][j] = E[i][j];
    }
    for (int i = 0; i < 3; ++i)
    {
      REAL *row = J + ((size_t)idx * 3 + i) * D;
      for (int j = 0; j < 6; ++j)
        row[j] = sw[i] * dX0[i][j];
      if (mode <= 1)
      {
        /* pose 1: -R1^T E(Xw), written out in the reference (:673-683) */
        for (int j = 0; j < 6; ++j)
          row[6 + j] = sw[i] * (-dX0[i][j]);
      }
      if (mode == 0)
      {
        const REAL *b0 = basis0 + (size_t)loc0[idx] * CS, *b1 = basis1 + (size_t)loc1[idx] * CS;
        if (loss == 3)
        {
          for (int j = 0; j < CS; ++j)
          {
            row[12 + j] = sw[i] * (rh[i] * b0[j] * scale0 / ss);       /* :560-563 */
            row[12 + CS + j] = sw[i] * (-h1[i] * b1[j] * scale1 / ss);
          }
          row[12 + 2 * CS] = sw[i] * (rh[i] * d0 * scale1 / (scale0 * ss) + h1[i] * d1 / ss);  /* :566-569 */
          row[13 + 2 * CS] = sw[i] * (-rh[i] * d0 / ss - h1[i] * d1 * scale0 / (scale1 * ss));
        }
        else
        {
          for (int j = 0; j < CS; ++j)
          {
            row[12 + j] = sw[i] * (rh[i] * scale0 * b0[j]);            /* :716-719 */
            row[12 + CS + j] = sw[i] * (-h1[i] * scale1 * b1[j]);
          }
          row[12 + 2 * CS] = sw[i] * (rh[i] * d0 / scale0);            /* :722-723 */
          row[13 + 2 * CS] = sw[i] * (-h1[i] * d1 / scale1);
        }
      }
      else if (mode == 1)
      {
        row[12] = sw[i] * (rh[i] * dpts0[idx]);                        /* :398-399 */
        row[13] = sw[i] * (-h1[i] * dpts1[idx]);
      }
      else if (mode == 3)
        row[6] = sw[i] * (rh[i] * dpts0[idx] / scale0);                /* :278 */
      r[(size_t)idx * 3 + i] = sw[i] * diff[i];
      if (sw_out)
        sw_out[(size_t)idx * 3 + i] = sw[i];
    }
  }
  const double sc = (double)weight / (double)N;
  *error = (REAL)(sc * se);
  for (int a = 0; a < D; ++a)
  {
    for (int b = 0; b < D; ++b)
    {
      double acc = 0;
      for (int k = 0; k < 3 * N; ++k)
        acc += (double)J[(size_t)k * D + a] * (double)J[(size_t)k * D + b];
      AtA[a * D + b] = (REAL)(sc * acc);
    }
    double accb = 0;
    for (int k = 0; k < 3 * N; ++k)
      accb += (double)J[(size_t)k * D + a] * (double)r[k];
    Atb[a] = (REAL)(sc * accb);
  }
  if (!J_out)
    free(J);
  if (!r_out)
    free(r);
}

REAL ORC(match_geom_error)(int mode, int loss, const REAL *R10, const REAL *t10, const REAL *bias0, const REAL *bias1,
                           const REAL *basis0, const REAL *basis1, const REAL *code0, const REAL *code1,
                           const REAL *dpts0, const REAL *dpts1, const REAL *homo0, const REAL *homo1,
                           const int32_t *loc0, const int32_t *loc1, REAL scale0, REAL scale1, int N, int CS,
                           REAL loss_param, REAL weight)
{
  double se = 0;
  for (int idx = 0; idx < N; ++idx)
  {
    const REAL *h0 = homo0 + (size_t)idx * 3, *h1 = homo1 + (size_t)idx * 3;
    REAL d0, d1;
    mg_depths(mode, loss, idx, bias0, bias1, basis0, basis1, code0, code1, dpts0, dpts1, loc0, loc1, scale0, scale1, CS,
              &d0, &d1);
    REAL diff[3], sw[3];
    for (int i = 0; i < 3; ++i)
    {
      const REAL X = d0 * (R10[i * 3 + 0] * h0[0] + R10[i * 3 + 1] * h0[1] + R10[i * 3 + 2] * h0[2]) + t10[i];
      diff[i] = d1 * h1[i] - X;
    }
    se += mg_loss(loss, diff, loss_param, sw);
  }
  return (REAL)((double)weight * se / (double)N);
}

/* ------------------------------------------------------------------------------------------------
 * f4: cycle-consistent descriptor matching (match_geometry_factor.cpp:62-97, camera_tracker.cpp:608-633)
 * ------------------------------------------------------------------------------------------------ */
static long long best_match(const REAL *query_desc, long long query_loc, const REAL *target, int C, long long HW)
{
  long long best = 0;
  REAL best_resp = 0;
  for (long long p = 0; p < HW; ++p)
  {
    REAL acc = 0;
    for (int c = 0; c < C; ++c)
    {
      const REAL d = query_desc[(size_t)c * HW + query_loc] - target[(size_t)c * HW + p];
      acc += d * d;
    }
    const REAL resp = -acc;
    if (p == 0 || resp > best_resp)
    {
      best_resp = resp;
      best = p;
    }
  }
  return best;
}

int ORC(cycle_match)(long long *raw_matched1, long long *cyc_matched0, int32_t *inlier, const REAL *desc0,
                     const REAL *desc1, const long long *kp_loc0, int K, int C, int H, int W, REAL cyc_thresh)
{
  const long long HW = (long long)H * W;
  int n = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) reduction(+ : n)
#endif
  for (int k = 0; k < K; ++k)
  {
    const long long m1 = best_match(desc0, kp_loc0[k], desc1, C, HW);
    const long long c0 = best_match(desc1, m1, desc0, C, HW);
    raw_matched1[k] = m1;
    cyc_matched0[k] = c0;
    const REAL dx = (REAL)(kp_loc0[k] % W) - (REAL)(c0 % W), dy = (REAL)(kp_loc0[k] / W) - (REAL)(c0 / W);
    inlier[k] = (dx * dx + dy * dy) <= cyc_thresh * cyc_thresh;
    n += inlier[k];
  }
  return n;
}
