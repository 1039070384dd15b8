// finalize_bodies.h -- per-edge finalize of the two dense factor types as device functions: the stand-alone finalize
// kernels (photo_kernels.hip / geo_kernels.hip: drop-in operators, factor cache) and the window's one-launch
// finalize + assembly (window.hip) share them.  One workgroup per edge; `s` is the workgroup's LDS scratch.
#pragma once
#include "sage_internal.h"

namespace sage
{

__device__ __forceinline__ int sidx6(int i, int j) // upper-triangular index, i <= j < 6
{
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}
__device__ __forceinline__ int gsidx6(int i, int j) { return sidx6(i, j); }

__host__ __device__ constexpr int photo_finalize_lds_doubles(int CS) { return kPhotoScalars + photo_tiles(CS) * 256; }
__host__ __device__ constexpr int geo_finalize_lds_doubles(int CS) { return geo_partial_floats(CS) + CS; } // + row 8 of a merged launch

// ------------------------------------------------------------------------------------------------
// photometric finalize: sum the workgroup partials of an edge in a fixed order (deterministic), expand the reduced
// blocks into the reference layout [pose0 pose1 code0 scale0] (photometric_factor_kernels.cpp:350-364),
// apply 1/num_inliers and the zero-overlap fallback (:1139-1161).
// ------------------------------------------------------------------------------------------------
struct PhotoFinalizeParams
{
  PhotoEdge single;
  const PhotoEdge *table;
  const int32_t *edge_first;
  const int32_t *edge_tiles;
  const float *partials;
  float *AtA, *Atb, *stats;
  float wsum;
  int edge_base; // blockIdx.x = edge - edge_base
  double *wide;  // optional [n_edges][D*D + D]: the results before rounding to fp32
};

__device__ __forceinline__ double tile_elem(const double *s, int base, int tile, int row, int col)
{
  return s[base + tile * 256 + (row & 3) * 64 + ((row >> 2) * 16 + col)];
}

template <int CS>
__device__ __forceinline__ void photo_finalize_body(const PhotoFinalizeParams &prm, const int e, double *s)
{
  constexpr int PP = kPhotoScalars + photo_tiles(CS) * 256; // entries of a summed record
  constexpr int D = 13 + CS;
  // s[PP] (LDS): partial sums and every derived product stay in double until the single final rounding
  const int tid = threadIdx.x;
  const PhotoEdge &E = prm.table ? prm.table[e] : prm.single;
  const float s0 = E.scale0 ? *E.scale0 : E.scale0_val;
  const int first = prm.edge_first[e], nt = prm.edge_tiles[e];
  // s[] keeps the historic index space: [0..39] scalars, then NT tiles of 256; the scalars and the cross tiles come from
  // the double part of the records, the code-code tiles from the fp32 part
  constexpr int NCCF = photo_cc_tiles(CS), DOFF = photo_partial_double_offset(CS), PF = photo_partial_floats(CS);
  for (int idx = tid; idx < PP; idx += (int)blockDim.x)
  {
    const bool dbl = idx < kPhotoScalars || idx >= kPhotoScalars + NCCF * 256;
    double a = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0; // four independent chains: the loads of a round are in flight together
    int t = 0;
    if (dbl)
    {
      const int di = idx < kPhotoScalars ? idx : idx - NCCF * 256;
      const double *pp = reinterpret_cast<const double *>(prm.partials + (size_t)first * PF + DOFF) + di;
      constexpr size_t STR = PF / 2; // record stride in doubles
      for (; t + 4 <= nt; t += 4)
      {
        a += pp[(size_t)t * STR]; a1 += pp[(size_t)(t + 1) * STR]; a2 += pp[(size_t)(t + 2) * STR]; a3 += pp[(size_t)(t + 3) * STR];
      }
      for (; t < nt; ++t)
        a += pp[(size_t)t * STR];
    }
    else
    {
      const float *pp = prm.partials + (size_t)first * PF + idx;
      for (; t + 4 <= nt; t += 4)
      {
        const float v0 = pp[(size_t)t * PF], v1 = pp[(size_t)(t + 1) * PF], v2 = pp[(size_t)(t + 2) * PF], v3 = pp[(size_t)(t + 3) * PF];
        a += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
      }
      for (; t < nt; ++t)
        a += (double)pp[(size_t)t * PF];
    }
    s[idx] = (a + a1) + (a2 + a3);
  }
  __syncthreads();
  const double s0d = (double)s0;
  const double n_in = s[36];
  const bool ok = n_in > 0.0;
  const double inv_n = ok ? 1.0 / n_in : 0.0;
  float *AtA = prm.AtA + (size_t)e * D * D;
  float *Atb = prm.Atb + (size_t)e * D;
  if (tid == 0)
  {
    prm.stats[2 * e + 0] = ok ? (float)(s[35] * inv_n) : 10.0f * prm.wsum;
    prm.stats[2 * e + 1] = (float)n_in;
  }
  // CS = 32: the contraction loads channel pairs per lane -> operand block = channel parity, row = channel / 2
  auto X = [&](int row, int col) -> double { // sum_n a_n[row] * b_n[col]
    if (CS == 32)
      return tile_elem(s, kPhotoScalars, (col & 1) ? 4 : 3, row, col >> 1);
    return tile_elem(s, kPhotoScalars, 1, row, col);
  };
  auto CC = [&](int i, int j) -> double { // sum_n sigma_n b_n[i] b_n[j], i <= j
    if (CS == 32)
    {
      const int ti = i & 1, tj = j & 1;
      if (ti <= tj)
        return tile_elem(s, kPhotoScalars, ti + tj, i >> 1, j >> 1); // (0,0)->0 (0,1)->1 (1,1)->2
      return tile_elem(s, kPhotoScalars, 1, j >> 1, i >> 1);
    }
    return tile_elem(s, kPhotoScalars, 0, i, j);
  };
  for (int idx = tid; idx < D * D + D; idx += (int)blockDim.x)
  {
    double val = 0.0;
    if (ok)
    {
      if (idx < D * D)
      {
        int i = idx / D, j = idx % D;
        if (i > j)
        {
          const int t = i;
          i = j;
          j = t;
        }
        // classes: pose (0..11), code (12..12+CS-1), scale (12+CS)
        if (j < 12)
        {
          const double sg = ((i >= 6) != (j >= 6)) ? -1.0 : 1.0;
          const int a = i % 6, b = j % 6;
          val = sg * s[a <= b ? sidx6(a, b) : sidx6(b, a)];
        }
        else if (i < 12)
        {
          const double sg = (i >= 6) ? -1.0 : 1.0;
          if (j < 12 + CS)
            val = sg * s0d * X(i % 6, j - 12);
          else
            val = sg * s[21 + i % 6] / s0d;
        }
        else if (i < 12 + CS)
        {
          if (j < 12 + CS)
            val = s0d * s0d * CC(i - 12, j - 12);
          else
            val = X(6, i - 12);
        }
        else
          val = s[27] / (s0d * s0d);
        val *= inv_n;
      }
      else
      {
        const int i = idx - D * D;
        if (i < 12)
          val = ((i >= 6) ? -1.0 : 1.0) * s[28 + i % 6];
        else if (i < 12 + CS)
          val = s0d * X(7, i - 12);
        else
          val = s[34] / s0d;
        val *= inv_n;
      }
    }
    if (idx < D * D)
      AtA[idx] = (float)val;
    else
      Atb[idx - D * D] = (float)val;
    if (prm.wide)
      prm.wide[(size_t)e * (D * D + D) + idx] = val;
  }
}

// ------------------------------------------------------------------------------------------------
// geometric finalize (geometric_factor_kernels.cpp:868-944: 1/num_inliers, weight, zero-overlap fallback)
// ------------------------------------------------------------------------------------------------
struct GeoFinalizeParams
{
  GeoEdge single;
  const GeoEdge *table;
  const int32_t *edge_first;
  const int32_t *edge_tiles;
  const float *partials;
  float *AtA, *Atb, *stats;
  float weight;
  int edge_base; // blockIdx.x = edge - edge_base
  double *wide;  // optional [n_edges][D*D + D]: the results before rounding to fp32
  // merged linearize (LaunchCommon::merge_geo_weight): the photometric launch's partial records of the same edges -- row 8
  // of their cross tiles is  sum_n w_g omega kappa D b_n  (the scale1-code0 block; every other code0 block of the
  // geometric edge rides in the photometric edge's result and reads zero here)
  const float *photo_partials;
  const int32_t *photo_rec_first, *photo_rec_count;
};

template <int CS>
__device__ __forceinline__ void geo_finalize_body(const GeoFinalizeParams &prm, const int e, double *s)
{
  constexpr int PP = geo_partial_floats(CS);
  constexpr int D = 14 + 2 * CS;
  constexpr int N16 = geo_n16(CS);
  constexpr int NTT = N16 * (N16 + 1) / 2;
  // s[PP] (LDS): partial sums and every derived product stay in double until the single final rounding
  const int tid = threadIdx.x;
  const GeoEdge &E = prm.table ? prm.table[e] : prm.single;
  const float s0 = E.scale0 ? *E.scale0 : E.scale0_val;
  const float s1 = E.scale1 ? *E.scale1 : E.scale1_val;
  const int first = prm.edge_first[e], nt = prm.edge_tiles[e];
  for (int idx = tid; idx < PP + (prm.photo_partials ? CS : 0); idx += (int)blockDim.x)
  {
    double a = 0.0; // the per-workgroup partials are summed in double: free (a few dozen adds), and it keeps the
                    // engine's accumulation noise below the reference's own fp32 floor
    double a1 = 0.0, a2 = 0.0, a3 = 0.0; // four independent chains: the loads of a round are in flight together
    if (idx >= PP)
    {
      // merged launch: s[PP + c] = sum over the photometric records of this edge of  sum_n w_g omega kappa D b_n[c]  (row 8
      // of the cross tile of channel c; doubles), in the records' fixed order
      constexpr int PF = photo_partial_floats(CS), DOFF = photo_partial_double_offset(CS);
      const int c = idx - PP;
      const int xt = CS == 32 ? (c & 1) : 0, xc = CS == 32 ? (c >> 1) : c; // cross tile, column inside it
      const int di = kPhotoScalars + xt * 256 + 32 + xc;                   // row 8 -> r = 0, 16-block 2
      const int pn = prm.photo_rec_count[e];
      const double *pp = reinterpret_cast<const double *>(prm.photo_partials + (size_t)prm.photo_rec_first[e] * PF + DOFF) + di;
      constexpr size_t STR = PF / 2;
      int t = 0;
      for (; t + 4 <= pn; t += 4)
      {
        a += pp[(size_t)t * STR]; a1 += pp[(size_t)(t + 1) * STR]; a2 += pp[(size_t)(t + 2) * STR]; a3 += pp[(size_t)(t + 3) * STR];
      }
      for (; t < pn; ++t)
        a += pp[(size_t)t * STR];
      s[idx] = (a + a1) + (a2 + a3);
      continue;
    }
    if (prm.photo_partials && idx >= kGeoScalars)
    {
      // merged launch: the t0 t0^T and y t0^T tiles were not contracted here (exact zeros in the records): not read
      constexpr int NBm = CS / 16;
      const int tile = (idx - kGeoScalars) >> 8;
      bool skipped = tile >= NTT && tile < NTT + NBm; // y t0^T
      for (int bi = 0; bi < NBm; ++bi)
        for (int bj = bi; bj < NBm; ++bj)
          skipped = skipped || tile == bi * N16 - (bi * (bi - 1)) / 2 + (bj - bi);
      if (skipped)
      {
        s[idx] = 0.0;
        continue;
      }
    }
    const float *pp = prm.partials + (size_t)first * PP + idx;
    int t = 0;
    for (; t + 4 <= nt; t += 4)
    {
      const float v0 = pp[(size_t)t * PP], v1 = pp[(size_t)(t + 1) * PP], v2 = pp[(size_t)(t + 2) * PP],
                  v3 = pp[(size_t)(t + 3) * PP];
      a += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; t < nt; ++t)
      a += (double)pp[(size_t)t * PP];
    s[idx] = (a + a1) + (a2 + a3);
  }
  __syncthreads();
  const double n_in = s[45];
  const bool ok = n_in > 0.0;
  const double wn = ok ? (double)prm.weight / n_in : 0.0;
  if (tid == 0)
  {
    prm.stats[2 * e + 0] = ok ? (float)(wn * s[44]) : 10.0f * prm.weight; // geometric_factor_kernels.cpp:934,944
    prm.stats[2 * e + 1] = (float)n_in;
  }
  auto telem = [&](int tile, int row, int col) -> double {
    return s[kGeoScalars + tile * 256 + (row & 3) * 64 + ((row >> 2) * 16 + col)];
  };
  // t index a in [0, 2CS): a < CS -> kappa*b0 channel a, else beta channel a-CS.  Operand block / row of a channel:
  // CS = 32 loads channel pairs per lane (block = parity, row = channel/2), CS = 16 one channel per lane.
  auto tblk = [&](int a) -> int { return CS == 32 ? (a / CS) * 2 + (a & 1) : a / CS; };
  auto trow = [&](int a) -> int { return CS == 32 ? (a % CS) >> 1 : a % CS; };
  auto TT = [&](int a, int b) -> double { // sum w t_a t_b
    int bi = tblk(a), bj = tblk(b), ra = trow(a), rb = trow(b);
    if (bi > bj || (bi == bj && ra > rb)) // always read the upper triangle: (w t_a) t_b != (w t_b) t_a in fp32
    {
      int t = bi; bi = bj; bj = t;
      t = ra; ra = rb; rb = t;
    }
    const int tile = bi * N16 - (bi * (bi - 1)) / 2 + (bj - bi);
    return telem(tile, ra, rb);
  };
  auto YT = [&](int r, int col) -> double { return telem(NTT + tblk(col), r, trow(col)); }; // sum w y_r t_col
  auto YY = [&](int a, int b) -> double { // sum w y_a y_b, a,b in 0..8 (never both 8)
    if (a > b)
    {
      const int t = a; a = b; b = t;
    }
    if (b < 6)
      return s[gsidx6(a, b)];
    if (b == 6)
      return a < 6 ? s[21 + a] : s[33];
    if (b == 7)
      return a < 6 ? s[27 + a] : (a == 6 ? s[34] : s[35]);
    return a < 6 ? s[36 + a] : (a == 6 ? s[42] : s[43]); // b == 8 (rho)
  };
  // column j -> (kind, index, coef): kind 0 = y entry, kind 1 = t entry
  auto column = [&](int j, int &kind, int &idx, double &coef) {
    if (j < 6) { kind = 0; idx = j; coef = 1.0; }
    else if (j < 12) { kind = 0; idx = j - 6; coef = -1.0; }
    else if (j < 12 + CS) { kind = 1; idx = j - 12; coef = (double)s0; }
    else if (j < 12 + 2 * CS) { kind = 1; idx = j - 12; coef = -(double)s1; }
    else if (j == 12 + 2 * CS) { kind = 0; idx = 6; coef = 1.0 / (double)s0; }
    else { kind = 0; idx = 7; coef = -1.0 / (double)s1; }
  };
  float *AtA = prm.AtA + (size_t)e * D * D;
  float *Atb = prm.Atb + (size_t)e * D;
  for (int q = tid; q < D * D + D; q += (int)blockDim.x)
  {
    double val = 0.0;
    if (ok)
    {
      if (q < D * D)
      {
        int ki, ii, kj, ij;
        double ci, cj;
        column(q / D, ki, ii, ci);
        column(q % D, kj, ij, cj);
        double mv;
        if (ki == 0 && kj == 0)
          mv = YY(ii, ij);
        else if (ki == 1 && kj == 1)
          mv = TT(ii, ij);
        else
          mv = ki == 0 ? YT(ii, ij) : YT(ij, ii);
        val = wn * (ci * cj) * mv; // (ci*cj) first: exactly symmetric in (i, j)
        if (prm.photo_partials && ki != kj)
        {
          // merged launch: (scale1, code0 channel c) = (1/n) (-1/s1) s0 * sum_n w_g omega kappa D b_n[c], summed over the
          // photometric records of this edge in their fixed order (row 8 of the cross tile of channel c)
          const int yrow = ki == 0 ? ii : ij, tcol = ki == 0 ? ij : ii;
          if (yrow == 7 && tcol < CS)
          {
            const double r8 = s[PP + tcol];
            val = (1.0 / n_in) * (ci * cj) * r8;
          }
        }
      }
      else
      {
        int k, ii;
        double c;
        column(q - D * D, k, ii, c);
        val = wn * c * (k == 0 ? YY(ii, 8) : YT(8, ii));
      }
    }
    if (q < D * D)
      AtA[q] = (float)val;
    else
      Atb[q - D * D] = (float)val;
    if (prm.wide)
      prm.wide[(size_t)e * (D * D + D) + q] = val;
  }
}

} // namespace sage
