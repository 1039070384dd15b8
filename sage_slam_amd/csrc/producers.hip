// producers.hip -- keyframe input producers feeding the hot path.
//   depth_and_grad : UpdateDepth (core/mapping/mapping_utils.h:215-222) + ComputeSpatialGrad (:236-252), the
//                    caller-side precompute of the geometric factor (core/gtsam/geometric_factor.cpp:317-347)
//   gaussian_pyramid_with_grad : Mapper::GenerateGaussianPyramidWithGrad (core/mapping/mapper.cpp:1384-1426)
// Plain HBM-bound image ops: one thread per output texel, coalesced along x.
#include "sage_device.h"
#include "sage_internal.h"

namespace sage
{

// dpt[i] = scale * (bias[i] + basis[i,:] . code) for a batch of keyframes (blockIdx.y = keyframe)
struct DepthBatch
{
  float *const *dpt;          // [K] -> [H*W]
  const float *const *bias;   // [K]
  const float *const *basis;  // [K]
  const float *const *code;   // [K] -> [CS]
  const float *const *scale;  // [K] -> 1
};

template <int CS>
__global__ __launch_bounds__(256) void depth_kernel(float *dpt, const float *__restrict__ bias,
                                                    const float *__restrict__ basis,
                                                    const float *__restrict__ code, const float *scale_dev,
                                                    float scale, int HW)
{
  // 8 (CS=32) or 4 (CS=16) lanes cooperate on one texel so the basis row is read as coalesced float4
  constexpr int F4 = CS / 4;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int px = gid / F4, c4 = gid % F4;
  const float s = scale_dev ? *scale_dev : scale;
  float part = 0.f;
  if (px < HW)
  {
    const f32x4 b = *reinterpret_cast<const f32x4 *>(basis + (size_t)px * CS + c4 * 4);
    part = b[0] * code[c4 * 4 + 0] + b[1] * code[c4 * 4 + 1] + b[2] * code[c4 * 4 + 2] + b[3] * code[c4 * 4 + 3];
  }
#pragma unroll
  for (int o = F4 / 2; o > 0; o >>= 1)
    part += __shfl_xor(part, o, 64);
  if (px < HW && c4 == 0)
    dpt[px] = s * (bias[px] + part);
}

// per-edge operator path with N << H*W samples: only the sampled pixels' depths, written at their map positions
// (dpt[loc[n]]), same arithmetic and lane cooperation as depth_kernel.  Locations outside [0, HW) are skipped.
template <int CS>
__global__ __launch_bounds__(256) void depth_samples_kernel(float *dpt, const float *__restrict__ bias,
                                                            const float *__restrict__ basis,
                                                            const float *__restrict__ code, float scale,
                                                            const void *__restrict__ loc, int loc_is_i64, int N, int HW)
{
  constexpr int F4 = CS / 4;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = gid / F4, c4 = gid % F4;
  long long px = -1;
  if (n < N)
    px = loc_is_i64 ? reinterpret_cast<const long long *>(loc)[n] : (long long)reinterpret_cast<const int *>(loc)[n];
  const bool ok = px >= 0 && px < HW;
  float part = 0.f;
  if (ok)
  {
    const f32x4 b = *reinterpret_cast<const f32x4 *>(basis + (size_t)px * CS + c4 * 4);
    part = b[0] * code[c4 * 4 + 0] + b[1] * code[c4 * 4 + 1] + b[2] * code[c4 * 4 + 2] + b[3] * code[c4 * 4 + 3];
  }
#pragma unroll
  for (int o = F4 / 2; o > 0; o >>= 1)
    part += __shfl_xor(part, o, 64);
  if (ok && c4 == 0)
    dpt[px] = scale * (bias[px] + part);
}

hipError_t launch_depth_samples(hipStream_t s, int CS, float *dpt, const float *bias, const float *basis,
                                const float *code, float scale, const void *loc, int loc_is_i64, int N, int HW)
{
  if (N <= 0)
    return hipSuccess;
  if (CS == 32)
    hipLaunchKernelGGL((depth_samples_kernel<32>), dim3((N * 8 + 255) / 256), dim3(256), 0, s, dpt, bias, basis, code,
                       scale, loc, loc_is_i64, N, HW);
  else if (CS == 16)
    hipLaunchKernelGGL((depth_samples_kernel<16>), dim3((N * 4 + 255) / 256), dim3(256), 0, s, dpt, bias, basis, code,
                       scale, loc, loc_is_i64, N, HW);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// batched variant over keyframes (blockIdx.y = keyframe): one launch per LM pass for the whole window
constexpr int kDepthUnroll = 4;
template <int CS>
__global__ __launch_bounds__(256) void depth_batch_kernel(const DepthItem *__restrict__ items, int HW)
{
  // F4 lanes share a pixel (one 16-byte piece of its basis row each); every thread walks kDepthUnroll pixels with all
  // its loads in flight together (one load per thread leaves the memory pipeline mostly idle: 4.3 TB/s)
  constexpr int F4 = CS / 4;
  constexpr int PX = 256 / F4; // pixels per workgroup and step
  const DepthItem it = items[blockIdx.y];
  const int c4 = threadIdx.x % F4, sub = threadIdx.x / F4;
  const int px0 = blockIdx.x * (PX * kDepthUnroll) + sub;
  const float cd[4] = {it.code[c4 * 4 + 0], it.code[c4 * 4 + 1], it.code[c4 * 4 + 2], it.code[c4 * 4 + 3]}; // (the code
                                                                     // lives at an odd float offset of the variable array)
  f32x4 b[kDepthUnroll];
#pragma unroll
  for (int u = 0; u < kDepthUnroll; ++u)
  {
    const int px = px0 + u * PX;
    b[u] = px < HW ? *reinterpret_cast<const f32x4 *>(it.basis + (size_t)px * CS + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float s = it.scale[0];
#pragma unroll
  for (int u = 0; u < kDepthUnroll; ++u)
  {
    const int px = px0 + u * PX;
    float part = b[u][0] * cd[0] + b[u][1] * cd[1] + b[u][2] * cd[2] + b[u][3] * cd[3];
#pragma unroll
    for (int o = F4 / 2; o > 0; o >>= 1)
      part += __shfl_xor(part, o, 64);
    if (px < HW && c4 == 0)
      it.dpt[px] = s * (it.bias[px] + part);
  }
}

__global__ void depth_grad_batch_kernel(const DepthItem *__restrict__ items, int H, int W)
{
  const DepthItem it = items[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= W)
    return;
  const float *a = it.dpt;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  it.grad[(size_t)y * W + x] = 0.5f * (a[(size_t)y * W + xp] - a[(size_t)y * W + xm]);
  it.grad[((size_t)H + y) * W + x] = 0.5f * (a[(size_t)yp * W + x] - a[(size_t)ym * W + x]);
}

hipError_t launch_depth_batch(hipStream_t s, int CS, const DepthItem *items_dev, int K, int H, int W, bool with_depth,
                              bool with_grad)
{
  const int HW = H * W;
  if (K <= 0)
    return hipSuccess;
  if (!with_depth)
    ;
  else if (CS == 32)
    hipLaunchKernelGGL((depth_batch_kernel<32>), dim3((HW + 32 * kDepthUnroll - 1) / (32 * kDepthUnroll), K), dim3(256), 0,
                       s, items_dev, HW);
  else if (CS == 16)
    hipLaunchKernelGGL((depth_batch_kernel<16>), dim3((HW + 64 * kDepthUnroll - 1) / (64 * kDepthUnroll), K), dim3(256), 0,
                       s, items_dev, HW);
  else
    return hipErrorInvalidValue;
  if (with_grad)
    hipLaunchKernelGGL(depth_grad_batch_kernel, dim3((W + 63) / 64, H, K), dim3(64), 0, s, items_dev, H, W);
  return hipGetLastError();
}

// pre-sampled source features f0s[l][g][n][0:4] = MINUS the 4-tap bilinear sample of the keyframe's own pyramid at its sampled
// pixels, with the Jacobian kernel's source coordinates (photometric_factor_kernels.cpp:101-139) -- pose independent,
// built once per keyframe (the tracker's cat_sampled_features_0, camera_tracker.cpp:1104-1123)
__global__ void presample_source_kernel(float *__restrict__ f0s, const float *__restrict__ feat_pk,
                                        const float *__restrict__ homo, int N, int G, const SagePyramid pyr, int exact_coord)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y, l = blockIdx.z;
  if (n >= N)
    return;
  const float fx0 = pyr.cam[0].fx, fy0 = pyr.cam[0].fy, cx0 = pyr.cam[0].cx, cy0 = pyr.cam[0].cy;
  const float fxl = pyr.cam[l].fx, fyl = pyr.cam[l].fy;
  const int Wl = (int)pyr.cam[l].w, Hl = (int)pyr.cam[l].h;
  const float su = homo[3 * n + 0] * fx0 + cx0 + 0.5f, sv = homo[3 * n + 1] * fy0 + cy0 + 0.5f;
  Taps ts;
  if (exact_coord) // non-dyadic pyramid: the reference's own expression (photometric_factor_kernels.cpp:101-103)
    make_taps(ts, (su * fxl) / fx0 - 0.5f, (sv * fyl) / fy0 - 0.5f, Wl, Hl);
  else
    make_taps(ts, su * (fxl / fx0) - 0.5f, sv * (fyl / fy0) - 0.5f, Wl, Hl);
  const f32x4 *src = reinterpret_cast<const f32x4 *>(feat_pk) + (size_t)g * pyr.P + pyr.level_offsets[l];
  f32x4 f = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    f += ts.w[k] * src[ts.off[k]];
  // stored NEGATED: the samplers start their interpolation chain from it (sum_k w_k t_k - f0)
  reinterpret_cast<f32x4 *>(f0s)[((size_t)l * G + g) * N + n] = -f;
}

hipError_t launch_presample_source(hipStream_t s, float *f0s, const float *feat_pk, const float *homo, int N, int FS,
                                   const SagePyramid &pyr)
{
  if (N <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(presample_source_kernel, dim3((N + 255) / 256, FS / 4, pyr.levels), dim3(256), 0, s, f0s, feat_pk,
                     homo, N, FS / 4, pyr, pyramid_is_dyadic(pyr) ? 0 : 1);
  return hipGetLastError();
}

// [C][P] channel-major -> [C/4][P][4] channel-group layout (engine-internal, once per keyframe).  axis 1 / 2: the texels of
// pyramid level l are multiplied by fx_l / fy_l on the way -- the photometric linearize samples h = (fx_l d/dx, fy_l d/dy)
// (photometric_factor_kernels.cpp:200-222 scales the sampled gradient by the level's focal lengths; here once per texel).
// level_scale (optional, per level): one more factor on every texel of the level -- the window engine passes sqrt(w_l), so
// that every product of two sampled quantities (h h^T, h r, r^2) carries the level's weight w_l (:1143-1149) by itself
__global__ void repack_groups_kernel(float *__restrict__ dst, const float *__restrict__ src, int C, int P, int axis,
                                     SagePyramid pyr, RepackScale ls)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (p >= P)
    return;
  float sc = 1.0f;
  if (axis != 0 || ls.on)
  {
    int l = 0;
    for (int i = 1; i < pyr.levels; ++i)
      l = p >= pyr.level_offsets[i] ? i : l;
    sc = axis == 0 ? 1.0f : (axis == 1 ? pyr.cam[l].fx : pyr.cam[l].fy);
    if (ls.on)
      sc *= ls.s[l];
  }
  f32x4 v;
  v[0] = sc * src[(size_t)(4 * g + 0) * P + p];
  v[1] = sc * src[(size_t)(4 * g + 1) * P + p];
  v[2] = sc * src[(size_t)(4 * g + 2) * P + p];
  v[3] = sc * src[(size_t)(4 * g + 3) * P + p];
  *reinterpret_cast<f32x4 *>(dst + ((size_t)g * P + p) * 4) = v;
}

hipError_t launch_repack_groups(hipStream_t s, float *dst, const float *src, int C, int P, int axis, const SagePyramid *pyr,
                                const float *level_scale)
{
  SagePyramid py{};
  if (pyr)
    py = *pyr;
  RepackScale ls{};
  if (pyr && level_scale)
  {
    ls.on = 1;
    for (int l = 0; l < pyr->levels; ++l)
      ls.s[l] = level_scale[l];
  }
  hipLaunchKernelGGL(repack_groups_kernel, dim3((P + 255) / 256, C / 4), dim3(256), 0, s, dst, src, C, P, pyr ? axis : 0, py, ls);
  return hipGetLastError();
}

// central differences with replicate padding (x then y) on [C,H,W]; out [2,C,H,W]
__global__ void spatial_grad_kernel(float *grad, const float *__restrict__ img, int C, int H, int W)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int c = blockIdx.z;
  if (x >= W)
    return;
  const float *a = img + (size_t)c * H * W;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  grad[((size_t)c * H + y) * W + x] = 0.5f * (a[(size_t)y * W + xp] - a[(size_t)y * W + xm]);
  grad[(((size_t)C + c) * H + y) * W + x] = 0.5f * (a[(size_t)yp * W + x] - a[(size_t)ym * W + x]);
}

hipError_t launch_depth_and_grad(hipStream_t s, int CS, float *dpt, float *grad, const float *bias,
                                 const float *basis, const float *code, const float *scale_dev, float scale,
                                 int H, int W)
{
  const int HW = H * W;
  if (CS == 32)
    hipLaunchKernelGGL((depth_kernel<32>), dim3((HW * 8 + 255) / 256), dim3(256), 0, s, dpt, bias, basis, code,
                       scale_dev, scale, HW);
  else if (CS == 16)
    hipLaunchKernelGGL((depth_kernel<16>), dim3((HW * 4 + 255) / 256), dim3(256), 0, s, dpt, bias, basis, code,
                       scale_dev, scale, HW);
  else
    return hipErrorInvalidValue;
  if (grad)
    hipLaunchKernelGGL(spatial_grad_kernel, dim3((W + 63) / 64, H, 1), dim3(64), 0, s, grad, dpt, 1, H, W);
  return hipGetLastError();
}

// one pyramid level: copy level image + its gradients into the concatenated [FS,P] / [2,FS,P] arrays
__global__ void store_level_kernel(float *pyr, float *grad, const float *__restrict__ img, int FS, int H, int W,
                                   int P, int lo)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, c = blockIdx.z;
  if (x >= W)
    return;
  const float *a = img + (size_t)c * H * W;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const size_t o = (size_t)lo + (size_t)y * W + x;
  pyr[(size_t)c * P + o] = a[(size_t)y * W + x];
  grad[(size_t)c * P + o] = 0.5f * (a[(size_t)y * W + xp] - a[(size_t)y * W + xm]);
  grad[((size_t)FS + c) * P + o] = 0.5f * (a[(size_t)yp * W + x] - a[(size_t)ym * W + x]);
}

// next level = conv3x3s2(img*mask)/(conv3x3s2(mask)+1e-8); next mask = mask[2y,2x]  (mapper.cpp:1407-1419)
__global__ void downsample_kernel(float *nxt, float *nmask, const float *__restrict__ img,
                                  const float *__restrict__ mask, int FS, int H, int W)
{
  const int nw = W / 2, nh = H / 2;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, c = blockIdx.z;
  if (x >= nw)
    return;
  const float G[3] = {1.f, 2.f, 1.f};
  float rf = 0.f, rm = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
    {
      const int yy = 2 * y + dy, xx = 2 * x + dx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
      {
        const float g = G[dy + 1] * G[dx + 1] / 16.0f;
        const float mk = mask[(size_t)yy * W + xx];
        rm += g * mk;
        rf += g * (img[((size_t)c * H + yy) * W + xx] * mk);
      }
    }
  nxt[((size_t)c * nh + y) * nw + x] = rf / (rm + 1.0e-8f);
  if (c == 0)
    nmask[(size_t)y * nw + x] = mask[(size_t)(2 * y) * W + 2 * x];
}

// scratch: FS*H*W/4*2 + H*W/4*2 floats (two ping-pong level images + masks)
hipError_t launch_gaussian_pyramid_with_grad(hipStream_t s, float *pyr, float *grad, const float *feat,
                                             const float *mask, const SagePyramid &p, int FS, float *scratch)
{
  const int H0 = (int)p.cam[0].h, W0 = (int)p.cam[0].w;
  const size_t img_sz = (size_t)FS * (H0 / 2) * (W0 / 2), m_sz = (size_t)(H0 / 2) * (W0 / 2);
  float *imgs[2] = {scratch, scratch + img_sz};
  float *masks[2] = {scratch + 2 * img_sz, scratch + 2 * img_sz + m_sz};
  const float *cur = feat, *curm = mask;
  for (int l = 0; l < p.levels; ++l)
  {
    const int H = (int)p.cam[l].h, W = (int)p.cam[l].w;
    hipLaunchKernelGGL(store_level_kernel, dim3((W + 63) / 64, H, FS), dim3(64), 0, s, pyr, grad, cur, FS, H, W,
                       p.P, p.level_offsets[l]);
    if (l == p.levels - 1)
      break;
    float *n = imgs[l & 1], *nm = masks[l & 1];
    hipLaunchKernelGGL(downsample_kernel, dim3((W / 2 + 63) / 64, H / 2, FS), dim3(64), 0, s, n, nm, cur, curm, FS,
                       H, W);
    cur = n;
    curm = nm;
  }
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// f1: valid-pixel enumeration (mapping_utils.h:254-287) -- ordered stream compaction by ONE workgroup (the mask is
// per video, this runs once): chunks of 1024 pixels, wave ballots + a running offset keep the ascending order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void valid_locations_kernel(const float *__restrict__ mask, SageCamera cam, int HW,
                                                               int W, long long *__restrict__ loc1d,
                                                               float *__restrict__ homo, int *__restrict__ n_out)
{
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0)
    s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < HW; c0 += 1024)
  {
    const int i = c0 + tid;
    const bool v = i < HW && mask[i] > 0.5f; // :267
    const unsigned long long b = __ballot(v);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0)
      s_wave[wave] = __popcll(b);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w)
      off += s_wave[w];
    if (v)
    {
      const int n = off + before;
      const float x = (float)(i % W), y = (float)(i / W); // :269-270
      loc1d[n] = i;
      homo[3 * n + 0] = (x - cam.cx) / cam.fx; // :279-280
      homo[3 * n + 1] = (y - cam.cy) / cam.fy;
      homo[3 * n + 2] = 1.0f;
    }
    __syncthreads();
    if (tid == 0)
    {
      int tot = 0;
      for (int w = 0; w < 16; ++w)
        tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (tid == 0)
    *n_out = s_base;
}

hipError_t launch_valid_locations(hipStream_t s, const float *mask, const SageCamera &cam, long long *loc1d, float *homo,
                                  int *n_out_dev)
{
  const int W = (int)cam.w, H = (int)cam.h;
  hipLaunchKernelGGL(valid_locations_kernel, dim3(1), dim3(1024), 0, s, mask, cam, H * W, W, loc1d, homo, n_out_dev);
  return hipGetLastError();
}

// mapper.cpp:1334-1340: sampled_locations = valid_locations[indexes]
__global__ void gather_locations_kernel(const long long *__restrict__ vloc, const float *__restrict__ vhomo,
                                        const long long *__restrict__ index, int n, long long *__restrict__ loc1d,
                                        float *__restrict__ homo)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const long long j = index[i];
  loc1d[i] = vloc[j];
  homo[3 * i + 0] = vhomo[3 * j + 0];
  homo[3 * i + 1] = vhomo[3 * j + 1];
  homo[3 * i + 2] = vhomo[3 * j + 2];
}

hipError_t launch_gather_locations(hipStream_t s, const long long *vloc, const float *vhomo, const long long *index_dev,
                                   int n, long long *loc1d, float *homo)
{
  if (n > 0)
    hipLaunchKernelGGL(gather_locations_kernel, dim3((n + 255) / 256), dim3(256), 0, s, vloc, vhomo, index_dev, n, loc1d,
                       homo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// engine-internal relayout of a keyframe's sampled locations: raster order.  The factor sums do not depend on the order
// of the samples, the vector L1 does: a wave whose 64 pixels are neighbours touches ~10 cache lines per tap load, a
// shuffled sample list (mapper.cpp:1326-1340 keeps the first N of a std::shuffle) 64 -- the photometric linearize is
// 7.5x slower on it.  Sort = scatter the sample index into a per-pixel mark plane, then an ordered compaction of the
// plane (one workgroup per keyframe).  status[2k] counts out-of-range locations, status[2k+1] the compacted samples
// (< n when a pixel is sampled twice: the caller then keeps the original order).
// ------------------------------------------------------------------------------------------------
__global__ void mark_locations_kernel(const SortItem *__restrict__ items, int HW, int *__restrict__ mark,
                                      int *__restrict__ status)
{
  const int k = blockIdx.y;
  const SortItem it = items[k];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= it.n)
    return;
  const long long l = it.loc[i];
  if (l < 0 || l >= HW)
    atomicAdd(&status[2 * k], 1);
  else
    mark[(size_t)k * HW + l] = i;
}

// walk order of the plane: raster (TW = 0) or image tiles of TW x TH pixels, tiles in raster order, pixels inside a tile
// in raster order (a wave's 64 consecutive samples then cover e.g. 16 x 4 pixels instead of a 64-pixel row segment: its
// four tap loads share texel rows -- (x, y+1) of one pixel row is (x, y) of the next -- fewer distinct lines per step)
__device__ __forceinline__ int walk_to_pixel(int c, int W, int H, int TW, int TH)
{
  if (TW <= 0)
    return c < W * H ? c : -1;
  const int per = TW * TH, tpr = (W + TW - 1) / TW;
  const int t = c / per, in = c - t * per;
  const int y = (t / tpr) * TH + in / TW, x = (t % tpr) * TW + in % TW;
  return (x < W && y < H) ? y * W + x : -1;
}

// r06 -- tile-padded order (pad_flags[k] != 0, 8 x 8 tiles only: a wave of the walk IS a tile).  The compaction above drops the
// empty pixels of a partially sampled tile, so every sample behind the first partial tile sits at a shifted position: a wave's 64
// consecutive samples then straddle two tiles, their warped footprint is twice as wide, the LDS-staged sampler of the photometric
// kernels refuses it and the whole keyframe goes through the texture path (measured: a 148 x 116 sample rectangle that starts at
// x = 6 instead of x = 8 runs the linearize 1.67 x slower than the aligned 144 x 112 one; any irregular mask -- an endoscope's
// circle -- does the same).  Padded: every tile that holds a sample keeps all 64 slots, a slot without a sample carries location -1
// (the kernels drop it like any out-of-image location) and the ray (0, 0, 1); waves and tiles coincide whatever the mask.
// tiles_out[k] (optional) = tiles with at least one sample, so that the host can price the padding before it asks for it.
__global__ __launch_bounds__(1024) void order_locations_kernel(const SortItem *__restrict__ items, int HW, int W, int TW,
                                                               int TH, const int *__restrict__ mark,
                                                               int *__restrict__ status, const int *__restrict__ pad_flags,
                                                               int *__restrict__ tiles_out)
{
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int k = blockIdx.x;
  const SortItem it = items[k];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = HW / W;
  const int n_walk = TW > 0 ? ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * TW * TH : HW;
  const bool pad = pad_flags && pad_flags[k] != 0 && TW * TH == 64;
  __shared__ int s_tiles;
  if (tid == 0)
  {
    s_base = 0;
    s_tiles = 0;
  }
  __syncthreads();
  for (int c0 = 0; c0 < n_walk; c0 += 1024)
  {
    const int c = c0 + tid;
    const int i = c < n_walk ? walk_to_pixel(c, W, H, TW, TH) : -1;
    const int src = i >= 0 ? mark[(size_t)k * HW + i] : -1;
    const bool v = src >= 0;
    const unsigned long long b = __ballot(v);
    const int before = pad ? lane : __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0)
    {
      s_wave[wave] = pad ? (b != 0ull ? 64 : 0) : __popcll(b);
      if (b != 0ull && TW * TH == 64)
        atomicAdd(&s_tiles, 1);
    }
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w)
      off += s_wave[w];
    if (v)
    {
      const int n = off + before;
      it.loc_out[n] = i;
      it.homo_out[3 * n + 0] = it.homo[3 * src + 0];
      it.homo_out[3 * n + 1] = it.homo[3 * src + 1];
      it.homo_out[3 * n + 2] = it.homo[3 * src + 2];
    }
    else if (pad && b != 0ull)
    {
      const int n = off + before; // a hole of a kept tile
      it.loc_out[n] = -1;
      it.homo_out[3 * n + 0] = 0.f;
      it.homo_out[3 * n + 1] = 0.f;
      it.homo_out[3 * n + 2] = 1.f;
    }
    __syncthreads();
    if (tid == 0)
    {
      int tot = 0;
      for (int w = 0; w < 16; ++w)
        tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (tid == 0)
  {
    status[2 * k + 1] = s_base; // samples written (compact order) / slots written (padded order)
    if (tiles_out)
      tiles_out[k] = s_tiles;
  }
}

hipError_t launch_sort_locations(hipStream_t s, const SortItem *items_dev, int K, int max_n, int HW, int *mark_dev,
                                 int *status_dev, int W, int tile_w, int tile_h, const int *pad_flags_dev, int *tiles_out_dev,
                                 bool keep_marks)
{
  if (W <= 0 || HW % W != 0 || tile_w <= 0 || tile_h <= 0)
  {
    W = HW; // raster walk
    tile_w = tile_h = 0;
  }
  hipError_t e;
  if (!keep_marks) // (keep_marks: a second ordering pass over the mark planes of the first -- the padded order)
  {
    if ((e = hipMemsetAsync(mark_dev, 0xff, (size_t)K * HW * sizeof(int), s)) != hipSuccess ||
        (e = hipMemsetAsync(status_dev, 0, (size_t)2 * K * sizeof(int), s)) != hipSuccess)
      return e;
    if (max_n > 0)
      hipLaunchKernelGGL(mark_locations_kernel, dim3((max_n + 255) / 256, K), dim3(256), 0, s, items_dev, HW, mark_dev,
                         status_dev);
  }
  hipLaunchKernelGGL(order_locations_kernel, dim3(K), dim3(1024), 0, s, items_dev, HW, W, tile_w, tile_h, mark_dev, status_dev,
                     pad_flags_dev, tiles_out_dev);
  return hipGetLastError();
}

// out[i] = mult * in[i]: the tracker's `guess_scale_0 * unscaled_dpts_0` (camera_tracker.cpp:264,273,431,453), one fp32
// multiply per sample like the reference's tensor expression
__global__ void scale_array_kernel(float *__restrict__ out, const float *__restrict__ in, float mult, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = mult * in[i];
}

hipError_t launch_scale_array(hipStream_t s, float *out, const float *in, float mult, int n)
{
  if (n > 0)
    hipLaunchKernelGGL(scale_array_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, in, mult, n);
  return hipGetLastError();
}

} // namespace sage
