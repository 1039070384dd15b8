// prepass_run -- EXECUTES integration/sage_gtsam_prepass.h (the gtsam side of the batched prepass) on a
// real window: real Eigen + Sophus from the reference's thirdparty tree, the recording stand-ins of this directory for
// gtsam::Values / gtsam::HessianFactor (gtsam itself needs Boost, absent in the image), libsage_ba.so for everything
// behind the C ABI.  Built in the build container by sage_slam_amd/build.py:build_glue_check() (needs /root/reference
// for the two header-only libraries), run on the GPU by tests/test_gpu_gtsam_glue.py:
//
//   prepass_run <window.bin> <out.bin> <psd_mode>
//
// window.bin (little endian; written by the test): int32 K,H,W,FS,CS,L,nlinks; float cam[6]; float photo_weights[L];
// float geo_weight, geo_loss_param, eps, code_prior, scale_prior, pose_prior; float mask[H*W]; per keyframe: int32 N,
// float feat[FS*P], grad[2*FS*P], bias[H*W], basis[H*W*CS], int64 loc1d[N], float homo[3N], pose12[12], code[CS], scale;
// int32 links[nlinks][2]; then the Values to evaluate at: per keyframe quaternion (w,x,y,z) float[4], t[3], code[CS], scale.
// out.bin: per keyframe the float pose12 the Sophus::SE3f in the Values actually holds (R row-major | t); int32
// recomputed count; then for type in {0,1}, edge in [0, 2*nlinks): int32 nkeys, uint64 keys[nkeys], int32 dims[nkeys],
// double info[(D+1)^2] row-major (the augmented information matrix the HessianFactor assembled), double error from
// SageWindowCache::Error at the same Values.
#include "../sage_gtsam_prepass.h"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                              \
  do                                                                          \
  {                                                                           \
    int _r = (int)(x);                                                        \
    if (_r != 0)                                                              \
    {                                                                         \
      fprintf(stderr, "prepass_run: %s failed with %d (line %d)\n", #x, _r, __LINE__); \
      return 3;                                                               \
    }                                                                         \
  } while (0)

template <class T> static std::vector<T> rd(FILE *f, size_t n)
{
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n)
  {
    fprintf(stderr, "prepass_run: short read\n");
    exit(4);
  }
  return v;
}
template <class T> static T *up(const std::vector<T> &v)
{
  void *p = nullptr;
  if (hipMalloc(&p, v.size() * sizeof(T) + 16) != hipSuccess ||
      hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
  {
    fprintf(stderr, "prepass_run: device upload failed\n");
    exit(5);
  }
  return static_cast<T *>(p);
}
template <class T> static void wr(FILE *f, const T *p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv)
{
  if (argc != 4)
  {
    fprintf(stderr, "usage: prepass_run window.bin out.bin psd_mode\n");
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f)
    return 2;
  const int psd_mode = atoi(argv[3]);
  auto hd = rd<int32_t>(f, 7);
  const int K = hd[0], H = hd[1], W = hd[2], FS = hd[3], CS = hd[4], L = hd[5], nlinks = hd[6];
  auto cam = rd<float>(f, 6);
  auto pw = rd<float>(f, L);
  auto sc = rd<float>(f, 6);
  SageCamera base{cam[0], cam[1], cam[2], cam[3], cam[4], cam[5]};
  SageWindowConfig cfg{};
  CHECK(sage_camera_pyramid(&base, L, &cfg.pyr));
  const int P = cfg.pyr.P;
  cfg.FS = FS;
  cfg.CS = CS;
  cfg.mask_dev = up(rd<float>(f, (size_t)H * W));
  for (int l = 0; l < L; ++l)
    cfg.photo_weights[l] = pw[l];
  cfg.geo_weight = sc[0]; cfg.geo_loss_param = sc[1]; cfg.eps = sc[2];
  cfg.code_prior_weight = sc[3]; cfg.scale_prior_weight = sc[4]; cfg.pose_prior_weight = sc[5];
  cfg.use_photo = cfg.use_geo = 1;
  SageWindow *win = nullptr;
  CHECK(sage_window_create(&cfg, nullptr, &win));
  for (int k = 0; k < K; ++k)
  {
    const int N = rd<int32_t>(f, 1)[0];
    SageKeyframeView v{};
    v.feat_pyr = up(rd<float>(f, (size_t)FS * P));
    v.grad_pyr = up(rd<float>(f, (size_t)2 * FS * P));
    v.bias = up(rd<float>(f, (size_t)H * W));
    v.basis = up(rd<float>(f, (size_t)H * W * CS));
    v.loc1d = up(rd<int64_t>(f, N));
    v.homo = up(rd<float>(f, (size_t)3 * N));
    v.N = N;
    auto pose = rd<float>(f, 12);
    auto code = rd<float>(f, CS);
    const float scale = rd<float>(f, 1)[0];
    if (sage_window_add_keyframe(win, &v, pose.data(), code.data(), scale) != k)
      return 6;
  }
  auto links = rd<int32_t>(f, (size_t)2 * nlinks);
  for (int l = 0; l < nlinks; ++l)
    if (sage_window_add_link(win, links[2 * l], links[2 * l + 1]) != l)
      return 7;
  CHECK(sage_window_finalize(win));

  // the gtsam side: keys as the mapper would number them, Values holding Sophus::SE3f / gtsam::Vector / float
  auto pk = [](int k) { return (gtsam::Key)(1000 + k); };
  auto ck = [](int k) { return (gtsam::Key)(2000 + k); };
  auto sk = [](int k) { return (gtsam::Key)(3000 + k); };
  std::vector<df::SageWindowCache::Keys> keys;
  gtsam::Values values;
  FILE *o = fopen(argv[2], "wb");
  if (!o)
    return 2;
  for (int k = 0; k < K; ++k)
  {
    keys.push_back({pk(k), ck(k), sk(k)});
    auto q = rd<float>(f, 4);
    auto t = rd<float>(f, 3);
    auto code = rd<float>(f, CS);
    const float scale = rd<float>(f, 1)[0];
    const Sophus::SE3f T(Eigen::Quaternionf(q[0], q[1], q[2], q[3]), Eigen::Vector3f(t[0], t[1], t[2]));
    values.insert(pk(k), T);
    gtsam::Vector c(CS);
    for (int i = 0; i < CS; ++i)
      c(i) = code[i];
    values.insert(ck(k), c);
    values.insert(sk(k), scale);
    // what the Values hold, formed here independently of the header under test
    const Eigen::Matrix3f R = T.so3().matrix();
    float p12[12];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        p12[i * 3 + j] = R(i, j);
    for (int i = 0; i < 3; ++i)
      p12[9 + i] = T.translation()[i];
    wr(o, p12, 12);
  }
  fclose(f);
  df::SageWindowCache cache(win, keys, CS, psd_mode);
  int32_t recomputed = cache.Prepare(values, true) ? 1 : 0;
  recomputed += cache.Prepare(values, true) ? 1 : 0; // second time: a cache hit
  wr(o, &recomputed, 1);
  for (int type = 0; type < 2; ++type)
    for (int e = 0; e < 2 * nlinks; ++e)
    {
      const int a = links[2 * (e / 2)], b = links[2 * (e / 2) + 1];
      const int k0 = (e & 1) ? b : a, k1 = (e & 1) ? a : b;
      // PhotometricFactor keys {pose0, pose1, code0, scale0} (photometric_factor.cpp:151-163), GeometricFactor keys
      // {pose0, pose1, code0, code1, scale0, scale1} (geometric_factor.cpp:120-135)
      gtsam::FastVector<gtsam::Key> fk = type == 0 ? gtsam::FastVector<gtsam::Key>{pk(k0), pk(k1), ck(k0), sk(k0)}
                                                   : gtsam::FastVector<gtsam::Key>{pk(k0), pk(k1), ck(k0), ck(k1), sk(k0), sk(k1)};
      boost::shared_ptr<gtsam::HessianFactor> hf = cache.Linearize(values, type, e, fk);
      const int32_t nk = (int32_t)hf->keys().size();
      wr(o, &nk, 1);
      wr(o, hf->keys().data(), nk);
      std::vector<int32_t> dims(hf->dims().begin(), hf->dims().end());
      wr(o, dims.data(), nk);
      const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> info = hf->augmentedInformation();
      wr(o, info.data(), (size_t)info.size());
      const double err = cache.Error(values, type, e);
      wr(o, &err, 1);
    }
  fclose(o);
  sage_window_destroy(win);
  return 0;
}
