// dev microbenchmark: latency of a launch-bound evaluation (H2D 48 B from pinned, 4 small dependent kernels, D2H 464 B to pinned,
// stream synchronise) issued call by call vs as one captured hipGraph -- the shape of one tracker LM evaluation.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(float *p, const float *pose, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = p[i] * 0.999f + pose[i % 12];
}
int main()
{
  hipStream_t s;
  hipStreamCreate(&s);
  float *d, *dpose, *hp, *hout;
  const int n = 3072;
  hipMalloc(&d, n * 4); hipMalloc(&dpose, 64);
  hipHostMalloc(&hp, 64); hipHostMalloc(&hout, 512);
  hipMemset(d, 0, n * 4);
  for (int i = 0; i < 12; ++i) hp[i] = 0.001f * i;
  auto eval_direct = [&]() {
    hipMemcpyAsync(dpose, hp, 48, hipMemcpyHostToDevice, s);
    for (int j = 0; j < 4; ++j)
      hipLaunchKernelGGL(k, dim3(12), dim3(256), 0, s, d, dpose, n);
    hipMemcpyAsync(hout, d, 464, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
  };
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  hipMemcpyAsync(dpose, hp, 48, hipMemcpyHostToDevice, s);
  for (int j = 0; j < 4; ++j)
    hipLaunchKernelGGL(k, dim3(12), dim3(256), 0, s, d, dpose, n);
  hipMemcpyAsync(hout, d, 464, hipMemcpyDeviceToHost, s);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  auto eval_graph = [&]() {
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
  };
  for (int rep = 0; rep < 3; ++rep)
  {
    for (int w = 0; w < 50; ++w) eval_direct();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) eval_direct();
    auto t1 = std::chrono::steady_clock::now();
    for (int w = 0; w < 50; ++w) eval_graph();
    auto t2 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) eval_graph();
    auto t3 = std::chrono::steady_clock::now();
    printf("evaluation of 4 dependent small kernels + H2D + D2H + sync: call by call %.1f us, one hipGraphLaunch %.1f us\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, std::chrono::duration<double, std::micro>(t3 - t2).count() / 2000);
  }
  return 0;
}
