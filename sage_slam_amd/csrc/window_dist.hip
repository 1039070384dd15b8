// window_dist.hip -- sharded windows: one process per GPU (NUMA / CCX placement of the host threads), the all-reduce hook and
// the native RCCL binding (librccl bound with dlopen; ncclAllReduce(double, sum) issued on the window's stream).
#include "runtime_internal.h"
#include <unistd.h>

static bool device_local_cpulist(int device, char *buf, size_t n)
{
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess)
    return false;
  for (char *p = bdf; *p; ++p)
    *p = (char)tolower((unsigned char)*p);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE *f = fopen(path, "r");
  if (!f)
    return false;
  buf[0] = 0;
  const bool got = fgets(buf, (int)n, f) != nullptr;
  fclose(f);
  return got;
}

static std::vector<int> parse_cpulist(const char *buf)
{
  std::vector<int> out;
  for (const char *p = buf; *p;)
  {
    char *end;
    const long a = strtol(p, &end, 10);
    if (end == p)
      break;
    long b = a;
    p = end;
    if (*p == '-')
    {
      b = strtol(p + 1, &end, 10);
      p = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      out.push_back((int)c);
    if (*p == ',')
      ++p;
  }
  return out;
}

// One process per GPU: keep the driving thread on the CPUs the GPU hangs off (its NUMA node: the window solve reads
// freshly DMA'd pinned memory), and -- when several GPUs share that node -- on its own L3 domain (CCX) of the node: the
// solve pins its helper / worker threads to the other cores of the caller's CCX (host_math.cpp), so two ranks whose
// driving threads shared a CCX would share those cores.  Returns the number of CPUs the thread is bound to (0: unchanged).
extern "C" int sage_bind_thread_to_device(int device)
{
  char buf[4096] = {0};
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return SAGE_E_INVALID;
  if (!device_local_cpulist(device, buf, sizeof(buf)))
    return 0;
  cpu_set_t allowed, want;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
    return 0;
  std::vector<int> cpus;
  for (int c : parse_cpulist(buf))
    if (CPU_ISSET(c, &allowed))
      cpus.push_back(c);
  if (cpus.empty())
    return 0;
  // devices on the same node, this one's position among them
  int on_node = 0, my_pos = 0;
  for (int d = 0; d < ndev; ++d)
  {
    char other[4096] = {0};
    if (d == device || (device_local_cpulist(d, other, sizeof(other)) && strcmp(other, buf) == 0))
    {
      if (d < device)
        ++my_pos;
      ++on_node;
    }
  }
  if (on_node > 1)
  {
    // L3 domains of the node, in the order of their first CPU
    std::vector<std::vector<int>> groups;
    std::vector<char> seen(CPU_SETSIZE, 0);
    for (int c : cpus)
    {
      if (seen[c])
        continue;
      char path[160], lb[4096] = {0};
      snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
      FILE *f = fopen(path, "r");
      std::vector<int> g;
      if (f)
      {
        if (fgets(lb, sizeof(lb), f))
          for (int x : parse_cpulist(lb))
            if (x < CPU_SETSIZE && CPU_ISSET(x, &allowed) && std::find(cpus.begin(), cpus.end(), x) != cpus.end())
              g.push_back(x);
        fclose(f);
      }
      if (g.empty())
        g.push_back(c);
      for (int x : g)
        seen[x] = 1;
      groups.push_back(g);
    }
    if (groups.size() > 1)
    {
      const size_t stride = std::max<size_t>(1, groups.size() / (size_t)on_node);
      cpus = groups[((size_t)my_pos * stride) % groups.size()];
    }
  }
  // r05: the box is a slice of a node whose other GPUs run other tenants' jobs on CPUs of the same NUMA node.  Look at the
  // load (250 ms of /proc/stat), keep to physical cores that are quiet on all their hardware threads, and -- alone on the node
  // -- move to the L3 domain with the most of them; the solve's helper threads are placed on quiet cores only
  // (placement_set_allowed: the whole node's quiet CPUs, so that a loop-closure plan still finds its second domain).
  if (!sage::env_flag("SAGE_BIND_NO_PROBE"))
  {
    const std::vector<int> busy = sage::placement_busy_cpus(250);
    std::vector<char> noisy(CPU_SETSIZE, 0);
    for (int b : busy)
      for (int sib : sage::placement_core_siblings(b))
        if (sib < CPU_SETSIZE)
          noisy[sib] = 1;
    // the node's quiet CPUs (of the CPUs this process may use)
    cpu_set_t quiet_node;
    CPU_ZERO(&quiet_node);
    int n_quiet_node = 0;
    for (int c : parse_cpulist(buf))
      if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed) && !noisy[c])
      {
        CPU_SET(c, &quiet_node);
        ++n_quiet_node;
      }
    auto quiet_cores_of = [&](const std::vector<int> &set) {
      int n = 0;
      std::vector<char> seen(CPU_SETSIZE, 0);
      for (int c : set)
      {
        if (c >= CPU_SETSIZE || seen[c] || noisy[c] || !CPU_ISSET(c, &allowed))
          continue;
        for (int sib : sage::placement_core_siblings(c))
          if (sib < CPU_SETSIZE)
            seen[sib] = 1;
        ++n;
      }
      return n;
    };
    std::vector<int> pick = cpus;
    if (on_node <= 1)
    {
      // L3 domains of the node: the one with the most quiet physical cores (ties: the first)
      // (among the domains within one core of the best the process id decides: processes started together -- they cannot see
      //  each other while all of them are looking -- spread out instead of landing on the same domain)
      std::vector<char> seen(CPU_SETSIZE, 0);
      std::vector<std::pair<int, std::vector<int>>> doms;
      int best = -1;
      for (int c : cpus)
      {
        if (seen[c])
          continue;
        std::vector<int> dom = sage::placement_l3_domain(c);
        if (dom.empty())
          dom.push_back(c);
        for (int x : dom)
          if (x < CPU_SETSIZE)
            seen[x] = 1;
        std::vector<int> in;
        for (int x : dom)
          if (std::find(cpus.begin(), cpus.end(), x) != cpus.end())
            in.push_back(x);
        const int q = quiet_cores_of(in);
        best = std::max(best, q);
        doms.emplace_back(q, std::move(in));
      }
      std::vector<const std::vector<int> *> good;
      for (const auto &d : doms)
        if (d.first >= std::max(best - 1, 4))
          good.push_back(&d.second);
      if (!good.empty())
        pick = *good[(size_t)getpid() % good.size()];
    }
    std::vector<int> quiet_pick;
    for (int c : pick)
      if (!noisy[c])
        quiet_pick.push_back(c);
    if (quiet_cores_of(quiet_pick) >= 4 && n_quiet_node >= 8)
    {
      cpus = quiet_pick;
      sage::placement_set_allowed(&quiet_node);
    }
    else
      sage::placement_set_allowed(nullptr); // too little room: placement as before
  }
  CPU_ZERO(&want);
  for (int c : cpus)
    CPU_SET(c, &want);
  if (sched_setaffinity(0, sizeof(want), &want) != 0)
    return 0;
  return (int)cpus.size();
}

extern "C" int sage_solver_helper_cpus(int *cpus, int n) { return cpus && n > 0 ? sage::placement_helper_cpus(cpus, n) : 0; }
extern "C" int sage_solver_placement_moves(void) { return sage::placement_monitor_moves(); }
extern "C" int sage_placement_monitor(int enable)
{
  sage::placement_monitor_enable(enable);
  return SAGE_OK;
}
extern "C" int sage_host_threads_running(void) { return sage::host_threads_running(); }
extern "C" void sage_shutdown(void) { sage::host_threads_shutdown(); }

extern "C" int sage_window_set_allreduce(SageWindow *w, SageAllReduceFn fn, void *user)
{
  if (!w)
    return SAGE_E_INVALID;
  w->allreduce = fn;
  w->allreduce2 = nullptr;
  w->allreduce_user = user;
  return SAGE_OK;
}

// =====================================================================================================
// native RCCL: the all-reduce of a sharded window as an ncclAllReduce on the window's own stream (xGMI), no Python
// and no torch in the loop.  RCCL is bound with dlopen so that the library loads on hosts without it; when the
// process already has an RCCL mapped (PyTorch-ROCm bundles one) that instance is reused.
// =====================================================================================================
namespace
{
struct RcclApi
{
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
};

RcclApi &rccl()
{
  static RcclApi api = [] {
    RcclApi a;
    const char *already[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : already)
      if (!a.handle)
        a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : fresh)
      if (!a.handle)
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle)
      return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.handle, "ncclCommCount"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(a.handle, "ncclCommUserRank"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    return a;
  }();
  return api;
}

struct RcclHook
{
  ncclComm_t comm;
  hipStream_t stream;
};

int rccl_allreduce_cb(double *buf, size_t n, void *user)
{
  RcclHook *h = static_cast<RcclHook *>(user);
  const ncclResult_t r = rccl().AllReduce(buf, buf, n, ncclDouble, ncclSum, h->comm, h->stream);
  if (r != ncclSuccess)
  {
    fprintf(stderr, "[sage] ncclAllReduce: %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    return 1;
  }
  return 0;
}
int rccl_allreduce2_cb(const double *send, double *recv, size_t n, void *user)
{
  RcclHook *h = static_cast<RcclHook *>(user);
  const ncclResult_t r = rccl().AllReduce(send, recv, n, ncclDouble, ncclSum, h->comm, h->stream);
  if (r != ncclSuccess)
  {
    fprintf(stderr, "[sage] ncclAllReduce: %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    return 1;
  }
  return 0;
}
} // namespace

extern "C" int sage_rccl_unique_id(unsigned char *id128)
{
  if (!id128)
    return SAGE_E_INVALID;
  if (!rccl().ok)
    return SAGE_E_UNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess)
    return SAGE_E_STATE;
  static_assert(sizeof(id) == SAGE_RCCL_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof(id));
  return SAGE_OK;
}

extern "C" int sage_rccl_comm_create(const unsigned char *id128, int rank, int world, void **comm_out)
{
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world)
    return SAGE_E_INVALID;
  if (!rccl().ok)
    return SAGE_E_UNSUPPORTED;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = rccl().CommInitRank(&c, world, id, rank);
  if (r != ncclSuccess)
  {
    fprintf(stderr, "[sage] ncclCommInitRank: %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    return SAGE_E_STATE;
  }
  *comm_out = c;
  return SAGE_OK;
}

extern "C" int sage_rccl_comm_info(void *comm, int *ranks_out, int *rank_out)
{
  if (!comm)
    return SAGE_E_INVALID;
  if (!rccl().ok || !rccl().CommCount || !rccl().CommUserRank)
    return SAGE_E_UNSUPPORTED;
  int n = 0, r = 0;
  if (rccl().CommCount(static_cast<ncclComm_t>(comm), &n) != ncclSuccess ||
      rccl().CommUserRank(static_cast<ncclComm_t>(comm), &r) != ncclSuccess)
    return SAGE_E_STATE;
  if (ranks_out)
    *ranks_out = n;
  if (rank_out)
    *rank_out = r;
  return SAGE_OK;
}

extern "C" void sage_rccl_comm_destroy(void *comm)
{
  if (comm && rccl().ok)
    (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm));
}

extern "C" int sage_window_use_rccl(SageWindow *w, void *nccl_comm)
{
  if (!w || !nccl_comm)
    return SAGE_E_INVALID;
  if (!rccl().ok)
    return SAGE_E_UNSUPPORTED;
  std::free(w->rccl_hook);
  RcclHook *h = static_cast<RcclHook *>(std::malloc(sizeof(RcclHook)));
  if (!h)
    return SAGE_E_STATE;
  h->comm = static_cast<ncclComm_t>(nccl_comm);
  h->stream = w->stream;
  w->rccl_hook = h;
  w->allreduce = rccl_allreduce_cb;
  w->allreduce2 = rccl_allreduce2_cb;
  w->allreduce_user = h;
  return SAGE_OK;
}

// Development aid for boxes with fewer GPUs than ranks (bench.py --emulate-shard): the share of the ranks that are not
// there.  rest_dev: n_iterates consecutive buffers of sage_window_packed_count doubles, entry i = the packed systems of all
// OTHER ranks summed, evaluated at the i-th LM iterate since sage_window_reset; after every all-reduce of the window (the
// real collective still runs: a one-rank communicator costs its launch, not its transfer) the entry of the iterate being
// reduced is added on the window's stream.  nullptr switches it off.  The caller keeps the table alive.
extern "C" int sage_window_emulate_peers(SageWindow *w, const double *rest_dev, int n_iterates)
{
  if (!w || (rest_dev && n_iterates < 1))
    return SAGE_E_INVALID;
  w->emu_rest = rest_dev;
  w->emu_n = rest_dev ? n_iterates : 0;
  w->emu_cur = 0;
  return SAGE_OK;
}

extern "C" int sage_window_set_shard(SageWindow *w, int rank, int world)
{
  if (!w || w->finalized || world < 1 || rank < 0 || rank >= world)
    return SAGE_E_INVALID;
  w->rank = rank;
  w->world = world;
  return SAGE_OK;
}

