// Data-parallel exchange of the additive WISKI statistics at the C ABI (SURVEY.md 8b table, row `wiski_allreduce_stats`;
// 8e): one grouped RCCL all-reduce(SUM) over the buffers that are sums over data points -- the half-stencil delta of
// W^T D^-1 W, W^T D^-1 y, the row-sum vector, and the [y^T D^-1 y, log|D|] / [count, weight sum] scalars (fp64).
// The reference has no distributed code; this is the one collective of the path, callable by a host program that owns
// its own ncclComm_t (the Python package uses torch.distributed's process group for the same exchange: distributed.py).
//
// librccl is resolved with dlopen at the first call, so libwiski_hip.so itself carries no link-time dependency on it
// (a torch process already holds its own copy; a plain C host gets the ROCm one).  WISKI_RCCL_LIB overrides the name.
#include "wiski_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;              // optional (reporting only)
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
  std::call_once(g_rccl_once, [] {
    const char* names[] = {getenv("WISKI_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n) continue;
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.h) break;
    }
    if (!g_rccl.h) return;
    auto sym = [](const char* s) { return dlsym(g_rccl.h, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))sym("ncclGetVersion");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))sym("ncclCommUserRank");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GroupStart && g_rccl.GroupEnd;
  });
  return g_rccl;
}

template <typename real>
int allreduce_stats_impl(void* comm, real* d_half, int64_t n_half, real* d_b, int64_t n_b, real* d_cnt, int64_t n_cnt, double* d_scal,
                         int64_t n_scal, void* stream) {
  const Rccl& R = rccl();
  if (!R.ok) return WISKI_E_LAUNCH;
  if (!comm || n_half < 0 || n_b < 0 || n_cnt < 0 || n_scal < 0) return WISKI_E_BADARG;
  const ncclDataType_t dt = sizeof(real) == 4 ? ncclFloat32 : ncclFloat64;
  ncclComm_t c = (ncclComm_t)comm;
  hipStream_t s = (hipStream_t)stream;
  if (R.GroupStart() != ncclSuccess) return WISKI_E_LAUNCH;
  bool good = true;
  if (d_half && n_half) good &= R.AllReduce(d_half, d_half, (size_t)n_half, dt, ncclSum, c, s) == ncclSuccess;
  if (d_b && n_b) good &= R.AllReduce(d_b, d_b, (size_t)n_b, dt, ncclSum, c, s) == ncclSuccess;
  if (d_cnt && n_cnt) good &= R.AllReduce(d_cnt, d_cnt, (size_t)n_cnt, dt, ncclSum, c, s) == ncclSuccess;
  if (d_scal && n_scal) good &= R.AllReduce(d_scal, d_scal, (size_t)n_scal, ncclFloat64, ncclSum, c, s) == ncclSuccess;
  if (R.GroupEnd() != ncclSuccess) return WISKI_E_LAUNCH;
  return good ? WISKI_OK : WISKI_E_LAUNCH;
}
}  // namespace

extern "C" {
int wiski_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int wiski_comm_unique_id(void* id_out) {
  const Rccl& R = rccl();
  if (!R.ok) return WISKI_E_LAUNCH;
  if (!id_out) return WISKI_E_BADARG;
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return WISKI_E_LAUNCH;
  memcpy(id_out, &id, sizeof(id));
  return WISKI_OK;
}

int wiski_comm_init_rank(const void* id_bytes, int32_t nranks, int32_t rank, void** comm_out) {
  const Rccl& R = rccl();
  if (!R.ok) return WISKI_E_LAUNCH;
  if (!id_bytes || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return WISKI_E_BADARG;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t c = nullptr;
  if (R.CommInitRank(&c, nranks, id, rank) != ncclSuccess) return WISKI_E_LAUNCH;
  *comm_out = (void*)c;
  return WISKI_OK;
}

int wiski_comm_destroy(void* comm) {
  const Rccl& R = rccl();
  if (!R.ok) return WISKI_E_LAUNCH;
  if (!comm) return WISKI_E_BADARG;
  return R.CommDestroy((ncclComm_t)comm) == ncclSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

int wiski_comm_info(void* comm, int32_t* version_out, int32_t* nranks_out, int32_t* rank_out) {
  const Rccl& R = rccl();
  if (!R.ok) return WISKI_E_LAUNCH;
  int v = 0, n = 0, r = -1;
  if (version_out && R.GetVersion && R.GetVersion(&v) != ncclSuccess) return WISKI_E_LAUNCH;
  if (comm) {
    if (R.CommCount && R.CommCount((ncclComm_t)comm, &n) != ncclSuccess) return WISKI_E_LAUNCH;
    if (R.CommUserRank && R.CommUserRank((ncclComm_t)comm, &r) != ncclSuccess) return WISKI_E_LAUNCH;
  }
  if (version_out) *version_out = v;
  if (nranks_out) *nranks_out = n;
  if (rank_out) *rank_out = r;
  return WISKI_OK;
}

int wiski_allreduce_stats_f32(void* comm, float* d_half, int64_t n_half, float* d_b, int64_t n_b, float* d_cnt, int64_t n_cnt, double* d_scal, int64_t n_scal, void* stream) {
  return allreduce_stats_impl<float>(comm, d_half, n_half, d_b, n_b, d_cnt, n_cnt, d_scal, n_scal, stream);
}
int wiski_allreduce_stats_f64(void* comm, double* d_half, int64_t n_half, double* d_b, int64_t n_b, double* d_cnt, int64_t n_cnt, double* d_scal, int64_t n_scal, void* stream) {
  return allreduce_stats_impl<double>(comm, d_half, n_half, d_b, n_b, d_cnt, n_cnt, d_scal, n_scal, stream);
}
}
