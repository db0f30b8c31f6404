// ics_group.hip -- the only cross-GPU step of the path: one process per GPU, every rank deconvolves its own
// images, and at the end the ranks exchange a small record each (time, iterations, checksum).  RCCL over xGMI, called
// directly (librccl.so is dlopen'ed on first use so that single-GPU users of libics_hip.so do not need it); no collective
// exists inside the iterations.  The RCCL unique id travels from rank 0 to the other ranks of the node through a file.
#include <dlfcn.h>
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <hip/hip_runtime.h>

#include "../../include/ics_hip.h"

// the slice of rccl.h this file needs (ROCm 7.2 /opt/rocm/include/rccl/rccl.h: the NCCL 2.x ABI)
namespace {
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_SUCCESS = 0, RCCL_SUM = 0, RCCL_MAX = 2, RCCL_UINT32 = 3, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8 };
struct Rccl {
  void* so;
  int (*GetUniqueId)(rcclUniqueId*);
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
  int (*CommDestroy)(rcclComm_t);
  int (*CommCount)(rcclComm_t, int*);
  const char* (*GetErrorString)(int);
  char path[256];   // what dlopen resolved
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t);
  int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
};
Rccl g_rccl = {};
}  // namespace

int ics_set_error(int code, const char* fmt, ...);   // ics_api.hip

struct ics_group {
  int rank, world, device;
  bool local;        // world == 1: no communicator (ICS_GROUP_FORCE_RCCL=1 builds one anyway -- plumbing test on a 1-GPU box)
  rcclComm_t comm;
  int nranks;        // ncclCommCount of the communicator
  hipStream_t stream;
  double* dbuf;      // device staging: (world + 1) * ICS_GROUP_MAX_COUNT doubles
};
#define ICS_GROUP_MAX_COUNT 49152   /* doubles per call: >= 3 * 127^2, the PSF-gradient sums of the largest PSF in ONE all-reduce */

static int load_rccl() {
  if (g_rccl.so) return ICS_OK;
  const char* names[] = {getenv("ICS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  const char* used = "";
  for (const char* n : names) if (n && (so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) { used = n; break; }
  if (!so) return ics_set_error(ICS_ENODEV, "librccl.so not found (%s); set ICS_RCCL_LIB", dlerror());
  Rccl r = {};
  r.so = so;
  snprintf(r.path, sizeof r.path, "%s", used);
  r.GetUniqueId = (int (*)(rcclUniqueId*))dlsym(so, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))dlsym(so, "ncclCommInitRank");
  r.CommDestroy = (int (*)(rcclComm_t))dlsym(so, "ncclCommDestroy");
  r.CommCount = (int (*)(rcclComm_t, int*))dlsym(so, "ncclCommCount");
  r.GetErrorString = (const char* (*)(int))dlsym(so, "ncclGetErrorString");
  r.AllReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(so, "ncclAllReduce");
  r.AllGather = (int (*)(const void*, void*, size_t, int, rcclComm_t, hipStream_t))dlsym(so, "ncclAllGather");
  r.Send = (int (*)(const void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(so, "ncclSend");
  r.Recv = (int (*)(void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(so, "ncclRecv");
  r.GroupStart = (int (*)())dlsym(so, "ncclGroupStart");
  r.GroupEnd = (int (*)())dlsym(so, "ncclGroupEnd");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.GetErrorString || !r.AllReduce || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
    dlclose(so);
    return ics_set_error(ICS_ENODEV, "librccl.so lacks an expected ncclXxx symbol");
  }
  {  // the file behind the handle (for ics_group_describe): the soname alone does not say which RCCL was loaded
    Dl_info info;
    if (dladdr((void*)r.GetUniqueId, &info) && info.dli_fname) snprintf(r.path, sizeof r.path, "%s", info.dli_fname);
  }
  g_rccl = r;
  return ICS_OK;
}

#define GHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return ics_set_error(ICS_EHIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)
#define GRCCL(x) do { int r_ = (x); if (r_ != RCCL_SUCCESS) return ics_set_error(ICS_EHIP, "%s failed: %s", #x, g_rccl.GetErrorString(r_)); } while (0)

extern "C" int ics_group_create(int device, int rank, int world, const char* rendezvous, int timeout_s, ics_group** out) {
  if (!out) return ics_set_error(ICS_EINVAL, "out is NULL");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return ics_set_error(ICS_EINVAL, "rank %d of %d", rank, world);
  ics_group* g = new ics_group();
  g->rank = rank; g->world = world; g->device = device; g->comm = nullptr; g->stream = nullptr; g->dbuf = nullptr; g->nranks = 0;
  const char* force = getenv("ICS_GROUP_FORCE_RCCL");
  g->local = world == 1 && !(force && force[0] == '1');
  if (g->local) { *out = g; return ICS_OK; }   // nothing to exchange: no RCCL, no device
  if (!rendezvous || !rendezvous[0]) { delete g; return ics_set_error(ICS_EINVAL, "a rendezvous file path is required for world > 1"); }
  int rc = load_rccl();
  if (rc != ICS_OK) { delete g; return rc; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    delete g;
    return ics_set_error(ICS_ENODEV, "rank %d wants device %d but %d HIP device(s) are visible: one GPU per rank is required", rank, device, n);
  }
  hipError_t he = hipSetDevice(device);
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipMalloc((void**)&g->dbuf, (size_t)(world + 1) * ICS_GROUP_MAX_COUNT * sizeof(double));
  if (he != hipSuccess) { ics_group_destroy(g); return ics_set_error(ICS_EHIP, "group setup on device %d: %s", device, hipGetErrorString(he)); }
  // unique id: rank 0 -> file (written under a temporary name, then renamed) -> the other ranks poll for it.  The path must be
  // unique to the launch (multi_gpu.rendezvous_path: master port + launcher pid + launcher start time): rank 0 removes whatever a
  // failed earlier launch left under that name BEFORE it publishes, and every failure path below removes the file again.
  rcclUniqueId id;
  memset(&id, 0, sizeof id);
  char tmp[1024];
  snprintf(tmp, sizeof tmp, "%s.tmp", rendezvous);
  if (rank == 0) {
    unlink(rendezvous); unlink(tmp);
    int r = g_rccl.GetUniqueId(&id);
    if (r != RCCL_SUCCESS) { ics_group_destroy(g); return ics_set_error(ICS_EHIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r)); }
    FILE* f = fopen(tmp, "wb");
    const bool ok = f && fwrite(&id, sizeof id, 1, f) == 1;
    const bool closed = f ? fclose(f) == 0 : false;
    if (!ok || !closed || rename(tmp, rendezvous) != 0) {
      const int en = errno;
      unlink(tmp); unlink(rendezvous);
      ics_group_destroy(g);
      return ics_set_error(ICS_EINVAL, "cannot write the rendezvous file %s: %s", rendezvous, strerror(en));
    }
  } else {
    bool got = false;
    for (int waited_ms = 0; waited_ms < timeout_s * 1000 && !got; waited_ms += 20) {
      FILE* f = fopen(rendezvous, "rb");
      if (f) { got = fread(&id, sizeof id, 1, f) == 1; fclose(f); }
      if (!got) usleep(20000);
    }
    if (!got) { ics_group_destroy(g); return ics_set_error(ICS_ESTATE, "rank %d: no RCCL id at %s after %d s (is rank 0 running?)", rank, rendezvous, timeout_s); }
  }
  int r = g_rccl.CommInitRank(&g->comm, world, id, rank);
  if (r != RCCL_SUCCESS) {
    g->comm = nullptr; ics_group_destroy(g);
    if (rank == 0) unlink(rendezvous);
    return ics_set_error(ICS_EHIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
  }
  {  // the communicator must span `world` ranks: a stale id or a partial group would otherwise go unnoticed
    int n = 0;
    r = g_rccl.CommCount(g->comm, &n);
    if (r != RCCL_SUCCESS || n != world) {
      ics_group_destroy(g);
      if (rank == 0) unlink(rendezvous);
      return ics_set_error(ICS_ESTATE, "RCCL communicator spans %d rank(s), expected %d", n, world);
    }
    g->nranks = n;
  }
  *out = g;
  // every rank has read the id once the communicator exists: the file can go
  if (rank == 0) unlink(rendezvous);
  return ICS_OK;
}

extern "C" void ics_group_destroy(ics_group* g) {
  if (!g) return;
  if (!g->local) {
    hipSetDevice(g->device);
    if (g->stream) hipStreamSynchronize(g->stream);
    if (g->comm) g_rccl.CommDestroy(g->comm);
    if (g->dbuf) hipFree(g->dbuf);
    if (g->stream) hipStreamDestroy(g->stream);
  }
  delete g;
}

extern "C" int ics_group_info(const ics_group* g, int* rank, int* world) {
  if (!g) return ics_set_error(ICS_EINVAL, "group is NULL");
  if (rank) *rank = g->rank;
  if (world) *world = g->world;
  return ICS_OK;
}

// recv[r * count + i] = send[i] of rank r.  count <= 64 doubles: these are per-job records, not frames.
extern "C" int ics_group_allgather(ics_group* g, const double* send, int count, double* recv) {
  if (!g || !send || !recv) return ics_set_error(ICS_EINVAL, "NULL argument");
  if (count < 1 || count > ICS_GROUP_MAX_COUNT) return ics_set_error(ICS_EINVAL, "count %d (1..%d)", count, ICS_GROUP_MAX_COUNT);
  if (g->local) { memcpy(recv, send, (size_t)count * sizeof(double)); return ICS_OK; }
  GHIP(hipSetDevice(g->device));
  double* dsend = g->dbuf;
  double* drecv = g->dbuf + ICS_GROUP_MAX_COUNT;
  GHIP(hipMemcpyAsync(dsend, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, g->stream));
  GRCCL(g_rccl.AllGather(dsend, drecv, (size_t)count, RCCL_FLOAT64, g->comm, g->stream));
  GHIP(hipMemcpyAsync(recv, drecv, (size_t)g->world * count * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  GHIP(hipStreamSynchronize(g->stream));
  return ICS_OK;
}

static int allreduce(ics_group* g, double* inout, int count, int op) {
  if (!g || !inout) return ics_set_error(ICS_EINVAL, "NULL argument");
  if (count < 1 || count > ICS_GROUP_MAX_COUNT) return ics_set_error(ICS_EINVAL, "count %d (1..%d)", count, ICS_GROUP_MAX_COUNT);
  if (g->local) return ICS_OK;
  GHIP(hipSetDevice(g->device));
  GHIP(hipMemcpyAsync(g->dbuf, inout, (size_t)count * sizeof(double), hipMemcpyHostToDevice, g->stream));
  GRCCL(g_rccl.AllReduce(g->dbuf, g->dbuf, (size_t)count, RCCL_FLOAT64, op, g->comm, g->stream));
  GHIP(hipMemcpyAsync(inout, g->dbuf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  GHIP(hipStreamSynchronize(g->stream));
  return ICS_OK;
}
extern "C" int ics_group_allreduce_max(ics_group* g, double* inout, int count) { return allreduce(g, inout, count, RCCL_MAX); }
// (host arrays; the row-band split's per-iteration collectives run in place on device buffers: ics_group_allreduce_device)
extern "C" int ics_group_allreduce_sum(ics_group* g, double* inout, int count) { return allreduce(g, inout, count, RCCL_SUM); }

extern "C" int ics_group_describe(const ics_group* g, int* backend, int* nranks, char* lib, size_t lib_len) {
  if (!g) return ics_set_error(ICS_EINVAL, "group is NULL");
  if (backend) *backend = g->local ? 0 : 1;
  if (nranks) *nranks = g->local ? 1 : g->nranks;
  if (lib && lib_len) { snprintf(lib, lib_len, "%s", g->local ? "" : g_rccl.path); }
  return ICS_OK;
}

// Point-to-point rows between the frames of two band jobs on different ranks (lib/banded.py, rank mode): `count` floats from
// device pointer `send` to rank send_peer and / or into `recv` from rank recv_peer (peer < 0: that side is absent), one
// ncclGroup so that neighbours exchanging both ways cannot deadlock.  The caller has drained the stream that produced `send`;
// the call returns when the transfer is complete.  (ics_rl_exchange_rows in ics_api.hip resolves frame rows to pointers.)
int ics_group_sendrecv_device(ics_group* g, const float* send, size_t send_count, int send_peer, float* recv, size_t recv_count, int recv_peer) {
  if (!g) return ics_set_error(ICS_EINVAL, "group is NULL");
  if (g->local) return ics_set_error(ICS_ESTATE, "a one-rank local group has no peers");
  if ((send_peer >= g->world) || (recv_peer >= g->world)) return ics_set_error(ICS_EINVAL, "peer out of range");
  GHIP(hipSetDevice(g->device));
  GRCCL(g_rccl.GroupStart());
  int r1 = RCCL_SUCCESS, r2 = RCCL_SUCCESS;
  if (send_peer >= 0 && send_count) r1 = g_rccl.Send(send, send_count, RCCL_FLOAT32, send_peer, g->comm, g->stream);
  if (recv_peer >= 0 && recv_count) r2 = g_rccl.Recv(recv, recv_count, RCCL_FLOAT32, recv_peer, g->comm, g->stream);
  const int r3 = g_rccl.GroupEnd();
  if (r1 != RCCL_SUCCESS || r2 != RCCL_SUCCESS || r3 != RCCL_SUCCESS)
    return ics_set_error(ICS_EHIP, "ncclSend / ncclRecv: %s", g_rccl.GetErrorString(r1 != RCCL_SUCCESS ? r1 : (r2 != RCCL_SUCCESS ? r2 : r3)));
  GHIP(hipStreamSynchronize(g->stream));
  return ICS_OK;
}

// In-place all-reduce of a DEVICE buffer on the caller's stream (no staging, no host synchronisation): the step-size keys and the
// PSF-gradient sums of the row-band split (ics_rl_allreduce_keys / ics_rl_allreduce_gradk in ics_api.hip).  kind: 0 = uint32 max,
// 1 = float64 sum.  A local one-rank group has nothing to do.
int ics_group_allreduce_device(ics_group* g, void* buf, size_t count, int kind, hipStream_t stream) {
  if (!g || !buf) return ics_set_error(ICS_EINVAL, "NULL argument");
  if (g->local) return ICS_OK;
  GHIP(hipSetDevice(g->device));
  GRCCL(g_rccl.AllReduce(buf, buf, count, kind == 0 ? RCCL_UINT32 : RCCL_FLOAT64, kind == 0 ? RCCL_MAX : RCCL_SUM, g->comm, stream));
  return ICS_OK;
}

int ics_group_info_local(const ics_group* g) { return g && g->local ? 1 : 0; }
int ics_group_device(const ics_group* g) { return g ? g->device : -1; }

// all ranks have reached this call (an all-reduce of one double)
extern "C" int ics_group_barrier(ics_group* g) {
  double x = 0.0;
  return ics_group_allreduce_max(g, &x, 1);
}
