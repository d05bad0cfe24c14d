// RCCL communicator behind pepsgpu_comm_* / pepsgpu_allreduce (include/pepsgpu.h).
//
// The path has ONE exchange step: the sum over ranks of the energy / gradient accumulators, which the
// reference does with MPI on host tensors (monte_carlo_tools/statistics_tensor.h:37-79 MPIMeanTensor,
// mc_energy_grad_evaluator.h:292-310, exact_summation_energy_evaluator.h:252-280 MPI_Send/Recv + reduce).
// Here it is one ncclAllReduce on the HBM-resident accumulators over xGMI, on the engine's own stream
// (ordered behind the accumulation kernels, no host hop).  librccl is opened at first use so that the
// library loads (and its symbols can be checked) on a machine without the RCCL runtime initialised.
#pragma once
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>
#include "common.h"

namespace pepsgpu {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;

  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;

  // Opens librccl and resolves every entry point into LOCALS first; the handle and the pointers are published together,
  // only after every dlsym has succeeded (a partial table is never visible: a later call retries from scratch).
  static RcclApi &get() {
    static RcclApi api;
    if (!api.handle) {
      const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      void *h = nullptr;
      std::string why;
      for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        const char *e = dlerror();          // one call: dlerror() clears the message it returns
        why = e ? e : "?";
      }
      PG_REQUIRE(h != nullptr, 2, std::string("cannot open librccl: ") + why);
      RcclApi loc;
      const char *missing = nullptr;
      auto sym = [&](const char *s) {
        void *p = dlsym(h, s);
        if (!p && !missing) missing = s;
        return p;
      };
      loc.GetUniqueId = reinterpret_cast<decltype(loc.GetUniqueId)>(sym("ncclGetUniqueId"));
      loc.CommInitRank = reinterpret_cast<decltype(loc.CommInitRank)>(sym("ncclCommInitRank"));
      loc.CommDestroy = reinterpret_cast<decltype(loc.CommDestroy)>(sym("ncclCommDestroy"));
      loc.CommCount = reinterpret_cast<decltype(loc.CommCount)>(sym("ncclCommCount"));
      loc.AllReduce = reinterpret_cast<decltype(loc.AllReduce)>(sym("ncclAllReduce"));
      loc.Broadcast = reinterpret_cast<decltype(loc.Broadcast)>(sym("ncclBroadcast"));
      loc.GetErrorString = reinterpret_cast<decltype(loc.GetErrorString)>(sym("ncclGetErrorString"));
      if (missing) {
        const std::string m = std::string("librccl lacks ") + missing;
        dlclose(h);
        throw ::pepsgpu::Error(2, m);
      }
      loc.handle = h;
      api = loc;
    }
    return api;
  }
};

#define PG_CHECK_RCCL(expr)                                                                                    \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess) {                                                                                   \
      auto _es = ::pepsgpu::RcclApi::get().GetErrorString;                                                     \
      throw ::pepsgpu::Error(2, std::string("RCCL error ") + (_es ? _es(_r) : "?") + " in " #expr);            \
    }                                                                                                          \
  } while (0)

// One communicator per context (= per GPU / rank).  A context that never called init (one rank) reduces by the identity.
struct Comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  void *scratch = nullptr;      // device staging for host buffers
  size_t scratch_bytes = 0;

  ~Comm() { destroy(); }
  void destroy() {
    if (comm) { (void)RcclApi::get().CommDestroy(comm); comm = nullptr; }
    if (scratch) { (void)hipFree(scratch); scratch = nullptr; scratch_bytes = 0; }
    nranks = 1; rank = 0;
  }
  void init(int n, int r, const void *id128) {
    PG_REQUIRE(n >= 1 && r >= 0 && r < n, 1, "pepsgpu_comm_init: bad rank / size");
    destroy();
    PG_REQUIRE(n == 1 || id128 != nullptr, 1, "pepsgpu_comm_init: null unique id");
    if (id128) {   // a single rank with an id still builds a real communicator (exercises RCCL on a one-GPU box)
      static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
      ncclUniqueId id;
      memcpy(&id, id128, sizeof(id));
      PG_CHECK_RCCL(RcclApi::get().CommInitRank(&comm, n, id, r));
      int cnt = 0;
      PG_CHECK_RCCL(RcclApi::get().CommCount(comm, &cnt));
      PG_REQUIRE(cnt == n, 2, "pepsgpu_comm_init: communicator reports a different rank count");
    }
    nranks = n; rank = r;
  }
  // state broadcast ("one broadcast of the parameter buffer after each optimizer update", replaces the MPI_Bcast
  // of every site tensor in split_index_tps_impl.h:778-880): `bytes` of HBM at `buf` from rank `root` to all, on stream s
  void bcast(hipStream_t s, void *buf, size_t bytes, int root) {
    PG_REQUIRE(root >= 0 && root < nranks, 1, "pepsgpu_bcast_state: bad root");
    if (!comm || bytes == 0) return;   // one rank: the identity
    PG_CHECK_RCCL(RcclApi::get().Broadcast(buf, buf, bytes, ncclChar, root, comm, s));
  }
  // in-place all-reduce of n elements; dtype 0 f32, 1 f64, 2 i32; op 0 sum, 1 max; buf in HBM (on_device) or on the host
  void allreduce(hipStream_t s, void *buf, size_t n, int dtype, int op, bool on_device) {
    PG_REQUIRE(dtype >= 0 && dtype <= 2 && (op == 0 || op == 1), 1, "pepsgpu_allreduce: bad dtype / op");
    if (!comm || n == 0) return;   // single rank without a communicator: the identity
    const size_t esz = dtype == 1 ? 8 : 4;
    const ncclDataType_t dt = dtype == 0 ? ncclFloat32 : dtype == 1 ? ncclFloat64 : ncclInt32;
    const ncclRedOp_t ro = op == 0 ? ncclSum : ncclMax;
    if (on_device) {
      PG_CHECK_RCCL(RcclApi::get().AllReduce(buf, buf, n, dt, ro, comm, s));
      return;
    }
    if (scratch_bytes < n * esz) {
      if (scratch) (void)hipFree(scratch);
      scratch = nullptr; scratch_bytes = 0;
      PG_CHECK_HIP(hipMalloc(&scratch, n * esz));
      scratch_bytes = n * esz;
    }
    PG_CHECK_HIP(hipMemcpyAsync(scratch, buf, n * esz, hipMemcpyHostToDevice, s));
    PG_CHECK_RCCL(RcclApi::get().AllReduce(scratch, scratch, n, dt, ro, comm, s));
    PG_CHECK_HIP(hipMemcpyAsync(buf, scratch, n * esz, hipMemcpyDeviceToHost, s));
    PG_CHECK_HIP(hipStreamSynchronize(s));
  }
};

}  // namespace pepsgpu
