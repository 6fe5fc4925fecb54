// Frame-shard collective of the multi-GPU path: ONE all-gather of the [B/G,300,6] detections per batch over
// RCCL (xGMI). The reference has no multi-GPU path; frames are independent inside `.predict` (reference yolo_seg/app.py:85-91).
//
// bench.py and parallel.py use torch.distributed's "nccl" backend (= RCCL) for this; the entry points below give a host that
// does not run PyTorch the same collective through the C-ABI. librccl is resolved at run time with dlopen (whichever copy the
// process already holds - PyTorch ships its own - is reused), so libyolop.so itself has no link-time dependency on it.
#include "common.h"
#include "../../include/yolop.h"
#include <dlfcn.h>
#include <cstring>

namespace {
struct Id128 { char b[128]; };      // ncclUniqueId is passed BY VALUE
}  // namespace

struct yp_comm { void* comm = nullptr; int rank = 0, world = 1, device = 0; };

namespace {

typedef int (*fn_uid)(Id128*);
typedef int (*fn_init)(void**, int, Id128, int);
typedef int (*fn_ag)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_err)(int);
struct Api { void* lib = nullptr; fn_uid uid = nullptr; fn_init init = nullptr; fn_ag ag = nullptr; fn_destroy destroy = nullptr; fn_err err = nullptr; };

static Api& api() {
    static Api a;
    if (a.lib) return a;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) return a;
    a.uid = (fn_uid)dlsym(a.lib, "ncclGetUniqueId");
    a.init = (fn_init)dlsym(a.lib, "ncclCommInitRank");
    a.ag = (fn_ag)dlsym(a.lib, "ncclAllGather");
    a.destroy = (fn_destroy)dlsym(a.lib, "ncclCommDestroy");
    a.err = (fn_err)dlsym(a.lib, "ncclGetErrorString");
    if (!a.uid || !a.init || !a.ag || !a.destroy) { dlclose(a.lib); a = Api{}; }
    return a;
}

}  // namespace

extern "C" {

int yp_fail_public(int code, const char* msg);     // engine.hip: sets yp_last_error

int yp_comm_unique_id(void* id128) {
    if (!id128) return yp_fail_public(YP_ERR_ARG, "yp_comm_unique_id: null buffer");
    Api& a = api();
    if (!a.lib) return yp_fail_public(YP_ERR_STATE, "yp_comm: librccl.so could not be loaded");
    Id128 id;
    const int rc = a.uid(&id);
    if (rc != 0) return yp_fail_public(YP_ERR_HIP, a.err ? a.err(rc) : "ncclGetUniqueId failed");
    memcpy(id128, id.b, 128);
    return YP_OK;
}

int yp_comm_create(const void* id128, int rank, int world, int device, yp_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return yp_fail_public(YP_ERR_ARG, "yp_comm_create: bad arguments");
    Api& a = api();
    if (!a.lib) return yp_fail_public(YP_ERR_STATE, "yp_comm: librccl.so could not be loaded");
    if (hipSetDevice(device) != hipSuccess) return yp_fail_public(YP_ERR_HIP, "yp_comm_create: hipSetDevice failed");
    Id128 id;
    memcpy(id.b, id128, 128);
    void* comm = nullptr;
    const int rc = a.init(&comm, world, id, rank);
    if (rc != 0) return yp_fail_public(YP_ERR_HIP, a.err ? a.err(rc) : "ncclCommInitRank failed");
    yp_comm* c = new yp_comm();
    c->comm = comm; c->rank = rank; c->world = world; c->device = device;
    *out = c;
    return YP_OK;
}

int yp_allgather(yp_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev) return yp_fail_public(YP_ERR_ARG, "yp_allgather: bad arguments");
    Api& a = api();
    const int rc = a.ag(send_dev, recv_dev, bytes_per_rank, /* ncclInt8 / ncclChar */ 0, c->comm, (hipStream_t)stream);
    if (rc != 0) return yp_fail_public(YP_ERR_HIP, a.err ? a.err(rc) : "ncclAllGather failed");
    return YP_OK;
}

int yp_comm_destroy(yp_comm* c) {
    if (!c) return YP_OK;
    Api& a = api();
    if (a.lib && c->comm) (void)a.destroy(c->comm);
    delete c;
    return YP_OK;
}

}  // extern "C"
