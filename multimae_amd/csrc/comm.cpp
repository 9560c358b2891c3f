// RCCL driver of the C ABI: communicator set-up and the bucketed gradient all-reduce of the data-parallel step
// (the reference reaches NCCL through torch's DistributedDataParallel: run_pretraining_multimae.py:224-226, utils/dist.py:94-107).
//
// librccl is not a link-time dependency: its entry points are resolved when the first communicator is created, from the
// copy already loaded into the process (PyTorch ships and loads its own librccl.so; two copies in one process would each
// keep their own state), else from librccl.so on the loader path.  Calls only enqueue on the caller's stream.
#include <dlfcn.h>
#include <link.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

#include <mutex>

#include "../../include/mmae.h"

void mmae_set_error(const char* fmt, ...);

namespace {

struct UniqueId { char internal[MMAE_COMM_ID_BYTES]; };  // ncclUniqueId
typedef void* Comm;                                      // ncclComm_t
enum { kSum = 0, kF32 = 7, kBf16 = 9 };                  // ncclSum, ncclFloat32, ncclBfloat16

struct Rccl {
    int (*GetUniqueId)(UniqueId*);
    int (*CommInitRank)(Comm*, int, UniqueId, int);
    int (*CommDestroy)(Comm);
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t);
    int (*CommCount)(Comm, int*);
    int (*CommUserRank)(Comm, int*);
    const char* (*GetErrorString)(int);
    bool ok;
};
Rccl g_rccl = {};
std::mutex g_mutex;

template <typename F>
bool resolve(void* lib, const char* name, F& out) {
    void* sym = lib ? dlsym(lib, name) : nullptr;
    if (!sym) sym = dlsym(RTLD_DEFAULT, name);
    out = reinterpret_cast<F>(sym);
    return sym != nullptr;
}

// Path of a librccl already mapped into the process, whatever scope it was loaded with (a Python extension's
// dependencies are local to it, so dlsym(RTLD_DEFAULT) does not see them).
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    const char* name = info->dlpi_name;
    const char* base = name ? strrchr(name, '/') : nullptr;
    base = base ? base + 1 : name;
    if (base && strncmp(base, "librccl.so", 10) == 0) {
        strncpy(static_cast<char*>(out), name, 1023);
        return 1;
    }
    return 0;
}

int load_rccl() {
    std::lock_guard<std::mutex> guard(g_mutex);
    if (g_rccl.ok) return MMAE_OK;
    void* lib = nullptr;
    char loaded[1024] = "";
    if (dl_iterate_phdr(find_loaded_rccl, loaded) && loaded[0]) {
        lib = dlopen(loaded, RTLD_NOW | RTLD_NOLOAD);  // a handle to that very copy: one RCCL state per process
        if (!lib) {
            mmae_set_error("comm: cannot take a handle to the loaded %s: %s", loaded, dlerror());
            return MMAE_ELAUNCH;
        }
    } else if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {  // no copy in the process yet
        lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) {
            mmae_set_error("comm: librccl.so is neither loaded in this process nor on the loader path: %s", dlerror());
            return MMAE_ELAUNCH;
        }
    }
    int found = 0;
    found += resolve(lib, "ncclGetUniqueId", g_rccl.GetUniqueId);
    found += resolve(lib, "ncclCommInitRank", g_rccl.CommInitRank);
    found += resolve(lib, "ncclCommDestroy", g_rccl.CommDestroy);
    found += resolve(lib, "ncclAllReduce", g_rccl.AllReduce);
    found += resolve(lib, "ncclBroadcast", g_rccl.Broadcast);
    found += resolve(lib, "ncclGetErrorString", g_rccl.GetErrorString);
    found += resolve(lib, "ncclCommCount", g_rccl.CommCount);
    found += resolve(lib, "ncclCommUserRank", g_rccl.CommUserRank);
    if (found != 8) {
        mmae_set_error("comm: librccl.so lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / Broadcast / "
                       "GetErrorString / CommCount / CommUserRank");
        return MMAE_ELAUNCH;
    }
    g_rccl.ok = true;
    return MMAE_OK;
}

struct Handle { Comm comm; int rank, world; };

int fail(const char* what, int rc) {
    mmae_set_error("comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return MMAE_ELAUNCH;
}

int dtype_of(int dtype, int* out) {
    if (dtype == MMAE_COMM_F32) *out = kF32;
    else if (dtype == MMAE_COMM_BF16) *out = kBf16;
    else {
        mmae_set_error("comm: dtype %d is neither MMAE_COMM_F32 nor MMAE_COMM_BF16", dtype);
        return MMAE_EINVAL;
    }
    return MMAE_OK;
}

}  // namespace

extern "C" int mmae_comm_unique_id(void* id_host) {
    if (!id_host) { mmae_set_error("comm_unique_id: null pointer"); return MMAE_EINVAL; }
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId id;
    rc = g_rccl.GetUniqueId(&id);
    if (rc) return fail("ncclGetUniqueId", rc);
    memcpy(id_host, &id, sizeof(id));
    return MMAE_OK;
}

extern "C" int mmae_comm_init(const void* id_host, int rank, int world, void** handle) {
    if (!id_host || !handle) { mmae_set_error("comm_init: null pointer"); return MMAE_EINVAL; }
    if (world < 1 || rank < 0 || rank >= world) { mmae_set_error("comm_init: rank %d of %d", rank, world); return MMAE_EINVAL; }
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId id;
    memcpy(&id, id_host, sizeof(id));
    Comm comm = nullptr;
    rc = g_rccl.CommInitRank(&comm, world, id, rank);  // collective over the ranks; uses the calling thread's current device
    if (rc) return fail("ncclCommInitRank", rc);
    *handle = new Handle{comm, rank, world};
    return MMAE_OK;
}

extern "C" int mmae_comm_allreduce_bucket(void* handle, void* ptr, size_t count, int dtype, mmae_stream_t stream) {
    if (!handle || !ptr) { mmae_set_error("comm_allreduce_bucket: null pointer"); return MMAE_EINVAL; }
    int dt = 0, rc = dtype_of(dtype, &dt);
    if (rc) return rc;
    if (count == 0) return MMAE_OK;
    rc = g_rccl.AllReduce(ptr, ptr, count, dt, kSum, static_cast<Handle*>(handle)->comm, (hipStream_t)stream);
    return rc ? fail("ncclAllReduce", rc) : MMAE_OK;
}

extern "C" int mmae_comm_broadcast(void* handle, void* ptr, size_t count, int dtype, int root, mmae_stream_t stream) {
    if (!handle || !ptr) { mmae_set_error("comm_broadcast: null pointer"); return MMAE_EINVAL; }
    Handle* h = static_cast<Handle*>(handle);
    if (root < 0 || root >= h->world) { mmae_set_error("comm_broadcast: root %d of %d ranks", root, h->world); return MMAE_EINVAL; }
    int dt = 0, rc = dtype_of(dtype, &dt);
    if (rc) return rc;
    if (count == 0) return MMAE_OK;
    rc = g_rccl.Broadcast(ptr, ptr, count, dt, root, h->comm, (hipStream_t)stream);
    return rc ? fail("ncclBroadcast", rc) : MMAE_OK;
}

extern "C" int mmae_comm_info(void* handle, int* rank, int* world) {
    if (!handle || !rank || !world) { mmae_set_error("comm_info: null pointer"); return MMAE_EINVAL; }
    Handle* h = static_cast<Handle*>(handle);
    int rc = g_rccl.CommCount(h->comm, world);  // what the communicator itself holds, not what the caller passed in
    if (rc) return fail("ncclCommCount", rc);
    rc = g_rccl.CommUserRank(h->comm, rank);
    return rc ? fail("ncclCommUserRank", rc) : MMAE_OK;
}

extern "C" int mmae_comm_destroy(void* handle) {
    if (!handle) return MMAE_OK;
    Handle* h = static_cast<Handle*>(handle);
    const int rc = g_rccl.CommDestroy(h->comm);
    delete h;
    return rc ? fail("ncclCommDestroy", rc) : MMAE_OK;
}
