// comm.cpp -- one-time broadcast between the contexts of a multi-GPU host process (aipt_comm_*, include/aiptd.h).
//
// The path shards at frame granularity: one context per GPU, contiguous frame chunks, and exactly one exchange --
// rank 0's packed scene (geometry + BVH, built once) and weight blob go to every other rank before the first frame.  The
// reference has nothing to cite here (single GPU).  Two transports behind one call:
//   * RCCL (ncclCommInitAll + grouped ncclBroadcast over xGMI) when every context sits on its own GPU.  librccl is dlopen'ed,
//     so the core library carries no link-time dependency on it;
//   * an in-process hipMemcpy shim when several contexts share a GPU (`aiptd --gpus 1 --ranks 8`: the way a one-GPU box checks
//     that sharded rendering is byte-identical, SURVEY 8e "Test without 8 GPUs").
#include "internal.h"

#include <dlfcn.h>

#include <set>

namespace {

typedef void* ncclComm_t_;
typedef int ncclResult_t_;
struct Rccl {
    void* lib = nullptr;
    ncclResult_t_ (*CommInitAll)(ncclComm_t_*, int, const int*) = nullptr;
    ncclResult_t_ (*CommDestroy)(ncclComm_t_) = nullptr;
    ncclResult_t_ (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    ncclResult_t_ (*GroupStart)() = nullptr;
    ncclResult_t_ (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t_) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Broadcast || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
constexpr int NCCL_UINT8 = 1;      // ncclUint8, rccl.h

}  // namespace

struct aipt_comm {
    std::vector<aipt_ctx*> ctxs;
    std::vector<ncclComm_t_> comms;   // empty: in-process shim
    std::string err;
};

extern "C" {

int aipt_comm_create(aipt_ctx* const* ctxs, int n, int force_shim, aipt_comm** out) {
    if (!out) return AIPT_E_INVALID;
    *out = nullptr;
    if (!ctxs || n < 1) return AIPT_E_INVALID;
    for (int r = 0; r < n; r++) if (!ctxs[r]) return AIPT_E_INVALID;
    aipt_comm* c = new (std::nothrow) aipt_comm();
    if (!c) return AIPT_E_NOMEM;
    c->ctxs.assign(ctxs, ctxs + n);
    std::set<int> devs;
    for (int r = 0; r < n; r++) devs.insert(ctxs[r]->device);
    const bool distinct = (int)devs.size() == n;
    if (distinct && !force_shim) {
        std::string err;
        if (!g_rccl.load(err)) { aipt::fail(ctxs[0], AIPT_E_STATE, "aipt_comm_create: %s", err.c_str()); delete c; return AIPT_E_STATE; }
        std::vector<int> devlist(n);
        for (int r = 0; r < n; r++) devlist[r] = ctxs[r]->device;
        c->comms.resize(n);
        const ncclResult_t_ rc = g_rccl.CommInitAll(c->comms.data(), n, devlist.data());
        if (rc) {
            aipt::fail(ctxs[0], AIPT_E_HIP, "ncclCommInitAll: %s", g_rccl.GetErrorString(rc));
            delete c;
            return AIPT_E_HIP;
        }
    }
    *out = c;
    return AIPT_OK;
}

int aipt_comm_is_rccl(const aipt_comm* c) { return c && !c->comms.empty(); }

int aipt_comm_broadcast(aipt_comm* c, void* const* d_bufs, size_t bytes, int root) {
    if (!c || !d_bufs) return AIPT_E_INVALID;
    const int n = (int)c->ctxs.size();
    if (root < 0 || root >= n) return aipt::fail(c->ctxs[0], AIPT_E_INVALID, "aipt_comm_broadcast: root %d of %d", root, n);
    for (int r = 0; r < n; r++) if (!d_bufs[r]) return aipt::fail(c->ctxs[0], AIPT_E_INVALID, "aipt_comm_broadcast: buffer %d is NULL", r);
    if (!c->comms.empty()) {
        ncclResult_t_ rc = g_rccl.GroupStart();
        for (int r = 0; r < n && !rc; r++) {
            hipSetDevice(c->ctxs[r]->device);
            rc = g_rccl.Broadcast(d_bufs[root], d_bufs[r], bytes, NCCL_UINT8, root, c->comms[r], c->ctxs[r]->stream);
        }
        const ncclResult_t_ rc2 = g_rccl.GroupEnd();
        if (rc || rc2) return aipt::fail(c->ctxs[0], AIPT_E_HIP, "ncclBroadcast: %s", g_rccl.GetErrorString(rc ? rc : rc2));
        for (int r = 0; r < n; r++) {
            hipSetDevice(c->ctxs[r]->device);
            const hipError_t e = hipStreamSynchronize(c->ctxs[r]->stream);
            if (e != hipSuccess) return aipt::fail(c->ctxs[r], AIPT_E_HIP, "broadcast sync: %s", hipGetErrorString(e));
        }
        return AIPT_OK;
    }
    // shim: ranks share GPUs inside one process -- plain copies from the root's buffer
    hipSetDevice(c->ctxs[root]->device);
    hipError_t e = hipStreamSynchronize(c->ctxs[root]->stream);
    for (int r = 0; r < n && e == hipSuccess; r++) {
        if (r == root) continue;
        e = hipMemcpy(d_bufs[r], d_bufs[root], bytes, hipMemcpyDeviceToDevice);    // peer copy when the devices differ
    }
    if (e != hipSuccess) return aipt::fail(c->ctxs[0], AIPT_E_HIP, "broadcast shim: %s", hipGetErrorString(e));
    return AIPT_OK;
}

void aipt_comm_destroy(aipt_comm* c) {
    if (!c) return;
    for (ncclComm_t_ k : c->comms) if (k) g_rccl.CommDestroy(k);
    delete c;
}

int aipt_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

}  // extern "C"
