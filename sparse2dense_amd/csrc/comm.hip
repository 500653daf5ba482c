// SyncBN statistics all-reduce on the compute stream (det3d/torchie/apis/train.py:281 converts every BatchNorm to apex
// SyncBatchNorm when training distributed: two tiny collectives per layer and step, 82 per step on this path).
// torch.distributed's NCCL process group runs its collectives on an internal stream with an event hand-off in both
// directions and ~20 us of host work per call; here the same RCCL library (already loaded by torch) is called directly
// on the stream the batch-norm kernels run on, between the reduction and the finalize kernel.  The symbols are resolved
// at run time (dlsym): without RCCL in the process the entry points report "unavailable" and the host side keeps using
// torch.distributed.  One communicator per process (one process per GPU).
#include "s2d_common.h"
#include <dlfcn.h>

namespace s2d {

struct RcclId {
    char internal[128];
};
typedef void *RcclComm;
typedef int (*GetUniqueIdFn)(RcclId *);
typedef int (*CommInitRankFn)(RcclComm *, int, RcclId, int);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, RcclComm, hipStream_t);
typedef int (*CommDestroyFn)(RcclComm);
typedef const char *(*GetErrorStringFn)(int);

static GetUniqueIdFn p_get_id = nullptr;
static CommInitRankFn p_init = nullptr;
static AllReduceFn p_allreduce = nullptr;
static CommDestroyFn p_destroy = nullptr;
static GetErrorStringFn p_errstr = nullptr;
static RcclComm g_comm = nullptr;
static int g_ranks = 0;

static void *g_lib = nullptr;   // handle of the RCCL the host framework already loaded (s2d_comm_load_library)

static bool comm_resolve() {
    if (p_get_id && p_init && p_allreduce && p_destroy) return true;
    // python loads its extension modules RTLD_LOCAL, so RTLD_DEFAULT does not see torch's RCCL: ask for the loaded
    // library by name (RTLD_NOLOAD: never a second copy), or take the handle the host passed in by path
    void *h = g_lib;
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = RTLD_DEFAULT;
    p_get_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    p_init = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    p_allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    p_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    p_errstr = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    return p_get_id && p_init && p_allreduce && p_destroy;
}

static int comm_fail(const char *what, int rc) {
    set_error("%s: RCCL error %d (%s)", what, rc, p_errstr ? p_errstr(rc) : "?");
    return S2D_ERR_COMM;
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_comm_load_library(const char *path) {
    S2D_CHECK_ARG(path, "comm_load_library: null path");
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);   // the file torch has mapped already: same handle, no second copy
    if (!h) {
        set_error("comm_load_library: %s", dlerror());
        return S2D_ERR_UNSUPPORTED;
    }
    g_lib = h;
    p_get_id = nullptr;
    return comm_resolve() ? S2D_OK : S2D_ERR_UNSUPPORTED;
}

extern "C" int s2d_comm_available(void) { return comm_resolve() ? 1 : 0; }

extern "C" int s2d_comm_unique_id(void *id128) {
    S2D_CHECK_ARG(id128, "comm_unique_id: null");
    if (!comm_resolve()) {
        set_error("comm_unique_id: RCCL is not loaded in this process");
        return S2D_ERR_UNSUPPORTED;
    }
    const int rc = p_get_id((RcclId *)id128);
    return rc ? comm_fail("comm_unique_id", rc) : S2D_OK;
}

extern "C" int s2d_comm_init(const void *id128, int nranks, int rank) {
    S2D_CHECK_ARG(id128 && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument");
    if (!comm_resolve()) {
        set_error("comm_init: RCCL is not loaded in this process");
        return S2D_ERR_UNSUPPORTED;
    }
    if (g_comm) {
        set_error("comm_init: already initialised");
        return S2D_ERR_INVALID_ARG;
    }
    RcclId id;
    memcpy(&id, id128, sizeof(id));
    RcclComm c = nullptr;
    const int rc = p_init(&c, nranks, id, rank);
    if (rc) return comm_fail("comm_init", rc);
    g_comm = c;
    g_ranks = nranks;
    return S2D_OK;
}

extern "C" int s2d_comm_ranks(void) { return g_comm ? g_ranks : 0; }

extern "C" int s2d_comm_shutdown(void) {
    if (g_comm && p_destroy) p_destroy(g_comm);
    g_comm = nullptr;
    g_ranks = 0;
    return S2D_OK;
}

// in-place sum over the ranks of `count` floats, enqueued on `stream`
extern "C" int s2d_comm_allreduce_sum_f32(float *buf, int64_t count, s2d_stream_t stream) {
    S2D_CHECK_ARG(buf && count > 0, "comm_allreduce: bad argument");
    if (!g_comm) {
        set_error("comm_allreduce: communicator not initialised");
        return S2D_ERR_INVALID_ARG;
    }
    const int rc = p_allreduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, g_comm, (hipStream_t)stream);
    return rc ? comm_fail("comm_allreduce", rc) : S2D_OK;
}
