// Collective entry points of the C ABI (comm_{init,allreduce_sum,destroy}): a thin wrapper over RCCL
// (xGMI on one node) for hosts that drive libp2pgan_hip.so without PyTorch.  The Python host of this repository reaches the
// same RCCL through torch.distributed (backend "nccl", parallel.py) so that its collectives are ordered with torch's
// streams; both paths all-reduce the same buffers: the flat generator / discriminator gradient buffers in buckets, the
// loss scalars and, for the histogram model, the Hellinger sum of squares.
// librccl.so is resolved at run time (dlopen) and only when p2p_comm_init is called: the library has no link-time
// dependency on it.  A process that already carries an RCCL (PyTorch ships one) must not get a second runtime mapped beside
// it: dlopen only re-uses a mapped library when the requested name equals its SONAME or path, so the already-loaded instance
// is looked for first (RTLD_NOLOAD under every name it may carry); only when none is mapped is a fresh copy loaded, with
// RTLD_LOCAL so that its symbols do not interpose on anything else in the process.
#include "p2p_common.hpp"
#include <dlfcn.h>
#include <string.h>

namespace {
typedef struct { char internal[128]; } nccl_id_t;         // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
typedef int (*get_id_fn)(nccl_id_t*);
typedef int (*init_fn)(nccl_comm_t*, int, nccl_id_t, int);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*destroy_fn)(nccl_comm_t);
typedef const char* (*errstr_fn)(int);

struct Rccl {
    void* h = nullptr;
    get_id_fn get_id = nullptr; init_fn init = nullptr; allreduce_fn allreduce = nullptr; destroy_fn destroy = nullptr;
    errstr_fn errstr = nullptr;
};
Rccl g_rccl;

bool rccl_load() {
    if (g_rccl.h) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }       // the instance the process already has
    if (!h)
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { p2p_set_error("p2p_comm: cannot load librccl.so (%s)", dlerror()); return false; }
    g_rccl.get_id = (get_id_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.init = (init_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (errstr_fn)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_id || !g_rccl.init || !g_rccl.allreduce || !g_rccl.destroy) {
        p2p_set_error("p2p_comm: librccl.so lacks the NCCL API");
        dlclose(h);
        return false;
    }
    g_rccl.h = h;
    return true;
}
int rccl_fail(const char* what, int rc) {
    p2p_set_error("%s: %s", what, g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error");
    return rc ? rc : -1;
}
}  // namespace

// Rank 0 calls p2p_comm_unique_id and hands the 128 bytes to every rank (file, socket, MPI ...); every rank then calls
// p2p_comm_init with the same bytes on the device it will use (hipSetDevice first).  One communicator per process.
extern "C" int p2p_comm_unique_id(void* id_out_128_bytes) {
    P2P_REQUIRE(id_out_128_bytes, "p2p_comm_unique_id: null pointer");
    if (!rccl_load()) return -1;
    nccl_id_t id;
    int rc = g_rccl.get_id(&id);
    if (rc) return rccl_fail("p2p_comm_unique_id", rc);
    memcpy(id_out_128_bytes, &id, sizeof(id));
    return 0;
}

extern "C" int p2p_comm_init(const void* id_128_bytes, int rank, int world, void** comm_out) {
    P2P_REQUIRE(id_128_bytes && comm_out && world >= 1 && rank >= 0 && rank < world, "p2p_comm_init: bad arguments");
    if (!rccl_load()) return -1;
    nccl_id_t id;
    memcpy(&id, id_128_bytes, sizeof(id));
    nccl_comm_t c = nullptr;
    int rc = g_rccl.init(&c, world, id, rank);
    if (rc) return rccl_fail("p2p_comm_init", rc);
    *comm_out = c;
    return 0;
}

// In-place SUM all-reduce of n f32 values, ordered on `stream` like every other entry point of the library.
extern "C" int p2p_comm_allreduce_sum(void* comm, float* buf, long long n, void* stream) {
    P2P_REQUIRE(comm && buf && n > 0, "p2p_comm_allreduce_sum: bad arguments");
    P2P_REQUIRE(((uintptr_t)buf % 4) == 0, "p2p_comm_allreduce_sum: the buffer must be 4-byte aligned");
    if (!rccl_load()) return -1;
    int rc = g_rccl.allreduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream);
    if (rc) return rccl_fail("p2p_comm_allreduce_sum", rc);
    return 0;
}

extern "C" int p2p_comm_destroy(void* comm) {
    P2P_REQUIRE(comm, "p2p_comm_destroy: null communicator");
    if (!rccl_load()) return -1;
    int rc = g_rccl.destroy(comm);
    if (rc) return rccl_fail("p2p_comm_destroy", rc);
    return 0;
}
