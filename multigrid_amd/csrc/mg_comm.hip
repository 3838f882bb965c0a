// Rank-to-rank data movement of the row-slab V-cycle over RCCL (xGMI between the GPUs of a node): the halo exchange
// of the distributed levels, the all-gather of the first replicated level's right-hand side and the MAX-reduction of
// the residual norm, issued as raw RCCL calls on the caller's HIP stream -- in order with the kernels of that stream,
// no event hops, no per-call bookkeeping.
//
// Replaces Buffers.__init__/exchange_buffers (persistent Send_init/Recv_init requests set up once per array and
// level, then Startall/Waitall per exchange; core/parallel_multigrid.py:19-97 of the reference), the four Bcasts of
// gather_blocks (:117-138) and the allreduce of the stop test (:342-345).
//
// The library does not link RCCL: a process has exactly one RCCL (the one torch.distributed loaded, whose
// communicator mg_comm_attach borrows), so the entry points are looked up in the copy that is already mapped.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h> // types and enumerations only; every function is called through RcclApi

#include "mg_common.h"

namespace {

struct RcclApi {
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *);
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GetVersion)(int *);
    const char *(*GetErrorString)(ncclResult_t);
    void *handle;
    bool ready;
};

RcclApi g_api;
std::mutex g_api_mutex;

// Look the entry points up: in `path` when given (dlopen of a library that is already mapped returns that mapping),
// else in whatever the process has loaded under RCCL's usual names, else among the global symbols.
int load_api(const char *path)
{
    std::lock_guard<std::mutex> lk(g_api_mutex);
    if (g_api.ready) return 0;
    void *h = nullptr;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        MG_REQUIRE(h, "mg_comm: cannot open the RCCL library '%s': %s", path, dlerror());
    } else {
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (!h) h = dlopen(nullptr, RTLD_NOW); // the global scope
        MG_REQUIRE(h, "mg_comm: no RCCL library is loaded in this process");
    }
    struct { const char *name; void **slot; } syms[] = {
        {"ncclGroupStart", (void **)&g_api.GroupStart},     {"ncclGroupEnd", (void **)&g_api.GroupEnd},
        {"ncclSend", (void **)&g_api.Send},                 {"ncclRecv", (void **)&g_api.Recv},
        {"ncclAllGather", (void **)&g_api.AllGather},       {"ncclAllReduce", (void **)&g_api.AllReduce},
        {"ncclCommCount", (void **)&g_api.CommCount},       {"ncclCommUserRank", (void **)&g_api.CommUserRank},
        {"ncclCommCuDevice", (void **)&g_api.CommCuDevice}, {"ncclGetUniqueId", (void **)&g_api.GetUniqueId},
        {"ncclCommInitRank", (void **)&g_api.CommInitRank}, {"ncclCommDestroy", (void **)&g_api.CommDestroy},
        {"ncclGetVersion", (void **)&g_api.GetVersion},     {"ncclGetErrorString", (void **)&g_api.GetErrorString},
    };
    for (auto &s : syms) {
        *s.slot = dlsym(h, s.name);
        if (!*s.slot) {
            mg_set_error("mg_comm: the RCCL library%s%s does not export %s", path ? " " : "", path ? path : "", s.name);
            return 1;
        }
    }
    g_api.handle = h;
    g_api.ready = true;
    return 0;
}

#define MG_NCCL(call)                                                                    \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) {                                                         \
            mg_set_error("%s failed: %s", #call, g_api.GetErrorString(r_));              \
            return 3;                                                                    \
        }                                                                                \
    } while (0)

struct Msg {
    double *ptr;
    size_t count;
    int peer;
};

} // namespace

struct mg_comm {
    ncclComm_t comm;
    int rank, world, device;
    bool owned;
};

struct mg_halo {
    mg_comm *c;
    std::vector<Msg> sends, recvs;
};

namespace {

int check_comm(const mg_comm *c, const char *fn)
{
    MG_REQUIRE(c && c->comm, "%s: null communicator", fn);
    int d = -1;
    MG_HIP(hipGetDevice(&d));
    MG_REQUIRE(d == c->device, "%s: the communicator lives on device %d, the current device is %d", fn, c->device, d);
    return 0;
}

int finish_comm(ncclComm_t comm, bool owned, mg_comm **out)
{
    mg_comm *c = new (std::nothrow) mg_comm();
    MG_REQUIRE(c, "mg_comm: out of host memory");
    c->comm = comm, c->owned = owned;
    ncclResult_t r = g_api.CommCount(comm, &c->world);
    if (r == ncclSuccess) r = g_api.CommUserRank(comm, &c->rank);
    if (r == ncclSuccess) r = g_api.CommCuDevice(comm, &c->device);
    if (r != ncclSuccess) {
        mg_set_error("mg_comm: the communicator does not answer (%s)", g_api.GetErrorString(r));
        delete c;
        return 3;
    }
    *out = c;
    return 0;
}

} // namespace

extern "C" int mg_comm_attach(void *nccl_comm, const char *rccl_path, mg_comm **out)
{
    MG_REQUIRE(out, "mg_comm_attach: null output");
    *out = nullptr;
    MG_REQUIRE(nccl_comm, "mg_comm_attach: null communicator");
    if (load_api(rccl_path)) return 1;
    return finish_comm((ncclComm_t)nccl_comm, false, out);
}

extern "C" int mg_comm_unique_id(const char *rccl_path, void *id128)
{
    MG_REQUIRE(id128, "mg_comm_unique_id: null output");
    if (load_api(rccl_path)) return 1;
    static_assert(sizeof(ncclUniqueId) == 128, "the id is passed around as 128 bytes");
    ncclUniqueId id;
    MG_NCCL(g_api.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int mg_comm_init(const void *id128, int rank, int world, const char *rccl_path, mg_comm **out)
{
    MG_REQUIRE(out, "mg_comm_init: null output");
    *out = nullptr;
    MG_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, "mg_comm_init: bad rank %d of %d", rank, world);
    if (load_api(rccl_path)) return 1;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    MG_NCCL(g_api.CommInitRank(&comm, world, id, rank));
    const int rc = finish_comm(comm, true, out);
    if (rc) (void)g_api.CommDestroy(comm);
    return rc;
}

extern "C" int mg_comm_info(mg_comm *c, int *rank, int *world, int *rccl_version)
{
    MG_REQUIRE(c, "mg_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) MG_NCCL(g_api.GetVersion(rccl_version));
    return 0;
}

extern "C" int mg_comm_destroy(mg_comm *c)
{
    if (!c) return 0;
    int rc = 0;
    if (c->owned && c->comm) {
        ncclResult_t r = g_api.CommDestroy(c->comm);
        if (r != ncclSuccess) {
            mg_set_error("ncclCommDestroy failed: %s", g_api.GetErrorString(r));
            rc = 3;
        }
    }
    delete c;
    return rc;
}

extern "C" int mg_halo_prepare(mg_comm *c, int nsend, double *const *send_ptr, const int64_t *send_count, const int *send_peer,
                               int nrecv, double *const *recv_ptr, const int64_t *recv_count, const int *recv_peer, mg_halo **out)
{
    MG_REQUIRE(out, "mg_halo_prepare: null output");
    *out = nullptr;
    MG_REQUIRE(c && c->comm, "mg_halo_prepare: null communicator");
    MG_REQUIRE(nsend >= 0 && nrecv >= 0 && nsend + nrecv <= 4096, "mg_halo_prepare: %d sends and %d receives", nsend, nrecv);
    MG_REQUIRE((nsend == 0 || (send_ptr && send_count && send_peer)) && (nrecv == 0 || (recv_ptr && recv_count && recv_peer)),
               "mg_halo_prepare: null message table");
    mg_halo *h = new (std::nothrow) mg_halo();
    MG_REQUIRE(h, "mg_halo_prepare: out of host memory");
    h->c = c;
    for (int dir = 0; dir < 2; ++dir) {
        const int n = dir ? nrecv : nsend;
        double *const *ptr = dir ? recv_ptr : send_ptr;
        const int64_t *cnt = dir ? recv_count : send_count;
        const int *peer = dir ? recv_peer : send_peer;
        for (int k = 0; k < n; ++k) {
            if (!ptr[k] || cnt[k] <= 0 || peer[k] < 0 || peer[k] >= c->world) {
                mg_set_error("mg_halo_prepare: %s %d: pointer %p, %lld doubles, peer %d of %d ranks", dir ? "receive" : "send", k,
                             (void *)ptr[k], (long long)cnt[k], peer[k], c->world);
                delete h;
                return 1;
            }
            (dir ? h->recvs : h->sends).push_back(Msg{ptr[k], (size_t)cnt[k], peer[k]});
        }
    }
    // a message must not land in rows another message of the same exchange reads or writes
    for (const Msg &r : h->recvs) {
        for (const Msg &s : h->sends)
            if (r.ptr < s.ptr + s.count && s.ptr < r.ptr + r.count) {
                mg_set_error("mg_halo_prepare: a received block overlaps a sent one");
                delete h;
                return 1;
            }
        for (const Msg &q : h->recvs)
            if (&q != &r && r.ptr < q.ptr + q.count && q.ptr < r.ptr + r.count) {
                mg_set_error("mg_halo_prepare: two received blocks overlap");
                delete h;
                return 1;
            }
    }
    *out = h;
    return 0;
}

extern "C" int mg_halo_exchange(mg_halo *h, void *stream)
{
    MG_REQUIRE(h, "mg_halo_exchange: null exchange");
    if (check_comm(h->c, "mg_halo_exchange")) return 1;
    if (h->sends.empty() && h->recvs.empty()) return 0;
    hipStream_t st = (hipStream_t)stream;
    ncclComm_t comm = h->c->comm;
    MG_NCCL(g_api.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (const Msg &m : h->sends)
        if (r == ncclSuccess) r = g_api.Send(m.ptr, m.count, ncclDouble, m.peer, comm, st);
    for (const Msg &m : h->recvs)
        if (r == ncclSuccess) r = g_api.Recv(m.ptr, m.count, ncclDouble, m.peer, comm, st);
    const ncclResult_t e = g_api.GroupEnd(); // always closed, so that a failed call leaves no open group behind
    if (r != ncclSuccess || e != ncclSuccess) {
        mg_set_error("mg_halo_exchange: %s", g_api.GetErrorString(r != ncclSuccess ? r : e));
        return 3;
    }
    return 0;
}

extern "C" int mg_halo_destroy(mg_halo *h)
{
    delete h;
    return 0;
}

extern "C" int mg_allgather_rows(mg_comm *c, const double *in, double *out, int64_t count, void *stream)
{
    if (check_comm(c, "mg_allgather_rows")) return 1;
    MG_REQUIRE(in && out && count > 0, "mg_allgather_rows: null array or empty block");
    MG_REQUIRE(in == out + (i64)c->rank * count || in + count <= out || out + (i64)c->world * count <= in,
               "mg_allgather_rows: the block to send lies inside the gathered array, but not at this rank's place");
    MG_NCCL(g_api.AllGather(in, out, (size_t)count, ncclDouble, c->comm, (hipStream_t)stream));
    return 0;
}

extern "C" int mg_allreduce_max(mg_comm *c, uint64_t *words, int64_t count, void *stream)
{
    if (check_comm(c, "mg_allreduce_max")) return 1;
    MG_REQUIRE(words && count > 0, "mg_allreduce_max: null array or no words");
    MG_NCCL(g_api.AllReduce(words, words, (size_t)count, ncclUint64, ncclMax, c->comm, (hipStream_t)stream));
    return 0;
}
