// Communicator of the sharded decode: one rank per GPU, collectives on the library's own stream.
//
//   * RCCL backend: librccl.so is bound directly (dlopen + the declarations of <rccl/rccl.h>); the communicator is
//     created from a 128-byte unique id that the CALLER distributes (rank 0 obtains it from wfx_comm_unique_id and hands it
//     to the other processes by whatever means it has: wefax_amd/sharded.py uses a TCP socket on the loopback interface).
//     Every collective is enqueued on the context's stream: no host synchronisation, no other library in between.
//   * local backend: all `world` ranks live in THIS process (one context each, any device): a collective completes when the
//     last rank has posted its part, and is carried out with device-to-device copies.  It exists so that the N-rank form of
//     the decode -- every index of every exchange -- runs on a one-GPU box and in the driver's `-m gpu` test suite; a byte
//     count that does not match between the two ends of a transfer is an error there, not a hang.
//
// Only three collectives are needed (wfx_shard.hip): a personalised exchange (grouped send / recv: the transposes of the
// distributed transforms and the final gather), a sum all-reduce of uint32 histograms, and an all-gather of equal blocks.
#include <dlfcn.h>

#include <cstring>
#include <deque>
#include <mutex>

#include <rccl/rccl.h>

#include "wfx_internal.h"

namespace {

struct rccl_api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string err;
};

rccl_api g_rccl;
std::mutex g_rccl_mutex;

bool rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        g_rccl.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
        return false;
    }
#define BIND(name)                                                      \
    g_rccl.name = (decltype(g_rccl.name))dlsym(h, "nccl" #name);        \
    if (!g_rccl.name) {                                                 \
        g_rccl.err = "librccl.so lacks nccl" #name;                     \
        dlclose(h);                                                     \
        return false;                                                   \
    }
    BIND(GetUniqueId)
    BIND(CommInitRank)
    BIND(CommDestroy)
    BIND(GetErrorString)
    BIND(AllReduce)
    BIND(AllGather)
    BIND(Send)
    BIND(Recv)
    BIND(GroupStart)
    BIND(GroupEnd)
#undef BIND
    g_rccl.handle = h;
    return true;
}

// ---- local backend: the ranks of one process ---------------------------------------------------
struct local_op {
    int kind = 0;                       // 1 exchange, 2 allreduce, 3 allgather
    std::vector<wfx_xfer> list;         // exchange
    void *buf = nullptr;                // allreduce (in place) / allgather receive buffer
    const void *send = nullptr;         // allgather
    size_t count = 0;                   // allreduce words / allgather bytes per rank
};

}  // namespace

struct wfx_comm_group {
    int world = 0;
    int refs = 0;
    std::vector<wfx_ctx *> ctx;                     // the context each rank last posted with
    std::vector<std::deque<local_op>> pending;      // per rank, in program order
    wfx_devbuf scratch;                             // all-reduce accumulator
    int device = 0;
};

struct wfx_comm {
    int world = 1, rank = 0;
    ncclComm_t nccl = nullptr;          // RCCL backend
    wfx_comm_group *group = nullptr;    // local backend
    int device = 0;
};

__global__ void __launch_bounds__(256) comm_add_u32(unsigned *__restrict__ acc, const unsigned *__restrict__ x, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc[i] += x[i];
}

static int fail_nccl(wfx_ctx *ctx, ncclResult_t r, const char *what)
{
    return wfx_fail(ctx, WFX_ERR_COMM, "RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
}

#define WFX_NCCL(ctx, call)                                        \
    do {                                                           \
        ncclResult_t r_ = (call);                                  \
        if (r_ != ncclSuccess) return fail_nccl(ctx, r_, #call);   \
    } while (0)

// Carry out the collective at the head of every rank's queue (called when the last rank has posted it).
static int local_execute(wfx_comm_group *g, wfx_ctx *ctx)
{
    const int W = g->world;
    for (int r = 0; r < W; ++r)
        if (g->pending[r].empty()) return 0;                          // someone has not posted yet
    const int kind = g->pending[0].front().kind;
    for (int r = 1; r < W; ++r)
        if (g->pending[r].front().kind != kind)
            return wfx_fail(ctx, WFX_ERR_COMM, "local communicator: rank %d posted collective kind %d while rank 0 posted %d", r,
                            g->pending[r].front().kind, kind);
    // everything enqueued before the collective must have finished on every rank's stream
    for (int r = 0; r < W; ++r) {
        (void)hipSetDevice(g->ctx[r]->device);
        WFX_HIP(ctx, hipStreamSynchronize(g->ctx[r]->stream));
    }
    if (kind == 1) {
        // match the k-th send of src to dst with the k-th receive dst posted for src
        for (int src = 0; src < W; ++src) {
            for (int dst = 0; dst < W; ++dst) {
                std::vector<const wfx_xfer *> snd, rcv;
                for (const wfx_xfer &x : g->pending[src].front().list)
                    if (x.peer == dst && x.send_bytes) snd.push_back(&x);
                for (const wfx_xfer &x : g->pending[dst].front().list)
                    if (x.peer == src && x.recv_bytes) rcv.push_back(&x);
                if (snd.size() != rcv.size())
                    return wfx_fail(ctx, WFX_ERR_COMM, "exchange: rank %d sends %zu messages to rank %d, which expects %zu", src, snd.size(), dst,
                                    rcv.size());
                for (size_t k = 0; k < snd.size(); ++k) {
                    if (snd[k]->send_bytes != rcv[k]->recv_bytes)
                        return wfx_fail(ctx, WFX_ERR_COMM, "exchange: message %zu from rank %d to rank %d has %zu bytes, the receiver expects %zu", k,
                                        src, dst, snd[k]->send_bytes, rcv[k]->recv_bytes);
                    if (snd[k]->send != rcv[k]->recv)
                        WFX_HIP(ctx, hipMemcpy(rcv[k]->recv, snd[k]->send, snd[k]->send_bytes, hipMemcpyDeviceToDevice));
                }
            }
        }
    } else if (kind == 2) {
        const size_t n = g->pending[0].front().count;
        for (int r = 1; r < W; ++r)
            if (g->pending[r].front().count != n) return wfx_fail(ctx, WFX_ERR_COMM, "all-reduce: counts differ between ranks");
        (void)hipSetDevice(g->ctx[0]->device);
        WFX_TRY(wfx_reserve(g->ctx[0], g->scratch, n * 4));
        WFX_HIP(ctx, hipMemcpy(g->scratch.p, g->pending[0].front().buf, n * 4, hipMemcpyDeviceToDevice));
        for (int r = 1; r < W; ++r) {
            hipLaunchKernelGGL(comm_add_u32, dim3(wfx_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)0, (unsigned *)g->scratch.p,
                               (const unsigned *)g->pending[r].front().buf, n);
            WFX_HIP(ctx, hipGetLastError());
        }
        WFX_HIP(ctx, hipDeviceSynchronize());
        for (int r = 0; r < W; ++r) WFX_HIP(ctx, hipMemcpy(g->pending[r].front().buf, g->scratch.p, n * 4, hipMemcpyDeviceToDevice));
    } else {
        const size_t nb = g->pending[0].front().count;
        for (int r = 1; r < W; ++r)
            if (g->pending[r].front().count != nb) return wfx_fail(ctx, WFX_ERR_COMM, "all-gather: block sizes differ between ranks");
        for (int dst = 0; dst < W; ++dst)
            for (int src = 0; src < W; ++src)
                WFX_HIP(ctx, hipMemcpy((char *)g->pending[dst].front().buf + (size_t)src * nb, g->pending[src].front().send, nb, hipMemcpyDeviceToDevice));
    }
    WFX_HIP(ctx, hipDeviceSynchronize());
    for (int r = 0; r < W; ++r) g->pending[r].pop_front();
    return 0;
}

static int local_post(wfx_comm *c, wfx_ctx *ctx, local_op &&op)
{
    wfx_comm_group *g = c->group;
    g->ctx[c->rank] = ctx;
    g->pending[c->rank].push_back(std::move(op));
    // the ranks of a local group run in one thread, phase by phase: carry out whatever has become complete
    for (;;) {
        bool all = true;
        for (int r = 0; r < g->world; ++r) all = all && !g->pending[r].empty();
        if (!all) return 0;
        WFX_TRY(local_execute(g, ctx));
    }
}

// ---- internal interface (wfx_shard.hip) ---------------------------------------------------------------
int wfx_comm_world(const wfx_comm *c) { return c ? c->world : 1; }
int wfx_comm_rank(const wfx_comm *c) { return c ? c->rank : 0; }
bool wfx_comm_is_local(const wfx_comm *c) { return c && c->group != nullptr; }

int wfx_comm_exchange(wfx_comm *c, wfx_ctx *ctx, const wfx_xfer *list, int n)
{
    if (!c) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    for (int i = 0; i < n; ++i)
        if (list[i].peer < 0 || list[i].peer >= c->world) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "exchange: peer %d out of range", list[i].peer);
    if (c->group) {
        local_op op;
        op.kind = 1;
        op.list.assign(list, list + n);
        return local_post(c, ctx, std::move(op));
    }
    bool any_remote = false;
    for (int i = 0; i < n; ++i) any_remote = any_remote || list[i].peer != c->rank;
    // a rank's message to itself is a copy on the stream (normally the packer has already written it in place)
    for (int i = 0; i < n; ++i)
        if (list[i].peer == c->rank && list[i].send_bytes) {
            if (list[i].send_bytes != list[i].recv_bytes) return wfx_fail(ctx, WFX_ERR_COMM, "exchange: self message of %zu bytes into %zu", list[i].send_bytes, list[i].recv_bytes);
            if (list[i].send != list[i].recv)
                WFX_HIP(ctx, hipMemcpyAsync(list[i].recv, list[i].send, list[i].send_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
    if (!any_remote) return 0;
    WFX_NCCL(ctx, g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        if (list[i].peer == c->rank) continue;
        if (list[i].send_bytes) WFX_NCCL(ctx, g_rccl.Send(list[i].send, list[i].send_bytes, ncclUint8, list[i].peer, c->nccl, ctx->stream));
        if (list[i].recv_bytes) WFX_NCCL(ctx, g_rccl.Recv(list[i].recv, list[i].recv_bytes, ncclUint8, list[i].peer, c->nccl, ctx->stream));
    }
    WFX_NCCL(ctx, g_rccl.GroupEnd());
    return 0;
}

int wfx_comm_allreduce_u32(wfx_comm *c, wfx_ctx *ctx, unsigned *buf, size_t count)
{
    if (!c) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    if (c->group) {
        local_op op;
        op.kind = 2;
        op.buf = buf;
        op.count = count;
        return local_post(c, ctx, std::move(op));
    }
    WFX_NCCL(ctx, g_rccl.AllReduce(buf, buf, count, ncclUint32, ncclSum, c->nccl, ctx->stream));
    return 0;
}

int wfx_comm_allgather(wfx_comm *c, wfx_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank)
{
    if (!c) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    if (c->group) {
        local_op op;
        op.kind = 3;
        op.send = send;
        op.buf = recv;
        op.count = bytes_per_rank;
        return local_post(c, ctx, std::move(op));
    }
    WFX_NCCL(ctx, g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, c->nccl, ctx->stream));
    return 0;
}

// ---- C ABI ---------------------------------------------------------------------------------------------
extern "C" {

int wfx_comm_unique_id(void *id_out)
{
    if (!id_out) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null argument");
    if (!rccl_load()) return wfx_fail(nullptr, WFX_ERR_COMM, "%s", g_rccl.err.c_str());
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == WFX_COMM_ID_BYTES, "unique id size");
    WFX_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int wfx_comm_create(wfx_ctx *ctx, const void *id, int world, int rank, wfx_comm **out)
{
    if (!ctx || !id || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad rank %d for world size %d", rank, world);
    if (!rccl_load()) return wfx_fail(ctx, WFX_ERR_COMM, "%s", g_rccl.err.c_str());
    WFX_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t nc = nullptr;
    WFX_NCCL(ctx, g_rccl.CommInitRank(&nc, world, uid, rank));
    wfx_comm *c = new wfx_comm();
    c->world = world;
    c->rank = rank;
    c->nccl = nc;
    c->device = ctx->device;
    *out = c;
    return 0;
}

int wfx_comm_create_local(int world, wfx_comm **out)
{
    if (!out || world < 1 || world > 64) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "local communicator: world size %d", world);
    wfx_comm_group *g = new wfx_comm_group();
    g->world = world;
    g->refs = world;
    g->ctx.assign(world, nullptr);
    g->pending.resize(world);
    for (int r = 0; r < world; ++r) {
        wfx_comm *c = new wfx_comm();
        c->world = world;
        c->rank = r;
        c->group = g;
        out[r] = c;
    }
    return 0;
}

int wfx_comm_info(wfx_comm *comm, int *world, int *rank, int *is_rccl)
{
    if (!comm) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null communicator");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    if (is_rccl) *is_rccl = comm->nccl != nullptr;
    return 0;
}

// ---- small host-side helpers for drivers (bench.py): a barrier and an all-gather of a few host bytes ----------
int wfx_comm_barrier(wfx_comm *comm, wfx_ctx *ctx)
{
    if (!comm || !ctx) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    (void)hipSetDevice(ctx->device);
    if (comm->group && comm->world > 1) return wfx_fail(ctx, WFX_ERR_STATE, "barrier: the ranks of a local communicator run in one thread");
    WFX_TRY(wfx_reserve(ctx, ctx->b_seg, 4096));
    WFX_HIP(ctx, hipMemsetAsync(ctx->b_seg.p, 0, 64, ctx->stream));
    if (!comm->group) WFX_TRY(wfx_comm_allreduce_u32(comm, ctx, (unsigned *)ctx->b_seg.p, 1));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int wfx_comm_allgather_host(wfx_comm *comm, wfx_ctx *ctx, const void *send_host, void *recv_host, size_t bytes)
{
    if (!comm || !ctx || !send_host || !recv_host) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    (void)hipSetDevice(ctx->device);
    if (comm->group && comm->world > 1) return wfx_fail(ctx, WFX_ERR_STATE, "all-gather: the ranks of a local communicator run in one thread");
    const size_t W = (size_t)comm->world;
    WFX_TRY(wfx_reserve(ctx, ctx->b_seg, 4096 + bytes * (W + 1)));
    char *snd = (char *)ctx->b_seg.p + 4096, *rcv = snd + bytes;
    WFX_HIP(ctx, hipMemcpyAsync(snd, send_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (comm->group)
        WFX_HIP(ctx, hipMemcpyAsync(rcv, snd, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    else
        WFX_TRY(wfx_comm_allgather(comm, ctx, snd, rcv, bytes));
    WFX_HIP(ctx, hipMemcpyAsync(recv_host, rcv, bytes * W, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int wfx_comm_destroy(wfx_comm *comm)
{
    if (!comm) return 0;
    if (comm->nccl) {
        int ndev = 0;
        // (a communicator destroyed after the HIP runtime has shut down -- a garbage collector at interpreter exit -- is left alone)
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > comm->device && hipSetDevice(comm->device) == hipSuccess && g_rccl.CommDestroy)
            g_rccl.CommDestroy(comm->nccl);
    }
    if (comm->group && --comm->group->refs == 0) {
        if (comm->group->scratch.p) (void)hipFree(comm->group->scratch.p);
        delete comm->group;
    }
    delete comm;
    return 0;
}

}  // extern "C"
