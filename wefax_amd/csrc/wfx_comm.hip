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
//   * shm backend: one PROCESS per rank on one host, any number of them per GPU; messages are staged through POSIX shared
//     memory (sender: device -> its outbox file, receiver: outbox -> device) between two barriers on a shared counter.
//     Slow (PCIe both ways) and blocking on the host, but everything the RCCL path has is real here: separate address
//     spaces, contexts and streams, ranks that reach a phase at different times, buffers reused by the next phase while a
//     peer is still a phase behind.  It is how `bench.py --gpus 2/4/8` runs end to end on a one-GPU box (RCCL refuses two
//     ranks on one device) and what the multi-process tests drive; a peer that dies or disagrees about a message surfaces as
//     WFX_ERR_COMM on every rank within the timeout -- never a hang.
//
// Only three collectives are needed (wfx_shard.hip): a personalised exchange (grouped send / recv: the transposes of the
// distributed transforms and the final gather), a sum all-reduce of uint32 histograms, and an all-gather of equal blocks.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
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

// ---- shm backend -------------------------------------------------------------------------------------------
struct shm_ctl {                        // /dev/shm/wfx_<job>_ctl, created by rank 0
    std::atomic<uint32_t> magic;        // set last
    uint32_t world;
    std::atomic<uint32_t> arrived, generation, failed, attached;
    std::atomic<uint64_t> box_bytes[64];    // size of each rank's outbox file
};
struct shm_msg {
    int32_t peer;
    uint32_t pad;
    uint64_t bytes, offset;
};
struct shm_box_hdr {
    uint64_t seq;
    uint32_t kind, nmsg;
    uint64_t count;
};
constexpr uint32_t SHM_MAGIC = 0x57465843u;
constexpr size_t SHM_ALIGN = 256;

struct wfx_shm {
    std::string job;
    int world = 1, rank = 0;
    double timeout = 120.0;
    shm_ctl *ctl = nullptr;
    int fd[64];
    void *map[64];
    size_t mapped[64];
    uint64_t seq = 0;
    bool host_mode = false;             // buffers are host memory (no context: the protocol tests on a machine without a GPU)
    wfx_shm()
    {
        for (int i = 0; i < 64; ++i) {
            fd[i] = -1;
            map[i] = nullptr;
            mapped[i] = 0;
        }
    }
};

struct wfx_comm {
    int world = 1, rank = 0;
    ncclComm_t nccl = nullptr;          // RCCL backend
    wfx_comm_group *group = nullptr;    // local backend
    wfx_shm *shm = nullptr;             // shm backend
    int device = 0;
    // what this rank has put on the wire since the last reset, one record per collective (counted where the collective is
    // posted, so every transport reports the same figures)
    std::vector<wfx_wire_entry> wire;
    unsigned long long wire_count = 0;
    char label[24] = {0};
    // optional timing of every collective (wfx_comm_wire_timing): a HIP-event pair on the stream the collective runs on (RCCL), the
    // host clock around the call (the transports that complete it before returning), and -- for an exchange on the communicator's
    // own stream -- a second pair on the CONTEXT's stream around the wait for it: the time the compute stream actually stood still
    struct wire_clock {
        hipEvent_t a = nullptr, b = nullptr, wa = nullptr, wb = nullptr;
        double host_us = -1.0;
        int on_comm_stream = 0;
    };
    bool timing = false;
    std::vector<wire_clock> clocks;          // parallel to `wire`
    std::vector<hipEvent_t> ev_free;
    int slot_entry[64];
    // RCCL backend: exchanges that may overlap compute go to a stream of the communicator's own (wfx_comm_exchange_async): an
    // event recorded on the context's stream gates them, an event per slot marks their completion (wfx_comm_wait)
    hipStream_t xstream = nullptr;
    hipEvent_t ready = nullptr;
    std::vector<hipEvent_t> done;
    std::vector<char> pending;
    unsigned long long async_count = 0;
};

void wfx_comm_label(wfx_comm *c, const char *name)
{
    if (c) snprintf(c->label, sizeof c->label, "%s", name ? name : "");
}

static hipEvent_t clock_event(wfx_comm *c)
{
    if (!c->ev_free.empty()) {
        hipEvent_t e = c->ev_free.back();
        c->ev_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static void clock_release(wfx_comm *c, wfx_comm::wire_clock &k)
{
    for (hipEvent_t *e : {&k.a, &k.b, &k.wa, &k.wb})
        if (*e) {
            c->ev_free.push_back(*e);
            *e = nullptr;
        }
    k.host_us = -1.0;
    k.on_comm_stream = 0;
}

// returns the index of the record (its clock is c->clocks[index] while timing is on)
static int wire_record(wfx_comm *c, const char *fallback, unsigned long long sent, unsigned long long received, unsigned long long largest)
{
    wfx_wire_entry e;
    memset(&e, 0, sizeof e);
    snprintf(e.name, sizeof e.name, "%s", c->label[0] ? c->label : fallback);
    e.total_bytes = sent;
    e.max_rank_bytes = received;
    e.max_link_bytes = largest;
    size_t at;
    if (c->wire.size() < 256) {
        at = c->wire.size();
        c->wire.push_back(e);
        c->clocks.emplace_back();
    } else {
        at = (size_t)(c->wire_count % 256);
        c->wire[at] = e;
        clock_release(c, c->clocks[at]);
    }
    ++c->wire_count;
    c->label[0] = 0;
    return (int)at;
}

// event pair around what the caller enqueues on `st` next (RCCL); no-ops while timing is off
static void clock_begin(wfx_comm *c, int at, hipStream_t st, int on_comm_stream)
{
    if (!c->timing) return;
    wfx_comm::wire_clock &k = c->clocks[(size_t)at];
    k.on_comm_stream = on_comm_stream;
    k.a = clock_event(c);
    k.b = clock_event(c);
    if (k.a) (void)hipEventRecord(k.a, st);
}
static void clock_end(wfx_comm *c, int at, hipStream_t st)
{
    if (!c->timing) return;
    wfx_comm::wire_clock &k = c->clocks[(size_t)at];
    if (k.b) (void)hipEventRecord(k.b, st);
}

static double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static std::string shm_name(const std::string &job, const char *what, int r = -1)
{
    std::string n = "/wfx_" + job + "_" + what;
    if (r >= 0) n += std::to_string(r);
    return n;
}

static int shm_fail(wfx_shm *s, wfx_ctx *ctx, const char *fmt, ...)
{
    if (s && s->ctl) s->ctl->failed.store(1, std::memory_order_release);
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return wfx_fail(ctx, WFX_ERR_COMM, "shm communicator, rank %d: %s", s ? s->rank : -1, buf);
}

// every rank has arrived (central counter, generation flips when the last one does); a failed peer or the timeout ends the wait
static int shm_barrier(wfx_shm *s, wfx_ctx *ctx)
{
    shm_ctl *c = s->ctl;
    if (c->failed.load(std::memory_order_acquire)) return shm_fail(s, ctx, "a peer reported a failure");
    const uint32_t gen = c->generation.load(std::memory_order_acquire);
    if (c->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)s->world) {
        c->arrived.store(0, std::memory_order_relaxed);
        c->generation.fetch_add(1, std::memory_order_acq_rel);
        return 0;
    }
    const double t0 = now_s();
    unsigned spins = 0;
    while (c->generation.load(std::memory_order_acquire) == gen) {
        if (c->failed.load(std::memory_order_acquire)) return shm_fail(s, ctx, "a peer reported a failure");
        if (++spins > 2000) {
            usleep(50);
            if (now_s() - t0 > s->timeout) return shm_fail(s, ctx, "barrier timed out after %lld s (a peer died or never arrived)", (long long)s->timeout);
        } else {
            sched_yield();
        }
    }
    return 0;
}

static int shm_map(wfx_shm *s, wfx_ctx *ctx, int r, size_t need)
{
    if (s->mapped[r] >= need) return 0;
    if (s->map[r]) munmap(s->map[r], s->mapped[r]);
    s->map[r] = nullptr;
    s->mapped[r] = 0;
    if (s->fd[r] < 0) {
        s->fd[r] = shm_open(shm_name(s->job, "box", r).c_str(), O_RDWR, 0600);
        if (s->fd[r] < 0) return shm_fail(s, ctx, "cannot open the outbox of rank %d", r);
    }
    void *m = mmap(nullptr, need, PROT_READ | PROT_WRITE, MAP_SHARED, s->fd[r], 0);
    if (m == MAP_FAILED) return shm_fail(s, ctx, "mmap of %lld bytes of rank %d's outbox failed", (long long)need, r);
    s->map[r] = m;
    s->mapped[r] = need;
    return 0;
}

// own outbox with room for `need` bytes (grown geometrically; the new size is published before the next barrier)
static int shm_own_box(wfx_shm *s, wfx_ctx *ctx, size_t need, char **base)
{
    const int r = s->rank;
    size_t have = (size_t)s->ctl->box_bytes[r].load(std::memory_order_acquire);
    if (need > have) {
        size_t cap = have ? have : (1u << 20);
        while (cap < need) cap *= 2;
        if (ftruncate(s->fd[r], (off_t)cap) != 0) return shm_fail(s, ctx, "cannot grow the outbox to %lld bytes (is /dev/shm full?)", (long long)cap);
        // tmpfs reserves nothing at ftruncate: without the pages a later copy into the mapping dies with SIGBUS (a container's
        // default /dev/shm is 64 MB) and the peers only notice at the barrier's timeout.  Reserve them now; failure is an error
        // on every rank (shm_fail sets the job's failure flag).
        int fe;
        while ((fe = posix_fallocate(s->fd[r], 0, (off_t)cap)) == EINTR) {}
        if (fe != 0) {
            (void)!ftruncate(s->fd[r], (off_t)have);
            return shm_fail(s, ctx, "cannot reserve %lld bytes of shared memory for the outbox (%s): /dev/shm is too small for this exchange", (long long)cap,
                            strerror(fe));
        }
        s->ctl->box_bytes[r].store(cap, std::memory_order_release);
        have = cap;
    }
    WFX_TRY(shm_map(s, ctx, r, have));
    *base = (char *)s->map[r];
    return 0;
}

static int shm_peer_box(wfx_shm *s, wfx_ctx *ctx, int p, const char **base, size_t *size)
{
    const size_t have = (size_t)s->ctl->box_bytes[p].load(std::memory_order_acquire);
    if (have < sizeof(shm_box_hdr)) return shm_fail(s, ctx, "rank %d has no outbox", p);
    WFX_TRY(shm_map(s, ctx, p, have));
    *base = (const char *)s->map[p];
    *size = have;
    return 0;
}

static int shm_copy(wfx_shm *s, wfx_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return 0;
    if (s->host_mode) {
        memmove(dst, src, bytes);
        return 0;
    }
    WFX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
    return 0;
}

static int shm_sync(wfx_shm *s, wfx_ctx *ctx)
{
    if (s->host_mode) return 0;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        s->ctl->failed.store(1, std::memory_order_release);
        return wfx_fail_hip(ctx, e, "hipStreamSynchronize (shm communicator)");
    }
    return 0;
}

static int shm_exchange(wfx_shm *s, wfx_ctx *ctx, const wfx_xfer *list, int n)
{
    const uint64_t seq = ++s->seq;
    WFX_TRY(shm_sync(s, ctx));                           // what the messages are made of has been computed
    // outbox: header, one descriptor per outgoing message (list order), the payloads
    int nmsg = 0;
    size_t payload = 0;
    for (int i = 0; i < n; ++i)
        if (list[i].peer != s->rank && list[i].send_bytes) {
            ++nmsg;
            payload += (list[i].send_bytes + SHM_ALIGN - 1) / SHM_ALIGN * SHM_ALIGN;
        }
    const size_t head = (sizeof(shm_box_hdr) + (size_t)nmsg * sizeof(shm_msg) + SHM_ALIGN - 1) / SHM_ALIGN * SHM_ALIGN;
    char *box = nullptr;
    WFX_TRY(shm_own_box(s, ctx, head + payload, &box));
    shm_box_hdr *h = (shm_box_hdr *)box;
    shm_msg *m = (shm_msg *)(box + sizeof(shm_box_hdr));
    size_t off = head;
    int k = 0;
    for (int i = 0; i < n; ++i)
        if (list[i].peer != s->rank && list[i].send_bytes) {
            m[k].peer = list[i].peer;
            m[k].pad = 0;
            m[k].bytes = list[i].send_bytes;
            m[k].offset = off;
            WFX_TRY(shm_copy(s, ctx, box + off, list[i].send, list[i].send_bytes, hipMemcpyDeviceToHost));
            off += (list[i].send_bytes + SHM_ALIGN - 1) / SHM_ALIGN * SHM_ALIGN;
            ++k;
        }
    h->seq = seq;
    h->kind = 1;
    h->nmsg = (uint32_t)nmsg;
    h->count = 0;
    WFX_TRY(shm_sync(s, ctx));
    WFX_TRY(shm_barrier(s, ctx));                        // every outbox of this collective is complete
    // a rank's message to itself is a copy (normally the packer has already written it in place)
    for (int i = 0; i < n; ++i)
        if (list[i].peer == s->rank && list[i].send_bytes) {
            if (list[i].send_bytes != list[i].recv_bytes) return shm_fail(s, ctx, "self message of %lld bytes into %lld", (long long)list[i].send_bytes, (long long)list[i].recv_bytes);
            if (list[i].send != list[i].recv) WFX_TRY(shm_copy(s, ctx, list[i].recv, list[i].send, list[i].send_bytes, hipMemcpyDeviceToDevice));
        }
    // the k-th receive posted for peer p takes the k-th message p addressed to this rank
    std::vector<int> taken((size_t)s->world, 0);
    for (int i = 0; i < n; ++i) {
        const int p = list[i].peer;
        if (p == s->rank || !list[i].recv_bytes) continue;
        const char *pb = nullptr;
        size_t psz = 0;
        WFX_TRY(shm_peer_box(s, ctx, p, &pb, &psz));
        const shm_box_hdr *ph = (const shm_box_hdr *)pb;
        if (ph->seq != seq || ph->kind != 1)
            return shm_fail(s, ctx, "rank %d is at collective %lld, this rank at exchange %lld: the ranks disagree about the sequence of collectives", p,
                            (long long)ph->seq, (long long)seq);
        const shm_msg *pm = (const shm_msg *)(pb + sizeof(shm_box_hdr));
        int want = taken[(size_t)p]++, found = -1;
        for (uint32_t q = 0; q < ph->nmsg; ++q)
            if (pm[q].peer == s->rank && want-- == 0) {
                found = (int)q;
                break;
            }
        if (found < 0) return shm_fail(s, ctx, "rank %d sent fewer messages to this rank than it expects (receive %d)", p, i);
        if (pm[found].bytes != list[i].recv_bytes)
            return shm_fail(s, ctx, "a message from rank %d has %lld bytes, the receiver expects %lld", p, (long long)pm[found].bytes, (long long)list[i].recv_bytes);
        if (pm[found].offset + pm[found].bytes > psz) return shm_fail(s, ctx, "rank %d's outbox is shorter than its descriptors say", p);
        WFX_TRY(shm_copy(s, ctx, list[i].recv, pb + pm[found].offset, pm[found].bytes, hipMemcpyHostToDevice));
    }
    // and nothing is left over: every message addressed to this rank was expected
    for (int p = 0; p < s->world; ++p) {
        if (p == s->rank) continue;
        const char *pb = nullptr;
        size_t psz = 0;
        WFX_TRY(shm_peer_box(s, ctx, p, &pb, &psz));
        const shm_box_hdr *ph = (const shm_box_hdr *)pb;
        if (ph->seq != seq || ph->kind != 1)
            return shm_fail(s, ctx, "rank %d is at collective %lld, this rank at exchange %lld", p, (long long)ph->seq, (long long)seq);
        const shm_msg *pm = (const shm_msg *)(pb + sizeof(shm_box_hdr));
        int sent = 0;
        for (uint32_t q = 0; q < ph->nmsg; ++q) sent += pm[q].peer == s->rank;
        if (sent != taken[(size_t)p]) return shm_fail(s, ctx, "rank %d sent %d messages to this rank, which expected %d", p, sent, taken[(size_t)p]);
    }
    WFX_TRY(shm_sync(s, ctx));
    return shm_barrier(s, ctx);                          // outboxes may be overwritten
}

static int shm_allreduce_u32(wfx_shm *s, wfx_ctx *ctx, unsigned *buf, size_t count)
{
    const uint64_t seq = ++s->seq;
    WFX_TRY(shm_sync(s, ctx));
    const size_t head = (sizeof(shm_box_hdr) + SHM_ALIGN - 1) / SHM_ALIGN * SHM_ALIGN;
    char *box = nullptr;
    WFX_TRY(shm_own_box(s, ctx, head + count * 4, &box));
    WFX_TRY(shm_copy(s, ctx, box + head, buf, count * 4, hipMemcpyDeviceToHost));
    shm_box_hdr *h = (shm_box_hdr *)box;
    h->seq = seq;
    h->kind = 2;
    h->nmsg = 0;
    h->count = count;
    WFX_TRY(shm_sync(s, ctx));
    WFX_TRY(shm_barrier(s, ctx));
    std::vector<unsigned> acc(count, 0u);
    for (int p = 0; p < s->world; ++p) {                 // rank order: the same sum on every rank
        const char *pb = nullptr;
        size_t psz = 0;
        WFX_TRY(shm_peer_box(s, ctx, p, &pb, &psz));
        const shm_box_hdr *ph = (const shm_box_hdr *)pb;
        if (ph->seq != seq || ph->kind != 2 || ph->count != count)
            return shm_fail(s, ctx, "all-reduce: rank %d is at collective %lld (kind %u), this rank at %lld", p, (long long)ph->seq, ph->kind, (long long)seq);
        const unsigned *v = (const unsigned *)(pb + head);
        for (size_t i = 0; i < count; ++i) acc[i] += v[i];
    }
    WFX_TRY(shm_copy(s, ctx, buf, acc.data(), count * 4, hipMemcpyHostToDevice));
    WFX_TRY(shm_sync(s, ctx));                           // (acc lives on this stack frame)
    return shm_barrier(s, ctx);
}

static int shm_allgather(wfx_shm *s, wfx_ctx *ctx, const void *send, void *recv, size_t bytes, bool host_ptrs)
{
    const uint64_t seq = ++s->seq;
    WFX_TRY(shm_sync(s, ctx));
    const size_t head = (sizeof(shm_box_hdr) + SHM_ALIGN - 1) / SHM_ALIGN * SHM_ALIGN;
    char *box = nullptr;
    WFX_TRY(shm_own_box(s, ctx, head + bytes, &box));
    if (host_ptrs)
        memcpy(box + head, send, bytes);
    else
        WFX_TRY(shm_copy(s, ctx, box + head, send, bytes, hipMemcpyDeviceToHost));
    shm_box_hdr *h = (shm_box_hdr *)box;
    h->seq = seq;
    h->kind = 3;
    h->nmsg = 0;
    h->count = bytes;
    WFX_TRY(shm_sync(s, ctx));
    WFX_TRY(shm_barrier(s, ctx));
    for (int p = 0; p < s->world; ++p) {
        const char *pb = nullptr;
        size_t psz = 0;
        WFX_TRY(shm_peer_box(s, ctx, p, &pb, &psz));
        const shm_box_hdr *ph = (const shm_box_hdr *)pb;
        if (ph->seq != seq || ph->kind != 3 || ph->count != bytes)
            return shm_fail(s, ctx, "all-gather: rank %d is at collective %lld (kind %u), this rank at %lld", p, (long long)ph->seq, ph->kind, (long long)seq);
        if (host_ptrs)
            memcpy((char *)recv + (size_t)p * bytes, pb + head, bytes);
        else
            WFX_TRY(shm_copy(s, ctx, (char *)recv + (size_t)p * bytes, pb + head, bytes, hipMemcpyHostToDevice));
    }
    WFX_TRY(shm_sync(s, ctx));
    return shm_barrier(s, ctx);
}

static void shm_close(wfx_shm *s)
{
    if (!s) return;
    for (int r = 0; r < 64; ++r) {
        if (s->map[r]) munmap(s->map[r], s->mapped[r]);
        if (s->fd[r] >= 0) close(s->fd[r]);
    }
    shm_unlink(shm_name(s->job, "box", s->rank).c_str());
    if (s->ctl) {
        // the last rank to leave removes the control block
        if (s->ctl->attached.fetch_sub(1, std::memory_order_acq_rel) == 1) shm_unlink(shm_name(s->job, "ctl").c_str());
        munmap(s->ctl, sizeof(shm_ctl));
    }
    delete s;
}

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
        const int rc = local_execute(g, ctx);
        if (rc != 0) {                                   // the failed collective does not stay queued: the group remains usable
            for (int r = 0; r < g->world; ++r)
                if (!g->pending[r].empty()) g->pending[r].pop_front();
            return rc;
        }
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
    int at = 0;
    {
        unsigned long long sent = 0, got = 0, big = 0;
        for (int i = 0; i < n; ++i)
            if (list[i].peer != c->rank) {
                sent += list[i].send_bytes;
                got += list[i].recv_bytes;
                if (list[i].send_bytes > big) big = list[i].send_bytes;
            }
        at = wire_record(c, "exchange", sent, got, big);
    }
    if (c->group) {
        local_op op;
        op.kind = 1;
        op.list.assign(list, list + n);
        return local_post(c, ctx, std::move(op));
    }
    if (c->shm) {
        const double t0 = now_s();
        const int rc = shm_exchange(c->shm, ctx, list, n);
        if (c->timing) c->clocks[(size_t)at].host_us = 1e6 * (now_s() - t0);
        return rc;
    }
    clock_begin(c, at, ctx->stream, 0);
    bool any_remote = false;
    for (int i = 0; i < n; ++i) any_remote = any_remote || list[i].peer != c->rank;
    // a rank's message to itself is a copy on the stream (normally the packer has already written it in place)
    for (int i = 0; i < n; ++i)
        if (list[i].peer == c->rank && list[i].send_bytes) {
            if (list[i].send_bytes != list[i].recv_bytes) return wfx_fail(ctx, WFX_ERR_COMM, "exchange: self message of %zu bytes into %zu", list[i].send_bytes, list[i].recv_bytes);
            if (list[i].send != list[i].recv)
                WFX_HIP(ctx, hipMemcpyAsync(list[i].recv, list[i].send, list[i].send_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
    if (!any_remote) {
        clock_end(c, at, ctx->stream);
        return 0;
    }
    WFX_NCCL(ctx, g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n && bad == ncclSuccess; ++i) {
        if (list[i].peer == c->rank) continue;
        if (list[i].send_bytes) bad = g_rccl.Send(list[i].send, list[i].send_bytes, ncclUint8, list[i].peer, c->nccl, ctx->stream);
        if (bad == ncclSuccess && list[i].recv_bytes) bad = g_rccl.Recv(list[i].recv, list[i].recv_bytes, ncclUint8, list[i].peer, c->nccl, ctx->stream);
    }
    const ncclResult_t end = g_rccl.GroupEnd();          // the group is closed on the error path too: the communicator stays usable
    if (bad != ncclSuccess) return fail_nccl(ctx, bad, "ncclSend / ncclRecv of an exchange");
    if (end != ncclSuccess) return fail_nccl(ctx, end, "ncclGroupEnd");
    clock_end(c, at, ctx->stream);
    return 0;
}

// An exchange that need not finish before the caller enqueues more work on its stream: RCCL communicators run it on their own
// stream, after everything the context's stream holds so far; wfx_comm_wait(slot) makes the context's stream wait for it.  The other
// transports (in-process, shared memory) complete it before returning, as wfx_comm_exchange does -- same results, no overlap.
// WFX_COMM_ASYNC=0 keeps RCCL exchanges in stream order too (A/B runs).
int wfx_comm_exchange_async(wfx_comm *c, wfx_ctx *ctx, const wfx_xfer *list, int n, int slot)
{
    if (!c) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    const char *env = getenv("WFX_COMM_ASYNC");
    const bool off = env && atoi(env) == 0;
    if (!c->nccl || off || slot < 0 || slot >= 64) return wfx_comm_exchange(c, ctx, list, n);
    for (int i = 0; i < n; ++i)
        if (list[i].peer < 0 || list[i].peer >= c->world) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "exchange: peer %d out of range", list[i].peer);
    if (!c->xstream) {
        // all or nothing: a communicator whose stream exists has every event it will be asked for
        hipStream_t xs = nullptr;
        hipEvent_t ready = nullptr;
        std::vector<hipEvent_t> done(64, nullptr);
        hipError_t e = hipStreamCreateWithFlags(&xs, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
        for (int k = 0; k < 64 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&done[(size_t)k], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (hipEvent_t ev : done)
                if (ev) (void)hipEventDestroy(ev);
            if (ready) (void)hipEventDestroy(ready);
            if (xs) (void)hipStreamDestroy(xs);
            return wfx_fail(ctx, WFX_ERR_HIP, "exchange stream / events: %s", hipGetErrorString(e));
        }
        c->xstream = xs;
        c->ready = ready;
        c->done = std::move(done);
        c->pending.assign(64, 0);
    }
    int at = 0;
    {
        unsigned long long sent = 0, got = 0, big = 0;
        for (int i = 0; i < n; ++i)
            if (list[i].peer != c->rank) {
                sent += list[i].send_bytes;
                got += list[i].recv_bytes;
                if (list[i].send_bytes > big) big = list[i].send_bytes;
            }
        at = wire_record(c, "exchange", sent, got, big);
    }
    WFX_HIP(ctx, hipEventRecord(c->ready, ctx->stream));
    WFX_HIP(ctx, hipStreamWaitEvent(c->xstream, c->ready, 0));
    clock_begin(c, at, c->xstream, 1);
    c->slot_entry[slot] = c->timing ? at : -1;
    bool any_remote = false;
    for (int i = 0; i < n; ++i) {
        any_remote = any_remote || list[i].peer != c->rank;
        if (list[i].peer == c->rank && list[i].send_bytes) {
            if (list[i].send_bytes != list[i].recv_bytes) return wfx_fail(ctx, WFX_ERR_COMM, "exchange: self message of %zu bytes into %zu", list[i].send_bytes, list[i].recv_bytes);
            if (list[i].send != list[i].recv)
                WFX_HIP(ctx, hipMemcpyAsync(list[i].recv, list[i].send, list[i].send_bytes, hipMemcpyDeviceToDevice, c->xstream));
        }
    }
    if (any_remote) {
        WFX_NCCL(ctx, g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int i = 0; i < n && bad == ncclSuccess; ++i) {
            if (list[i].peer == c->rank) continue;
            if (list[i].send_bytes) bad = g_rccl.Send(list[i].send, list[i].send_bytes, ncclUint8, list[i].peer, c->nccl, c->xstream);
            if (bad == ncclSuccess && list[i].recv_bytes) bad = g_rccl.Recv(list[i].recv, list[i].recv_bytes, ncclUint8, list[i].peer, c->nccl, c->xstream);
        }
        const ncclResult_t end = g_rccl.GroupEnd();
        if (bad != ncclSuccess) return fail_nccl(ctx, bad, "ncclSend / ncclRecv of an exchange");
        if (end != ncclSuccess) return fail_nccl(ctx, end, "ncclGroupEnd");
    }
    clock_end(c, at, c->xstream);
    WFX_HIP(ctx, hipEventRecord(c->done[(size_t)slot], c->xstream));
    c->pending[(size_t)slot] = 1;
    ++c->async_count;
    return 0;
}

// the context's stream waits for the exchange recorded under `slot` (no-op when it ran in stream order or has been waited for)
int wfx_comm_wait(wfx_comm *c, wfx_ctx *ctx, int slot)
{
    if (!c || slot < 0 || slot >= (int)c->pending.size() || !c->pending[(size_t)slot]) return 0;
    wfx_comm::wire_clock *k = (c->timing && c->slot_entry[slot] >= 0 && (size_t)c->slot_entry[slot] < c->clocks.size()) ? &c->clocks[(size_t)c->slot_entry[slot]] : nullptr;
    if (k && !k->wa) {
        k->wa = clock_event(c);
        k->wb = clock_event(c);
        if (k->wa) (void)hipEventRecord(k->wa, ctx->stream);
    } else
        k = nullptr;
    WFX_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->done[(size_t)slot], 0));
    if (k && k->wb) (void)hipEventRecord(k->wb, ctx->stream);
    c->slot_entry[slot] = -1;
    c->pending[(size_t)slot] = 0;
    return 0;
}

unsigned long long wfx_comm_async_count(const wfx_comm *c) { return c ? c->async_count : 0; }

int wfx_comm_allreduce_u32(wfx_comm *c, wfx_ctx *ctx, unsigned *buf, size_t count)
{
    if (!c) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    // (a ring all-reduce moves 2 (W - 1) / W of the buffer out of and into every rank)
    const int at = wire_record(c, "all-reduce", 2ull * (unsigned long long)(c->world - 1) * count * 4 / (unsigned long long)c->world,
                               2ull * (unsigned long long)(c->world - 1) * count * 4 / (unsigned long long)c->world, count * 4 / (unsigned long long)c->world);
    if (c->group) {
        local_op op;
        op.kind = 2;
        op.buf = buf;
        op.count = count;
        return local_post(c, ctx, std::move(op));
    }
    if (c->shm) {
        const double t0 = now_s();
        const int rc = shm_allreduce_u32(c->shm, ctx, buf, count);
        if (c->timing) c->clocks[(size_t)at].host_us = 1e6 * (now_s() - t0);
        return rc;
    }
    clock_begin(c, at, ctx->stream, 0);
    WFX_NCCL(ctx, g_rccl.AllReduce(buf, buf, count, ncclUint32, ncclSum, c->nccl, ctx->stream));
    clock_end(c, at, ctx->stream);
    return 0;
}

int wfx_comm_allgather(wfx_comm *c, wfx_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank)
{
    if (!c) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    const int at = wire_record(c, "all-gather", (unsigned long long)(c->world - 1) * bytes_per_rank, (unsigned long long)(c->world - 1) * bytes_per_rank, bytes_per_rank);
    if (c->group) {
        local_op op;
        op.kind = 3;
        op.send = send;
        op.buf = recv;
        op.count = bytes_per_rank;
        return local_post(c, ctx, std::move(op));
    }
    if (c->shm) {
        const double t0 = now_s();
        const int rc = shm_allgather(c->shm, ctx, send, recv, bytes_per_rank, false);
        if (c->timing) c->clocks[(size_t)at].host_us = 1e6 * (now_s() - t0);
        return rc;
    }
    clock_begin(c, at, ctx->stream, 0);
    WFX_NCCL(ctx, g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, c->nccl, ctx->stream));
    clock_end(c, at, ctx->stream);
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

static int shm_attach(wfx_ctx *ctx, const char *job, int world, int rank, double timeout_s, double t0, wfx_comm **out, bool *leftover);

int wfx_comm_create_shm(wfx_ctx *ctx, const char *job, int world, int rank, double timeout_s, wfx_comm **out)
{
    if (!job || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad rank %d for world size %d (shm: at most 64)", rank, world);
    for (const char *q = job; *q; ++q)
        if (!((*q >= '0' && *q <= '9') || (*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || *q == '-' || *q == '.'))
            return wfx_fail(ctx, WFX_ERR_BAD_ARG, "shm job name: letters, digits, '-' and '.' only");
    if (strlen(job) < 1 || strlen(job) > 100) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "shm job name: 1..100 characters");
    // A peer can pass every check on the control block of an EARLIER job of this name a moment before rank 0 takes that block away
    // (marks it failed, unlinks it, creates its own): its first barrier then fails on a dead block while rank 0 waits for it.  Such a
    // peer notices -- the barrier failed AND the block it holds is no longer linked under the job's name -- lets go of everything
    // WITHOUT touching the names (they belong to the new job now) and attaches again.
    const double t_start = now_s();
    for (;;) {
        bool leftover = false;
        const int rc = shm_attach(ctx, job, world, rank, timeout_s, t_start, out, &leftover);
        if (rc == 0 || !leftover) return rc;
        usleep(2000);
    }
}

static int shm_attach(wfx_ctx *ctx, const char *job, int world, int rank, double timeout_s, double t0, wfx_comm **out, bool *leftover)
{
    wfx_shm *s = new wfx_shm();
    s->job = job;
    s->world = world;
    s->rank = rank;
    s->timeout = timeout_s > 0 ? timeout_s : 120.0;
    s->host_mode = ctx == nullptr;
    const std::string cn = shm_name(s->job, "ctl");
    int cfd = -1;
    if (rank == 0) {
        // a leftover of a crashed job of the same name: a peer of THIS launch may already have opened it -- take its magic away
        // and mark it failed before unlinking, so that such a peer finds out (it re-opens: see below) instead of sitting on a
        // dead control block; stale outboxes of that job go too
        {
            const int ofd = shm_open(cn.c_str(), O_RDWR, 0600);
            if (ofd >= 0) {
                struct stat st;
                if (fstat(ofd, &st) == 0 && (size_t)st.st_size >= sizeof(shm_ctl)) {
                    void *om = mmap(nullptr, sizeof(shm_ctl), PROT_READ | PROT_WRITE, MAP_SHARED, ofd, 0);
                    if (om != MAP_FAILED) {
                        ((shm_ctl *)om)->magic.store(0, std::memory_order_release);
                        ((shm_ctl *)om)->failed.store(1, std::memory_order_release);
                        munmap(om, sizeof(shm_ctl));
                    }
                }
                close(ofd);
            }
            shm_unlink(cn.c_str());
            for (int r = 0; r < 64; ++r) shm_unlink(shm_name(s->job, "box", r).c_str());
        }
        cfd = shm_open(cn.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (cfd < 0 || ftruncate(cfd, (off_t)sizeof(shm_ctl)) != 0) {
            if (cfd >= 0) close(cfd);
            delete s;
            return wfx_fail(ctx, WFX_ERR_COMM, "shm communicator: cannot create %s", cn.c_str());
        }
    }
    void *m = MAP_FAILED;
    for (;;) {
        if (rank != 0) {
            while ((cfd = shm_open(cn.c_str(), O_RDWR, 0600)) < 0) {
                if (now_s() - t0 > s->timeout) {
                    delete s;
                    return wfx_fail(ctx, WFX_ERR_COMM, "shm communicator, rank %d: rank 0 never created %s", rank, cn.c_str());
                }
                usleep(2000);
            }
            struct stat st;
            while (fstat(cfd, &st) == 0 && (size_t)st.st_size < sizeof(shm_ctl)) {
                if (now_s() - t0 > s->timeout) break;
                usleep(1000);
            }
        }
        m = mmap(nullptr, sizeof(shm_ctl), PROT_READ | PROT_WRITE, MAP_SHARED, cfd, 0);
        if (m == MAP_FAILED) {
            close(cfd);
            delete s;
            return wfx_fail(ctx, WFX_ERR_COMM, "shm communicator: cannot map %s", cn.c_str());
        }
        if (rank == 0) break;
        // wait for rank 0's magic -- on a block that is still linked under the job's name: one that was unlinked meanwhile is the
        // leftover of an earlier job which rank 0 has just replaced (it took the magic away first), so open the name again
        bool stale = false, ready = false;
        while (!ready && !stale) {
            struct stat st;
            if (fstat(cfd, &st) != 0 || st.st_nlink == 0) stale = true;
            else if (((shm_ctl *)m)->magic.load(std::memory_order_acquire) == SHM_MAGIC) {
                ready = fstat(cfd, &st) == 0 && st.st_nlink > 0 && ((shm_ctl *)m)->magic.load(std::memory_order_acquire) == SHM_MAGIC;
                stale = !ready;
            } else if (now_s() - t0 > s->timeout) {
                munmap(m, sizeof(shm_ctl));
                close(cfd);
                delete s;
                return wfx_fail(ctx, WFX_ERR_COMM, "shm communicator, rank %d: the control block was never initialised", rank);
            } else
                usleep(1000);
        }
        if (ready) break;
        munmap(m, sizeof(shm_ctl));
        close(cfd);
        usleep(2000);
    }
    int ctl_fd = -1;                         // (a peer keeps its descriptor until the first barrier is through: see wfx_comm_create_shm)
    if (rank == 0)
        close(cfd);
    else
        ctl_fd = cfd;
    s->ctl = (shm_ctl *)m;
    if (rank == 0) {
        s->ctl->world = (uint32_t)world;
        s->ctl->arrived.store(0);
        s->ctl->generation.store(0);
        s->ctl->failed.store(0);
        s->ctl->attached.store(0);
        for (int r = 0; r < 64; ++r) s->ctl->box_bytes[r].store(0);
        s->ctl->magic.store(SHM_MAGIC, std::memory_order_release);
    } else {
        if ((int)s->ctl->world != world) {
            const unsigned theirs = s->ctl->world;
            munmap(m, sizeof(shm_ctl));
            s->ctl = nullptr;
            close(ctl_fd);
            delete s;
            return wfx_fail(ctx, WFX_ERR_COMM, "shm communicator: job %s has world size %u, this rank was told %d", job, theirs, world);
        }
    }
    s->ctl->attached.fetch_add(1, std::memory_order_acq_rel);
    const std::string bn = shm_name(s->job, "box", rank);
    shm_unlink(bn.c_str());
    s->fd[rank] = shm_open(bn.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (s->fd[rank] < 0) {
        const int rc = shm_fail(s, ctx, "cannot create the outbox %s", bn.c_str());
        if (ctl_fd >= 0) close(ctl_fd);
        shm_close(s);
        return rc;
    }
    wfx_comm *c = new wfx_comm();
    c->world = world;
    c->rank = rank;
    c->shm = s;
    c->device = ctx ? ctx->device : 0;
    char *box = nullptr;
    int rc = shm_own_box(s, ctx, 1u << 20, &box);        // outboxes exist before anyone looks for them
    if (rc == 0) rc = shm_barrier(s, ctx);
    if (rc != 0) {
        struct stat st;
        if (ctl_fd >= 0 && fstat(ctl_fd, &st) == 0 && st.st_nlink == 0 && now_s() - t0 < s->timeout) {
            // the block this peer joined was a leftover that rank 0 has replaced since: let go of the mappings only -- the names
            // (control block, outboxes) are the new job's, and so is the count of attached ranks it will keep in ITS block
            *leftover = true;
            for (int r = 0; r < 64; ++r) {
                if (s->map[r]) munmap(s->map[r], s->mapped[r]);
                if (s->fd[r] >= 0) close(s->fd[r]);
            }
            munmap(s->ctl, sizeof(shm_ctl));
            delete s;
        } else
            shm_close(s);
        if (ctl_fd >= 0) close(ctl_fd);
        delete c;
        return rc;
    }
    if (ctl_fd >= 0) close(ctl_fd);
    *out = c;
    return 0;
}

// A randomised sequence of collectives with known answers (every rank derives the same plan from `seed`): exchanges with 0-2
// messages of up to 1 MB per ordered pair of ranks, all-reduces, all-gathers, with a random pause before each so that the
// ranks arrive out of step.  Device buffers when ctx is given, host buffers when it is NULL (shm communicators only).
static uint64_t st_mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

int wfx_comm_selftest(wfx_comm *comm, wfx_ctx *ctx, int rounds, uint64_t seed)
{
    if (!comm) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null communicator");
    if (comm->group) return wfx_fail(ctx, WFX_ERR_STATE, "selftest: the ranks of a local communicator run in one thread");
    if (!ctx && !comm->shm) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "selftest without a context needs a shm communicator");
    if (ctx) (void)hipSetDevice(ctx->device);
    const int W = comm->world, me = comm->rank;
    auto dalloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (!ctx) return malloc(bytes ? bytes : 8);
        return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? p : nullptr;
    };
    auto dfree = [&](void *p) {
        if (!ctx) free(p);
        else (void)hipFree(p);
    };
    auto put = [&](void *dst, const void *src, size_t bytes) {      // host -> buffer
        if (!ctx) memcpy(dst, src, bytes);
        else (void)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    auto get = [&](void *dst, const void *src, size_t bytes) {      // buffer -> host
        if (!ctx) memcpy(dst, src, bytes);
        else (void)hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    };
    for (int round = 0; round < rounds; ++round) {
        usleep((useconds_t)(st_mix(seed ^ (uint64_t)(round * 64 + me)) % 3000));
        // ---- exchange ----
        struct msg {
            int src, dst;
            size_t bytes;
            uint64_t key;
        };
        std::vector<msg> plan;
        for (int a = 0; a < W; ++a)
            for (int b = 0; b < W; ++b) {
                const uint64_t h = st_mix(seed * 1315423911ull + (uint64_t)round * 4099 + (uint64_t)a * 67 + (uint64_t)b);
                const int cnt = (int)(h % 3);
                for (int k = 0; k < cnt; ++k) {
                    const uint64_t hk = st_mix(h + (uint64_t)k);
                    plan.push_back({a, b, (size_t)(8 + 8 * (hk % ((round % 4 == 3) ? 131072 : 2048))), hk});
                }
            }
        std::vector<wfx_xfer> list;
        std::vector<void *> sbuf, rbuf;
        std::vector<const msg *> rmsg;
        for (const msg &m : plan) {
            if (m.src == me) {
                std::vector<uint64_t> host(m.bytes / 8);
                for (size_t i = 0; i < host.size(); ++i) host[i] = st_mix(m.key + i);
                void *d = dalloc(m.bytes);
                if (!d) return wfx_fail(ctx, WFX_ERR_OOM, "selftest: allocation failed");
                put(d, host.data(), m.bytes);
                sbuf.push_back(d);
                if (m.dst != me) list.push_back({m.dst, d, m.bytes, nullptr, 0});
            }
            if (m.dst == me) {
                void *d = dalloc(m.bytes);
                if (!d) return wfx_fail(ctx, WFX_ERR_OOM, "selftest: allocation failed");
                rbuf.push_back(d);
                rmsg.push_back(&m);
                if (m.src != me) list.push_back({m.src, nullptr, 0, d, m.bytes});
                else list.push_back({me, sbuf.back(), m.bytes, d, m.bytes});      // (the send entry of a self message was skipped above)
            }
        }
        if (ctx && (round & 1)) {      // odd rounds: on the communicator's own stream (RCCL), ordered back by wfx_comm_wait
            WFX_TRY(wfx_comm_exchange_async(comm, ctx, list.data(), (int)list.size(), round % 64));
            WFX_TRY(wfx_comm_wait(comm, ctx, round % 64));
        } else
            WFX_TRY(wfx_comm_exchange(comm, ctx, list.data(), (int)list.size()));
        if (ctx) WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t q = 0; q < rbuf.size(); ++q) {
            std::vector<uint64_t> host(rmsg[q]->bytes / 8);
            get(host.data(), rbuf[q], rmsg[q]->bytes);
            for (size_t i = 0; i < host.size(); ++i)
                if (host[i] != st_mix(rmsg[q]->key + i))
                    return wfx_fail(ctx, WFX_ERR_COMM, "selftest round %d: word %zu of a %zu-byte message from rank %d arrived wrong on rank %d", round, i,
                                    rmsg[q]->bytes, rmsg[q]->src, me);
        }
        for (void *p : sbuf) dfree(p);
        for (void *p : rbuf) dfree(p);
        // ---- all-reduce ----
        usleep((useconds_t)(st_mix(seed ^ (uint64_t)(round * 64 + me) ^ 0x55) % 2000));
        const size_t nw = 257 + (size_t)(st_mix(seed + (uint64_t)round) % 4000);
        std::vector<unsigned> hv(nw), want(nw, 0u);
        for (int r = 0; r < W; ++r)
            for (size_t i = 0; i < nw; ++i) {
                const unsigned v = (unsigned)(st_mix(seed + (uint64_t)round * 131 + (uint64_t)r * 7919 + i) % 100000);
                if (r == me) hv[i] = v;
                want[i] += v;
            }
        void *ar = dalloc(nw * 4);
        put(ar, hv.data(), nw * 4);
        WFX_TRY(wfx_comm_allreduce_u32(comm, ctx, (unsigned *)ar, nw));
        if (ctx) WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        get(hv.data(), ar, nw * 4);
        dfree(ar);
        for (size_t i = 0; i < nw; ++i)
            if (hv[i] != want[i]) return wfx_fail(ctx, WFX_ERR_COMM, "selftest round %d: all-reduce word %zu is %u, expected %u", round, i, hv[i], want[i]);
        // ---- all-gather ----
        const size_t gb = 64 + 8 * (size_t)(st_mix(seed + 17 * (uint64_t)round) % 2000);
        std::vector<uint64_t> gs(gb / 8), gr(gb / 8 * (size_t)W);
        for (size_t i = 0; i < gs.size(); ++i) gs[i] = st_mix(seed + (uint64_t)round * 977 + (uint64_t)me * 31 + i);
        void *ds = dalloc(gb), *dr = dalloc(gb * (size_t)W);
        put(ds, gs.data(), gb);
        WFX_TRY(wfx_comm_allgather(comm, ctx, ds, dr, gb));
        if (ctx) WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        get(gr.data(), dr, gb * (size_t)W);
        dfree(ds);
        dfree(dr);
        for (int r = 0; r < W; ++r)
            for (size_t i = 0; i < gs.size(); ++i)
                if (gr[(size_t)r * gs.size() + i] != st_mix(seed + (uint64_t)round * 977 + (uint64_t)r * 31 + i))
                    return wfx_fail(ctx, WFX_ERR_COMM, "selftest round %d: all-gather block of rank %d arrived wrong on rank %d", round, r, me);
    }
    return 0;
}

uint64_t wfx_comm_async_exchanges(wfx_comm *comm) { return comm ? comm->async_count : 0; }

int wfx_comm_wire_reset(wfx_comm *comm)
{
    if (!comm) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null communicator");
    comm->wire.clear();
    comm->wire_count = 0;
    for (auto &k : comm->clocks) clock_release(comm, k);
    comm->clocks.clear();
    for (int &e : comm->slot_entry) e = -1;
    return 0;
}

int wfx_comm_wire_timing(wfx_comm *comm, int on)
{
    if (!comm) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null communicator");
    comm->timing = on != 0;
    return wfx_comm_wire_reset(comm);
}

// The streams the collectives ran on must have been synchronised (wfx_sync / wfx_shard_result); entries whose events have not
// completed report -1.
int wfx_comm_wire_times(wfx_comm *comm, wfx_wire_time *out, int cap)
{
    if (!comm) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null communicator");
    const int n = (int)comm->clocks.size();
    const size_t first = comm->wire_count > 256 ? (size_t)(comm->wire_count % 256) : 0;
    for (int i = 0; i < n && i < cap && out; ++i) {
        const wfx_comm::wire_clock &k = comm->clocks[(first + (size_t)i) % (size_t)n];
        wfx_wire_time t;
        t.us = t.wait_us = -1.0;
        t.on_comm_stream = k.on_comm_stream;
        t.timed = 0;
        float ms = 0.0f;
        if (k.host_us >= 0.0) {           // completed before the call returned: all of it kept the caller waiting
            t.us = t.wait_us = k.host_us;
            t.timed = 2;
        } else if (k.a && k.b && hipEventElapsedTime(&ms, k.a, k.b) == hipSuccess) {
            t.us = 1e3 * (double)ms;
            t.timed = 1;
            if (!k.on_comm_stream)
                t.wait_us = t.us;         // in stream order: the compute stream did nothing else meanwhile
            else if (k.wa && k.wb && hipEventElapsedTime(&ms, k.wa, k.wb) == hipSuccess)
                t.wait_us = 1e3 * (double)ms;
        }
        out[i] = t;
    }
    return n;
}

int wfx_comm_wire_stats(wfx_comm *comm, wfx_wire_entry *out, int cap)
{
    if (!comm) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null communicator");
    const int n = (int)comm->wire.size();
    // (more than 256 collectives since the reset: the ring holds the last 256, handed out oldest first)
    const size_t first = comm->wire_count > 256 ? (size_t)(comm->wire_count % 256) : 0;
    for (int i = 0; i < n && i < cap && out; ++i) out[i] = comm->wire[(first + (size_t)i) % (size_t)n];
    return (int)(comm->wire_count < 0x7fffffffull ? comm->wire_count : 0x7fffffffull);
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
    if (!comm || (!ctx && !(comm->shm && comm->shm->host_mode))) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    if (ctx) (void)hipSetDevice(ctx->device);
    if (comm->group && comm->world > 1) return wfx_fail(ctx, WFX_ERR_STATE, "barrier: the ranks of a local communicator run in one thread");
    if (comm->shm) {
        WFX_TRY(shm_sync(comm->shm, ctx));
        return shm_barrier(comm->shm, ctx);
    }
    WFX_TRY(wfx_reserve(ctx, ctx->b_seg, 4096));
    WFX_HIP(ctx, hipMemsetAsync(ctx->b_seg.p, 0, 64, ctx->stream));
    if (!comm->group) WFX_TRY(wfx_comm_allreduce_u32(comm, ctx, (unsigned *)ctx->b_seg.p, 1));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int wfx_comm_allgather_host(wfx_comm *comm, wfx_ctx *ctx, const void *send_host, void *recv_host, size_t bytes)
{
    if (!comm || (!ctx && !(comm->shm && comm->shm->host_mode)) || !send_host || !recv_host) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    if (ctx) (void)hipSetDevice(ctx->device);
    if (comm->group && comm->world > 1) return wfx_fail(ctx, WFX_ERR_STATE, "all-gather: the ranks of a local communicator run in one thread");
    if (comm->shm) return shm_allgather(comm->shm, ctx, send_host, recv_host, bytes, true);
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
    {   // the events of wfx_comm_wire_timing: those still attached to records and the free list
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > comm->device && hipSetDevice(comm->device) == hipSuccess) {
            for (auto &k : comm->clocks) clock_release(comm, k);
            for (hipEvent_t e : comm->ev_free)
                if (e) (void)hipEventDestroy(e);
        }
        comm->clocks.clear();
        comm->ev_free.clear();
    }
    if (comm->xstream) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > comm->device && hipSetDevice(comm->device) == hipSuccess) {
            (void)hipStreamSynchronize(comm->xstream);
            for (hipEvent_t e : comm->done)
                if (e) (void)hipEventDestroy(e);
            if (comm->ready) (void)hipEventDestroy(comm->ready);
            (void)hipStreamDestroy(comm->xstream);
        }
    }
    if (comm->nccl) {
        int ndev = 0;
        // (a communicator destroyed after the HIP runtime has shut down -- a garbage collector at interpreter exit -- is left alone)
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > comm->device && hipSetDevice(comm->device) == hipSuccess && g_rccl.CommDestroy)
            g_rccl.CommDestroy(comm->nccl);
    }
    if (comm->shm) shm_close(comm->shm);
    if (comm->group && --comm->group->refs == 0) {
        if (comm->group->scratch.p) (void)hipFree(comm->group->scratch.p);
        delete comm->group;
    }
    delete comm;
    return 0;
}

}  // extern "C"
