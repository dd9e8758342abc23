// comm.hip -- the multi-GPU entry points of the C ABI: one process per GPU, RCCL over xGMI.
//
// No reference counterpart (the reference is single-process, shared-memory rayon); this is SURVEY 8b's "8-GPU entry
// point" and the north star's "RCCL all-gather of the low-dim coordinate array over xGMI".  A host (the Rust shim of
// INTEGRATION.md, bench.py) creates one communicator per process and attaches it to an EntropyOptim whose node range
// [node_lo, node_hi) is this rank's shard; from then on ae_entropy_optim_gradient_iteration exchanges the owned rows of
// the coordinate array itself -- in place, on the library's stream, ordered between the round launches by the stream,
// no host synchronisation -- `exchanges_per_batch` times per batch.
//
// RCCL is loaded at run time (dlopen of librccl.so.1) on the first ae_comm_* call: a single-GPU process never loads it
// and the library carries no link-time dependency on it.  A process that has already loaded an RCCL (PyTorch ships one
// under the same soname) gets that copy -- one RCCL per process.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>

#include "ce_internal.h"

namespace {

typedef struct { char internal[128]; } nccl_unique_id;  // ncclUniqueId, rccl.h: NCCL_UNIQUE_ID_BYTES = 128
typedef void* nccl_comm_t;
enum { kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclUint64 = 5, kNcclSum = 0 };  // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    if (r.lib) return r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) fail(AE_ERR_NO_DEVICE, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    auto sym = [&](const char* n) {
        void* p = dlsym(r.lib, n);
        if (!p) fail(AE_ERR_NO_DEVICE, "RCCL symbol %s not found", n);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}

void nccl_check(int rc, const char* what) {
    if (rc != 0) fail(AE_ERR_NO_DEVICE, "RCCL %s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
}

// Second transport, for validation and for hosts where RCCL cannot be loaded: the ranks of ONE machine exchange through a POSIX
// shared-memory segment (device -> segment -> device, a sense-reversing barrier of two atomics in the segment).  Same
// collectives, same call sites; several ranks may share one GPU (RCCL refuses that), which is how the multi-process path is
// exercised end to end on a single-GPU box.  Not a performance path.
struct HostMem {
    int fd = -1;
    char name[96] = {0};
    uint8_t* base = nullptr;
    size_t bytes = 0, data_bytes = 0;
    uint32_t sense = 0;
    struct Header {
        std::atomic<uint32_t> arrived;
        std::atomic<uint32_t> generation;
        std::atomic<uint32_t> ready;  // set by rank 0 once the segment is sized
    };
    Header* hdr() { return reinterpret_cast<Header*>(base); }
    uint8_t* slot(int rank) { return base + 64 + (size_t)rank * 64; }
    uint8_t* data(int world) { return base + 64 + (size_t)world * 64; }
    void barrier(int world) {
        Header* h = hdr();
        const uint32_t gen = h->generation.load(std::memory_order_acquire);
        if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            h->arrived.store(0, std::memory_order_relaxed);
            h->generation.store(gen + 1, std::memory_order_release);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            while (h->generation.load(std::memory_order_acquire) == gen) {
                sched_yield();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) fail(AE_ERR_STATE, "host-memory communicator: a rank did not reach the barrier within 120 s");
            }
        }
    }
};

}  // namespace

struct ae_comm {
    nccl_comm_t nccl = nullptr;
    std::unique_ptr<HostMem> host;  // non-null: the shared-memory transport
    int rank = 0, world = 1;
};

namespace ae {

static bool comm_active(const ae_comm* c) { return c && (c->world > 1 || debug_knob("AE_COMM_FORCE")); }  // (the knob: a single-GPU box exercises the RCCL calls)

bool comm_shares_devices(const ae_comm* c) { return c && c->host != nullptr; }
int comm_rank(const ae_comm* c) { return c ? c->rank : 0; }
int comm_world(const ae_comm* c) { return c ? c->world : 1; }

// `count` floats at device pointer p: root's content on every rank afterwards (library stream; returns when it is there)
void comm_broadcast_f32(ae_comm* c, float* p, uint64_t count, int root) {
    if (!comm_active(c)) return;
    if (c->host) {
        HostMem& h = *c->host;
        if (count * sizeof(float) > h.data_bytes) fail(AE_ERR_INVALID_ARG, "host-memory communicator: broadcast of %llu bytes exceeds the segment", (unsigned long long)(count * 4));
        sync();  // every rank: what the library's (non-blocking) stream still does to p comes before the host-path copies below
        if (c->rank == root) AE_HIP(hipMemcpy(h.data(c->world), p, count * sizeof(float), hipMemcpyDeviceToHost));
        h.barrier(c->world);
        if (c->rank != root) AE_HIP(hipMemcpy(p, h.data(c->world), count * sizeof(float), hipMemcpyHostToDevice));
        h.barrier(c->world);
        return;
    }
    nccl_check(rccl().Broadcast(p, p, count, kNcclFloat32, root, c->nccl, stream()), "broadcast");
    sync();
}

// (bytes are bytes: both transports move a float32 buffer verbatim)
void comm_broadcast_u32(ae_comm* c, uint32_t* p, uint64_t count, int root) { comm_broadcast_f32(c, reinterpret_cast<float*>(p), count, root); }

// every rank's two words, in rank order
static std::vector<uint64_t> comm_all_gather_u64x2(ae_comm* c, const uint64_t (&mine)[2]) {
    std::vector<uint64_t> all(2 * (size_t)c->world);
    if (!comm_active(c)) { all[0] = mine[0]; all[1] = mine[1]; return all; }
    if (c->host) {
        HostMem& h = *c->host;
        memcpy(h.slot(c->rank), mine, 16);
        h.barrier(c->world);
        for (int q = 0; q < c->world; q++) memcpy(&all[2 * q], h.slot(q), 16);
        h.barrier(c->world);
        return all;
    }
    DevBuf<uint64_t> d_mine(2), d_all(2 * (size_t)c->world);
    d_mine.upload(mine, 2);
    nccl_check(rccl().AllGather(d_mine.p, d_all.p, 2, kNcclUint64, c->nccl, stream()), "all-gather of the node ranges");
    return d_all.to_host();
}

double comm_all_reduce_sum(ae_comm* c, double value) {
    if (!comm_active(c)) return value;
    if (c->host) {
        HostMem& h = *c->host;
        memcpy(h.slot(c->rank), &value, 8);
        h.barrier(c->world);
        double s = 0.;
        for (int q = 0; q < c->world; q++) { double v; memcpy(&v, h.slot(q), 8); s += v; }  // rank order: the same sum on every rank
        h.barrier(c->world);
        return s;
    }
    DevBuf<double> d(1);
    d.upload(&value, 1);
    nccl_check(rccl().AllReduce(d.p, d.p, 1, kNcclFloat64, kNcclSum, c->nccl, stream()), "all-reduce");
    d.download(&value, 1);
    return value;
}

// all-gather of the owned rows of o->y, in place, on the library's stream
void ce_comm_exchange(ae_entropy_optim* o) {
    ae_comm* c = o->comm;
    if (!comm_active(c)) return;
    const uint64_t dim = o->dev.dim;
    o->comm_bytes += (o->dev.n - (o->dev.node_hi - o->dev.node_lo)) * dim * sizeof(float);   // received: the other ranks' rows
    float* const yb = o->comm_y ? o->comm_y : o->y.p;   // (rows in the caller's labels, or the time-sliced mode's internal copy: the same ranges)
    if (c->host) {
        HostMem& h = *c->host;
        if (o->dev.n * dim * sizeof(float) > h.data_bytes) fail(AE_ERR_INVALID_ARG, "host-memory communicator: the coordinate array exceeds the segment (max_bytes of ae_comm_init_hostmem)");
        float* seg = reinterpret_cast<float*>(h.data(c->world));
        sync();
        AE_HIP(hipMemcpy(seg + o->dev.node_lo * dim, yb + o->dev.node_lo * dim, (o->dev.node_hi - o->dev.node_lo) * dim * sizeof(float), hipMemcpyDeviceToHost));
        h.barrier(c->world);
        for (int q = 0; q < c->world; q++) {
            if (q == c->rank) continue;
            const uint64_t lo = o->comm_ranges[2 * q], hi = o->comm_ranges[2 * q + 1];
            AE_HIP(hipMemcpy(yb + lo * dim, seg + lo * dim, (hi - lo) * dim * sizeof(float), hipMemcpyHostToDevice));
        }
        h.barrier(c->world);
        return;
    }
    Rccl& r = rccl();
    if (o->comm_equal) {
        const uint64_t rows = o->comm_ranges[1] - o->comm_ranges[0];
        nccl_check(r.AllGather(yb + o->dev.node_lo * dim, yb, rows * dim, kNcclFloat32, c->nccl, stream()), "all-gather");
    } else {  // unequal shards: one in-place broadcast per owner, fused by the group
        nccl_check(r.GroupStart(), "group start");
        for (int q = 0; q < c->world; q++) {
            float* p = yb + o->comm_ranges[2 * q] * dim;
            nccl_check(r.Broadcast(p, p, (o->comm_ranges[2 * q + 1] - o->comm_ranges[2 * q]) * dim, kNcclFloat32, q, c->nccl, stream()), "broadcast");
        }
        nccl_check(r.GroupEnd(), "group end");
    }
}

// attaches the communicator to a shard handle: the ranks' node ranges must tile [0, n) in rank order
void entropy_optim_attach_comm(ae_entropy_optim* o, ae_comm* c, uint32_t exchanges_per_batch) {
    if (!c) { o->comm = nullptr; return; }
    if (o->params.ce_mode != AE_CE_HOGWILD && o->params.ce_mode != AE_CE_SLICED)
        fail(AE_ERR_INVALID_ARG, "the time-sliced mode (AE_CE_SLICED: faithful, for node orders with few cross-shard edges) and the rounds mode (AE_CE_HOGWILD: approximate) "
                                 "shard over devices; this handle runs mode %u", o->params.ce_mode);
    const uint64_t h[2] = {o->dev.node_lo, o->dev.node_hi};
    o->comm_ranges = comm_all_gather_u64x2(c, h);  // every rank learns every rank's node range
    uint64_t expect = 0;
    bool equal = true;
    for (int q = 0; q < c->world; q++) {
        if (o->comm_ranges[2 * q] != expect || o->comm_ranges[2 * q + 1] <= o->comm_ranges[2 * q])
            fail(AE_ERR_INVALID_ARG, "the ranks' node ranges must tile [0, n) in rank order");
        expect = o->comm_ranges[2 * q + 1];
        equal = equal && (o->comm_ranges[2 * q + 1] - o->comm_ranges[2 * q]) == (o->comm_ranges[1] - o->comm_ranges[0]);
    }
    if (expect != o->dev.n) fail(AE_ERR_INVALID_ARG, "the ranks' node ranges must tile [0, n) in rank order");
    o->comm_equal = equal;
    o->comm = c;
    o->comm_exchanges = exchanges_per_batch ? exchanges_per_batch : 4u;   // (0: the library's choice)
    if (exchanges_per_batch == 1 && c->world >= 4 && o->dev.n >= (1ull << 20))
        set_last_warning("exchanges_per_batch = 1 with " + std::to_string(c->world) + " ranks and " + std::to_string(o->dev.n) +
                         " nodes is outside the validated envelope: the other ranks' rows are a whole batch old while the layout still moves fast (measured at 11 M nodes "
                         "in 8 shards: edges 10-21 % short, cross entropy +2.5 %); ask for 4 or more, or 0 for the library's choice");
    // The time-sliced mode now knows every rank's range: its internal numbering (a random relabelling inside every rank's range, the
    // same on every rank) and everything built on it are made now (a sharded handle defers its preparation to this point or to its
    // first batch: ce.hip).  A preparation that fails on ONE rank (or a range whose cross-shard mass is over the limit on one rank
    // and under it on another) must not leave the others waiting in the first collective: the ranks agree on the outcome first.
    if (o->params.ce_mode == AE_CE_SLICED && c->world > 1) {
        int32_t code = AE_OK;
        std::string msg;
        try {
            ce_slice_prepare(o);
        } catch (const Error& e) {
            code = e.code;
            msg = e.msg;
        }
        uint64_t mine[2] = {0, (uint64_t)(uint32_t)code};
        memcpy(&mine[0], &o->sl_cross_frac, 8);
        const std::vector<uint64_t> all = comm_all_gather_u64x2(c, mine);
        double worst = 0.;
        int worst_rank = 0, failed_rank = -1;
        for (int q = 0; q < c->world; q++) {
            double f;
            memcpy(&f, &all[2 * q], 8);
            if (f > worst) { worst = f; worst_rank = q; }
            if (all[2 * q + 1] != 0 && failed_rank < 0) failed_rank = q;
        }
        // the slices of a batch are cut alike on every rank: what enters the cut must be the same number everywhere, and the overflow
        // share of the (scheduling-dependent) parallel colouring is not -- the ranks agree on the largest
        {
            uint64_t ov[2] = {0, (uint64_t)o->sl_classes};
            memcpy(&ov[0], &o->sl_ov_frac, 8);
            const std::vector<uint64_t> allov = comm_all_gather_u64x2(c, ov);
            double ov_max = 0.;
            for (int q = 0; q < c->world; q++) { double f; memcpy(&f, &allov[2 * q], 8); ov_max = std::max(ov_max, f); }
            o->sl_ov_frac_sched = ov_max;
        }
        if (code != AE_OK) { o->comm = nullptr; fail(code, "%s", msg.c_str()); }
        if (failed_rank >= 0) { o->comm = nullptr; fail((int32_t)all[2 * failed_rank + 1], "rank %d could not prepare its shard of the time-sliced mode (its own message says why)", failed_rank); }
        if (worst > ce_slice_max_cross_mass() && !debug_knob("AE_SL_ANY_PARTITION")) {
            o->comm = nullptr;
            fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED over %d ranks: %.1f %% of rank %d's edge probability mass lies on cross-shard edges (limit %.0f %%): order the nodes by "
                                     "locality first (ae_kgraph_partition + ae_kgraph_permuted; Embedder::embed does it itself), or ask for the approximate rounds mode (AE_CE_HOGWILD)",
                 c->world, 100. * worst, worst_rank, 100. * ce_slice_max_cross_mass());
        }
    }
}

}  // namespace ae

extern "C" {

int32_t ae_comm_unique_id(uint8_t* id128) {
    return guard([&] {
        if (!id128) fail(AE_ERR_INVALID_ARG, "null argument");
        nccl_unique_id id;
        nccl_check(rccl().GetUniqueId(&id), "get unique id");
        memcpy(id128, id.internal, 128);
    });
}

int32_t ae_comm_init(int32_t rank, int32_t world, const uint8_t* id128, ae_comm** out) {
    return guard([&] {
        require_device();
        if (!id128 || !out || world < 1 || rank < 0 || rank >= world) fail(AE_ERR_INVALID_ARG, "bad argument");
        std::unique_ptr<ae_comm> c(new ae_comm);
        c->rank = rank;
        c->world = world;
        nccl_unique_id id;
        memcpy(id.internal, id128, 128);
        nccl_check(rccl().CommInitRank(&c->nccl, world, id, rank), "communicator init");
        *out = c.release();
    });
}

int32_t ae_comm_init_hostmem(int32_t rank, int32_t world, const char* name, uint64_t max_bytes, ae_comm** out) {
    return guard([&] {
        require_device();
        if (!name || !out || world < 1 || rank < 0 || rank >= world || strlen(name) == 0 || strlen(name) > 80) fail(AE_ERR_INVALID_ARG, "bad argument");
        std::unique_ptr<ae_comm> c(new ae_comm);
        c->rank = rank;
        c->world = world;
        c->host.reset(new HostMem);
        HostMem& h = *c->host;
        snprintf(h.name, sizeof(h.name), "/%s", name);
        h.data_bytes = (size_t)max_bytes;
        h.bytes = 64 + (size_t)world * 64 + h.data_bytes;
        // Rank 0 creates a fresh segment (O_EXCL after unlinking whatever a crashed run left under the name).  A rank != 0 may open the
        // STALE segment first (same name, large enough, ready == 1): it therefore proves that rank 0 lives in the mapping it holds -- it
        // writes a fresh token into its slot and proceeds only when rank 0 has echoed it; a mapping nobody answers in is dropped and the
        // name opened again (rank 0's unlink + create make the next open the live one).
        auto map_segment = [&]() {
            void* m = mmap(nullptr, h.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, h.fd, 0);
            if (m == MAP_FAILED) fail(AE_ERR_OOM, "host-memory communicator: mmap of %zu bytes failed", h.bytes);
            h.base = static_cast<uint8_t*>(m);
        };
        auto token_of = [&](int q) { return reinterpret_cast<std::atomic<uint64_t>*>(h.slot(q)); };      // written by rank q
        auto ack_of = [&](int q) { return reinterpret_cast<std::atomic<uint64_t>*>(h.slot(q) + 8); };    // written by rank 0
        const auto t_start = std::chrono::steady_clock::now();
        auto expired = [&] { return std::chrono::steady_clock::now() - t_start > std::chrono::seconds(120); };
        if (rank == 0) {
            shm_unlink(h.name);  // a stale segment of a crashed run
            h.fd = shm_open(h.name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (h.fd < 0 || ftruncate(h.fd, (off_t)h.bytes) != 0) fail(AE_ERR_STATE, "host-memory communicator: cannot create the segment %s (%s)", h.name, strerror(errno));
            map_segment();  // (a fresh segment is zero-filled: arrived = generation = 0, no tokens)
            h.hdr()->ready.store(1, std::memory_order_release);
            for (int q = 1; q < world; q++) {
                uint64_t t;
                while ((t = token_of(q)->load(std::memory_order_acquire)) == 0) {
                    usleep(500);
                    if (expired()) fail(AE_ERR_STATE, "host-memory communicator: rank %d did not join the segment %s within 120 s", q, h.name);
                }
                ack_of(q)->store(t, std::memory_order_release);
            }
        } else {
            const uint64_t token = ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)rank << 56) | 1ull;
            for (;;) {
                if (expired()) fail(AE_ERR_STATE, "host-memory communicator: rank 0 did not create / answer in the segment %s within 120 s", h.name);
                h.fd = shm_open(h.name, O_RDWR, 0600);
                struct stat st;
                if (h.fd < 0 || fstat(h.fd, &st) != 0 || (size_t)st.st_size < h.bytes) {  // not created / not sized yet
                    if (h.fd >= 0) { close(h.fd); h.fd = -1; }
                    usleep(2000);
                    continue;
                }
                map_segment();
                bool live = false;
                const auto t0 = std::chrono::steady_clock::now();
                while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(500)) {
                    if (h.hdr()->ready.load(std::memory_order_acquire) == 1) {
                        token_of(rank)->store(token, std::memory_order_release);
                        if (ack_of(rank)->load(std::memory_order_acquire) == token) { live = true; break; }
                    }
                    usleep(500);
                }
                if (live) break;
                munmap(h.base, h.bytes);  // nobody answers here: a stale segment (or rank 0 is slow: the same name is opened again)
                h.base = nullptr;
                close(h.fd);
                h.fd = -1;
            }
        }
        h.barrier(world);
        *out = c.release();
    });
}

int32_t ae_comm_destroy(ae_comm* c) {
    return guard([&] {
        if (!c) return;
        if (c->nccl) (void)rccl().CommDestroy(c->nccl);
        if (c->host) {
            if (c->host->base) munmap(c->host->base, c->host->bytes);
            if (c->host->fd >= 0) close(c->host->fd);
            if (c->rank == 0) shm_unlink(c->host->name);
        }
        delete c;
    });
}

int32_t ae_comm_all_reduce_sum(ae_comm* c, double* value) {
    return guard([&] {
        require_device();
        if (!c || !value) fail(AE_ERR_INVALID_ARG, "null argument");
        *value = comm_all_reduce_sum(c, *value);
    });
}

int32_t ae_entropy_optim_comm_bytes(const ae_entropy_optim* o, uint64_t* bytes) {
    return guard([&] {
        if (!o || !bytes) fail(AE_ERR_INVALID_ARG, "null argument");
        *bytes = o->comm_bytes;
    });
}

int32_t ae_entropy_optim_set_comm(ae_entropy_optim* o, ae_comm* c, uint32_t exchanges_per_batch) {
    return guard([&] {
        require_device();
        if (!o) fail(AE_ERR_INVALID_ARG, "null argument");
        entropy_optim_attach_comm(o, c, exchanges_per_batch);
    });
}

}  // extern "C"
