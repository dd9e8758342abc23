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

}  // namespace

struct ae_comm {
    nccl_comm_t nccl = nullptr;
    int rank = 0, world = 1;
};

namespace ae {

// all-gather of the owned rows of o->y, in place, on the library's stream
void ce_comm_exchange(ae_entropy_optim* o) {
    ae_comm* c = o->comm;
    if (!c || (c->world == 1 && !debug_knob("AE_COMM_FORCE"))) return;  // (the knob: a single-GPU box exercises the RCCL calls)
    Rccl& r = rccl();
    const uint64_t dim = o->dev.dim;
    if (o->comm_equal) {
        const uint64_t rows = o->comm_ranges[1] - o->comm_ranges[0];
        nccl_check(r.AllGather(o->y.p + o->dev.node_lo * dim, o->y.p, rows * dim, kNcclFloat32, c->nccl, stream()), "all-gather");
    } else {  // unequal shards: one in-place broadcast per owner, fused by the group
        nccl_check(r.GroupStart(), "group start");
        for (int q = 0; q < c->world; q++) {
            float* p = o->y.p + o->comm_ranges[2 * q] * dim;
            nccl_check(r.Broadcast(p, p, (o->comm_ranges[2 * q + 1] - o->comm_ranges[2 * q]) * dim, kNcclFloat32, q, c->nccl, stream()), "broadcast");
        }
        nccl_check(r.GroupEnd(), "group end");
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

int32_t ae_comm_destroy(ae_comm* c) {
    return guard([&] {
        if (!c) return;
        if (c->nccl) (void)rccl().CommDestroy(c->nccl);
        delete c;
    });
}

int32_t ae_comm_all_reduce_sum(ae_comm* c, double* value) {
    return guard([&] {
        require_device();
        if (!c || !value) fail(AE_ERR_INVALID_ARG, "null argument");
        if (c->world == 1 && !debug_knob("AE_COMM_FORCE")) return;
        DevBuf<double> d(1);
        d.upload(value, 1);
        nccl_check(rccl().AllReduce(d.p, d.p, 1, kNcclFloat64, kNcclSum, c->nccl, stream()), "all-reduce");
        d.download(value, 1);
    });
}

int32_t ae_entropy_optim_set_comm(ae_entropy_optim* o, ae_comm* c, uint32_t exchanges_per_batch) {
    return guard([&] {
        require_device();
        if (!o) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!c) { o->comm = nullptr; return; }
        if (o->params.ce_mode != AE_CE_HOGWILD)
            fail(AE_ERR_INVALID_ARG, "only the rounds mode (AE_CE_HOGWILD) shards over devices; this handle runs mode %u", o->params.ce_mode);
        // every rank learns every rank's node range
        DevBuf<uint64_t> mine(2), all(2 * (size_t)c->world);
        const uint64_t h[2] = {o->dev.node_lo, o->dev.node_hi};
        mine.upload(h, 2);
        if (c->world > 1 || debug_knob("AE_COMM_FORCE")) nccl_check(rccl().AllGather(mine.p, all.p, 2, kNcclUint64, c->nccl, stream()), "all-gather of the node ranges");
        else AE_HIP(hipMemcpyAsync(all.p, mine.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream()));
        o->comm_ranges = all.to_host();
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
        o->comm_exchanges = exchanges_per_batch ? exchanges_per_batch : 1u;
    });
}

}  // extern "C"
