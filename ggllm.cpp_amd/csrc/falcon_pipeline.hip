// falcon_pipeline.hip -- layer-pipelined multi-GPU decode in C++: one process per GPU, rank r holds a contiguous
// range of Falcon blocks (falcon_hip_model with layer_begin / layer_end), and the ONLY exchange between ranks is the residual
// rows [B][n_embd] f32 from stage to stage plus the B greedy-sampled token ids from the last stage back to stage 0 -- RCCL
// ncclSend / ncclRecv inside one ncclGroupStart / ncclGroupEnd per slot, over xGMI, one link per hop. No all-reduce, no
// all-gather, no host staging. What it replaces in the reference: the per-op device loop with peer copies of
// ggml-cuda.cu:2586-2608, 2713-2732 (every mat-mul split over the devices, results gathered on the main device).
//
// A single decode stream is serial through the stages, so the pipeline keeps G GROUPS of B lock-step sequences in flight
// (falcon_hip_context_create_seqs: one pass over the stage's weights serves the B tokens of a group through the N <= 4
// mat-vec). Work item w = k * G + g is group g in round k. Two slot schedules (the same as bench_pipeline.PipelineRunner,
// whose gloo tests pin them; falcon_hip_pipeline_schedule exposes this file's version to tests/test_pipeline_gloo.py):
//
//   G <  2P  rank r computes item t - r in slot t. At the start of the slot ONE grouped exchange sends the result of slot t-1
//            to the next rank (last rank: the tokens, to rank 0) and receives the input of item t - r; all on the library stream.
//   G >= 2P  a hop gets a whole slot: rank r computes item t - 2r in slot t; the exchange of slot t (result of slot t-1 out,
//            input of slot t+1 in) runs on a second stream and is only waited for by the compute of slot t+1, so every
//            transfer overlaps a stage step.
//
// Send and receive of a slot are posted in one group, so the ring cannot deadlock. RCCL is bound at run time (dlopen of
// librccl.so.1, the copy already in the process if there is one): single-GPU users of libggml_hip.so do not need it.
// A LOCAL transport (every rank a falcon_hip_pipeline of the same process and device, hand-off by device copies) runs the
// identical schedule and stage code on one GPU: falcon_hip_pipeline_run_local, used by tests/test_gpu_pipeline.py.
// A HOST-STAGED transport between PROCESSES of one node (FALCON_PIPE_TRANSPORT=shm, round 5): the ranks are real processes -- spawned,
// handed the unique id, each with its own HIP context, stage graphs and slot schedule, exactly as the RCCL job -- and only the bytes of a
// hand-off travel differently: device -> pinned host -> a POSIX shared-memory mailbox of the receiver -> pinned host -> device, sequenced
// by a {seq, ack} word pair per mailbox. It exists because RCCL refuses two ranks on one device ("Duplicate GPU detected"): with it
// `bench.py --gpus N` runs end to end with every rank on GPU 0 of a one-GPU box (FALCON_PIPE_SAME_DEVICE=1), tests/test_gpu_pipeline_procs.py.
#include "../../include/falcon-hip.h"
#include "../../include/ggml-hip-ops.h"
#include "fq_device.h"
#include "hip_context.h"

#include "rccl_dyn.h"

#include <mutex>
#include <stdio.h>
#include <string.h>
#include <vector>

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

// ------------------------------------------------------------------------------------------------ RCCL, bound at run time (rccl_dyn.h)
rccl_api * fq_rccl() {
    static rccl_api api; static std::once_flag once; static bool ok = false;
    std::call_once(once, [] {
        for (const char * name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) { fprintf(stderr, "ggml-hip: librccl.so.1 not found (%s)\n", dlerror()); return; }
        bool all = true;
        auto bind = [&](const char * sym) { void * p = dlsym(api.lib, sym); if (!p) { fprintf(stderr, "ggml-hip: %s missing from RCCL\n", sym); all = false; } return p; };
        api.ncclGetUniqueId    = (decltype(api.ncclGetUniqueId))    bind("ncclGetUniqueId");
        api.ncclCommInitRank   = (decltype(api.ncclCommInitRank))   bind("ncclCommInitRank");
        api.ncclCommDestroy    = (decltype(api.ncclCommDestroy))    bind("ncclCommDestroy");
        api.ncclCommCount      = (decltype(api.ncclCommCount))      bind("ncclCommCount");
        api.ncclSend           = (decltype(api.ncclSend))           bind("ncclSend");
        api.ncclRecv           = (decltype(api.ncclRecv))           bind("ncclRecv");
        api.ncclAllGather      = (decltype(api.ncclAllGather))      bind("ncclAllGather");
        api.ncclGroupStart     = (decltype(api.ncclGroupStart))     bind("ncclGroupStart");
        api.ncclGroupEnd       = (decltype(api.ncclGroupEnd))       bind("ncclGroupEnd");
        api.ncclGetErrorString = (decltype(api.ncclGetErrorString)) bind("ncclGetErrorString");
        ok = all;
    });
    return ok ? &api : nullptr;
}

namespace {

// ------------------------------------------------------------------------------------------------ the slot schedule (host only)
enum { OP_SEND_HIDDEN = 0, OP_SEND_TOKEN = 1, OP_RECV_HIDDEN = 2, OP_RECV_TOKEN = 3 };
struct pipe_op   { int kind, group, peer; };
struct pipe_slot { int n_ops = 0; pipe_op op[2]; int group = -1, round = -1; };

bool overlapped(int P, int G) { return P > 1 && G >= 2 * P; }
int  slot_count(int P, int G, int W) { return P == 1 ? W : (overlapped(P, G) ? W + 2 * P : W + P); }

pipe_slot slot_of(int r, int P, int G, int W, int t) {
    pipe_slot s;
    if (P == 1) { s.group = t % G; s.round = t / G; return s; }
    const bool first = r == 0, last = r == P - 1;
    const int hop = overlapped(P, G) ? 2 : 1;                       // slots between the computes of one item on neighbouring ranks
    const int w_prev = t - 1 - hop * r;                              // the item this rank computed in the previous slot
    if (w_prev >= 0 && w_prev < W) s.op[s.n_ops++] = last ? pipe_op{ OP_SEND_TOKEN, w_prev % G, 0 } : pipe_op{ OP_SEND_HIDDEN, w_prev % G, r + 1 };
    if (hop == 1) {
        if (first) { const int x = t - P;          if (x >= 0 && x < W) s.op[s.n_ops++] = { OP_RECV_TOKEN,  x % G, P - 1 }; }
        else       { const int w = t - r;          if (w >= 0 && w < W) s.op[s.n_ops++] = { OP_RECV_HIDDEN, w % G, r - 1 }; }
    } else {
        if (first) { const int x = t - 2 * P + 1;  if (x >= 0 && x < W) s.op[s.n_ops++] = { OP_RECV_TOKEN,  x % G, P - 1 }; }   // the tokens the last rank sends in this slot
        else       { const int w = t + 1 - 2 * r;  if (w >= 0 && w < W) s.op[s.n_ops++] = { OP_RECV_HIDDEN, w % G, r - 1 }; }   // the input of the NEXT slot
    }
    const int w = t - hop * r;
    if (w >= 0 && w < W) { s.group = w % G; s.round = w / G; }
    return s;
}

}   // namespace

// ------------------------------------------------------------------------------------------------ the pipeline object
struct falcon_hip_pipeline {
    falcon_hip_model * m = nullptr;
    falcon_hip_hparams hp{};
    int rank = 0, world = 1, G = 1, B = 1, n_ctx = 0;
    bool first = true, last = true;
    std::vector<falcon_hip_context *> ctx;                          // one lock-step context per group
    std::vector<float *>   hidden_in, hidden_out;                   // [B][n_embd] per group
    std::vector<int32_t *> tok_in, tok_out;                         // [B] per group
    std::vector<float *>   mb_hidden; std::vector<int32_t *> mb_tok; // local transport: this rank's mailboxes, per group
    int32_t * hist = nullptr;                                       // last rank: [n_ctx][G * B] sampled tokens, by round
    int rounds_done = 0;
    ncclComm_t comm = nullptr;                                      // world > 1 and not the local transport (loop-back: rank 0 of the local job owns the one-rank communicator)
    bool local = false;
    falcon_hip_pipeline * loop = nullptr;                           // local transport over RCCL's self send / recv: the rank that owns the communicator
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_compute[4] = {}, ev_comm[4] = {}, ev_start = nullptr;
    std::vector<void *> allocs;
    struct shm_transport * shm = nullptr;                           // FALCON_PIPE_TRANSPORT=shm: mailboxes in POSIX shared memory (ranks = processes of one node)
};

// ------------------------------------------------------------------------------------------------ the host-staged transport (shm)
// One segment per job, named after the job's unique id: a header, then one mailbox per (receiving rank, group) and kind:
//   { seq, ack } (64 bytes)  |  payload (residual rows [B][n_embd] f32, or B token ids)
// A sender waits for ack == its count (the receiver has taken the previous message of that mailbox: always true already, by the
// schedule's own data dependencies -- the check makes it hold by construction), writes the payload, then seq = ++count (release). A
// receiver waits for seq == its count + 1 (acquire), copies the payload to the device, then ack = ++count. Waits are bounded
// (FALCON_PIPE_SHM_TIMEOUT_S, default 300): a dead peer ends the job with a message instead of hanging it.
struct shm_box { volatile uint32_t seq, ack; uint32_t pad[14]; };
static_assert(sizeof(shm_box) == 64, "mailbox header");
// The DEVICE-TO-DEVICE variant (FALCON_PIPE_TRANSPORT=ipc, round 6): the same segment, mailboxes and { seq, ack } protocol, but a mailbox's payload lives in
// the RECEIVING rank's device memory -- every rank exports one allocation by hipIpcGetMemHandle (the handle table sits behind the header), opens its
// successor's, and a send is ONE device-to-device copy into the peer's mailbox (hipMemcpyAsync over xGMI between two GPUs; an on-device copy when the ranks
// share GPU 0); a receive copies out of this rank's own mailbox. Nothing is staged through the host. The fall-back of a job whose RCCL pre-flight fails,
// tried before the host-staged form.
struct shm_header { volatile uint32_t ready, attached; uint32_t world, G, B, E; volatile uint32_t ipc_ok, ipc_bad; uint32_t pad[8]; };
static_assert(sizeof(shm_header) == 64, "segment header");
struct shm_transport {
    int fd = -1; uint8_t * base = nullptr; size_t bytes = 0; char name[64] = {0}; bool owner = false;
    bool ipc = false; size_t table_bytes = 0;                       // ipc: world x 64-byte hipIpcMemHandle_t behind the header
    uint8_t * dev_box = nullptr, * peer_box = nullptr;              // ipc: this rank's device mailboxes [G][hid | tok]; the successor's, opened through its handle
    size_t dev_stride = 0; bool ipc_failed = false;
    size_t hid_bytes = 0, tok_bytes = 0, box_stride = 0;            // payload sizes (rounded to 64), bytes per (rank, group)
    void * stage_out = nullptr, * stage_in = nullptr;               // pinned staging rows
    std::vector<uint32_t> sent_h, sent_t, got_h, got_t;             // per group: messages this rank has sent to / taken from a mailbox
    double timeout_s = 300.0;
    size_t tok_hdr_off = 0;                                         // offset of a mailbox's token header behind its residual-row header
    shm_box * box(int rank, int G, int g, bool token) const { return (shm_box *)(base + 64 + table_bytes + ((size_t) rank * G + g) * box_stride + (token ? tok_hdr_off : 0)); }
    size_t dev_off(int g, bool token) const { return (size_t) g * dev_stride + (token ? hid_bytes : 0); }
};

namespace {

void * palloc(falcon_hip_pipeline * p, size_t bytes) {
    void * d = nullptr;
    HIP_CHECK(hipMalloc(&d, bytes ? bytes : 16));
    HIP_CHECK(hipMemset(d, 0, bytes ? bytes : 16));
    p->allocs.push_back(d);
    return d;
}

void compute(falcon_hip_pipeline * p, const pipe_slot & s, int n_past0, hipStream_t st) {
    if (s.group < 0) return;
    const int g = s.group;
    falcon_hip_stage_step(p->ctx[(size_t) g], p->tok_in[(size_t) g], p->hidden_in[(size_t) g], n_past0 + s.round, p->hidden_out[(size_t) g], p->tok_out[(size_t) g]);
    if (p->last) {
        const int slot = p->rounds_done + s.round;
        if (slot < p->n_ctx)
            HIP_CHECK(hipMemcpyAsync(p->hist + ((size_t) slot * p->G + g) * p->B, p->tok_out[(size_t) g], (size_t) p->B * 4, hipMemcpyDeviceToDevice, st));
    }
}

// one grouped RCCL exchange: every send and receive of the slot between ncclGroupStart and ncclGroupEnd
void exchange_rccl(falcon_hip_pipeline * p, const pipe_slot & s, hipStream_t st) {
    if (!s.n_ops) return;
    rccl_api * R = fq_rccl();
    const size_t nh = (size_t) p->B * p->hp.n_embd, nt = (size_t) p->B;
    RCCL_CHECK(R->ncclGroupStart());
    for (int i = 0; i < s.n_ops; ++i) {
        const pipe_op & o = s.op[i];
        const size_t g = (size_t) o.group;
        switch (o.kind) {
            case OP_SEND_HIDDEN: RCCL_CHECK(R->ncclSend(p->hidden_out[g], nh, ncclFloat32, o.peer, p->comm, st)); break;
            case OP_SEND_TOKEN:  RCCL_CHECK(R->ncclSend(p->tok_out[g],    nt, ncclInt32,   o.peer, p->comm, st)); break;
            case OP_RECV_HIDDEN: RCCL_CHECK(R->ncclRecv(p->hidden_in[g],  nh, ncclFloat32, o.peer, p->comm, st)); break;
            case OP_RECV_TOKEN:  RCCL_CHECK(R->ncclRecv(p->tok_in[g],     nt, ncclInt32,   o.peer, p->comm, st)); break;
        }
    }
    RCCL_CHECK(R->ncclGroupEnd());
}

double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec; }
// spin (politely) until *word == want; false after the transport's time-out
bool shm_wait(const shm_transport * t, const volatile uint32_t * word, uint32_t want, const char * what, int rank, int group) {
    const double t0 = now_s();
    for (unsigned spins = 0; __atomic_load_n(word, __ATOMIC_ACQUIRE) != want; ++spins) {
        if (spins > 2000) usleep(20); else if (spins > 200) sched_yield();
        if ((spins & 1023) == 1023 && now_s() - t0 > t->timeout_s) {
            fprintf(stderr, "falcon-hip: pipeline (shm transport): rank %d waited %.0f s for %s of group %d (have %u, want %u) -- a peer is gone or stuck\n",
                    rank, t->timeout_s, what, group, (unsigned) *word, want);
            return false;
        }
    }
    return true;
}
uint64_t fnv64(const void * data, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= ((const uint8_t *) data)[i]; h *= 1099511628211ull; } return h; }

// every rank of the job calls this with the same unique id: rank 0 creates the segment, the others attach; returns when all have
bool shm_attach(falcon_hip_pipeline * p, const void * unique_id, bool ipc) {
    shm_transport * t = new shm_transport();
    if (const char * e = getenv("FALCON_PIPE_SHM_TIMEOUT_S")) t->timeout_s = atof(e) > 0 ? atof(e) : t->timeout_s;
    const size_t E = (size_t) p->hp.n_embd, G = (size_t) p->G, B = (size_t) p->B;
    t->hid_bytes = (B * E * 4 + 63) & ~(size_t) 63; t->tok_bytes = (B * 4 + 63) & ~(size_t) 63;
    t->ipc = ipc;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle table entry");
    t->table_bytes = ipc ? (size_t) p->world * 64 : 0;
    t->tok_hdr_off = ipc ? 64 : 64 + t->hid_bytes;                           // (ipc: the two { seq, ack } headers only; the payloads are in device memory)
    t->box_stride = t->tok_hdr_off + 64 + (ipc ? 0 : t->tok_bytes);
    t->dev_stride = t->hid_bytes + t->tok_bytes;
    t->bytes = 64 + t->table_bytes + (size_t) p->world * G * t->box_stride;
    snprintf(t->name, sizeof(t->name), "/falcon_pipe_%016llx", (unsigned long long) fnv64(unique_id, FALCON_HIP_PIPELINE_ID_BYTES));
    t->owner = p->rank == 0;
    const double t0 = now_s();
    if (t->owner) {
        shm_unlink(t->name);                                        // (a stale segment of a job that died under this very id)
        t->fd = shm_open(t->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (t->fd < 0 || ftruncate(t->fd, (off_t) t->bytes) != 0) {
            fprintf(stderr, "falcon-hip: pipeline: shm_open / ftruncate(%s, %zu): %s\n", t->name, t->bytes, strerror(errno));
            if (t->fd >= 0) { close(t->fd); shm_unlink(t->name); }
            delete t; return false;
        }
    } else {
        for (;;) {
            t->fd = shm_open(t->name, O_RDWR, 0600);
            struct stat st;
            if (t->fd >= 0 && fstat(t->fd, &st) == 0 && (size_t) st.st_size >= t->bytes) break;
            if (t->fd >= 0) { close(t->fd); t->fd = -1; }
            if (now_s() - t0 > t->timeout_s) { fprintf(stderr, "falcon-hip: pipeline: rank %d found no segment %s of %zu bytes within %.0f s\n", p->rank, t->name, t->bytes, t->timeout_s); delete t; return false; }
            usleep(2000);
        }
    }
    t->base = (uint8_t *) mmap(nullptr, t->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
    if (t->base == (uint8_t *) MAP_FAILED) { fprintf(stderr, "falcon-hip: pipeline: mmap(%s): %s\n", t->name, strerror(errno)); close(t->fd); delete t; return false; }
    shm_header * h = (shm_header *) t->base;
    if (t->owner) {
        memset(t->base, 0, t->bytes);
        h->world = (uint32_t) p->world; h->G = (uint32_t) G; h->B = (uint32_t) B; h->E = (uint32_t) E;
        __atomic_store_n(&h->ready, 0x46504950u, __ATOMIC_RELEASE);
    } else if (!shm_wait(t, &h->ready, 0x46504950u, "the segment header", p->rank, -1)) { munmap(t->base, t->bytes); close(t->fd); delete t; return false; }
    if (h->world != (uint32_t) p->world || h->G != G || h->B != B || h->E != E) {
        fprintf(stderr, "falcon-hip: pipeline: rank %d: the job's segment says world %u, %u groups of %u, n_embd %u -- this rank was created with %d, %zu, %zu, %zu\n",
                p->rank, h->world, h->G, h->B, h->E, p->world, G, B, E);
        munmap(t->base, t->bytes); close(t->fd); delete t; return false;
    }
    if (ipc) {
        // this rank's device mailboxes, exported BEFORE it counts itself in: when every rank has attached, every handle is in the table
        hipIpcMemHandle_t hnd;
        bool ok = hipMalloc((void **) &t->dev_box, G * t->dev_stride) == hipSuccess && hipMemset(t->dev_box, 0, G * t->dev_stride) == hipSuccess &&
                  hipDeviceSynchronize() == hipSuccess && hipIpcGetMemHandle(&hnd, t->dev_box) == hipSuccess;
        if (ok) memcpy(t->base + 64 + (size_t) p->rank * 64, &hnd, 64);
        else { (void) hipGetLastError(); t->ipc_failed = true; }
    }
    __atomic_add_fetch(&h->attached, 1u, __ATOMIC_ACQ_REL);
    if (!shm_wait(t, &h->attached, (uint32_t) p->world, "every rank to attach", p->rank, -1)) { if (t->dev_box) (void) hipFree(t->dev_box); munmap(t->base, t->bytes); close(t->fd); if (t->owner) shm_unlink(t->name); delete t; return false; }
    if (ipc) {
        // open the successor's mailboxes (the only rank this one sends to: stage r -> r + 1, the last stage's tokens -> stage 0); every rank reports, and ALL
        // of them give the transport up together if one could not (the caller falls back to the host-staged form as one job)
        const int next = (p->rank + 1) % p->world;
        bool ok = !t->ipc_failed;
        if (ok) {
            hipIpcMemHandle_t hnd;
            memcpy(&hnd, t->base + 64 + (size_t) next * 64, 64);
            const hipError_t e = hipIpcOpenMemHandle((void **) &t->peer_box, hnd, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { fprintf(stderr, "falcon-hip: pipeline (ipc transport): rank %d could not open rank %d's mailboxes: %s\n", p->rank, next, hipGetErrorString(e)); (void) hipGetLastError(); t->peer_box = nullptr; ok = false; }
        }
        __atomic_add_fetch(ok ? &h->ipc_ok : &h->ipc_bad, 1u, __ATOMIC_ACQ_REL);
        const double t1 = now_s();
        while (__atomic_load_n(&h->ipc_ok, __ATOMIC_ACQUIRE) + __atomic_load_n(&h->ipc_bad, __ATOMIC_ACQUIRE) < (uint32_t) p->world && now_s() - t1 < t->timeout_s) usleep(200);
        if (__atomic_load_n(&h->ipc_ok, __ATOMIC_ACQUIRE) != (uint32_t) p->world) {
            if (p->rank == 0) fprintf(stderr, "falcon-hip: pipeline (ipc transport): %u of %d ranks could not export / open device mailboxes -- the transport is not available to this job\n",
                                      (unsigned) __atomic_load_n(&h->ipc_bad, __ATOMIC_ACQUIRE), p->world);
            if (t->peer_box) (void) hipIpcCloseMemHandle(t->peer_box);
            if (t->dev_box) (void) hipFree(t->dev_box);
            munmap(t->base, t->bytes); close(t->fd); if (t->owner) shm_unlink(t->name); delete t; return false;
        }
    }
    if (t->owner) { shm_unlink(t->name); t->name[0] = 0; }           // every rank holds its mapping now: the name can go, and a job that dies later leaves nothing in /dev/shm
    if (!ipc) {
        HIP_CHECK(hipHostMalloc(&t->stage_out, t->hid_bytes, hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc(&t->stage_in, t->hid_bytes, hipHostMallocDefault));
    }
    t->sent_h.assign(G, 0); t->sent_t.assign(G, 0); t->got_h.assign(G, 0); t->got_t.assign(G, 0);
    p->shm = t;
    return true;
}
void shm_detach(falcon_hip_pipeline * p) {
    shm_transport * t = p->shm;
    if (!t) return;
    if (t->stage_out) HIP_CHECK(hipHostFree(t->stage_out));
    if (t->stage_in) HIP_CHECK(hipHostFree(t->stage_in));
    if (t->peer_box) (void) hipIpcCloseMemHandle(t->peer_box);
    if (t->dev_box) (void) hipFree(t->dev_box);
    munmap(t->base, t->bytes); close(t->fd);
    if (t->owner && t->name[0]) shm_unlink(t->name);                // (normally gone since every rank attached)
    delete t; p->shm = nullptr;
}
// the slot's exchange over the mailboxes, BLOCKING on the host: sends first (they wait for nobody but the mailbox's ack), then the
// receives. `st` is the stream the copies run on; the caller has made it wait for the compute whose result is sent.
bool exchange_shm(falcon_hip_pipeline * p, const pipe_slot & s, hipStream_t st) {
    shm_transport * t = p->shm;
    const size_t nh = (size_t) p->B * p->hp.n_embd * 4, nt = (size_t) p->B * 4;
    for (int i = 0; i < s.n_ops; ++i) {
        const pipe_op & o = s.op[i];
        if (o.kind != OP_SEND_HIDDEN && o.kind != OP_SEND_TOKEN) continue;
        const bool tok = o.kind == OP_SEND_TOKEN; const size_t g = (size_t) o.group, n = tok ? nt : nh;
        shm_box * b = t->box(o.peer, p->G, o.group, tok);
        uint32_t & cnt = tok ? t->sent_t[g] : t->sent_h[g];
        if (t->ipc) {
            // device to device: wait until the receiver has taken the mailbox's previous message, copy straight into ITS device memory, publish
            if (!shm_wait(t, &b->ack, cnt, tok ? "the receiver to take the previous tokens" : "the receiver to take the previous residual rows", p->rank, o.group)) return false;
            HIP_CHECK(hipMemcpyAsync(t->peer_box + t->dev_off(o.group, tok), tok ? (const void *) p->tok_out[g] : (const void *) p->hidden_out[g], n, hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
            __atomic_store_n(&b->seq, ++cnt, __ATOMIC_RELEASE);
            continue;
        }
        HIP_CHECK(hipMemcpyAsync(t->stage_out, tok ? (const void *) p->tok_out[g] : (const void *) p->hidden_out[g], n, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (!shm_wait(t, &b->ack, cnt, tok ? "the receiver to take the previous tokens" : "the receiver to take the previous residual rows", p->rank, o.group)) return false;
        memcpy((uint8_t *) b + 64, t->stage_out, n);
        __atomic_store_n(&b->seq, ++cnt, __ATOMIC_RELEASE);
    }
    for (int i = 0; i < s.n_ops; ++i) {
        const pipe_op & o = s.op[i];
        if (o.kind != OP_RECV_HIDDEN && o.kind != OP_RECV_TOKEN) continue;
        const bool tok = o.kind == OP_RECV_TOKEN; const size_t g = (size_t) o.group, n = tok ? nt : nh;
        shm_box * b = t->box(p->rank, p->G, o.group, tok);
        uint32_t & cnt = tok ? t->got_t[g] : t->got_h[g];
        if (!shm_wait(t, &b->seq, cnt + 1, tok ? "the sampled tokens" : "the residual rows", p->rank, o.group)) return false;
        if (t->ipc) {
            HIP_CHECK(hipMemcpyAsync(tok ? (void *) p->tok_in[g] : (void *) p->hidden_in[g], t->dev_box + t->dev_off(o.group, tok), n, hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
            __atomic_store_n(&b->ack, ++cnt, __ATOMIC_RELEASE);
            continue;
        }
        memcpy(t->stage_in, (const uint8_t *) b + 64, n);
        HIP_CHECK(hipMemcpyAsync(tok ? (void *) p->tok_in[g] : (void *) p->hidden_in[g], t->stage_in, n, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        __atomic_store_n(&b->ack, ++cnt, __ATOMIC_RELEASE);
    }
    return true;
}

falcon_hip_pipeline * create(falcon_hip_model * m, int rank, int world, int n_groups, int batch, int n_ctx) {
    if (!m || world < 1 || rank < 0 || rank >= world || n_groups < 1 || batch < 1 || batch > 256 || n_ctx < 1) {
        fprintf(stderr, "falcon-hip: pipeline: bad arguments (rank %d of %d, %d groups of %d sequences, n_ctx %d)\n", rank, world, n_groups, batch, n_ctx);
        return nullptr;
    }
    if (world > 1 && n_groups < world) { fprintf(stderr, "falcon-hip: pipeline: %d stages need at least %d groups in flight (got %d)\n", world, world, n_groups); return nullptr; }
    falcon_hip_pipeline * p = new falcon_hip_pipeline();
    p->m = m; p->rank = rank; p->world = world; p->G = n_groups; p->B = batch; p->n_ctx = n_ctx;
    falcon_hip_model_get_hparams(m, &p->hp);
    p->first = p->hp.layer_begin == 0; p->last = p->hp.layer_end == p->hp.n_layer;
    if ((rank == 0) != p->first || (rank == world - 1) != p->last) {
        fprintf(stderr, "falcon-hip: pipeline: rank %d of %d holds blocks [%d, %d) of %d: the first rank must hold block 0, the last rank the last block\n",
                rank, world, p->hp.layer_begin, p->hp.layer_end, p->hp.n_layer);
        delete p; return nullptr;
    }
    const size_t E = (size_t) p->hp.n_embd;
    for (int g = 0; g < n_groups; ++g) {
        p->ctx.push_back(falcon_hip_context_create_seqs(m, n_ctx, batch, n_ctx));
        p->hidden_in.push_back((float *) palloc(p, (size_t) batch * E * 4));
        p->hidden_out.push_back((float *) palloc(p, (size_t) batch * E * 4));
        p->tok_in.push_back((int32_t *) palloc(p, (size_t) batch * 4));
        p->tok_out.push_back((int32_t *) palloc(p, (size_t) batch * 4));
        p->mb_hidden.push_back((float *) palloc(p, (size_t) batch * E * 4));
        p->mb_tok.push_back((int32_t *) palloc(p, (size_t) batch * 4));
    }
    if (p->last) p->hist = (int32_t *) palloc(p, (size_t) n_ctx * n_groups * batch * 4);
    HIP_CHECK(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; ++i) {
        HIP_CHECK(hipEventCreateWithFlags(&p->ev_compute[i], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&p->ev_comm[i], hipEventDisableTiming));
    }
    HIP_CHECK(hipEventCreateWithFlags(&p->ev_start, hipEventDisableTiming));
    return p;
}

}   // namespace

extern "C" {

static bool ipc_transport_selected() { const char * e = getenv("FALCON_PIPE_TRANSPORT"); return e && !strcmp(e, "ipc"); }
static bool shm_transport_selected() { const char * e = getenv("FALCON_PIPE_TRANSPORT"); return e && (!strcmp(e, "shm") || !strcmp(e, "ipc")); }      // (both ride on the segment)

int falcon_hip_pipeline_unique_id(void * id_out) {
    rccl_api * R = fq_rccl();
    if (!R && shm_transport_selected()) {                           // the host-staged transport only needs 128 bytes no other job has
        int fd = open("/dev/urandom", O_RDONLY);
        const bool ok = fd >= 0 && read(fd, id_out, FALCON_HIP_PIPELINE_ID_BYTES) == FALCON_HIP_PIPELINE_ID_BYTES;
        if (fd >= 0) close(fd);
        return ok ? 0 : -1;
    }
    if (!R) return -1;
    ncclUniqueId id;
    if (R->ncclGetUniqueId(&id) != ncclSuccess) return -1;
    memcpy(id_out, &id, FALCON_HIP_PIPELINE_ID_BYTES);
    return 0;
}

falcon_hip_pipeline * falcon_hip_pipeline_create(falcon_hip_model * m, int rank, int world, const void * unique_id, int n_groups, int batch, int n_ctx) {
    static_assert(sizeof(ncclUniqueId) == FALCON_HIP_PIPELINE_ID_BYTES, "unique id size");
    falcon_hip_pipeline * p = create(m, rank, world, n_groups, batch, n_ctx);
    if (!p || world == 1) return p;
    if (shm_transport_selected()) {                                 // ranks = processes of one node, hand-offs through host shared memory
        if (!unique_id || !shm_attach(p, unique_id, ipc_transport_selected())) { fprintf(stderr, "falcon-hip: pipeline: the %s transport could not be set up (rank %d of %d)\n", ipc_transport_selected() ? "ipc" : "shm", rank, world); falcon_hip_pipeline_free(p); return nullptr; }
        return p;
    }
    rccl_api * R = fq_rccl();
    if (!R || !unique_id) { fprintf(stderr, "falcon-hip: pipeline: %s\n", R ? "no unique id" : "RCCL is not available"); falcon_hip_pipeline_free(p); return nullptr; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t rc = R->ncclCommInitRank(&p->comm, world, id, rank);
    if (rc != ncclSuccess) { fprintf(stderr, "falcon-hip: pipeline: ncclCommInitRank(rank %d of %d): %s\n", rank, world, R->ncclGetErrorString(rc)); p->comm = nullptr; falcon_hip_pipeline_free(p); return nullptr; }
    return p;
}

falcon_hip_pipeline * falcon_hip_pipeline_create_local(falcon_hip_model * m, int rank, int world, int n_groups, int batch, int n_ctx) {
    falcon_hip_pipeline * p = create(m, rank, world, n_groups, batch, n_ctx);
    if (p) p->local = true;
    return p;
}

// The local transport with every hand-off sent through RCCL instead of a device copy: ONE communicator of one rank (this process,
// this GPU), every message an ncclSend addressed to rank 0 itself matched by an ncclRecv from rank 0, all sends and receives of a slot
// inside one ncclGroupStart / ncclGroupEnd on the pipeline's second stream, ordered against the stage steps by the events of the
// overlapped schedule. It runs the library's real RCCL binding (rccl_dyn.h: dlopen, ncclGetUniqueId, ncclCommInitRank, grouped
// ncclSend / ncclRecv of f32 rows and i32 token ids, ncclCommCount, ncclCommDestroy) on a box with a single GPU, where a job of two
// ranks is refused ("Duplicate GPU detected"). Returns 0, or -1 when RCCL is not available / refuses.
int falcon_hip_pipeline_local_attach_rccl(falcon_hip_pipeline ** ranks, int world) {
    if (world < 1 || !ranks || !ranks[0]) return -1;
    for (int r = 0; r < world; ++r) if (!ranks[r] || !ranks[r]->local || ranks[r]->loop) return -1;
    rccl_api * R = fq_rccl();
    if (!R) return -1;
    ncclUniqueId id;
    ncclResult_t rc = R->ncclGetUniqueId(&id);
    if (rc != ncclSuccess) { fprintf(stderr, "falcon-hip: pipeline: ncclGetUniqueId: %s\n", R->ncclGetErrorString(rc)); return -1; }
    rc = R->ncclCommInitRank(&ranks[0]->comm, 1, id, 0);
    if (rc != ncclSuccess) { fprintf(stderr, "falcon-hip: pipeline: ncclCommInitRank(rank 0 of 1): %s\n", R->ncclGetErrorString(rc)); ranks[0]->comm = nullptr; return -1; }
    for (int r = 0; r < world; ++r) ranks[r]->loop = ranks[0];
    return 0;
}

// ranks of the RCCL communicator the pipeline exchanges over (ncclCommCount): 1 for a one-rank pipeline, 0 for the local transport
// (the loop-back transport: its one-rank communicator's count)
int falcon_hip_pipeline_rccl_ranks(falcon_hip_pipeline * p) {
    if (p && p->local && p->loop && p->loop->comm) {
        int n = -1;
        if (fq_rccl()->ncclCommCount(p->loop->comm, &n) != ncclSuccess) return -1;
        return n;
    }
    if (!p || p->local || p->shm) return 0;
    if (!p->comm) return 1;
    int n = -1;
    if (fq_rccl()->ncclCommCount(p->comm, &n) != ncclSuccess) return -1;
    return n;
}

// A ring of one small message over a FRESH communicator: rank r sends 256 floats to r + 1 and receives from r - 1 (grouped ncclSend / ncclRecv, the pipeline's
// own call pattern), checks what arrived, destroys the communicator. The launcher runs it in a CHILD process per rank under a time-out before the job proper
// (bench_pipeline.py): RCCL between these ranks has never run where this library is developed, and a hung or refused collective must cost the job a
// fall-back to the host-staged transport, not its life. Returns 0, or a non-zero code with a message on stderr.
int falcon_hip_rccl_selftest(int rank, int world, const void * unique_id, int device) {
    if (world < 2 || rank < 0 || rank >= world || !unique_id) return 1;
    rccl_api * R = fq_rccl();
    if (!R) return 2;
    if (hipSetDevice(device) != hipSuccess) { fprintf(stderr, "falcon-hip: rccl selftest: hipSetDevice(%d) failed\n", device); return 3; }
    ncclUniqueId id; memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t rc = R->ncclCommInitRank(&comm, world, id, rank);
    if (rc != ncclSuccess) { fprintf(stderr, "falcon-hip: rccl selftest: ncclCommInitRank(rank %d of %d): %s\n", rank, world, R->ncclGetErrorString(rc)); return 4; }
    const int n = 256, next = (rank + 1) % world, prev = (rank + world - 1) % world;
    float host[256], * out = nullptr, * in = nullptr;
    hipStream_t st = nullptr;
    int bad = 0;
    if (hipMalloc((void **) &out, n * 4) != hipSuccess || hipMalloc((void **) &in, n * 4) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) bad = 5;
    if (!bad) {
        for (int i = 0; i < n; ++i) host[i] = (float)(1000 * rank + i);
        if (hipMemcpy(out, host, n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(in, 0, n * 4) != hipSuccess) bad = 5;
    }
    if (!bad) {
        if (R->ncclGroupStart() != ncclSuccess) bad = 6;
        if (!bad && R->ncclSend(out, n, ncclFloat32, next, comm, st) != ncclSuccess) bad = 6;
        if (!bad && R->ncclRecv(in, n, ncclFloat32, prev, comm, st) != ncclSuccess) bad = 6;
        if (R->ncclGroupEnd() != ncclSuccess) bad = 6;
        if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = 7;
        if (!bad && hipMemcpy(host, in, n * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 7;
        if (!bad) for (int i = 0; i < n; ++i) if (host[i] != (float)(1000 * prev + i)) { bad = 8; break; }
    }
    if (bad) fprintf(stderr, "falcon-hip: rccl selftest: rank %d of %d failed at step %d\n", rank, world, bad);
    if (st) (void) hipStreamDestroy(st);
    if (out) (void) hipFree(out);
    if (in) (void) hipFree(in);
    R->ncclCommDestroy(comm);
    return bad;
}

// how this rank's hand-offs travel: 0 = nowhere (one stage), 1 = RCCL send / recv, 2 = device copies in one process (local), 3 = the local job
// over a one-rank RCCL communicator (loop-back), 4 = host shared memory between processes (FALCON_PIPE_TRANSPORT=shm), 5 = device-to-device copies into
// the peer's IPC-exported mailboxes (FALCON_PIPE_TRANSPORT=ipc)
int falcon_hip_pipeline_transport(falcon_hip_pipeline * p) {
    if (!p) return -1;
    if (p->local) return p->loop ? 3 : 2;
    if (p->shm) return p->shm->ipc ? 5 : 4;
    return p->comm ? 1 : 0;
}

void falcon_hip_pipeline_free(falcon_hip_pipeline * p) {
    if (!p) return;
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
    if (p->comm_stream) HIP_CHECK(hipStreamSynchronize(p->comm_stream));
    if (p->comm) fq_rccl()->ncclCommDestroy(p->comm);
    shm_detach(p);
    for (falcon_hip_context * c : p->ctx) falcon_hip_context_free(c);
    for (int i = 0; i < 4; ++i) { if (p->ev_compute[i]) HIP_CHECK(hipEventDestroy(p->ev_compute[i])); if (p->ev_comm[i]) HIP_CHECK(hipEventDestroy(p->ev_comm[i])); }
    if (p->ev_start) HIP_CHECK(hipEventDestroy(p->ev_start));
    if (p->comm_stream) HIP_CHECK(hipStreamDestroy(p->comm_stream));
    for (void * d : p->allocs) HIP_CHECK(hipFree(d));
    delete p;
}

int falcon_hip_pipeline_set_tokens(falcon_hip_pipeline * p, const int32_t * tokens) {
    for (int i = 0; i < p->G * p->B; ++i) if (tokens[i] < 0 || tokens[i] >= p->hp.n_vocab) {
        fprintf(stderr, "falcon-hip: pipeline: token id %d of sequence %d is outside [0, %d)\n", tokens[i], i, p->hp.n_vocab);
        return 2;
    }
    hipStream_t st = fq_ctx().stream;
    for (int g = 0; g < p->G; ++g) HIP_CHECK(hipMemcpyAsync(p->tok_in[(size_t) g], tokens + (size_t) g * p->B, (size_t) p->B * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// Advance every sequence by `rounds` tokens from position n_past0: all device work is enqueued, nothing is waited for
// (falcon_hip_pipeline_get_history or ggml_hip_synchronize waits). Every rank of the job makes the same call.
int falcon_hip_pipeline_run(falcon_hip_pipeline * p, int rounds, int n_past0) {
    if (rounds < 1 || n_past0 < 0 || n_past0 + rounds > p->n_ctx) { fprintf(stderr, "falcon-hip: pipeline: %d rounds from position %d exceed n_ctx %d\n", rounds, n_past0, p->n_ctx); return 1; }
    if (p->world > 1 && (p->local || (!p->comm && !p->shm))) { fprintf(stderr, "falcon-hip: pipeline: this rank has no RCCL communicator (local transport: falcon_hip_pipeline_run_local)\n"); return 1; }
    hipStream_t st = fq_ctx().stream;
    const int P = p->world, G = p->G, W = rounds * G, T = slot_count(P, G, W);
    if (P > 1 && p->shm) {
        // the same slots, the exchange blocking on the host. Overlapped schedule: the exchange of slot t (result of slot t - 1 out, input of slot
        // t + 1 in) is run AFTER slot t's stage step has been enqueued, on the second stream behind the event of slot t - 1's step -- the host
        // waits in it while the device computes slot t, the overlap the RCCL form gets from its second stream.
        if (!overlapped(P, G)) {
            for (int t = 0; t < T; ++t) {
                const pipe_slot s = slot_of(p->rank, P, G, W, t);
                if (!exchange_shm(p, s, st)) return 4;
                compute(p, s, n_past0, st);
            }
        } else {
            HIP_CHECK(hipEventRecord(p->ev_start, st));
            HIP_CHECK(hipStreamWaitEvent(p->comm_stream, p->ev_start, 0));
            for (int t = 0; t < T; ++t) {
                const pipe_slot s = slot_of(p->rank, P, G, W, t);
                if (t > 0) HIP_CHECK(hipStreamWaitEvent(st, p->ev_comm[(t - 1) & 3], 0));                  // the input it received one slot ago
                compute(p, s, n_past0, st);
                HIP_CHECK(hipEventRecord(p->ev_compute[t & 3], st));
                if (t > 0) HIP_CHECK(hipStreamWaitEvent(p->comm_stream, p->ev_compute[(t - 1) & 3], 0));   // the result it sends
                if (!exchange_shm(p, s, p->comm_stream)) return 4;
                HIP_CHECK(hipEventRecord(p->ev_comm[t & 3], p->comm_stream));
            }
            HIP_CHECK(hipStreamWaitEvent(st, p->ev_comm[(T - 1) & 3], 0));
        }
        p->rounds_done += rounds;
        return 0;
    }
    if (P == 1) {
        for (int t = 0; t < T; ++t) {
            const pipe_slot s = slot_of(0, 1, G, W, t);
            compute(p, s, n_past0, st);
            HIP_CHECK(hipMemcpyAsync(p->tok_in[(size_t) s.group], p->tok_out[(size_t) s.group], (size_t) p->B * 4, hipMemcpyDeviceToDevice, st));   // the sampled tokens are the next inputs
        }
    } else if (!overlapped(P, G)) {
        for (int t = 0; t < T; ++t) {
            const pipe_slot s = slot_of(p->rank, P, G, W, t);
            exchange_rccl(p, s, st);
            compute(p, s, n_past0, st);
        }
    } else {
        HIP_CHECK(hipEventRecord(p->ev_start, st));
        HIP_CHECK(hipStreamWaitEvent(p->comm_stream, p->ev_start, 0));                 // everything enqueued before this call
        for (int t = 0; t < T; ++t) {
            const pipe_slot s = slot_of(p->rank, P, G, W, t);
            if (t > 0) HIP_CHECK(hipStreamWaitEvent(p->comm_stream, p->ev_compute[(t - 1) & 3], 0));   // the result it sends
            exchange_rccl(p, s, p->comm_stream);
            HIP_CHECK(hipEventRecord(p->ev_comm[t & 3], p->comm_stream));
            if (t > 0) HIP_CHECK(hipStreamWaitEvent(st, p->ev_comm[(t - 1) & 3], 0));                  // the input it received one slot ago
            compute(p, s, n_past0, st);
            HIP_CHECK(hipEventRecord(p->ev_compute[t & 3], st));
        }
        HIP_CHECK(hipStreamWaitEvent(st, p->ev_comm[(T - 1) & 3], 0));
    }
    p->rounds_done += rounds;
    return 0;
}

// The same schedule with every rank in THIS process on one device: hand-off through the receiver's mailboxes by device copies
// on the library stream (sends of a slot, then its receives, then its stage steps -- what one grouped exchange guarantees).
int falcon_hip_pipeline_run_local(falcon_hip_pipeline ** ranks, int world, int rounds, int n_past0) {
    if (world < 1) return 1;
    const int G = ranks[0]->G, B = ranks[0]->B, W = rounds * G, T = slot_count(world, G, W);
    for (int r = 0; r < world; ++r) {
        falcon_hip_pipeline * p = ranks[r];
        if (!p->local || p->world != world || p->rank != r || p->G != G || p->B != B || n_past0 + rounds > p->n_ctx || n_past0 < 0 || rounds < 1) {
            fprintf(stderr, "falcon-hip: pipeline: run_local needs the %d local-transport ranks of one job, in rank order, with room for the rounds\n", world);
            return 1;
        }
    }
    hipStream_t st = fq_ctx().stream;
    const size_t nh = (size_t) B * ranks[0]->hp.n_embd * 4, nt = (size_t) B * 4;
    std::vector<pipe_slot> s((size_t) world);
    falcon_hip_pipeline * lp = ranks[0]->loop;
    if (lp) {
        // loop-back over RCCL: a message's send and its receive fall into the same slot on both ends (slot_of), so the slot's whole traffic is
        // one group on the second stream; message k of the group = send k and receive k, posted in the same order (RCCL matches them in order)
        rccl_api * R = fq_rccl();
        hipStream_t cs = lp->comm_stream;
        HIP_CHECK(hipEventRecord(lp->ev_start, st));
        HIP_CHECK(hipStreamWaitEvent(cs, lp->ev_start, 0));
        for (int t = 0; t < T; ++t) {
            for (int r = 0; r < world; ++r) s[(size_t) r] = slot_of(r, world, G, W, t);
            if (t > 0) HIP_CHECK(hipStreamWaitEvent(cs, lp->ev_compute[(t - 1) & 3], 0));              // the results it sends
            int n_msgs = 0;
            for (int r = 0; r < world; ++r) for (int i = 0; i < s[(size_t) r].n_ops; ++i) n_msgs += s[(size_t) r].op[i].kind <= OP_SEND_TOKEN;
            if (n_msgs) {
                RCCL_CHECK(R->ncclGroupStart());
                for (int r = 0; r < world; ++r) for (int i = 0; i < s[(size_t) r].n_ops; ++i) {
                    const pipe_op & o = s[(size_t) r].op[i]; const size_t g = (size_t) o.group;
                    if (o.kind == OP_SEND_HIDDEN) {
                        RCCL_CHECK(R->ncclSend(ranks[r]->hidden_out[g], nh / 4, ncclFloat32, 0, lp->comm, cs));
                        RCCL_CHECK(R->ncclRecv(ranks[o.peer]->hidden_in[g], nh / 4, ncclFloat32, 0, lp->comm, cs));
                    } else if (o.kind == OP_SEND_TOKEN) {
                        RCCL_CHECK(R->ncclSend(ranks[r]->tok_out[g], nt / 4, ncclInt32, 0, lp->comm, cs));
                        RCCL_CHECK(R->ncclRecv(ranks[o.peer]->tok_in[g], nt / 4, ncclInt32, 0, lp->comm, cs));
                    }
                }
                RCCL_CHECK(R->ncclGroupEnd());
            }
            HIP_CHECK(hipEventRecord(lp->ev_comm[t & 3], cs));
            HIP_CHECK(hipStreamWaitEvent(st, lp->ev_comm[t & 3], 0));                                   // the inputs this slot's steps read
            for (int r = 0; r < world; ++r) {
                compute(ranks[r], s[(size_t) r], n_past0, st);
                if (world == 1) {                                                                       // one stage: the sampled tokens are the next inputs, through RCCL too
                    HIP_CHECK(hipEventRecord(lp->ev_start, st));
                    HIP_CHECK(hipStreamWaitEvent(cs, lp->ev_start, 0));
                    RCCL_CHECK(R->ncclGroupStart());
                    RCCL_CHECK(R->ncclSend(ranks[0]->tok_out[(size_t) s[0].group], nt / 4, ncclInt32, 0, lp->comm, cs));
                    RCCL_CHECK(R->ncclRecv(ranks[0]->tok_in[(size_t) s[0].group], nt / 4, ncclInt32, 0, lp->comm, cs));
                    RCCL_CHECK(R->ncclGroupEnd());
                }
            }
            HIP_CHECK(hipEventRecord(lp->ev_compute[t & 3], st));
        }
        HIP_CHECK(hipEventRecord(lp->ev_comm[T & 3], cs));
        HIP_CHECK(hipStreamWaitEvent(st, lp->ev_comm[T & 3], 0));
        for (int r = 0; r < world; ++r) ranks[r]->rounds_done += rounds;
        return 0;
    }
    for (int t = 0; t < T; ++t) {
        for (int r = 0; r < world; ++r) s[(size_t) r] = slot_of(r, world, G, W, t);
        for (int r = 0; r < world; ++r) for (int i = 0; i < s[(size_t) r].n_ops; ++i) {
            const pipe_op & o = s[(size_t) r].op[i]; const size_t g = (size_t) o.group;
            if (o.kind == OP_SEND_HIDDEN) HIP_CHECK(hipMemcpyAsync(ranks[o.peer]->mb_hidden[g], ranks[r]->hidden_out[g], nh, hipMemcpyDeviceToDevice, st));
            if (o.kind == OP_SEND_TOKEN)  HIP_CHECK(hipMemcpyAsync(ranks[o.peer]->mb_tok[g],    ranks[r]->tok_out[g],    nt, hipMemcpyDeviceToDevice, st));
        }
        for (int r = 0; r < world; ++r) for (int i = 0; i < s[(size_t) r].n_ops; ++i) {
            const pipe_op & o = s[(size_t) r].op[i]; const size_t g = (size_t) o.group;
            if (o.kind == OP_RECV_HIDDEN) HIP_CHECK(hipMemcpyAsync(ranks[r]->hidden_in[g], ranks[r]->mb_hidden[g], nh, hipMemcpyDeviceToDevice, st));
            if (o.kind == OP_RECV_TOKEN)  HIP_CHECK(hipMemcpyAsync(ranks[r]->tok_in[g],    ranks[r]->mb_tok[g],    nt, hipMemcpyDeviceToDevice, st));
        }
        for (int r = 0; r < world; ++r) {
            compute(ranks[r], s[(size_t) r], n_past0, st);
            if (world == 1) HIP_CHECK(hipMemcpyAsync(ranks[0]->tok_in[(size_t) s[0].group], ranks[0]->tok_out[(size_t) s[0].group], nt, hipMemcpyDeviceToDevice, st));
        }
    }
    for (int r = 0; r < world; ++r) ranks[r]->rounds_done += rounds;
    return 0;
}

// last rank: the tokens sampled in rounds [first_round, first_round + n_rounds) since the pipeline was created, [round][G * B]
// (sequence i = g * B + b). Waits for the device. Returns 0, -1 on another rank, 3 if an in-launch hand-off timed out.
int falcon_hip_pipeline_get_history(falcon_hip_pipeline * p, int32_t * out, int first_round, int n_rounds) {
    hipStream_t st = fq_ctx().stream;
    if (!p->last) { HIP_CHECK(hipStreamSynchronize(st)); return -1; }
    if (first_round < 0 || n_rounds < 0 || first_round + n_rounds > p->rounds_done || first_round + n_rounds > p->n_ctx) return 1;
    HIP_CHECK(hipMemcpyAsync(out, p->hist + (size_t) first_round * p->G * p->B, (size_t) n_rounds * p->G * p->B * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (falcon_hip_context * c : p->ctx) if (falcon_hip_context_sync_error(c)) return 3;
    return 0;
}

// host only (tests): slot `slot` of rank `rank` when `rounds` rounds of n_groups groups run over `world` ranks:
// out[0] = number of exchange ops (<= 2), out[1 + 3 i ..] = { kind (0 send hidden, 1 send tokens, 2 recv hidden, 3 recv tokens),
// group, peer }, out[7] = group computed in this slot (-1: none), out[8] = its round. Returns the number of slots of the run.
int falcon_hip_pipeline_schedule(int rank, int world, int n_groups, int rounds, int slot, int * out) {
    const int W = rounds * n_groups, T = slot_count(world, n_groups, W);
    if (slot < 0 || slot >= T) return T;
    const pipe_slot s = slot_of(rank, world, n_groups, W, slot);
    for (int i = 0; i < 9; ++i) out[i] = -1;
    out[0] = s.n_ops;
    for (int i = 0; i < s.n_ops; ++i) { out[1 + 3 * i] = s.op[i].kind; out[2 + 3 * i] = s.op[i].group; out[3 + 3 * i] = s.op[i].peer; }
    out[7] = s.group; out[8] = s.round;
    return T;
}

}
