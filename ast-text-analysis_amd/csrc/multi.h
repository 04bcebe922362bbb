// multi.h -- several devices in ONE process (include/east_hip.h, "Several devices in one process").
//
// Every document is an independent annotated suffix array (east/relevance.py:41-46) and every (keyphrase, document)
// score is independent (east/applications.py:43-52): a collection shards at document granularity, one handle -- one
// AST shard -- per device, each driven by a host thread of the library's own (distinct handles are thread-safe, the
// process-wide east_hip_debug_* knobs are never touched here).  No collective on the build path; the K x D_local
// score blocks are assembled with ONE all-gather: RCCL (librccl.so, loaded at run time: ncclCommInitAll + a grouped
// ncclAllGather over xGMI, every block padded to the widest shard), or device-to-device copies to the first shard's
// device where RCCL cannot run (logical shards that share a device -- the single-GPU tests -- or no librccl.so).
// No torch, no process spawn: what `east -g N` runs by default (east/main.py).
#pragma once
#include <dlfcn.h>

typedef int (*nccl_comm_init_all_fn)(void **, int, const int *);
typedef int (*nccl_comm_destroy_fn)(void *);
typedef int (*nccl_all_gather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_group_fn)(void);
typedef const char *(*nccl_error_string_fn)(int);
#define EAST_NCCL_DOUBLE 8          // rccl.h: ncclFloat64 = ncclDouble = 8

struct RcclApi {
    void *lib = nullptr;
    nccl_comm_init_all_fn comm_init_all = nullptr;
    nccl_comm_destroy_fn comm_destroy = nullptr;
    nccl_all_gather_fn all_gather = nullptr;
    nccl_group_fn group_start = nullptr, group_end = nullptr;
    nccl_error_string_fn error_string = nullptr;
    bool overridden = false;    // EAST_HIP_RCCL_LIB names the library (the tests: a stub that records the calls it gets)
    bool load()
    {
        if (lib) return true;
        const char *forced_lib = getenv("EAST_HIP_RCCL_LIB");
        if (forced_lib && *forced_lib) {
            lib = dlopen(forced_lib, RTLD_NOW | RTLD_LOCAL);
            overridden = lib != nullptr;
        } else {
            // (a copy the process already holds -- torch brings its own -- before the system's)
            const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
            for (int pass = 0; pass < 2 && !lib; pass++)
                for (const char *name : names) {
                    lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                    if (lib) break;
                }
        }
        if (!lib) return false;
        comm_init_all = (nccl_comm_init_all_fn)dlsym(lib, "ncclCommInitAll");
        comm_destroy = (nccl_comm_destroy_fn)dlsym(lib, "ncclCommDestroy");
        all_gather = (nccl_all_gather_fn)dlsym(lib, "ncclAllGather");
        group_start = (nccl_group_fn)dlsym(lib, "ncclGroupStart");
        group_end = (nccl_group_fn)dlsym(lib, "ncclGroupEnd");
        error_string = (nccl_error_string_fn)dlsym(lib, "ncclGetErrorString");
        if (!comm_init_all || !comm_destroy || !all_gather || !group_start || !group_end) { lib = nullptr; return false; }
        return true;
    }
};

struct east_hip_group {
    std::vector<east_hip_index *> shard;        // owned; shard[s] lives on devices[s]
    std::vector<int> devices;
    std::vector<int32_t> first_doc;             // n_shards + 1: shard s holds the documents [first_doc[s], first_doc[s + 1])
    int32_t n_docs = 0;
    bool built = false;
    // assembly buffers: per shard the padded block (K x widest) and the gathered blocks (G x K x widest), on the first
    // shard's device also the K x D table and the shard bounds
    struct Buf { double *send = nullptr, *recv = nullptr; size_t send_cap = 0, recv_cap = 0; };
    std::vector<Buf> buf;
    double *packed = nullptr;
    size_t packed_cap = 0;
    u32 *d_first = nullptr;
    std::vector<hipEvent_t> done;               // per shard: its block is in place (copy path)
    RcclApi rccl;
    std::vector<void *> comms;
    int gather_mode = -1;                       // -1: not decided; 1: RCCL all-gather; 2: copies to the first shard's device
    double build_ms = -1.0, score_ms = -1.0, gather_ms = -1.0;
};

// Contiguous blocks of documents balanced by size (the rule of east/parallel.py: shard_documents -- block r ends with the
// first document at which the running size reaches r / G of the total); blocks may be empty with fewer documents than shards.
static void shard_by_size(const i64 *sizes, int32_t n, int32_t n_shards, std::vector<int32_t> &first)
{
    first.assign((size_t)n_shards + 1, 0);
    std::vector<double> csum((size_t)n);
    double run = 0.0;
    for (int32_t i = 0; i < n; i++) { run += (double)sizes[i]; csum[i] = run; }
    for (int32_t r = 1; r < n_shards; r++) {
        int32_t cut = 0;
        if (n) {
            const double target = run * (double)r / (double)n_shards;
            cut = (int32_t)(std::lower_bound(csum.begin(), csum.end(), target) - csum.begin()) + 1;
        }
        first[r] = std::min(std::max(cut, first[r - 1]), n);
    }
    first[n_shards] = n;
}

// fn(s) for every shard, each on a thread of its own (one shard: the caller's); the first error is thrown in the caller
template <class F> static void on_every_shard(east_hip_group *g, F fn)
{
    const size_t G = g->shard.size();
    std::vector<EastError> errs(G, EastError{0, ""});
    auto body = [&](size_t s) {
        try {
            fn((int)s);
        } catch (const EastError &e) {
            errs[s] = e;
        } catch (const std::exception &e) {
            errs[s] = EastError{EAST_HIP_ERR_INTERNAL, e.what()};
        }
        restore_device();                        // (the worker's own thread-local "device on entry")
    };
    if (G == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        for (size_t s = 0; s < G; s++) th.emplace_back(body, s);
        for (auto &t : th) t.join();
    }
    for (size_t s = 0; s < G; s++)
        if (errs[s].code) east_throw(errs[s].code, "shard " + std::to_string(s) + " (device " + std::to_string(g->devices[s]) + "): " + errs[s].msg);
}

static void group_check(east_hip_group *g, bool need_built)
{
    if (!g) east_throw(EAST_HIP_ERR_INVALID, "null group");
    if (need_built && !g->built) east_throw(EAST_HIP_ERR_NOT_BUILT, "no collection has been built on this group");
}

static double wall_ms_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// table (K x D_s, row-major) -> the padded block (K x width), zeros behind a row's D_s entries
__global__ __launch_bounds__(BLOCK) void group_pad_kernel(const double *__restrict__ table, u32 K, u32 Ds, u32 width,
                                                          double *__restrict__ send)
{
    const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (u64)K * width) return;
    const u32 k = (u32)(i / width), j = (u32)(i - (u64)k * width);
    send[i] = j < Ds ? table[(u64)k * Ds + j] : 0.0;
}

// the gathered blocks (G x K x width) -> the K x D table: out[k * D + first[s] + j] = recv[(s * K + k) * width + j]
__global__ __launch_bounds__(BLOCK) void group_pack_kernel(const double *__restrict__ recv, const u32 *__restrict__ first, u32 G,
                                                           u32 K, u32 D, u32 width, double *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (u64)K * D) return;
    const u32 k = (u32)(i / D), d = (u32)(i - (u64)k * D);
    u32 s = 0;
    while (s + 1 < G && first[s + 1] <= d) s++;
    out[i] = recv[((u64)s * K + k) * width + (d - first[s])];
}

static void grow(double *&p, size_t &cap, size_t bytes)
{
    if (bytes <= cap) return;
    if (p) HIP_CHECK(hipFree(p));
    p = nullptr;
    cap = 0;
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); east_throw(EAST_HIP_ERR_OOM, "hipMalloc of the score blocks of a group failed"); }
    p = (double *)q;
    cap = bytes;
}

// RCCL where every shard has a device of its own (EAST_HIP_GROUP_GATHER=copy / rccl overrides), copies otherwise
static int decide_gather_mode(east_hip_group *g)
{
    if (g->gather_mode > 0) return g->gather_mode;
    const char *forced = getenv("EAST_HIP_GROUP_GATHER");
    bool distinct = true;
    for (size_t a = 0; a < g->devices.size(); a++)
        for (size_t b = a + 1; b < g->devices.size(); b++) distinct = distinct && g->devices[a] != g->devices[b];
    int mode = distinct ? 1 : 2;
    if (forced && !strcmp(forced, "copy")) mode = 2;
    // (RCCL itself refuses a communicator with one device twice; a library named by EAST_HIP_RCCL_LIB -- the stub of the
    // gpu tier, which moves the blocks with device-to-device copies -- takes logical shards too: the grouped all-gather
    // with G > 1, its call order, counts and device switches then run on a one-GPU box)
    if (forced && !strcmp(forced, "rccl") && !distinct && g->rccl.load() && g->rccl.overridden) mode = 1;
    if (mode == 1) {
        bool ok = g->rccl.load();
        if (ok) {
            g->comms.assign(g->devices.size(), nullptr);
            const int rc = g->rccl.comm_init_all(g->comms.data(), (int)g->devices.size(), g->devices.data());
            if (rc != 0) {
                g->comms.clear();
                ok = false;
                if (forced && !strcmp(forced, "rccl"))
                    east_throw(EAST_HIP_ERR_HIP, std::string("ncclCommInitAll failed: ") + (g->rccl.error_string ? g->rccl.error_string(rc) : "?"));
            }
        } else if (forced && !strcmp(forced, "rccl")) {
            east_throw(EAST_HIP_ERR_HIP, "librccl.so cannot be loaded");
        }
        if (!ok) mode = 2;
    }
    g->gather_mode = mode;
    return mode;
}

static void group_destroy(east_hip_group *g)
{
    if (!g) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (void *c : g->comms)
        if (c) (void)g->rccl.comm_destroy(c);
    for (size_t s = 0; s < g->shard.size(); s++) {
        if (!g->shard[s]) continue;
        (void)hipSetDevice(g->devices[s]);
        if (g->shard[s]->stream) (void)hipStreamSynchronize(g->shard[s]->stream);
        if (s < g->buf.size()) {
            if (g->buf[s].send) (void)hipFree(g->buf[s].send);
            if (g->buf[s].recv) (void)hipFree(g->buf[s].recv);
        }
        if (s < g->done.size() && g->done[s]) (void)hipEventDestroy(g->done[s]);
        if (s == 0) {
            if (g->packed) (void)hipFree(g->packed);
            if (g->d_first) (void)hipFree(g->d_first);
        }
        east_hip_destroy(g->shard[s]);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    delete g;
}

// the all-gather of the shards' padded blocks + the K x D table on the first shard's device (left in g->packed)
static void group_allgather(east_hip_group *g, u32 K)
{
    const u32 G = (u32)g->shard.size(), D = (u32)g->n_docs;
    u32 width = 1;
    for (u32 s = 0; s < G; s++) width = std::max<u32>(width, (u32)(g->first_doc[s + 1] - g->first_doc[s]));
    const size_t block = (size_t)K * width * 8;
    const int mode = decide_gather_mode(g);
    // buffers and the padded blocks, every shard on its own stream
    on_every_shard(g, [&](int s) {
        east_hip_index *h = g->shard[s];
        use_device(h);
        grow(g->buf[s].send, g->buf[s].send_cap, block);
        if (mode == 1 || s == 0) grow(g->buf[s].recv, g->buf[s].recv_cap, block * G);
        const u32 Ds = (u32)(g->first_doc[s + 1] - g->first_doc[s]);
        if (Ds) {
            hipLaunchKernelGGL(group_pad_kernel, dim3(ceil_div_u32((u64)K * width, BLOCK)), dim3(BLOCK), 0, h->stream,
                               (const double *)h->table, K, Ds, width, g->buf[s].send);
            HIP_CHECK(hipGetLastError());
        } else {
            HIP_CHECK(hipMemsetAsync(g->buf[s].send, 0, block, h->stream));
        }
    });
    east_hip_index *h0 = g->shard[0];
    if (mode == 1) {
        // one grouped all-gather: every shard's block to every device, over xGMI between distinct devices
        // (nothing throws between group start and group end: a group left open would poison the communicator's next call)
        int rc = g->rccl.group_start();
        hipError_t dev_rc = hipSuccess;
        const bool opened = rc == 0;
        for (u32 s = 0; s < G && rc == 0 && dev_rc == hipSuccess; s++) {
            dev_rc = hipSetDevice(g->devices[s]);
            if (dev_rc == hipSuccess)
                rc = g->rccl.all_gather(g->buf[s].send, g->buf[s].recv, (size_t)K * width, EAST_NCCL_DOUBLE, g->comms[s], g->shard[s]->stream);
        }
        if (opened) {
            const int rc_end = g->rccl.group_end();
            if (rc == 0) rc = rc_end;
        }
        if (rc != 0 || dev_rc != hipSuccess) {
            // the other shards' blocks may be in flight: every stream is drained before the error goes up, and the first
            // shard's device is current again (guarded() then restores the caller's)
            for (u32 s = 0; s < G; s++)
                if (hipSetDevice(g->devices[s]) == hipSuccess) (void)hipStreamSynchronize(g->shard[s]->stream);
            (void)hipSetDevice(g->devices[0]);
            (void)hipGetLastError();
            if (dev_rc != hipSuccess) east_throw(EAST_HIP_ERR_HIP, std::string("hipSetDevice inside the grouped all-gather failed: ") + hipGetErrorString(dev_rc));
            east_throw(EAST_HIP_ERR_HIP, std::string("ncclAllGather failed: ") + (g->rccl.error_string ? g->rccl.error_string(rc) : "?"));
        }
    } else {
        // every block to the first shard's device, each on its own stream; the first shard's stream waits for all of them
        for (u32 s = 0; s < G; s++) {
            HIP_CHECK(hipSetDevice(g->devices[s]));
            if (g->devices[s] == g->devices[0])
                HIP_CHECK(hipMemcpyAsync(g->buf[0].recv + (size_t)s * K * width, g->buf[s].send, block, hipMemcpyDeviceToDevice, g->shard[s]->stream));
            else
                HIP_CHECK(hipMemcpyPeerAsync(g->buf[0].recv + (size_t)s * K * width, g->devices[0], g->buf[s].send, g->devices[s], block,
                                             g->shard[s]->stream));
            if (s > 0) HIP_CHECK(hipEventRecord(g->done[s], g->shard[s]->stream));
        }
        HIP_CHECK(hipSetDevice(g->devices[0]));
        for (u32 s = 1; s < G; s++) HIP_CHECK(hipStreamWaitEvent(h0->stream, g->done[s], 0));
    }
    HIP_CHECK(hipSetDevice(g->devices[0]));
    grow(g->packed, g->packed_cap, (size_t)K * D * 8);
    std::vector<u32> first(g->first_doc.begin(), g->first_doc.end());
    HIP_CHECK(hipMemcpyAsync(g->d_first, first.data(), first.size() * 4, hipMemcpyHostToDevice, h0->stream));
    hipLaunchKernelGGL(group_pack_kernel, dim3(ceil_div_u32((u64)K * D, BLOCK)), dim3(BLOCK), 0, h0->stream,
                       (const double *)g->buf[0].recv, (const u32 *)g->d_first, G, K, D, width, g->packed);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(h0->stream));         // (also covers `first`)
    if (mode == 1)                                        // the other shards' part of the collective
        for (u32 s = 1; s < G; s++) { HIP_CHECK(hipSetDevice(g->devices[s])); HIP_CHECK(hipStreamSynchronize(g->shard[s]->stream)); }
}

extern "C" {

int east_hip_group_create(const int32_t *devices, int32_t n_shards, east_hip_group_t *out)
{
    if (out) *out = nullptr;
    return guarded([&] {
        if (!out || !devices || n_shards < 1 || n_shards > 1024) east_throw(EAST_HIP_ERR_INVALID, "null argument or no shards");
        east_hip_group *g = new east_hip_group();
        try {
            for (int32_t s = 0; s < n_shards; s++) {
                east_hip_handle_t h = nullptr;
                const int rc = east_hip_create(devices[s], 0, &h);
                if (rc != 0) east_throw(rc, g_last_error);
                g->shard.push_back(h);
                g->devices.push_back(devices[s]);
            }
            g->buf.resize((size_t)n_shards);
            g->done.assign((size_t)n_shards, nullptr);
            for (int32_t s = 0; s < n_shards; s++) {
                use_device_ordinal(devices[s]);
                HIP_CHECK(hipEventCreateWithFlags(&g->done[s], hipEventDisableTiming));
            }
            use_device_ordinal(devices[0]);
            void *p = nullptr;
            HIP_CHECK(hipMalloc(&p, ((size_t)n_shards + 1) * 4));
            g->d_first = (u32 *)p;
        } catch (...) {
            group_destroy(g);
            throw;
        }
        *out = g;
    });
}

void east_hip_group_destroy(east_hip_group_t g) { group_destroy(g); }

int east_hip_group_shards(east_hip_group_t g, int32_t *first_doc)
{
    if (!g) return EAST_HIP_ERR_INVALID;
    if (first_doc)
        for (size_t s = 0; s < g->first_doc.size(); s++) first_doc[s] = g->first_doc[s];
    return (int)g->shard.size();
}

east_hip_handle_t east_hip_group_handle(east_hip_group_t g, int32_t shard)
{
    return g && shard >= 0 && (size_t)shard < g->shard.size() ? g->shard[(size_t)shard] : nullptr;
}

int east_hip_debug_shard_documents(const int64_t *sizes, int32_t n_docs, int32_t n_shards, int32_t *first_doc)
{
    if (!sizes || !first_doc || n_docs < 0 || n_shards < 1) return EAST_HIP_ERR_INVALID;
    std::vector<int32_t> first;
    shard_by_size(sizes, n_docs, n_shards, first);
    for (size_t s = 0; s < first.size(); s++) first_doc[s] = first[s];
    return EAST_HIP_OK;
}

int east_hip_group_build(east_hip_group_t g, const uint32_t *symbols, int64_t n_total, const int64_t *doc_offsets,
                         const int32_t *n_strings, int32_t n_docs, int32_t encoding)
{
    return guarded([&] {
        group_check(g, false);
        use_device_ordinal(g->devices[0]);               // (remembers the caller's device: guarded() puts it back)
        if (!symbols) east_throw(EAST_HIP_ERR_INVALID, "null symbols");
        check_build_args(n_total, doc_offsets, n_strings, n_docs);
        g->built = false;
        std::vector<i64> sizes((size_t)n_docs);
        for (int32_t d = 0; d < n_docs; d++) sizes[d] = doc_offsets[d + 1] - doc_offsets[d];
        shard_by_size(sizes.data(), n_docs, (int32_t)g->shard.size(), g->first_doc);
        g->n_docs = n_docs;
        const auto t0 = std::chrono::steady_clock::now();
        on_every_shard(g, [&](int s) {
            const int32_t b = g->first_doc[s], e = g->first_doc[s + 1];
            east_hip_index *h = g->shard[s];
            h->built = false;
            if (e == b) return;
            std::vector<i64> off((size_t)(e - b) + 1);
            for (int32_t d = b; d <= e; d++) off[d - b] = doc_offsets[d] - doc_offsets[b];
            build_common(h, symbols + doc_offsets[b], true, off[e - b], off.data(), n_strings + b, e - b,
                         encoding == EAST_HIP_SYMBOLS_TAGGED);
        });
        g->build_ms = wall_ms_since(t0);
        g->built = true;
    });
}

int east_hip_group_build_texts_v(east_hip_group_t g, const uint8_t *const *texts, const int64_t *lengths, int32_t n_docs,
                                 const uint8_t *cp_class, const uint32_t *cp_upper, const uint32_t *word_hi,
                                 const uint32_t *digit_hi, const uint32_t *hi_upper_from, const uint32_t *hi_upper_to,
                                 int32_t n_hi_upper)
{
    return guarded([&] {
        group_check(g, false);
        use_device_ordinal(g->devices[0]);
        if (!texts || !lengths || n_docs < 1) east_throw(EAST_HIP_ERR_INVALID, "null argument or no documents");
        for (int32_t d = 0; d < n_docs; d++)
            if (lengths[d] < 0) east_throw(EAST_HIP_ERR_INVALID, "negative text length");
        g->built = false;
        shard_by_size(lengths, n_docs, (int32_t)g->shard.size(), g->first_doc);
        g->n_docs = n_docs;
        const auto t0 = std::chrono::steady_clock::now();
        on_every_shard(g, [&](int s) {
            const int32_t b = g->first_doc[s], e = g->first_doc[s + 1];
            east_hip_index *h = g->shard[s];
            h->built = false;
            if (e == b) return;
            std::vector<i64> off((size_t)(e - b) + 1, 0);
            for (int32_t d = b; d < e; d++) off[d - b + 1] = off[d - b] + lengths[d] + 1;      // + the separator
            build_from_texts(h, nullptr, off[e - b], off.data(), e - b, cp_class, cp_upper, word_hi, digit_hi, hi_upper_from,
                             hi_upper_to, n_hi_upper, texts + b);
        });
        g->build_ms = wall_ms_since(t0);
        g->built = true;
    });
}

// One call for the whole table: every shard scores its documents (east_hip_set_keyphrases + east_hip_score_resident on a
// thread of its own), the K x D_local blocks are assembled by one all-gather (see the head of this file) and the K x D
// table goes to the host once, from the first shard's device.
int east_hip_score_table_multi(east_hip_group_t g, const uint32_t *q_symbols, const int64_t *q_offsets, int32_t n_keyphrases,
                               int normalized, double *out)
{
    return guarded([&] {
        group_check(g, true);
        use_device_ordinal(g->devices[0]);
        if (!out) east_throw(EAST_HIP_ERR_INVALID, "null output table");
        if (n_keyphrases < 1 || !q_symbols || !q_offsets) east_throw(EAST_HIP_ERR_INVALID, "no keyphrases");
        const auto t0 = std::chrono::steady_clock::now();
        on_every_shard(g, [&](int s) {
            east_hip_index *h = g->shard[s];
            if (g->first_doc[s + 1] == g->first_doc[s]) return;
            set_keyphrases(h, q_symbols, q_offsets, n_keyphrases);
            score_resident(h, normalized);
            HIP_CHECK(hipStreamSynchronize(h->stream));
            HIP_CHECK(hipEventElapsedTime(&h->last_score_ms, h->ev0, h->ev1));
        });
        g->score_ms = wall_ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        group_allgather(g, (u32)n_keyphrases);
        HIP_CHECK(hipSetDevice(g->devices[0]));
        HIP_CHECK(hipMemcpy(out, g->packed, (size_t)n_keyphrases * g->n_docs * 8, hipMemcpyDeviceToHost));
        g->gather_ms = wall_ms_since(t1);
    });
}

// [0] build wall ms, [1] score wall ms (slowest shard), [2] all-gather + pack + copy to the host wall ms, [3] how the blocks
// were assembled (1: RCCL all-gather, 2: copies to the first shard's device, -1: no score call yet), [4] shards
int east_hip_group_info(east_hip_group_t g, double *out, int32_t cap)
{
    if (!g || !out) return EAST_HIP_ERR_INVALID;
    const double v[5] = {g->build_ms, g->score_ms, g->gather_ms, (double)g->gather_mode, (double)g->shard.size()};
    for (int i = 0; i < 5 && i < cap; i++) out[i] = v[i];
    return 5;
}

}  // extern "C"
