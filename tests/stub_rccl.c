/* A stand-in for librccl.so, for the gpu tier on a ONE-GPU box (tests/test_gpu_multi.py).
 *
 * RCCL refuses a communicator that names one device twice, so csrc/multi.h's grouped all-gather --
 * ncclGroupStart, one ncclAllGather per shard with hipSetDevice in front of each, ncclGroupEnd -- had only ever run
 * with a group of ONE.  This library implements the six entry points multi.h binds (ncclCommInitAll, ncclAllGather,
 * ncclGroupStart, ncclGroupEnd, ncclCommDestroy, ncclGetErrorString) with device-to-device copies on the callers' own
 * streams, takes any device list, and writes every call it gets to the file EAST_STUB_RCCL_LOG names (with the device
 * that was current at the call), so that a test can check call order, counts, data type and device switching for
 * G = 2, 3, 8 logical shards (EAST_STUB_RCCL_FAIL_AT=<i> makes the i-th all-gather of a group fail).  Selected through EAST_HIP_RCCL_LIB=<this library> + EAST_HIP_GROUP_GATHER=rccl.
 * Test infrastructure only: the product never loads it by itself.
 *
 * Build: gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stub_rccl.c -o <out>.so -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define STUB_MAX 64
typedef struct StubWorld StubWorld;
typedef struct { int rank, device; StubWorld *world; } StubComm;
struct StubWorld { int n, alive; StubComm comm[STUB_MAX]; };
typedef struct { const void *send; void *recv; size_t count; int dtype, device_at_call; StubComm *comm; hipStream_t stream; } StubCall;

static int g_depth = 0, g_calls = 0;
static StubCall g_call[STUB_MAX];

static void stub_log(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void stub_log(const char *fmt, ...)
{
    const char *path = getenv("EAST_STUB_RCCL_LOG");
    if (!path || !*path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fputc('\n', f);
    fclose(f);
}

static size_t stub_size(int dtype) { return dtype == 8 ? 8 : dtype == 7 ? 4 : dtype == 2 || dtype == 3 ? 4 : dtype == 4 || dtype == 5 ? 8 : 1; }

/* every rank's block to every rank: recv_r[s] <- send_s, on rank r's stream, behind an event on rank s's stream */
static int stub_run(void)
{
    if (g_calls == 0) return 0;
    StubWorld *w = g_call[0].comm->world;
    int seen[STUB_MAX] = {0}, cur = 0;
    if (g_calls != w->n) return 5;                      /* ncclInvalidUsage: one call per rank */
    for (int i = 0; i < g_calls; i++) {
        if (g_call[i].comm->world != w || g_call[i].count != g_call[0].count || g_call[i].dtype != g_call[0].dtype) return 5;
        if (seen[g_call[i].comm->rank]++) return 5;
    }
    if (hipGetDevice(&cur) != hipSuccess) return 1;
    hipEvent_t ready[STUB_MAX];
    const size_t bytes = g_call[0].count * stub_size(g_call[0].dtype);
    for (int i = 0; i < g_calls; i++) {
        if (hipSetDevice(g_call[i].comm->device) != hipSuccess) return 1;
        if (hipEventCreateWithFlags(&ready[i], hipEventDisableTiming) != hipSuccess) return 1;
        if (hipEventRecord(ready[i], g_call[i].stream) != hipSuccess) return 1;
    }
    for (int r = 0; r < g_calls; r++) {
        if (hipSetDevice(g_call[r].comm->device) != hipSuccess) return 1;
        for (int s = 0; s < g_calls; s++) {
            if (hipStreamWaitEvent(g_call[r].stream, ready[s], 0) != hipSuccess) return 1;
            if (hipMemcpyAsync((char *)g_call[r].recv + (size_t)g_call[s].comm->rank * bytes, g_call[s].send, bytes, hipMemcpyDefault,
                               g_call[r].stream) != hipSuccess) return 1;
        }
    }
    for (int i = 0; i < g_calls; i++) (void)hipEventDestroy(ready[i]);     /* (released once the waits have passed) */
    (void)hipSetDevice(cur);
    return 0;
}

int ncclCommInitAll(void **comms, int n, const int *devices)
{
    if (!comms || n < 1 || n > STUB_MAX) return 4;      /* ncclInvalidArgument */
    StubWorld *w = (StubWorld *)calloc(1, sizeof(StubWorld));
    if (!w) return 2;
    w->n = w->alive = n;
    char list[STUB_MAX * 4 + 8] = "";
    for (int r = 0; r < n; r++) {
        w->comm[r].rank = r;
        w->comm[r].device = devices ? devices[r] : r;
        w->comm[r].world = w;
        comms[r] = &w->comm[r];
        snprintf(list + strlen(list), sizeof(list) - strlen(list), "%s%d", r ? "," : "", w->comm[r].device);
    }
    stub_log("init n=%d devices=%s", n, list);
    return 0;
}

int ncclCommDestroy(void *comm)
{
    StubComm *c = (StubComm *)comm;
    if (!c) return 4;
    stub_log("destroy rank=%d", c->rank);
    if (--c->world->alive == 0) free(c->world);
    return 0;
}

int ncclGroupStart(void)
{
    if (g_depth++ == 0) g_calls = 0;
    stub_log("group_start");
    return 0;
}

int ncclGroupEnd(void)
{
    if (g_depth < 1) return 5;
    int rc = 0;
    if (--g_depth == 0) { rc = stub_run(); stub_log("group_end calls=%d rc=%d", g_calls, rc); g_calls = 0; }
    return rc;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream)
{
    StubComm *c = (StubComm *)comm;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (!c || !send || !recv) return 4;
    stub_log("allgather rank=%d count=%zu dtype=%d device_at_call=%d comm_device=%d grouped=%d", c->rank, count, dtype, cur, c->device, g_depth > 0);
    if (g_calls >= STUB_MAX) return 5;
    {   /* EAST_STUB_RCCL_FAIL_AT=<i>: the i-th all-gather of a group fails (the caller's error path) */
        const char *fail = getenv("EAST_STUB_RCCL_FAIL_AT");
        if (fail && *fail && atoi(fail) == g_calls) return 3;   /* ncclInternalError */
    }
    g_call[g_calls].send = send; g_call[g_calls].recv = recv; g_call[g_calls].count = count; g_call[g_calls].dtype = dtype;
    g_call[g_calls].device_at_call = cur; g_call[g_calls].comm = c; g_call[g_calls].stream = stream;
    g_calls++;
    if (g_depth == 0) { const int rc = stub_run(); g_calls = 0; return rc; }
    return 0;
}

const char *ncclGetErrorString(int rc)
{
    return rc == 0 ? "no error" : rc == 1 ? "stub: a HIP call failed" : rc == 4 ? "stub: invalid argument" : rc == 5 ? "stub: invalid usage" : "stub: error";
}
