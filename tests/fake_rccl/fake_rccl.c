/* TEST-ONLY stand-in for librccl, loaded through AMX_RCCL_LIBRARY so that the multi-rank branches of amx_gather_outputs
 * (allophant_amd/csrc/amx_dist.hip; contract include/allophant_amx.h "amx_gather_outputs") execute without multi-GPU
 * hardware.  It implements the five entry points that file binds -- ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv /
 * ncclGetErrorString -- in two modes:
 *
 *   record     (fake_rccl_set_mode(0), CPU tests): nothing moves; every call is appended to a log the test reads back
 *              (fake_rccl_count / fake_rccl_get), so peers, counts, dtype codes, buffer offsets and the group bracketing of
 *              any (rank, world, root) can be checked on a machine without a GPU.  Pointers are never dereferenced.
 *   transport  (fake_rccl_set_mode(1), GPU test): the communicator is a fake_comm made by fake_comm_create(rank, world, dir);
 *              ncclGroupEnd waits for the stream, then performs the group's sends (device -> host -> a file in `dir`, renamed
 *              into place) and its receives (poll for the file, host -> device).  Two processes on ONE GPU can so play two
 *              ranks; the bytes take the route  rank r's HBM -> host file -> root's HBM.  HIP is resolved with dlopen at
 *              the first transport call: this file builds with plain gcc and needs no GPU in record mode.
 *
 * Not a product file and not an RCCL re-implementation: no rings, no channels, no streams of its own. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define FAKE_MAX_CALLS 4096
#define FAKE_MAGIC 0x46414b45

typedef struct {
    int kind;          /* 0 group start, 1 group end, 2 send, 3 recv */
    uint64_t buffer;   /* pointer value */
    uint64_t count;
    int dtype, peer;
    uint64_t comm, stream;
    int group_depth;   /* nesting depth at the time of the call (sends / receives must see 1) */
} fake_call;

typedef struct {
    int magic, rank, world;
    char dir[512];
    unsigned long long send_seq[64], recv_seq[64];
} fake_comm;

static int g_mode = 0, g_depth = 0, g_n = 0;
static fake_call g_calls[FAKE_MAX_CALLS];
/* ops of the open group (transport mode) */
static fake_call g_pending[256];
static int g_npending = 0;

void fake_rccl_set_mode(int mode) { g_mode = mode; }
void fake_rccl_reset(void) { g_n = 0; g_depth = 0; g_npending = 0; }
int fake_rccl_count(void) { return g_n; }
int fake_rccl_get(int i, fake_call* out) {
    if (i < 0 || i >= g_n) return -1;
    *out = g_calls[i];
    return 0;
}
void* fake_comm_create(int rank, int world, const char* dir) {
    fake_comm* c = (fake_comm*)calloc(1, sizeof(fake_comm));
    c->magic = FAKE_MAGIC; c->rank = rank; c->world = world;
    strncpy(c->dir, dir, sizeof(c->dir) - 1);
    return c;
}

static void log_call(int kind, const void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
    if (g_n >= FAKE_MAX_CALLS) return;
    fake_call* c = &g_calls[g_n++];
    c->kind = kind; c->buffer = (uint64_t)(uintptr_t)buf; c->count = count; c->dtype = dtype; c->peer = peer;
    c->comm = (uint64_t)(uintptr_t)comm; c->stream = (uint64_t)(uintptr_t)stream; c->group_depth = g_depth;
}

/* ---- HIP, resolved lazily (transport mode only) ---- */
static int (*p_sync)(void*) = 0;
static int (*p_memcpy)(void*, const void*, size_t, int) = 0;
static int hip_load(void) {
    if (p_sync) return 0;
    /* by SONAME first: that returns the HIP runtime the process has already loaded (torch's bundled copy in a Python
     * process) -- a second copy would not know the caller's device pointers */
    void* h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return -1;
    p_sync = (int (*)(void*))dlsym(h, "hipStreamSynchronize");
    p_memcpy = (int (*)(void*, const void*, size_t, int))dlsym(h, "hipMemcpy");
    return p_sync && p_memcpy ? 0 : -1;
}
static size_t dtype_bytes(int dtype) { return dtype == 7 ? 4 : dtype == 4 ? 8 : 0; } /* ncclFloat32 = 7, ncclInt64 = 4 */

int ncclGroupStart(void) {
    log_call(0, 0, 0, -1, -1, 0, 0);
    ++g_depth;
    return 0;
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
    log_call(2, buf, count, dtype, peer, comm, stream);
    if (g_mode == 1) {
        if (g_depth < 1 || g_npending >= 256) return 5;
        g_pending[g_npending++] = g_calls[g_n - 1];
    }
    return 0;
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
    log_call(3, buf, count, dtype, peer, comm, stream);
    if (g_mode == 1) {
        if (g_depth < 1 || g_npending >= 256) return 5;
        g_pending[g_npending++] = g_calls[g_n - 1];
    }
    return 0;
}

static int transport_flush(void) {
    if (g_npending == 0) return 0;
    if (hip_load()) return 2;
    fake_comm* c = (fake_comm*)(uintptr_t)g_pending[0].comm;
    if (!c || c->magic != FAKE_MAGIC) return 4;
    /* stream order: everything enqueued before the group has finished */
    if (p_sync((void*)(uintptr_t)g_pending[0].stream)) return 1;
    for (int pass = 0; pass < 2; ++pass) {          /* sends first: a rank may receive from itself */
        for (int i = 0; i < g_npending; ++i) {
            fake_call* op = &g_pending[i];
            const size_t bytes = op->count * dtype_bytes(op->dtype);
            if (!bytes || op->peer < 0 || op->peer >= c->world || op->peer >= 64) return 4;
            char path[768], tmp[800];
            if (pass == 0 && op->kind == 2) {
                snprintf(path, sizeof(path), "%s/m_%d_%d_%llu", c->dir, c->rank, op->peer, c->send_seq[op->peer]++);
                snprintf(tmp, sizeof(tmp), "%s.tmp", path);
                void* host = malloc(bytes);
                if (p_memcpy(host, (const void*)(uintptr_t)op->buffer, bytes, 2 /* D2H */)) { free(host); return 1; }
                FILE* f = fopen(tmp, "wb");
                if (!f || fwrite(host, 1, bytes, f) != bytes) { free(host); return 2; }
                fclose(f);
                free(host);
                if (rename(tmp, path)) return 2;
            } else if (pass == 1 && op->kind == 3) {
                snprintf(path, sizeof(path), "%s/m_%d_%d_%llu", c->dir, op->peer, c->rank, c->recv_seq[op->peer]++);
                FILE* f = 0;
                for (int tries = 0; tries < 600000 && !(f = fopen(path, "rb")); ++tries) usleep(1000);  /* <= 10 min */
                if (!f) return 2;
                void* host = malloc(bytes);
                const size_t got = fread(host, 1, bytes, f);
                fclose(f);
                if (got != bytes) { free(host); return 2; }
                if (p_memcpy((void*)(uintptr_t)op->buffer, host, bytes, 1 /* H2D */)) { free(host); return 1; }
                free(host);
                unlink(path);
            }
        }
    }
    g_npending = 0;
    return 0;
}

int ncclGroupEnd(void) {
    --g_depth;
    log_call(1, 0, 0, -1, -1, 0, 0);
    if (g_mode == 1 && g_depth == 0) return transport_flush();
    return 0;
}

const char* ncclGetErrorString(int code) {
    switch (code) {
        case 0: return "fake rccl: success";
        case 1: return "fake rccl: HIP call failed";
        case 2: return "fake rccl: transport (file / dlopen) failure";
        case 4: return "fake rccl: invalid argument";
        default: return "fake rccl: invalid usage";
    }
}
