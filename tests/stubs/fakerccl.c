/* Test infrastructure (tests/test_multi_stub.py): the six RCCL entry points libtrayhip.so resolves with dlopen("librccl.so"), over host
 * memory: ncclReduce(sum, float) really adds the send buffers into the root's receive buffer at ncclGroupEnd, every call is logged
 * (FAKEHIP_LOG) with the current "device" of the fake HIP runtime, and FAKERCCL_FAIL=reduce makes ncclReduce fail (error path). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int rank, n, dev; } Comm;
static struct { const float* send; float* recv; size_t count; int root, rank; } g_ops[64];
static int g_n_ops = 0, g_in_group = 0;
static void log_(const char* msg) {
    const char* path = getenv("FAKEHIP_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (f) { fputs(msg, f); fputc('\n', f); fclose(f); }
}
static int current_device(void) { int (*fn)(void) = (int (*)(void))dlsym(RTLD_DEFAULT, "fakehip_current_device"); return fn ? fn() : -1; }
int ncclCommInitAll(void** comms, int n, const int* devs) {
    char b[128];
    for (int i = 0; i < n; ++i) { Comm* c = malloc(sizeof *c); c->rank = i; c->n = n; c->dev = devs[i]; comms[i] = c; }
    snprintf(b, sizeof b, "nccl_comm_init_all n=%d", n); log_(b);
    return 0;
}
int ncclCommDestroy(void* c) { free(c); log_("nccl_comm_destroy"); return 0; }
int ncclGroupStart(void) { g_in_group = 1; g_n_ops = 0; log_("nccl_group_start"); return 0; }
int ncclReduce(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, void* stream) {
    const Comm* c = comm;
    char b[256];
    snprintf(b, sizeof b, "nccl_reduce rank=%d comm_dev=%d current_dev=%d count=%zu dtype=%d op=%d root=%d stream=%p in_group=%d", c->rank, c->dev, current_device(), count, dtype, op, root, stream, g_in_group);
    log_(b);
    const char* fail = getenv("FAKERCCL_FAIL");
    if (fail && strcmp(fail, "reduce") == 0) return 1;
    if (g_n_ops < 64) { g_ops[g_n_ops].send = send; g_ops[g_n_ops].recv = recv; g_ops[g_n_ops].count = count; g_ops[g_n_ops].root = root; g_ops[g_n_ops].rank = c->rank; ++g_n_ops; }
    return 0;
}
int ncclGroupEnd(void) {
    /* sum of all ranks' send buffers into the root's receive buffer (in place on the root, as the library calls it) */
    float* dst = NULL; size_t count = 0; int root = -1;
    for (int i = 0; i < g_n_ops; ++i) if (g_ops[i].rank == g_ops[i].root) { dst = g_ops[i].recv; count = g_ops[i].count; root = g_ops[i].root; }
    if (dst)
        for (int i = 0; i < g_n_ops; ++i)
            if (g_ops[i].rank != root) for (size_t k = 0; k < count; ++k) dst[k] += g_ops[i].send[k];
    g_in_group = 0;
    log_("nccl_group_end");
    return 0;
}
const char* ncclGetErrorString(int e) { return e ? "fakerccl: unhandled system error" : "no error"; }
