/* Test infrastructure (tests/test_multi_stub.py): a stand-in for the HIP runtime entry points libtrayhip.so's HOST code calls, preloaded
 * (LD_PRELOAD) in a container without a GPU so that the multi-device code path of the library -- tray_multi_create /
 * tray_render_frame_multi / tray_multi_update_frame: one host thread and stream per device, the grouped ncclReduce, the restore of the
 * caller's current device -- executes somewhere before the first 8-GPU run. "Devices" are host memory, streams and events are dummy
 * handles, kernel launches are recorded and do nothing (the films stay zero: this checks the plumbing, not pixels). Every call that
 * matters for the plumbing is appended to the log file named by FAKEHIP_LOG. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef struct { uint32_t x, y, z; } dim3;
static __thread int t_device = 0;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static int n_devices(void) { const char* e = getenv("FAKEHIP_DEVICES"); return e ? atoi(e) : 2; }
static void logf_(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void logf_(const char* fmt, ...) {
    const char* path = getenv("FAKEHIP_LOG");
    if (!path) return;
    pthread_mutex_lock(&g_mu);
    FILE* f = fopen(path, "a");
    if (f) { va_list ap; va_start(ap, fmt); vfprintf(f, fmt, ap); va_end(ap); fputc('\n', f); fclose(f); }
    pthread_mutex_unlock(&g_mu);
}
int fakehip_current_device(void) { return t_device; }   /* for the RCCL stand-in */

hipError_t hipGetDeviceCount(int* n) { *n = n_devices(); return 0; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= n_devices()) return 101; t_device = d; return 0; }
hipError_t hipGetDevice(int* d) { *d = t_device; return 0; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
hipError_t hipFree(void* p) { free(p); return 0; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) { (void)flags; *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
hipError_t hipHostFree(void* p) { free(p); return 0; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) { (void)kind; memmove(d, s, n); return 0; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) { (void)kind; (void)st; memmove(d, s, n); return 0; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { (void)st; memset(d, v, n); return 0; }
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = (size_t)1 << 36; *tot = (size_t)1 << 37; return 0; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = malloc(16); *(int*)*s = t_device; logf_("stream_create dev=%d stream=%p", t_device, *s); return 0; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { (void)f; return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return 0; }
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return 0; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned f) { (void)s; (void)e; (void)f; return 0; }
hipError_t hipDeviceSynchronize(void) { return 0; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(8); return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned f) { (void)f; return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)e; (void)s; return 0; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return 0; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { (void)a; (void)b; *ms = 1.0f; return 0; }
hipError_t hipGetLastError(void) { return 0; }
hipError_t hipPeekAtLastError(void) { return 0; }
const char* hipGetErrorString(hipError_t e) { return e ? "fakehip error" : "no error"; }
hipError_t hipFuncSetAttribute(const void* f, int a, int v) { (void)f; (void)a; (void)v; return 0; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void* f, int bs, size_t sh) { (void)f; (void)bs; (void)sh; *n = 2; return 0; }
/* hipDeviceProp_t is large and versioned: the library reads multiProcessorCount only when the call succeeds, so it fails here */
hipError_t hipGetDevicePropertiesR0600(void* prop, int dev) { (void)prop; (void)dev; return 1; }
hipError_t hipGetDeviceProperties(void* prop, int dev) { (void)prop; (void)dev; return 1; }
/* kernel launches: hipcc's host stubs push the configuration, then call hipLaunchKernel after popping it */
static __thread struct { dim3 g, b; size_t sh; hipStream_t st; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { t_cfg.g = g; t_cfg.b = b; t_cfg.sh = sh; t_cfg.st = st; return 0; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) { *g = t_cfg.g; *b = t_cfg.b; *sh = t_cfg.sh; *st = t_cfg.st; return 0; }
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t sh, hipStream_t st) {
    (void)f; (void)sh;
    /* FAKEHIP_TILE_KERNEL=1: the launch is k_path_tiles(scene, tiles, tile_count, chunk, chunk_stride, spp, kf, slice_shift, rgbw, counter, stats)
     * (kernels.hip): log its shard arguments and leave a mark in the film -- word 0 += device + 1 -- so that the sum-reduce has something to sum */
    if (getenv("FAKEHIP_TILE_KERNEL")) {
        const uint32_t tile_count = *(uint32_t*)args[2], chunk = *(uint32_t*)args[3], chunk_stride = *(uint32_t*)args[4], spp = *(uint32_t*)args[5];
        float* film = *(float**)args[8];
        film[0] += (float)(t_device + 1);
        logf_("launch dev=%d grid=%u block=%u stream=%p tile_count=%u chunk=%u chunk_stride=%u spp=%u film=%p", t_device, g.x, b.x, st, tile_count, chunk, chunk_stride, spp, (void*)film);
    } else logf_("launch dev=%d grid=%u block=%u stream=%p", t_device, g.x, b.x, st);
    return 0;
}
