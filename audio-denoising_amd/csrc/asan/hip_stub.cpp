// Stand-in for the HIP runtime, for the host-side sanitizer build only (make asan): "device" memory is
// host memory from malloc, copies and memsets are memcpy / memset (so AddressSanitizer sees every
// byte range the library's host code hands to the runtime), kernel launches do nothing.  Never
// linked into libadn_hip.so and never run on a GPU box.
//
// Knobs (environment): ADN_STUB_MALLOC_LIMIT = bytes; a hipMalloc above it fails with
// hipErrorOutOfMemory (drives the library's ADN_ERR_NOMEM path).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

static long g_launches = 0;
static int g_device = 0;
static dim3 g_grid, g_block;
static size_t g_shmem = 0;
static hipStream_t g_stream = nullptr;

extern "C" {

// what the host objects' module constructors reference (there is no device code in this build)
__attribute__((visibility("default"))) const char __hip_fatbin[16] = {0};

void **__hipRegisterFatBinary(const void *) {
    static void *handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned int, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_grid = grid;
    g_block = block;
    g_shmem = shmem;
    g_stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
    *grid = g_grid;
    *block = g_block;
    *shmem = g_shmem;
    *stream = g_stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t shmem, hipStream_t) {
    // the launch geometry the host code computed must be one a GPU would accept
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 ||
        block.x * block.y * block.z > 1024 || shmem > 160 * 1024 || grid.y > 65535 || grid.z > 65535)
        abort();
    ++g_launches;
    return hipSuccess;
}
long adn_stub_launches(void) { return g_launches; }

hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = g_device;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) {
    if (d != 0) return hipErrorInvalidDevice;
    g_device = d;
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int d) {
    if (d != 0) return hipErrorInvalidDevice;
    memset(p, 0, sizeof(*p));
    const char *arch = getenv("ADN_STUB_ARCH");
    strncpy(p->gcnArchName, arch ? arch : "gfx950:sramecc+:xnack-", sizeof(p->gcnArchName) - 1);
    p->multiProcessorCount = 256;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipErrorOutOfMemory ? "out of memory (stub)" : "error (stub)"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t bytes) {
    const char *lim = getenv("ADN_STUB_MALLOC_LIMIT");
    if (lim && bytes > (size_t)atoll(lim)) return hipErrorOutOfMemory;
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) {
    memcpy(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) {
    memset(p, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int value) {
    return value > 160 * 1024 ? hipErrorInvalidValue : hipSuccess;
}
hipError_t hipGetSymbolAddress(void **p, const void *) {
    static void *scratch = nullptr;
    if (!scratch) scratch = malloc(64u << 20);
    *p = scratch;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) {
    *e = (hipEvent_t)malloc(8);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    free((void *)e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}
}
