// k_runtime.hip -- the runtime part of the thin extern-"C" shim (lsk.h): device memory, copies, streams, events.
#include "lsk_dev.hpp"

// ---------------------------------------------------------------------------------------------
// runtime shim
// ---------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";


extern "C" char const *lsk_last_error(void) { return g_err; }
extern "C" char *lsk_error_buffer(size_t *capacity) { *capacity = sizeof(g_err); return g_err; } // the other translation units report through it
extern "C" int lsk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int lsk_set_device(int device) { LSK_CHECK(hipSetDevice(device)); return 0; }
extern "C" int lsk_malloc(void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // an allocation failure is recoverable: do not leave it for the next launch check
        snprintf(g_err, sizeof(g_err), "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        *p = nullptr;
        return -1;
    }
    return 0;
}
extern "C" int lsk_free(void *p) { if (p) LSK_CHECK(hipFree(p)); return 0; }
extern "C" int lsk_mem_info(size_t *free_bytes, size_t *total_bytes) { LSK_CHECK(hipMemGetInfo(free_bytes, total_bytes)); return 0; }
extern "C" int lsk_h2d(void *dst, void const *src, size_t bytes) {
    if (bytes) LSK_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int lsk_d2h(void *dst, void const *src, size_t bytes) {
    if (bytes) LSK_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int lsk_d2d_async(void *dst, void const *src, size_t bytes, void *stream) {
    if (bytes) LSK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int lsk_memset_async(void *p, int value, size_t bytes, void *stream) {
    if (bytes) LSK_CHECK(hipMemsetAsync(p, value, bytes, (hipStream_t)stream));
    return 0;
}
extern "C" int lsk_sync(void *stream) { LSK_CHECK(hipStreamSynchronize((hipStream_t)stream)); return 0; }
extern "C" int lsk_device_sync(void) { LSK_CHECK(hipDeviceSynchronize()); return 0; }

extern "C" int lsk_event_create(void **ev) { hipEvent_t e; LSK_CHECK(hipEventCreate(&e)); *ev = (void *)e; return 0; }
extern "C" int lsk_event_destroy(void *ev) { if (ev) LSK_CHECK(hipEventDestroy((hipEvent_t)ev)); return 0; }
extern "C" int lsk_event_record(void *ev, void *stream) { LSK_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
extern "C" int lsk_event_elapsed_ms(void *start, void *stop, float *ms) {
    LSK_CHECK(hipEventSynchronize((hipEvent_t)stop));
    LSK_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

