// A torch pluggable allocator for tools/probe_guard_alloc.py: every tensor gets its own hipMalloc with GUARD bytes of
// 0xFF (NaN in f32 / bf16 / f16) in front of it and behind it, and a zeroed body. A kernel that reads outside the tensors
// it was handed -- up to GUARD bytes either side -- reads NaN instead of a neighbour's plausible values.
#include <hip/hip_runtime.h>
#include <sys/types.h>
#include <cstdio>
#include <cstdlib>

static const size_t GUARD = 2u << 20;

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    hipSetDevice(device);
    size_t body = ((size_t)size + 255) & ~(size_t)255;
    char* p = nullptr;
    if (hipMalloc((void**)&p, body + 2 * GUARD) != hipSuccess) { fprintf(stderr, "guard_malloc: hipMalloc failed\n"); abort(); }
    hipMemset(p, 0xFF, GUARD);
    hipMemset(p + GUARD, 0, body);
    hipMemset(p + GUARD + body, 0xFF, GUARD);
    hipDeviceSynchronize();
    return p + GUARD;
}

extern "C" void guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)stream;
    hipSetDevice(device);
    hipDeviceSynchronize();
    hipFree((char*)ptr - GUARD);
}
