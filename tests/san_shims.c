/* tests/san_shims.c -- linked into the ThreadSanitizer build of the library only (Makefile `tsan`).
 * The ROCm clang instruments memcpy / memset / memmove as calls to __tsan_mem*; gcc 11's libtsan (the only TSan runtime in
 * the image) predates those entry points.  Forwarding to the libc functions keeps the checks: libtsan intercepts them. */
#include <string.h>
void *__tsan_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
void *__tsan_memset(void *d, int c, size_t n) { return memset(d, c, n); }
void *__tsan_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
