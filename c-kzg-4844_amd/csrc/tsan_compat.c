/* clang >= 15 instruments memory intrinsics as calls to __tsan_mem{cpy,move,set}; GCC 11's libtsan (the runtime the
 * ThreadSanitizer pass preloads, see Makefile) predates them and intercepts the libc functions instead.  Only linked
 * into libckzg_hip_tsan.so. */
#include <string.h>
void *__tsan_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
void *__tsan_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
void *__tsan_memset(void *d, int c, size_t n) { return memset(d, c, n); }
