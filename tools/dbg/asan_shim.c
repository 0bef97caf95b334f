/* clang-instrumented objects run against gcc's libasan (the ROCm compiler-rt's ASan runtime intercepts HSA allocations and
   needs the instrumented ROCm stack): the three helpers newer clang emits and libasan.so.6 lacks. */
#include <string.h>
void *__sanitizer_internal_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
void *__sanitizer_internal_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
void *__sanitizer_internal_memset(void *d, int c, size_t n) { return memset(d, c, n); }
