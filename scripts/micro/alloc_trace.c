/* LD_PRELOAD shim (diagnostics): malloc / realloc / mmap calls of MSX_ALLOC_TRACE_MIN bytes and more (default 100 MB) with their
 * call stacks on stderr -- which code asks for the large mappings a command holds when it ends.
 * gcc -O1 -shared -fPIC -o /tmp/alloc_trace.so scripts/micro/alloc_trace.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
static __thread int inside;
static size_t min_bytes(void) { static size_t m; if (!m) { const char *e = getenv("MSX_ALLOC_TRACE_MIN"); m = e ? (size_t)atoll(e) : (size_t)100 << 20; } return m; }
static void report(const char *what, size_t n) {
	void *bt[24];
	char line[96];
	int k, len;
	if (inside) return;
	inside = 1;
	len = snprintf(line, sizeof line, "## %s %zu bytes\n", what, n);
	(void)!write(2, line, (size_t)len);
	k = backtrace(bt, 24);
	backtrace_symbols_fd(bt, k, 2);
	inside = 0;
}
void *malloc(size_t n) {
	static void *(*real)(size_t);
	if (!real) real = (void *(*)(size_t))dlsym(RTLD_NEXT, "malloc");
	if (n >= min_bytes()) report("malloc", n);
	return real(n);
}
void *realloc(void *p, size_t n) {
	static void *(*real)(void *, size_t);
	if (!real) real = (void *(*)(void *, size_t))dlsym(RTLD_NEXT, "realloc");
	if (n >= min_bytes()) report("realloc", n);
	return real(p, n);
}
void *mmap(void *a, size_t n, int prot, int flags, int fd, off_t off) {
	static void *(*real)(void *, size_t, int, int, int, off_t);
	if (!real) real = (void *(*)(void *, size_t, int, int, int, off_t))dlsym(RTLD_NEXT, "mmap");
	if (n >= min_bytes()) report("mmap", n);
	return real(a, n, prot, flags, fd, off);
}
