/* write_rate.c -- how fast can one process put N bytes of ready-made output into a regular file?
 * Decides how the encode stage hands device-framed BGZF blocks to the file system:
 *   a  one thread, write() of 64 MB pieces              (what chunk_write does today, minus writev's iovecs)
 *   b  T threads, pwrite() of disjoint ranges           (serialised by the inode lock on most file systems?)
 *   c  ftruncate + mmap(MAP_SHARED) + T threads memcpy  (page faults run in parallel)
 *   d  as c with MAP_POPULATE on the mapping            (faults taken up front by one thread)
 *   e  fallocate + T threads pwrite                      (pages exist already)
 * usage: write_rate <dir> <GB> <threads>
 * gcc -O2 -o write_rate write_rate.c -lpthread */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statfs.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { int fd; const uint8_t *src; uint8_t *map; size_t lo, hi; int mode; } job;

static void *work(void *arg) {
	job *j = (job *)arg;
	const size_t piece = (size_t)16 << 20;
	size_t p;
	for (p = j->lo; p < j->hi; p += piece) {
		const size_t n = j->hi - p < piece ? j->hi - p : piece;
		if (j->mode == 0) {
			size_t done = 0;
			while (done < n) { ssize_t k = pwrite(j->fd, j->src + p + done, n - done, (off_t)(p + done)); if (k <= 0) { perror("pwrite"); exit(1); } done += (size_t)k; }
		} else {
			memcpy(j->map + p, j->src + p, n);
		}
	}
	return NULL;
}

static void run_threads(int T, int fd, const uint8_t *src, uint8_t *map, size_t n, int mode) {
	pthread_t th[64];
	job jb[64];
	int t;
	for (t = 0; t < T; t++) {
		jb[t].fd = fd; jb[t].src = src; jb[t].map = map; jb[t].mode = mode;
		jb[t].lo = n / (size_t)T * (size_t)t; jb[t].hi = t == T - 1 ? n : n / (size_t)T * (size_t)(t + 1);
		pthread_create(&th[t], NULL, work, &jb[t]);
	}
	for (t = 0; t < T; t++) pthread_join(th[t], NULL);
}

int main(int argc, char **argv) {
	const char *dir = argc > 1 ? argv[1] : "/tmp";
	const size_t n = (size_t)((argc > 2 ? atof(argv[2]) : 2.0) * (double)(1 << 30));
	const int T = argc > 3 ? atoi(argv[3]) : 8;
	char path[512];
	struct statfs sf;
	uint8_t *src = (uint8_t *)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	size_t i;
	int v;
	if (src == MAP_FAILED) { perror("mmap"); return 1; }
	madvise(src, n, MADV_HUGEPAGE);
	for (i = 0; i < n; i += 4096) src[i] = (uint8_t)(i >> 12);
	if (statfs(dir, &sf) == 0) printf("dir %s f_type 0x%lx\n", dir, (unsigned long)sf.f_type);
	snprintf(path, sizeof path, "%s/write_rate.%d", dir, (int)getpid());
	for (v = 0; v < 5; v++) {
		int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0644);
		double t0, t1;
		if (fd < 0) { perror("open"); return 1; }
		t0 = now();
		if (v == 0) {
			size_t p = 0;
			while (p < n) { ssize_t k = write(fd, src + p, n - p < ((size_t)64 << 20) ? n - p : ((size_t)64 << 20)); if (k <= 0) { perror("write"); return 1; } p += (size_t)k; }
		} else if (v == 1) {
			run_threads(T, fd, src, NULL, n, 0);
		} else if (v == 2 || v == 3) {
			uint8_t *m;
			if (ftruncate(fd, (off_t)n) != 0) { perror("ftruncate"); return 1; }
			m = (uint8_t *)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED | (v == 3 ? MAP_POPULATE : 0), fd, 0);
			if (m == MAP_FAILED) { perror("mmap file"); return 1; }
			run_threads(T, fd, src, m, n, 1);
			munmap(m, n);
		} else {
			if (fallocate(fd, 0, 0, (off_t)n) != 0) perror("fallocate");
			run_threads(T, fd, src, NULL, n, 0);
		}
		t1 = now();
		close(fd);
		printf("%c  %-46s %6.3f s  %6.2f GB/s\n", 'a' + v,
		       v == 0 ? "one thread write()" : v == 1 ? "T threads pwrite()" : v == 2 ? "ftruncate + mmap + T threads memcpy" :
		       v == 3 ? "same, MAP_POPULATE" : "fallocate + T threads pwrite()", t1 - t0, (double)n / (t1 - t0) / 1e9);
		unlink(path);
	}
	return 0;
}
