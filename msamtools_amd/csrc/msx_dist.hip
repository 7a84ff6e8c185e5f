// msx_dist.hip -- one process per GPU: the exchange step of `profile` over RCCL (xGMI).
//
// Records shard by QNAME pool across ranks; filter / best-hit / insert counting need no
// communication.  The profile has one real exchange (msam_profile.c:65-243 accumulate
// per-reference counts over ALL inserts, :331-389 iterate over ALL multi-mappers): the
// per-reference count vector `ui` (u32, exact under any summation order) and the three
// counters are summed once, and per proportional-sharing iteration the vector `share`
// (f64[n_features]: sum over this rank's multi-mappers of w/S) is summed; every rank then
// applies the same update to the same numbers and takes the same convergence decision, so
// no rank ever waits for the host.  All collectives are enqueued on the context's stream.
//
// librccl is opened at run time (dlopen), only when a distributed context is initialised:
// the single-GPU command line neither links nor loads it.
#include "msx_internal.h"
#include "msx_count.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <ctime>

// ---- the part of the RCCL (NCCL 2) API used here, bound by dlsym ----------------------------
typedef struct { char internal[128]; } msx_nccl_id;        // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *msx_nccl_comm;
enum { MSX_NCCL_INT32 = 2, MSX_NCCL_UINT32 = 3, MSX_NCCL_INT64 = 4, MSX_NCCL_FLOAT64 = 8 };   // ncclDataType_t
enum { MSX_NCCL_SUM = 0, MSX_NCCL_MAX = 2 };                                                  // ncclRedOp_t

struct msx_rccl {
	void *so = nullptr;
	int (*GetUniqueId)(msx_nccl_id *) = nullptr;
	int (*CommInitRank)(msx_nccl_comm *, int, msx_nccl_id, int) = nullptr;
	int (*CommDestroy)(msx_nccl_comm) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, msx_nccl_comm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
};

static msx_rccl g_rccl;

static int rccl_load(msx_ctx *ctx) {
	if (g_rccl.so) return MSX_OK;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *so = nullptr;
	for (const char *nm : names)
		if ((so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
	if (!so) return msx_fail(ctx, MSX_ERR_DIST, "cannot load librccl: %s", dlerror());
#define BIND(field, sym)                                                                   \
	if (!(*(void **)(&g_rccl.field) = dlsym(so, sym))) {                                   \
		dlclose(so);                                                                       \
		return msx_fail(ctx, MSX_ERR_DIST, "librccl has no symbol %s", sym);               \
	}
	BIND(GetUniqueId, "ncclGetUniqueId")
	BIND(CommInitRank, "ncclCommInitRank")
	BIND(CommDestroy, "ncclCommDestroy")
	BIND(AllReduce, "ncclAllReduce")
	BIND(GroupStart, "ncclGroupStart")
	BIND(GroupEnd, "ncclGroupEnd")
	BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
	g_rccl.so = so;
	return MSX_OK;
}

#define MSX_NCCL(ctx, call)                                                                        \
	do {                                                                                           \
		int r_ = (call);                                                                           \
		if (r_ != 0)                                                                               \
			return msx_fail((ctx), MSX_ERR_DIST, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
	} while (0)

struct msx_dist {
	int rank = 0, world = 1;
	msx_nccl_comm comm = nullptr;
	void *scratch = nullptr;        // device, 64 bytes: scalars of barrier / reductions of host values
	hipStream_t side = nullptr;     // MSX_DIST_SLICES: a slice's all-reduce travels here while the next slice is computed
	hipEvent_t ev_ready = nullptr, ev_done = nullptr;
};

// ---- rendezvous: rank 0 hands the 128-byte id to the others over TCP -------------------------
// MASTER_ADDR / MSX_DIST_PORT (default MASTER_PORT + 17: MASTER_PORT itself belongs to the launcher's
// store).  Plain blocking sockets; the connect side retries while rank 0 is not listening yet.
static int send_all(int fd, const void *buf, size_t n) {
	const char *p = (const char *)buf;
	while (n) {
		ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
		if (k <= 0) { if (k < 0 && errno == EINTR) continue; return -1; }
		p += k; n -= (size_t)k;
	}
	return 0;
}
static int recv_all(int fd, void *buf, size_t n) {
	char *p = (char *)buf;
	while (n) {
		ssize_t k = recv(fd, p, n, 0);
		if (k <= 0) { if (k < 0 && errno == EINTR) continue; return -1; }
		p += k; n -= (size_t)k;
	}
	return 0;
}

extern "C" int msx_dist_rendezvous(const char *addr, int port, int rank, int world, void *payload, size_t bytes,
                                   int timeout_s) {
	if (!payload || world < 1 || rank < 0 || rank >= world) return MSX_ERR_ARG;
	if (world == 1) return MSX_OK;
	if (!addr || !*addr) addr = "127.0.0.1";
	char portstr[16];
	snprintf(portstr, sizeof portstr, "%d", port);
	if (rank == 0) {
		int ls = socket(AF_INET, SOCK_STREAM, 0);
		if (ls < 0) return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: socket: %s", strerror(errno));
		int one = 1;
		setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
		sockaddr_in sa;
		memset(&sa, 0, sizeof sa);
		sa.sin_family = AF_INET;
		sa.sin_port = htons((uint16_t)port);
		// listen on the interface the ranks were told to connect to (MASTER_ADDR), not on every one
		sa.sin_addr.s_addr = htonl(INADDR_ANY);
		{
			addrinfo h2, *r2 = nullptr;
			memset(&h2, 0, sizeof h2);
			h2.ai_family = AF_INET;
			h2.ai_socktype = SOCK_STREAM;
			if (getaddrinfo(addr, nullptr, &h2, &r2) == 0 && r2) {
				sa.sin_addr = ((sockaddr_in *)r2->ai_addr)->sin_addr;
				freeaddrinfo(r2);
			}
		}
		if (bind(ls, (sockaddr *)&sa, sizeof sa) != 0) {       // (an address of another host's view of this one: any interface)
			sa.sin_addr.s_addr = htonl(INADDR_ANY);
			if (bind(ls, (sockaddr *)&sa, sizeof sa) != 0) {
				int e = errno;
				close(ls);
				return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: cannot bind port %d: %s", port, strerror(e));
			}
		}
		if (listen(ls, world + 8) != 0) {
			int e = errno;
			close(ls);
			return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: cannot listen on port %d: %s", port, strerror(e));
		}
		const int limit = timeout_s > 0 ? timeout_s : 300;
		timeval tv = {limit, 0};
		setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
		std::vector<char> seen((size_t)world, 0);
		const time_t t_end = time(nullptr) + limit;
		for (int got = 1; got < world;) {
			int fd = accept(ls, nullptr, nullptr);
			if (fd < 0) {
				int e = errno;
				if (e == EINTR) continue;
				close(ls);
				return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: %d of %d ranks connected: %s", got, world, strerror(e));
			}
			// a connection that is not a rank's (a port scanner, a stale client) must neither block the hand-over
			// nor end it: 5 s for the hello, then on to the next connection
			timeval tc = {5, 0};
			setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tc, sizeof tc);
			setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tc, sizeof tc);
			int32_t hello[2] = {0, -1};
			const bool ok = recv_all(fd, hello, 8) == 0 && hello[0] == 0x4d535831 /* "MSX1" */ && hello[1] > 0 &&
			                hello[1] < world && !seen[(size_t)hello[1]] && send_all(fd, payload, bytes) == 0;
			close(fd);
			if (ok) { seen[(size_t)hello[1]] = 1; got++; }
			else if (time(nullptr) > t_end) {
				close(ls);
				return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: %d of %d ranks connected in %d s", got, world, limit);
			}
		}
		close(ls);
		return MSX_OK;
	}
	addrinfo hints, *res = nullptr;
	memset(&hints, 0, sizeof hints);
	hints.ai_family = AF_INET;
	hints.ai_socktype = SOCK_STREAM;
	if (getaddrinfo(addr, portstr, &hints, &res) != 0 || !res)
		return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: cannot resolve %s", addr);
	const int tries = (timeout_s > 0 ? timeout_s : 300) * 10;
	int rc = -1;
	for (int t = 0; t < tries && rc != 0; t++) {
		int fd = socket(AF_INET, SOCK_STREAM, 0);
		if (fd < 0) break;
		if (connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
			int32_t hello[2] = {0x4d535831, rank};
			timeval tc = {30, 0};
			setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tc, sizeof tc);
			rc = (send_all(fd, hello, 8) || recv_all(fd, payload, bytes)) ? -2 : 0;
			close(fd);
			if (rc == -2) { rc = -1; usleep(200000); }      // (rank 0 dropped us -- a duplicate, or it was busy: try again)
		} else {
			close(fd);
			usleep(100000);
		}
	}
	freeaddrinfo(res);
	if (rc != 0) return msx_fail(nullptr, MSX_ERR_DIST, "rendezvous: rank %d could not reach %s:%d", rank, addr, port);
	return MSX_OK;
}

// ---- communicator ----------------------------------------------------------------------------
extern "C" int msx_dist_unique_id(uint8_t id[MSX_DIST_ID_BYTES]) {
	if (!id) return MSX_ERR_ARG;
	int rc = rccl_load(nullptr);
	if (rc) return rc;
	msx_nccl_id u;
	MSX_NCCL(nullptr, g_rccl.GetUniqueId(&u));
	memcpy(id, &u, MSX_DIST_ID_BYTES);
	return MSX_OK;
}

extern "C" int msx_dist_init(msx_ctx *ctx, const uint8_t id[MSX_DIST_ID_BYTES], int rank, int world) {
	if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return MSX_ERR_ARG;
	if (ctx->dist) return msx_fail(ctx, MSX_ERR_ARG, "msx_dist_init: this context already has a communicator");
	int rc = rccl_load(ctx);
	if (rc) return rc;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	msx_dist *d = new msx_dist();
	d->rank = rank;
	d->world = world;
	msx_nccl_id u;
	memcpy(&u, id, MSX_DIST_ID_BYTES);
	int r = g_rccl.CommInitRank(&d->comm, world, u, rank);
	if (r != 0) {
		delete d;
		return msx_fail(ctx, MSX_ERR_DIST, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
		                g_rccl.GetErrorString(r));
	}
	if (hipMalloc(&d->scratch, 64) != hipSuccess) {
		g_rccl.CommDestroy(d->comm);
		delete d;
		return msx_fail(ctx, MSX_ERR_NOMEM, "msx_dist_init: scratch allocation failed");
	}
	ctx->dist = d;
	return MSX_OK;
}

extern "C" int msx_dist_init_env(msx_ctx *ctx) {
	if (!ctx) return MSX_ERR_ARG;
	const char *ws = getenv("WORLD_SIZE"), *rk = getenv("RANK");
	const int world = ws ? atoi(ws) : 1, rank = rk ? atoi(rk) : 0;
	if (world < 1 || rank < 0 || rank >= world)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_dist_init_env: RANK=%s WORLD_SIZE=%s", rk ? rk : "", ws ? ws : "");
	uint8_t id[MSX_DIST_ID_BYTES];
	memset(id, 0, sizeof id);
	int rc;
	if (rank == 0 && (rc = msx_dist_unique_id(id))) return rc;
	const char *addr = getenv("MASTER_ADDR"), *mp = getenv("MASTER_PORT"), *dp = getenv("MSX_DIST_PORT");
	const int port = dp ? atoi(dp) : (mp ? atoi(mp) : 29500) + 17;
	if ((rc = msx_dist_rendezvous(addr, port, rank, world, id, sizeof id, 300))) {
		ctx->err = msx_tls_err;
		return rc;
	}
	return msx_dist_init(ctx, id, rank, world);
}

extern "C" void msx_dist_finalize(msx_ctx *ctx) {
	if (!ctx || !ctx->dist) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	if (ctx->dist->comm) g_rccl.CommDestroy(ctx->dist->comm);
	if (ctx->dist->side) { (void)hipStreamSynchronize(ctx->dist->side); (void)hipStreamDestroy(ctx->dist->side); }
	if (ctx->dist->ev_ready) (void)hipEventDestroy(ctx->dist->ev_ready);
	if (ctx->dist->ev_done) (void)hipEventDestroy(ctx->dist->ev_done);
	if (ctx->dist->scratch) (void)hipFree(ctx->dist->scratch);
	delete ctx->dist;
	ctx->dist = nullptr;
}

extern "C" int msx_dist_rank(const msx_ctx *ctx) { return (ctx && ctx->dist) ? ctx->dist->rank : 0; }
extern "C" int msx_dist_world(const msx_ctx *ctx) { return (ctx && ctx->dist) ? ctx->dist->world : 1; }

// ---- small collectives on host values (bench harness: barrier, max of the elapsed time, totals) ----
static int reduce_host(msx_ctx *ctx, void *val, size_t bytes, int dtype, int op) {
	msx_dist *d = ctx->dist;
	if (!d) return MSX_OK;          // one rank: the value is its own reduction
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	MSX_HIP(ctx, hipMemcpyAsync(d->scratch, val, bytes, hipMemcpyHostToDevice, ctx->stream));
	MSX_NCCL(ctx, g_rccl.AllReduce(d->scratch, d->scratch, 1, dtype, op, d->comm, ctx->stream));
	MSX_HIP(ctx, hipMemcpyAsync(val, d->scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}

extern "C" int msx_dist_barrier(msx_ctx *ctx) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	int32_t one = 1;
	if (!ctx->dist) { MSX_HIP(ctx, hipStreamSynchronize(ctx->stream)); return MSX_OK; }
	return reduce_host(ctx, &one, 4, MSX_NCCL_INT32, MSX_NCCL_SUM);
}

extern "C" int msx_dist_max_f64(msx_ctx *ctx, double *value) {
	if (!ctx || !value) return MSX_ERR_ARG;
	msx_join(ctx);
	return reduce_host(ctx, value, 8, MSX_NCCL_FLOAT64, MSX_NCCL_MAX);
}

extern "C" int msx_dist_sum_i64(msx_ctx *ctx, int64_t *value) {
	if (!ctx || !value) return MSX_ERR_ARG;
	msx_join(ctx);
	return reduce_host(ctx, value, 8, MSX_NCCL_INT64, MSX_NCCL_SUM);
}

// ---- the profile's exchange ---------------------------------------------------------------------
// sum over ranks of ui (and d for --multi=equal) and of {inserts, uniq, multi}: msam_profile.c:65-243
// counts every insert of the file, whichever rank read it.  Integer sums are order-free: bit-exact.
extern "C" int msx_profile_allreduce_counts(msx_ctx *ctx, msx_profile *p) {
	if (!ctx || !p) return MSX_ERR_ARG;
	msx_join(ctx);
	{ int frc = msx_profile_fold_equal(ctx, p); if (frc) return frc; }        // (--multi equal: d[] complete before it travels)
	msx_dist *d = ctx->dist;
	if (!d) return MSX_OK;          // (a one-rank communicator still runs the collective: the same code path)
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	MSX_NCCL(ctx, g_rccl.GroupStart());
	int r1 = g_rccl.AllReduce(p->ui, p->ui, (size_t)p->n_features, MSX_NCCL_UINT32, MSX_NCCL_SUM, d->comm, ctx->stream);
	int r2 = g_rccl.AllReduce(p->counters, p->counters, 3, MSX_NCCL_UINT32, MSX_NCCL_SUM, d->comm, ctx->stream);
	int r3 = (p->d && p->share_type == MSX_MULTI_SHARE_EQUAL)
	             ? g_rccl.AllReduce(p->d, p->d, (size_t)p->n_features, MSX_NCCL_FLOAT64, MSX_NCCL_SUM, d->comm, ctx->stream)
	             : 0;
	MSX_NCCL(ctx, g_rccl.GroupEnd());
	if (r1 || r2 || r3)
		return msx_fail(ctx, MSX_ERR_DIST, "ncclAllReduce(counts) failed: %s", g_rccl.GetErrorString(r1 ? r1 : r2 ? r2 : r3));
	return MSX_OK;
}

int msx_dist_allreduce_share(msx_ctx *ctx, msx_profile *p) {
	msx_dist *d = ctx->dist;
	if (!d) return MSX_OK;          // (a one-rank communicator still runs the collective: the same code path)
	MSX_NCCL(ctx, g_rccl.AllReduce(p->share, p->share, (size_t)p->n_features, MSX_NCCL_FLOAT64, MSX_NCCL_SUM, d->comm,
	                                ctx->stream));
	return MSX_OK;
}

int msx_dist_allreduce_share_side(msx_ctx *ctx, msx_profile *p, int32_t first, int32_t count) {
	msx_dist *d = ctx->dist;
	if (!d || count <= 0) return MSX_OK;
	if (!d->side) {
		MSX_HIP(ctx, hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
		MSX_HIP(ctx, hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming));
		MSX_HIP(ctx, hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
	}
	MSX_HIP(ctx, hipEventRecord(d->ev_ready, ctx->stream));
	MSX_HIP(ctx, hipStreamWaitEvent(d->side, d->ev_ready, 0));
	MSX_NCCL(ctx, g_rccl.AllReduce(p->share + first, p->share + first, (size_t)count, MSX_NCCL_FLOAT64, MSX_NCCL_SUM, d->comm, d->side));
	return MSX_OK;
}
int msx_dist_side_join(msx_ctx *ctx) {
	msx_dist *d = ctx->dist;
	if (!d || !d->side) return MSX_OK;
	MSX_HIP(ctx, hipEventRecord(d->ev_done, d->side));
	MSX_HIP(ctx, hipStreamWaitEvent(ctx->stream, d->ev_done, 0));
	return MSX_OK;
}

int msx_dist_allreduce_u32(msx_ctx *ctx, uint32_t *dev, size_t count) {
	msx_dist *d = ctx->dist;
	if (!d) return MSX_OK;          // (a one-rank communicator still runs the collective: the same code path)
	MSX_NCCL(ctx, g_rccl.AllReduce(dev, dev, count, MSX_NCCL_UINT32, MSX_NCCL_SUM, d->comm, ctx->stream));
	return MSX_OK;
}
