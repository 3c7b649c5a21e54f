/*
 * wr_ring.hip -- the halo ring of BASELINE config 5 behind the C ABI (include/webradio_amd.h,
 * wr_ring_*): ONE stream cut in time, chunk c on rank c mod world, every chunk's halo (the last
 * H frames of the chunk before it) from the ring neighbour.
 *
 * The reference has no counterpart -- its pipeline is one thread that walks the front ends in
 * turn (radio.cxx:56-59), and a LowPass carries its 63-frame history from block to block in a
 * member (dsp/lowpass.cxx:133-142).  Cutting the stream over GPUs turns that member into a
 * message: one ncclSend / ncclRecv pair per chunk inside one group, rank r -> r + 1, over a
 * single xGMI link (a 2 MB halo is latency-class: ~13 us at 153 GB/s).  No other collective.
 *
 * The exchange runs on the ring's OWN stream, so it can be issued a round ahead of the chunk that
 * needs it (the halo is input, not a result): wr_ring_exchange makes that stream wait for what
 * the device's stream has enqueued so far, posts the pair and returns; wr_ring_wait makes the
 * device's stream wait for the pair -- event to event, the host never blocks.
 *
 * RCCL is looked up at run time (dlopen of librccl.so.1: the copy the process already has, e.g.
 * PyTorch's, else ROCm's), so the library and every path that does not shard in time load and run
 * without it.
 */
#include "wr_internal.h"

#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <rccl/rccl.h>

namespace {

struct Rccl {
	void *h = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclGetVersion) GetVersion = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_lock;

template <typename F> bool sym(void *h, const char *name, F &f)
{
	f = reinterpret_cast<F>(dlsym(h, name));
	return f != nullptr;
}

/* 0 or a WR_ERR_* code (message set) */
int rccl_load()
{
	std::lock_guard<std::mutex> g(g_rccl_lock);
	if (g_rccl.h)
		return WR_OK;
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!h)
		h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!h)
		return wrc_fail(WR_ERR_NODEV, "wr_ring: RCCL (librccl.so.1) cannot be loaded: %s", dlerror());
	Rccl r;
	r.h = h;
	if (!sym(h, "ncclGetUniqueId", r.GetUniqueId) || !sym(h, "ncclCommInitRank", r.CommInitRank) ||
	    !sym(h, "ncclCommDestroy", r.CommDestroy) || !sym(h, "ncclSend", r.Send) || !sym(h, "ncclRecv", r.Recv) ||
	    !sym(h, "ncclGroupStart", r.GroupStart) || !sym(h, "ncclGroupEnd", r.GroupEnd) ||
	    !sym(h, "ncclGetErrorString", r.GetErrorString) || !sym(h, "ncclGetVersion", r.GetVersion))
		return wrc_fail(WR_ERR_NODEV, "wr_ring: librccl.so.1 lacks an entry point: %s", dlerror());
	g_rccl = r;
	return WR_OK;
}

} // namespace

struct wr_ring {
	wr_dev *dev;
	int rank, world;
	ncclComm_t comm;
	hipStream_t side;          /* the exchanges' own stream */
	hipEvent_t ready;          /* device stream -> side: the data to send are there */
	hipEvent_t done;           /* side -> device stream: the pair has completed */
	bool pending;              /* an exchange wr_ring_wait has not been called for */
	unsigned long long exchanges;
};

/* `ready` only keeps the exchange from running ahead of the device's stream (the data it guards were written long before): a
 * device-scope release is all it needs.  With the default
 * system-scope fence every record on the device's stream cost the chunk's launch 11 us (r03: 452 -> 606 Gsps at world 1) */
#define WR_RING_EVENT_FLAGS (hipEventDisableTiming | hipEventReleaseToDevice)

#define RCCL_TRY(expr)                                                                                   \
	do {                                                                                                 \
		ncclResult_t r_ = (expr);                                                                        \
		if (r_ != ncclSuccess)                                                                           \
			return wrc_fail(WR_ERR_HIP, "%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
	} while (0)
#define HIP_TRY_R(expr)                                                                                  \
	do {                                                                                                 \
		hipError_t e_ = (expr);                                                                          \
		if (e_ != hipSuccess)                                                                            \
			return wrc_fail(WR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

extern "C" int wr_ring_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int wr_ring_version(int *version)
{
	if (!version)
		return wrc_fail(WR_ERR_ARG, "wr_ring_version: NULL");
	if (int rc = rccl_load())
		return rc;
	RCCL_TRY(g_rccl.GetVersion(version));
	return WR_OK;
}

extern "C" int wr_ring_make_id(void *id, size_t nbytes)
{
	if (!id || nbytes != sizeof(ncclUniqueId))
		return wrc_fail(WR_ERR_ARG, "wr_ring_make_id: id must hold wr_ring_id_bytes() = %zu bytes", sizeof(ncclUniqueId));
	if (int rc = rccl_load())
		return rc;
	ncclUniqueId u;
	RCCL_TRY(g_rccl.GetUniqueId(&u));
	memcpy(id, &u, sizeof(u));
	return WR_OK;
}

extern "C" int wr_ring_create(wr_ring **out, wr_dev *dev, const void *id, size_t nbytes, int rank, int world)
{
	if (!out || !dev || !id || nbytes != sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world)
		return wrc_fail(WR_ERR_ARG, "wr_ring_create: bad argument (rank %d of %d, id %zu bytes)", rank, world, nbytes);
	*out = nullptr;
	if (int rc = rccl_load())
		return rc;
	HIP_TRY_R(hipSetDevice(wrc_dev_index(dev)));
	wr_ring *r = new (std::nothrow) wr_ring();
	if (!r)
		return wrc_fail(WR_ERR_NOMEM, "wr_ring_create: out of memory");
	r->dev = dev;
	r->rank = rank;
	r->world = world;
	r->pending = false;
	r->exchanges = 0;
	ncclUniqueId u;
	memcpy(&u, id, sizeof(u));
	ncclResult_t nr = g_rccl.CommInitRank(&r->comm, world, u, rank);
	if (nr != ncclSuccess) {
		delete r;
		return wrc_fail(WR_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(nr));
	}
	hipError_t e = hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&r->ready, WR_RING_EVENT_FLAGS);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&r->done, hipEventDisableTiming);   /* publishes the received halo: the default fence (it is
		                                                                    recorded on the ring's own stream) */
	if (e != hipSuccess) {
		g_rccl.CommDestroy(r->comm);
		delete r;
		return wrc_fail(WR_ERR_HIP, "wr_ring_create: %s", hipGetErrorString(e));
	}
	*out = r;
	return WR_OK;
}

static int ring_exchange(wr_ring *r, wr_tuner *after, const float *send_dev, float *recv_dev, size_t nfloats)
{
	if (!r || !send_dev || !recv_dev || !nfloats)
		return wrc_fail(WR_ERR_ARG, "wr_ring_exchange: bad argument");
	if (send_dev == recv_dev)
		return wrc_fail(WR_ERR_ARG, "wr_ring_exchange: send and receive buffers must differ");
	HIP_TRY_R(hipSetDevice(wrc_dev_index(r->dev)));
	hipStream_t main = wrc_dev_stream(r->dev);
	hipEvent_t launch = nullptr;
	if (after)
		if (int rc = wrc_tuner_launch_mark(after, &launch))
			return rc;                                      /* (not marking its launches: WR_ERR_STATE) */
	if (launch) {
		/* behind the tuner's last launch: its own completion signal, nothing put on the device's stream (an event
		 * record there sits between two launches and costs the chunk 11 us: 452 -> 606 Gsps at world 1, r03) */
		HIP_TRY_R(hipStreamWaitEvent(r->side, launch, 0));
	} else if (!after) {
		/* what the device's stream has been given so far (the kernel or copy that produced `send_dev`,
		 * the last reader of `recv_dev`) comes first; nothing enqueued later is waited for */
		HIP_TRY_R(hipEventRecord(r->ready, main));
		HIP_TRY_R(hipStreamWaitEvent(r->side, r->ready, 0));
	}	/* (a tuner that has launched nothing yet: nothing to wait for) */
	const int next = (r->rank + 1) % r->world, prev = (r->rank + r->world - 1) % r->world;
	RCCL_TRY(g_rccl.GroupStart());
	ncclResult_t s = g_rccl.Send(send_dev, nfloats, ncclFloat, next, r->comm, r->side);
	ncclResult_t v = g_rccl.Recv(recv_dev, nfloats, ncclFloat, prev, r->comm, r->side);
	RCCL_TRY(g_rccl.GroupEnd());
	RCCL_TRY(s);
	RCCL_TRY(v);
	HIP_TRY_R(hipEventRecord(r->done, r->side));
	r->pending = true;
	++r->exchanges;
	return WR_OK;
}

extern "C" int wr_ring_exchange(wr_ring *r, const float *send_dev, float *recv_dev, size_t nfloats)
{
	return ring_exchange(r, nullptr, send_dev, recv_dev, nfloats);
}

extern "C" int wr_ring_exchange_after(wr_ring *r, wr_tuner *tuner, const float *send_dev, float *recv_dev, size_t nfloats)
{
	if (!tuner)
		return wrc_fail(WR_ERR_ARG, "wr_ring_exchange_after: tuner is NULL");
	return ring_exchange(r, tuner, send_dev, recv_dev, nfloats);
}

extern "C" int wr_ring_wait(wr_ring *r)
{
	if (!r)
		return wrc_fail(WR_ERR_ARG, "wr_ring_wait: NULL");
	if (!r->pending)
		return WR_OK;
	HIP_TRY_R(hipSetDevice(wrc_dev_index(r->dev)));
	/* posted a round ahead, the pair has normally completed by now: then there is nothing to wait for, and a wait that
	 * is enqueued all the same is one more packet between two launches */
	const hipError_t q = hipEventQuery(r->done);
	if (q == hipSuccess) {
		r->pending = false;
		return WR_OK;
	}
	if (q != hipErrorNotReady)
		return wrc_fail(WR_ERR_HIP, "wr_ring_wait: %s", hipGetErrorString(q));
	(void)hipGetLastError();
	HIP_TRY_R(hipStreamWaitEvent(wrc_dev_stream(r->dev), r->done, 0));
	r->pending = false;
	return WR_OK;
}

extern "C" int wr_ring_info(wr_ring *r, int *rank, int *world, unsigned long long *exchanges)
{
	if (!r)
		return wrc_fail(WR_ERR_ARG, "wr_ring_info: NULL");
	if (rank)
		*rank = r->rank;
	if (world)
		*world = r->world;
	if (exchanges)
		*exchanges = r->exchanges;
	return WR_OK;
}

extern "C" int wr_ring_destroy(wr_ring *r)
{
	if (!r)
		return WR_OK;
	(void)hipSetDevice(wrc_dev_index(r->dev));
	(void)hipStreamSynchronize(r->side);
	if (g_rccl.CommDestroy)
		(void)g_rccl.CommDestroy(r->comm);
	(void)hipEventDestroy(r->ready);
	(void)hipEventDestroy(r->done);
	(void)hipStreamDestroy(r->side);
	delete r;
	return WR_OK;
}
