/*
 * gpubatch.cxx -- see gpubatch.h.  Everything numerical happens behind the C ABI; this
 * file only decides WHAT is submitted WHEN, so that DspBlock::run()'s depth-first walk
 * (dspblock.cxx:207-209 upstream) ends up as one launch sequence per tuner block.
 */
#include "gpubatch.h"

#include <stdint.h>

#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdio.h>

#include "debug.h"
#include "demodulator.h"
#include "downconverter.h"
#include "filetuner.h"
#include "lowpass.h"

namespace wrhost {

namespace {
std::mutex g_devLock;
std::vector<wr_dev *> g_devs;
int g_batches = 0;

unsigned int envUnsigned(const char *name, unsigned int fallback)
{
	const char *v = getenv(name);
	return (v && *v) ? (unsigned int)strtoul(v, NULL, 0) : fallback;
}
}

namespace {
std::vector<TraceEvent> g_trace;
std::mutex g_traceLock;
bool traceOn()
{
	static const bool on = envUnsigned("WEBRADIO_TRACE", 0) != 0;
	return on;
}
void traceAdd(const void *src, char kind)
{
	if (!traceOn())
		return;
	std::lock_guard<std::mutex> g(g_traceLock);
	TraceEvent e = { src, kind };
	g_trace.push_back(e);
}
}

const std::vector<TraceEvent> &trace() { return g_trace; }
void traceClear()
{
	std::lock_guard<std::mutex> g(g_traceLock);
	g_trace.clear();
}

int deviceCount()
{
	int n = 0;
	if (wr_device_count(&n) != WR_OK)
		return 0;
	return n;
}

wr_dev *device(int index)
{
	std::lock_guard<std::mutex> g(g_devLock);
	if (index < 0)
		return NULL;
	if ((size_t)index >= g_devs.size())
		g_devs.resize(index + 1, NULL);
	if (!g_devs[index]) {
		wr_dev *d = NULL;
		if (wr_dev_open(&d, index, NULL) != WR_OK) {
			/* no CPU path: the caller's init()/process() reports failure */
			LOG_ERROR("GPU device %d unavailable: %s\n", index, wr_last_error());
			return NULL;
		}
		g_devs[index] = d;
	}
	return g_devs[index];
}

/* the device copy of a source's current block, and the host buffers it has been uploaded from:
 * sources hand out the same one or two vectors block after block (RtlSdrTuner swaps two,
 * rtlsdrtuner.cxx:280), so they are page-locked once and the copy is a DMA nobody waits for */
struct SourceStage {
	DevBuf buf;
	DevBuf buf2;                /* a float source's blocks alternate between `buf` and `buf2`: block b + 1 crosses PCIe on the
	                               upload stream while block b is being worked on (wr_dev_upload_ahead) */
	bool second;
	DevBuf raw;                 /* the block as the source holds it in the RTL-SDR byte format (RawU8Block), on the device */
	unsigned long epoch;
	const void *host;
	size_t floats;
	wr_dev *dev;
	struct Pinned { void *ptr; size_t bytes; };
	std::vector<Pinned> pinned;
	const float *cur;           /* the staged block: in `buf`, or `buf2` */
	bool rawStaged;             /* the block now staged came from the source's raw bytes: its float vector was not read */
	/* sparse staging (stagedWindows / stagedTail): of the source's block of `sparseEpoch` only the windows
	 * [k * sparsePeriod - (sparseLen - 1), k * sparsePeriod] and the last `sparseTail` frames lie in `buf` */
	unsigned long sparseEpoch;
	unsigned int sparsePeriod, sparseLen;
	size_t sparseTail;
	size_t tailWanted;          /* the longest tail a consumer has asked for so far (a SpectrumSink: its frame + hop) */
	SourceStage() : second(false), epoch(0), host(NULL), floats(0), dev(NULL), cur(NULL), rawStaged(false), sparseEpoch(0),
	                sparsePeriod(0), sparseLen(0), sparseTail(0), tailWanted(0) {}
	~SourceStage() { unpin(); }
	/* the page locks must go before the memory does: freed but still registered, it is handed out
	 * again by the allocator and a later copy out of it fails ("invalid argument") */
	void unpin() {
		if (dev)
			wr_dev_wait_uploads(dev);
		for (size_t n = 0; n < pinned.size(); n++)
			wr_dev_host_unregister(dev, pinned[n].ptr);
		pinned.clear();
		epoch = 0;
		host = NULL;
	}
	bool pin(wr_dev *d, const void *p, size_t bytes) {
		for (size_t n = 0; n < pinned.size(); n++)
			if (pinned[n].ptr == p && pinned[n].bytes >= bytes)
				return true;
		if (envUnsigned("WEBRADIO_NO_PINNING", 0))
			return false;
		/* Small blocks are not page-locked: the allocator serves them from the heap, where a registered range shares
		 * its first and last page with its neighbours and its address is soon handed out again -- and on this ROCm a
		 * registered range that overlaps memory the runtime page-locks on the fly (the destination of any copy into
		 * pageable memory) can abort the process (r04: profiles/r04_abort_hunt.txt).  Large vectors get a mapping of
		 * their own from the allocator; small ones are copied through the runtime's staging buffers, which at their
		 * size costs nothing that matters. */
		static const size_t minBytes = envUnsigned("WEBRADIO_PIN_MIN_BYTES", 1u << 20);
		if (bytes < minBytes)
			return false;
		/* ... and of the large ones only those that start where a mapping of their own starts: glibc hands out a
		 * request above its mmap threshold as a private mapping with the user pointer 16 bytes into its first page
		 * (an aligned_alloc / mmap / hipHostMalloc'ed buffer starts ON a page); once the threshold has grown -- it
		 * follows the largest mapped block freed so far, up to 32 MB -- even megabytes come from the heap, in the
		 * middle of a page somebody else also lives on.  WEBRADIO_PIN_ANY=1 lifts the check. */
		static const bool pinAny = envUnsigned("WEBRADIO_PIN_ANY", 0) != 0 || minBytes == 0;
		if (!pinAny && ((uintptr_t)p & 4095u) > 64u)
			return false;
		for (size_t n = 0; n < pinned.size(); n++)
			if (pinned[n].ptr == p) {                      /* same address, grown: register afresh */
				wr_dev_wait_uploads(d);                    /* (a copy out of it may still be in flight) */
				wr_dev_host_unregister(d, pinned[n].ptr);
				pinned.erase(pinned.begin() + n);
				break;
			}
		if (pinned.size() >= 8) {                          /* (RtlSdrTuner rotates five vectors: its ring of N_BUFFERS = 4,
		                                                      rtlsdrtuner.cxx:33-34, and the one handed out)
		                                                      a source that allocates per block: give up on the oldest */
			wr_dev_wait_uploads(d);
			wr_dev_host_unregister(d, pinned[0].ptr);
			pinned.erase(pinned.begin());
		}
		if (wr_dev_host_register(d, const_cast<void *>(p), bytes) != WR_OK)
			return false;
		Pinned e = { const_cast<void *>(p), bytes };
		pinned.push_back(e);
		dev = d;
		return true;
	}
};

static void releaseSource(DspSource *src)
{
	delete static_cast<SourceStage *>(src->gpuStage());
	src->setGpuStage(NULL);
	TunerBatch::destroyFor(src);
}

/* top of DspSource::stop(): the block vector is about to be released (dspblock.cxx:153-167) */
static void beforeSourceStop(DspSource *src)
{
	SourceStage *st = static_cast<SourceStage *>(src->gpuStage());
	if (st)
		st->unpin();
}

/* top of DspSource::run(): the source is about to refill its block vector */
static void beforeSourceRun(DspSource *src)
{
	SourceStage *st = static_cast<SourceStage *>(src->gpuStage());
	if (!st || !st->dev)
		return;
	/* a source that alternates between raw buffers (RawU8Block::rawU8Buffers) may refill one while the transfer out
	 * of the other is still in flight: the host then runs a block ahead of the GPU instead of in step with it */
	unsigned int inflight = 0;
	if (st->rawStaged) {
		const RawU8Block *raw = dynamic_cast<const RawU8Block *>(src);
		const unsigned int nbuf = raw ? raw->rawU8Buffers() : 1u;
		inflight = nbuf > 1u ? (nbuf - 1u > 3u ? 3u : nbuf - 1u) : 0u;
	}
	wr_dev_wait_uploads_but(st->dev, inflight);
}

wr_dev *deviceFor(const DspBlock *block)
{
	DspSource *src = TunerBatch::rootSource(block);
	if (!src)
		return device((int)envUnsigned("WEBRADIO_DEVICE", 0));
	if (src->gpuIndex() < 0) {
		int count = deviceCount();
		if (count <= 0) {
			LOG_ERROR("no GPU: %s\n", wr_last_error());
			return NULL;
		}
		std::lock_guard<std::mutex> g(g_devLock);
		int index = getenv("WEBRADIO_DEVICE") ? (int)envUnsigned("WEBRADIO_DEVICE", 0) : (g_batches % count);
		g_batches++;
		src->setGpuIndex(index);
		src->setGpuCleanup(releaseSource);
	}
	return device(src->gpuIndex());
}

static std::mutex g_stageLock;

/* a source whose block already lies on `dev` (DeviceBlock): where */
static const float *residentBlock(const DspSource *src, wr_dev *dev, size_t floats)
{
	const DeviceBlock *db = dynamic_cast<const DeviceBlock *>(src);
	if (!db)
		return NULL;
	wr_dev *bd = NULL;
	size_t frames = 0;
	const float *p = db->deviceBlock(&bd, &frames);
	return (p && bd == dev && frames * 2 == floats) ? p : NULL;
}

/* The source's current block staged in `pieces` equal parts, part `p` (0, 1, ... in order) now: part p crosses PCIe on the
 * upload stream -- as bytes where the source still holds them (RawU8Block), converted on the device -- while the caller has
 * the parts before it worked on.  Returns the device address of the part inside the staged block, which after the last
 * part is the same staged block stagedBlock() makes in one piece (the SpectrumSink then finds it there).  NULL: this block
 * cannot be staged in parts (host memory that cannot be page-locked, no memory): before part 0 nothing has happened and the
 * caller takes the one-piece path. */
static const float *stagePiece(DspSource *src, wr_dev *dev, const vector<sample_t> &host, unsigned int pieces, unsigned int p)
{
	std::lock_guard<std::mutex> g(g_stageLock);
	SourceStage *st = static_cast<SourceStage *>(src->gpuStage());
	if (!st) {
		st = new SourceStage();
		src->setGpuStage(st);
		src->setGpuCleanup(releaseSource);
		src->setGpuBeforeRun(beforeSourceRun);
		src->setGpuBeforeStop(beforeSourceStop);
	}
	const size_t bytes = host.size() * sizeof(float);
	const RawU8Block *rawsrc = dynamic_cast<const RawU8Block *>(src);
	size_t rawFrames = 0;
	const uint8_t *rawBytes = rawsrc ? rawsrc->rawU8(&rawFrames) : NULL;
	const bool raw = rawBytes && rawFrames * 2 == host.size() && !envUnsigned("WEBRADIO_NO_U8_STAGING", 0);
	if (p == 0) {
		if (st->epoch == src->epoch() && st->host == host.data() && st->floats == host.size())
			return NULL;                                /* somebody staged the whole block already */
		if (!raw && !src->hostBlockValid())
			return NULL;
		if (!(raw ? st->pin(dev, rawBytes, host.size()) : st->pin(dev, host.data(), bytes)))
			return NULL;
		st->dev = dev;
		if (!raw)
			st->second = !st->second;                   /* float blocks alternate between two device copies (see stagedBlock) */
		DevBuf &dst = (!raw && st->second) ? st->buf2 : st->buf;
		if (!dst.reserve(dev, bytes))
			return NULL;
		st->cur = (const float *)dst.ptr;
		st->rawStaged = raw;
		st->epoch = 0;                                  /* not complete until the last part is on its way */
	}
	const size_t part = host.size() / pieces, off = part * p;      /* floats */
	float *dstp = const_cast<float *>(st->cur) + off;
	const int rc = raw ? wr_u8_to_f32_from_host(dev, rawBytes + off, dstp, part)
	                   : wr_dev_upload_ahead(dev, dstp, host.data() + off, part * sizeof(float));
	if (rc != WR_OK) {
		LOG_ERROR("staging part %u of %u of the source block failed: %s\n", p, pieces, wr_last_error());
		return NULL;
	}
	if (p + 1 == pieces) {
		st->epoch = src->epoch();
		st->host = host.data();
		st->floats = host.size();
	}
	return dstp;
}

static bool sparseEnabled()
{
	static const bool on = envUnsigned("WEBRADIO_SPARSE", 1) != 0;
	return on;
}

/* Sparse staging of the source's current block (wr_stage_windows_from_host): of a block whose every receiver decimates by
 * `period` through a channel filter of `length` taps only the frames under the taps -- 64 of every 400 at BASELINE config 2 --
 * and the tail cross PCIe, read by a kernel on the device's own stream, in order with the tuner's launches behind it; there
 * is no transfer to wait for and no second stream to hand over from.  period = 0: the tail alone (a SpectrumSink's frame).
 * Returns the address the WHOLE block would have on the device (the frames not staged keep what the buffer held), or NULL when
 * this block cannot be staged that way (host memory that cannot be page-locked, windows too long for the kernel): the caller
 * stages the whole block. */
static const float *stagedSparse(const DspBlock *consumer, const vector<sample_t> &host, wr_dev **dev_out, unsigned int period,
                                 unsigned int length, size_t tail)
{
	DspSource *src = TunerBatch::rootSource(consumer);
	if (!sparseEnabled() || !src || host.empty() || host.data() != src->currentBlock().data())
		return NULL;
	wr_dev *dev = deviceFor(consumer);
	if (!dev)
		return NULL;
	if (const float *there = residentBlock(src, dev, host.size())) {
		if (dev_out)
			*dev_out = dev;
		return there;                                   /* nothing to stage: the source produced it in device memory */
	}
	std::lock_guard<std::mutex> g(g_stageLock);
	SourceStage *st = static_cast<SourceStage *>(src->gpuStage());
	if (!st) {
		st = new SourceStage();
		src->setGpuStage(st);
		src->setGpuCleanup(releaseSource);
		src->setGpuBeforeRun(beforeSourceRun);
		src->setGpuBeforeStop(beforeSourceStop);
	}
	const size_t nframes = host.size() / 2;
	if (tail > nframes)
		tail = nframes;
	if (!period && tail > st->tailWanted)
		st->tailWanted = tail;                          /* the next block's window pass brings it along */
	if (st->epoch == src->epoch() && st->host == host.data() && st->floats == host.size() && st->dev == dev) {
		if (dev_out)
			*dev_out = dev;
		return st->cur;                                 /* the whole block is there already */
	}
	const bool have = st->sparseEpoch == src->epoch() && st->host == host.data() && st->floats == host.size() && st->dev == dev;
	if (have && (!period || (st->sparsePeriod == period && st->sparseLen == length)) && st->sparseTail >= tail) {
		if (dev_out)
			*dev_out = dev;
		return st->cur;
	}
	if (have && period && st->sparsePeriod)
		return NULL;                                    /* two consumers with different windows: the whole block then */
	const RawU8Block *rawsrc = dynamic_cast<const RawU8Block *>(src);
	size_t rawFrames = 0;
	const uint8_t *rawBytes = rawsrc ? rawsrc->rawU8(&rawFrames) : NULL;
	const bool raw = rawBytes && rawFrames == nframes && !envUnsigned("WEBRADIO_NO_U8_STAGING", 0);
	if (!raw && !src->hostBlockValid())
		return NULL;
	const void *from = raw ? (const void *)rawBytes : (const void *)host.data();
	const size_t bytes = host.size() * sizeof(float);
	if ((uintptr_t)from & 15u)
		return NULL;
	if (!st->pin(dev, from, raw ? host.size() : bytes) || !st->buf.reserve(dev, bytes))
		return NULL;
	if (period && st->tailWanted > tail)
		tail = st->tailWanted > nframes ? nframes : st->tailWanted;
	if (wr_stage_windows_from_host(dev, from, raw ? 1 : 0, (float *)st->buf.ptr, nframes, period ? period : (unsigned int)nframes + 1u,
	                               period ? length : 1u, tail) != WR_OK) {
		LOG_ERROR("sparse staging of the source block failed: %s\n", wr_last_error());
		return NULL;
	}
	st->dev = dev;
	st->host = host.data();
	st->floats = host.size();
	st->cur = (const float *)st->buf.ptr;
	st->rawStaged = raw;
	st->epoch = 0;                                      /* (not the whole block) */
	if (!have || period) {
		st->sparsePeriod = period;
		st->sparseLen = period ? length : 0;
	}
	st->sparseTail = (have && st->sparseTail > tail) ? st->sparseTail : tail;
	st->sparseEpoch = src->epoch();
	if (dev_out)
		*dev_out = dev;
	return st->cur;
}

const float *stagedTail(const DspBlock *consumer, const vector<sample_t> &host, wr_dev **dev_out, size_t tail_frames)
{
	return stagedSparse(consumer, host, dev_out, 0, 0, tail_frames);
}

const float *stagedBlock(const DspBlock *consumer, const vector<sample_t> &host, wr_dev **dev_out,
                         bool only_if_present)
{
	DspSource *src = TunerBatch::rootSource(consumer);
	if (!src || host.empty() || host.data() != src->currentBlock().data())
		return NULL;                    /* not fed straight from the source: caller uploads itself */
	wr_dev *dev = deviceFor(consumer);
	if (!dev)
		return NULL;
	if (const float *there = residentBlock(src, dev, host.size())) {
		if (dev_out)
			*dev_out = dev;
		return there;
	}
	std::lock_guard<std::mutex> g(g_stageLock);
	SourceStage *st = static_cast<SourceStage *>(src->gpuStage());
	if (!st && only_if_present)
		return NULL;
	if (!st) {
		st = new SourceStage();
		src->setGpuStage(st);
		src->setGpuCleanup(releaseSource);
		src->setGpuBeforeRun(beforeSourceRun);
		src->setGpuBeforeStop(beforeSourceStop);
	}
	if (st->epoch != src->epoch() || st->host != host.data() || st->floats != host.size()) {
		if (only_if_present)
			return NULL;
		const size_t bytes = host.size() * sizeof(float);
		st->dev = dev;
		/* A source that still holds the block in the RTL-SDR byte format (FileTuner; what RtlSdrTuner::dataReady
		 * receives, rtlsdrtuner.cxx:86-117): ship the BYTES -- a quarter of the float block over PCIe -- and
		 * convert on the device with the reference's rule (u8 - 128) / 128 (rtlsdrtuner.cxx:106: exact in
		 * float, so the staged block is the float block bit for bit).  Such a source need not even fill its
		 * float vector when every consumer reads the block on the device (DspBlock::readsSourceOnDevice). */
		const RawU8Block *rawsrc = dynamic_cast<const RawU8Block *>(src);
		size_t rawFrames = 0;
		const uint8_t *rawBytes = rawsrc ? rawsrc->rawU8(&rawFrames) : NULL;
		if (rawBytes && rawFrames * 2 == host.size() && !envUnsigned("WEBRADIO_NO_U8_STAGING", 0)) {
			const bool pinned = st->pin(dev, rawBytes, host.size());
			/* page-locked: the bytes cross PCIe as a DMA copy on the library's upload stream, beside the kernels of
			 * the block before, and are converted on the device (wr_u8_to_f32_from_host: nothing the host waits
			 * for); else a staged copy and the kernel on the device copy */
			DevBuf &dst = st->buf;
			if (!dst.reserve(dev, bytes) ||
			    (pinned ? wr_u8_to_f32_from_host(dev, rawBytes, (float *)dst.ptr, host.size())
			            : (!st->raw.reserve(dev, host.size()) ||
			               wr_dev_upload(dev, st->raw.ptr, rawBytes, host.size()) != WR_OK)
			                  ? WR_ERR_HIP
			                  : wr_u8_to_f32(dev, (const uint8_t *)st->raw.ptr, (float *)dst.ptr, host.size())) != WR_OK) {
				LOG_ERROR("staging the source block (byte format) failed: %s\n", wr_last_error());
				return NULL;
			}
			st->epoch = src->epoch();
			st->host = host.data();
			st->floats = host.size();
			st->rawStaged = pinned;
			st->cur = (const float *)dst.ptr;
			if (dev_out)
				*dev_out = dev;
			return st->cur;
		}
		st->rawStaged = false;
		st->cur = NULL;
		/* out of page-locked memory the copy is enqueued and the graph walk goes on beside it; the
		 * source's next run() waits for it before it touches the vector again (beforeSourceRun) */
		const bool pinned = st->pin(dev, host.data(), bytes);
		/* page-locked: on the upload stream, into the device copy the LAST block was not staged in -- the 32 MB of block
		 * b + 1 cross PCIe while the kernels of block b run (the source's next run() still waits for the copy before it
		 * touches the vector again: a float source refills the vector it swapped out at once, rtlsdrtuner.cxx:265-285) */
		if (pinned)
			st->second = !st->second;
		DevBuf &dst = (pinned && st->second) ? st->buf2 : st->buf;
		if (!dst.reserve(dev, bytes) ||
		    (pinned ? wr_dev_upload_ahead(dev, dst.ptr, host.data(), bytes)
		            : wr_dev_upload(dev, dst.ptr, host.data(), bytes)) != WR_OK) {
			LOG_ERROR("staging the source block failed: %s\n", wr_last_error());
			return NULL;
		}
		st->epoch = src->epoch();
		st->host = host.data();
		st->floats = host.size();
		st->cur = (const float *)dst.ptr;
	}
	if (dev_out)
		*dev_out = dev;
	return st->cur;
}

void submitBatchFirst(const DspBlock *consumer, const vector<sample_t> &host)
{
	DspSource *src = TunerBatch::rootSource(consumer);
	if (!src || host.empty() || host.data() != src->currentBlock().data())
		return;
	TunerBatch *batch = src->batch();
	if (batch && batch->channels())
		(void)batch->submitOnce(host, (unsigned int)(host.size() / 2));      /* (the receivers see its verdict when they ask) */
}

bool streamInfo(const DspBlock *block, bool *live, unsigned long long *launches, unsigned long long *blocks)
{
	DspSource *src = TunerBatch::rootSource(block);
	TunerBatch *batch = src ? src->batch() : NULL;
	return batch && batch->streamInfo(live, launches, blocks);
}

bool hostBlockValid(const DspBlock *block)
{
	DspSource *src = TunerBatch::rootSource(block);
	return !src || src->hostBlockValid();
}

bool DevBuf::reserve(wr_dev *d, size_t nbytes)
{
	if (dev == d && bytes >= nbytes && ptr)
		return true;
	release();
	if (!d || wr_dev_malloc(d, nbytes ? nbytes : 4, &ptr) != WR_OK) {
		ptr = NULL;
		return false;
	}
	dev = d;
	bytes = nbytes;
	return true;
}

void DevBuf::release()
{
	if (ptr && dev)
		wr_dev_free(dev, ptr);
	ptr = NULL;
	bytes = 0;
	dev = NULL;
}

/* ------------------------------------------------------------------ TunerBatch -- */

/* WEBRADIO_TIMES=1 (measurement only): where the wall time of an on-time block goes inside the batch -- averaged and printed
 * when the batch goes (profiles/r05_host_times.txt) */
static const bool g_times = getenv("WEBRADIO_TIMES") && atoi(getenv("WEBRADIO_TIMES")) != 0;
static double g_tacc[8];
static unsigned long g_tn;
static double g_tlast;
static inline double nowUs()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

TunerBatch::TunerBatch(DspSource *source, wr_dev *dev)
	: _source(source), _dev(dev), _tuner(NULL), _rate(0), _maxFrames(0), _submittedEpoch(0),
	  _submitOk(false), _audioPtr(NULL), _ringHeld(false), _audioStride(0), _audioFrames(0), _audioSlots(0),
	  _late(envUnsigned("WEBRADIO_AUDIO_LATE", 0) != 0), _lateDepth(envUnsigned("WEBRADIO_AUDIO_LATE", 0) >= 2 ? 2u : 1u),
	  _lateQueued(false), _silence(false), _lateSeq(0), _pieces(envUnsigned("WEBRADIO_PIECES", 2)), _delivered(false), _partSeq0(0),
	  _streaming(false)
{
	if (_pieces < 1 || _late)
		_pieces = 1;
	if (_pieces > 8)
		_pieces = 8;
}

TunerBatch::~TunerBatch()
{
	if (g_times && g_tn)
		fprintf(stderr, "WEBRADIO_TIMES, us per block over %lu blocks: walk+source %.1f | stage enqueued %.1f | submits+flush %.1f (from the submit's start) | "
		        "wait part 0 %.1f copy %.1f | wait part 1 %.1f copy %.1f\n", g_tn, g_tacc[0] / g_tn, g_tacc[1] / g_tn, g_tacc[2] / g_tn,
		        g_tacc[3] / g_tn, g_tacc[4] / g_tn, g_tacc[5] / g_tn, g_tacc[6] / g_tn);
	if (_tuner)
		wr_tuner_destroy(_tuner);
}

DspSource *TunerBatch::rootSource(const DspBlock *block)
{
	const DspBlock *b = block;
	while (b && b->_producer)
		b = b->_producer;
	return const_cast<DspSource *>(dynamic_cast<const DspSource *>(b));
}

wr_dev *TunerBatch::batchDeviceOf(const DspBlock *block)
{
	DspSource *src = rootSource(block);
	return (src && src->batch()) ? src->batch()->dev() : NULL;
}

void TunerBatch::destroyFor(DspSource *src)
{
	TunerBatch *b = src->batch();
	if (!b)
		return;
	src->setBatch(NULL);
	for (size_t n = 0; n < b->_channels.size(); n++)
		delete b->_channels[n];          /* their blocks are gone or stopping */
	b->_channels.clear();
	delete b;
}

/* The shape radio.cxx:68-76 builds, with nothing else attached along the way.  Only
 * then is it safe to skip the intermediate buffers. */

Channel *TunerBatch::enrol(DownConverter *mixer)
{
	DspBlock::gpuUnfuse = TunerBatch::unfuseChainOf;
	if (envUnsigned("WEBRADIO_NO_FUSION", 0))
		return NULL;
	DspSource *src = dynamic_cast<DspSource *>(mixer->_producer);
	if (!src || mixer->_consumers.size() != 1)
		return NULL;
	LowPass *f1 = dynamic_cast<LowPass *>(mixer->_consumers[0]);
	if (!f1 || f1->_consumers.size() != 1)
		return NULL;
	/* optionally a second LowPass before the demodulator: how the reference's own means cut a narrow
	 * channel out of a fast stream (one 64-tap stage gives bin 0 below fs/128, lowpass.cxx:167) */
	LowPass *f1b = dynamic_cast<LowPass *>(f1->_consumers[0]);
	if (f1b && f1b->_consumers.size() != 1)
		return NULL;
	Demodulator *dm = dynamic_cast<Demodulator *>(f1b ? f1b->_consumers[0] : f1->_consumers[0]);
	if (!dm || dm->_consumers.size() != 1)
		return NULL;
	LowPass *f2 = dynamic_cast<LowPass *>(dm->_consumers[0]);
	if (!f2)
		return NULL;
	if (f1->_channel || dm->_channel || f2->_channel || (f1b && f1b->_channel))
		return NULL;
	if (f1->firLength() > WR_FIR_FUSED_MAX || f2->firLength() > WR_FIR_FUSED_MAX ||
	    (f1b && f1b->firLength() > WR_FIR_FUSED_MAX))
		return NULL;                         /* the tuner takes the reference's 64 taps (shorter filters ride as 64 taps
		                                        with the oldest ones zero) and 128 or 256 for any of the three filters
		                                        (r05: the audio filter and the second channel stage too) */

	TunerBatch *batch = src->batch();
	if (!batch) {
		wr_dev *dev = deviceFor(mixer);      /* the source's GPU (tuners shard one per GPU) */
		if (!dev)
			return NULL;
		batch = new TunerBatch(src, dev);
		src->setBatch(batch);
	}

	std::lock_guard<std::mutex> g(batch->_lock);
	{
		/* The wr_tuner outlives stop()/start() of the source (its channels park their state in
		 * their blocks meanwhile).  What a stopped source may have changed -- setSampleRate,
		 * setBlockSize (dspblock.h:60-66; refused while running) -- is baked into the tuner: the
		 * NCO step of every channel follows from the input rate (downconverter.cxx:80) and the
		 * device buffers are sized for the block.  With no channel enrolled nothing is lost by
		 * building a new one. */
		size_t wantFrames = src->blockSize() / 2;
		if (wantFrames == 0)
			wantFrames = 1;
		if (batch->_tuner && batch->_channels.empty() &&
		    (batch->_rate != src->outputSampleRate() || batch->_maxFrames != wantFrames)) {
			wr_tuner_destroy(batch->_tuner);
			batch->_tuner = NULL;
			batch->_ringHeld = false;
			batch->_audioSlots = 0;
			batch->_audioPtr = NULL;
			batch->_submitOk = false;
		}
	}
	if (!batch->_tuner) {
		batch->_rate = src->outputSampleRate();
		batch->_maxFrames = src->blockSize() / 2;
		if (batch->_maxFrames == 0)
			batch->_maxFrames = 1;
		unsigned int maxch = envUnsigned("WEBRADIO_MAX_CHANNELS", 1024);
		int nco = WR_NCO_ROTATE;
		const char *want = getenv("WEBRADIO_NCO");
		if (envUnsigned("WEBRADIO_NCO_EXACT", 0) || (want && !strcmp(want, "exact")))
			nco = WR_NCO_EXACT;
		else if (want && !strcmp(want, "split"))
			nco = WR_NCO_SPLIT;
		if (wr_tuner_create(&batch->_tuner, batch->_dev, batch->_rate, maxch, batch->_maxFrames, nco) != WR_OK) {
			LOG_ERROR("wr_tuner_create: %s\n", wr_last_error());
			batch->_tuner = NULL;
			return NULL;
		}
		wr_tuner_audio_ring(batch->_tuner, batch->_late ? 2 + batch->_lateDepth : 1 + (batch->_pieces > 1 ? batch->_pieces : 1));
		/* r06: a source that produces its blocks in device memory (DeviceBlock) is STREAMED: its blocks ring the doorbell of
		 * one persistent launch (k_tuner_stream, what bench.py's headline times) instead of launching a kernel each; a block's
		 * audio is in the pinned ring when WrStreamCtl::done says so, no flush.  Blocks that come out of HOST memory are
		 * staged by work on the device's own stream, which an open launch would hold up: they go the ordinary way.
		 * WEBRADIO_STREAM=0 turns it off. */
		batch->_streaming = dynamic_cast<DeviceBlock *>(src) != NULL && envUnsigned("WEBRADIO_STREAM", 1) != 0;
		if (batch->_streaming && wr_tuner_set_streaming(batch->_tuner, 1) != WR_OK)
			batch->_streaming = false;
		batch->_lateQueued = false;
		batch->_lateSeq = 0;
	}
	int id = -1;
	if (wr_chan_add(batch->_tuner, &id) != WR_OK) {
		LOG_ERROR("wr_chan_add: %s\n", wr_last_error());
		return NULL;
	}
	Channel *ch = new Channel();
	ch->batch = batch;
	ch->id = id;
	ch->slot = -1;
	ch->lateSeq = ~0ull;
	ch->mixer = mixer;
	ch->chanFilter = f1;
	ch->chanFilter2 = f1b;
	ch->demod = dm;
	ch->audioFilter = f2;
	ch->dirty = true;
	batch->_channels.push_back(ch);
	/* NCO phase and Demodulator prev_i/q outlive stop()/start() upstream (Q5) */
	wr_chan_set_state(batch->_tuner, id, mixer->_phase, dm->_prev);
	mixer->_channel = f1->_channel = dm->_channel = f2->_channel = ch;
	f1->_stage = 0;
	f2->_stage = 1;
	if (f1b) {
		f1b->_channel = ch;
		f1b->_stage = 2;
		f1b->elideOutput(true);
	}
	/* nobody reads these three on the host any more */
	mixer->elideOutput(true);
	f1->elideOutput(true);
	dm->elideOutput(true);
	return ch;
}

/* DspBlock::connect on a block whose output the fusion had elided */
void TunerBatch::unfuseChainOf(DspBlock *block)
{
	withdraw(block->gpuChannel());
}

void TunerBatch::withdraw(Channel *ch)
{
	if (!ch)
		return;
	TunerBatch *batch = ch->batch;
	if (batch && batch->_tuner) {
		std::lock_guard<std::mutex> g(batch->_lock);
		wr_chan_get_state(batch->_tuner, ch->id, &ch->mixer->_phase, ch->demod->_prev);
		wr_chan_remove(batch->_tuner, ch->id);
		for (size_t n = 0; n < batch->_channels.size(); n++)
			if (batch->_channels[n] == ch) {
				batch->_channels.erase(batch->_channels.begin() + n);
				break;
			}
	}
	ch->mixer->_channel = NULL;
	if (ch->chanFilter2) {
		ch->chanFilter2->_channel = NULL;
		ch->chanFilter2->elideOutput(false);
	}
	ch->chanFilter->_channel = NULL;
	ch->demod->_channel = NULL;
	ch->audioFilter->_channel = NULL;
	ch->mixer->elideOutput(false);
	ch->chanFilter->elideOutput(false);
	ch->demod->elideOutput(false);
	delete ch;
}

void TunerBatch::markDirty(Channel *ch)
{
	if (!ch)
		return;
	std::lock_guard<std::mutex> g(ch->batch->_lock);
	ch->dirty = true;
}

/* stage the block-level parameters of one channel (setters may have run on other
 * threads since the last block; they take effect here, at a block boundary) */
bool TunerBatch::pushParams(Channel *ch)
{
	LowPass *f1 = ch->chanFilter, *f2 = ch->audioFilter;
	if (!f1->isRunning() || !f2->isRunning() || !ch->demod->isRunning() ||
	    (ch->chanFilter2 && !ch->chanFilter2->isRunning()))
		return true;                    /* the chain is still starting: next block */
	if (wr_chan_set_if(_tuner, ch->id, ch->mixer->_ifHz) != WR_OK)
		return false;
	LowPass *fs[3] = {f1, ch->chanFilter2, f2};
	const int stages[3] = {WR_FILTER_CHANNEL, WR_FILTER_CHANNEL2, WR_FILTER_AUDIO};
	for (int n = 0; n < 3; n++) {
		if (!fs[n])
			continue;
		vector<float> taps;
		{
			std::lock_guard<std::mutex> g(fs[n]->_coeffLock);          /* see LowPass::recalculate */
			taps = fs[n]->_coeff;
		}
		if (!taps.empty() && taps.size() <= (size_t)WR_FIR_FUSED_MAX &&
		    wr_chan_set_taps_n(_tuner, ch->id, stages[n], taps.data(), (unsigned int)taps.size(),
		                       fs[n]->decimation()) != WR_OK)
			return false;
	}
	if (wr_chan_set_mode(_tuner, ch->id, (int)ch->demod->_mode) != WR_OK)
		return false;
	ch->dirty = false;
	return true;
}

bool TunerBatch::submitOnce(const vector<sample_t> &tunerBuffer, unsigned int nframes)
{
	std::lock_guard<std::mutex> g(_lock);
	if (_submittedEpoch == _source->epoch())
		return _submitOk;
	_submittedEpoch = _source->epoch();
	_submitOk = false;
	bool pushed = false;
	for (size_t n = 0; n < _channels.size(); n++)
		if (_channels[n]->dirty || _channels[n]->slot < 0) {
			pushed = true;
			if (_channels[n]->dirty && !pushParams(_channels[n])) {
				LOG_ERROR("channel parameters rejected: %s\n", wr_last_error());
				return false;
			}
		}
	if (nframes > _maxFrames) {
		LOG_ERROR("block of %u frames exceeds the %zu the tuner batch was sized for\n", nframes, _maxFrames);
		return false;
	}
	if (_ringHeld) {
		wr_tuner_audio_ring_release(_tuner);      /* last block's audio has been handed out */
		_ringHeld = false;
		_audioSlots = 0;
	}
	_delivered = false;
	/* On time (the default: a block's audio is handed on within the run() that brought the block) the block goes through
	 * in P equal parts: part p + 1 crosses PCIe while part p's kernels run, part p's audio comes back -- and is put where
	 * the audio filters' consumers will read it -- while part p + 1 computes.  What run() then waits for after the link
	 * has delivered the block is one part's kernels and transfer, not the block's (r03: transfer, kernels, audio copy
	 * and hand-out in series, 0.43 ms a C2 block of bytes; the bits do not depend on how a stream is cut into blocks:
	 * tests/test_gpu_tuner.py::test_block_split_invariance). */
	/* (r04) Where every receiver decimates alike and by at least twice its channel filter's length, only the frames under
	 * the taps are brought over (stagedSparse): a sixth of the block at BASELINE config 2 -- and no transfer to wait for. */
	unsigned int sp = 0, sl = 0;
	const double tt0 = g_times ? nowUs() : 0.0;
	if (g_times && g_tlast > 0.0)
		g_tacc[0] += tt0 - g_tlast;                 /* from the end of the last collectParts to this submit: the walk + the source */
	if (!pushed && sparseWindows(&sp, &sl)) {
		wr_dev *sdev = NULL;
		const float *staged = stagedSparse(_channels[0]->mixer, tunerBuffer, &sdev, sp, sl, sl - 1u);
		if (g_times)
			g_tacc[1] += nowUs() - tt0;             /* staging enqueued */
		if (staged && sdev == _dev) {
			/* on time, the staged block goes through in P parts all the same: a part's audio is put where the audio
			 * filters' consumers will read it while the parts behind it compute (collectParts) */
			const unsigned int parts = piecesFor(nframes);
			wr_tuner_submit_count(_tuner, &_partSeq0);      /* part p's ring entry will be numbered _partSeq0 + p: the library's count */
			for (unsigned int p = 0; p < parts; p++)
				if (wr_tuner_submit(_tuner, staged + (size_t)2 * (nframes / parts) * p, nframes / parts, WR_DEVICE) != WR_OK) {
					LOG_ERROR("wr_tuner_submit: %s\n", wr_last_error());
					if (p)
						drainRing();                /* the parts that did go out must not be taken for the next block's */
					return false;
				}
			if (parts == 1)
				return afterSubmit(false);
			++_lateSeq;
			if (!_streaming)
				wr_tuner_flush(_tuner);                     /* (a streaming launch's post stage starts by itself) */
			traceAdd(_source, 'S');
			if (g_times)
				g_tacc[2] += nowUs() - tt0;         /* ... parts submitted and flushed */
			return collectParts(parts);
		}
	}
	unsigned int P = pushed ? 1u : piecesFor(nframes);
	if (P > 1) {
		wr_tuner_submit_count(_tuner, &_partSeq0);
		for (unsigned int p = 0; p < P; p++) {
			const float *part = stagePiece(_source, _dev, tunerBuffer, P, p);
			if (!part) {
				if (p == 0) {
					P = 1;                      /* cannot be staged in parts: the one-piece path below */
					break;
				}
				return false;
			}
			if (wr_tuner_submit(_tuner, part, nframes / P, WR_DEVICE) != WR_OK) {
				LOG_ERROR("wr_tuner_submit (part %u of %u): %s\n", p, P, wr_last_error());
				if (p)
					drainRing();
				return false;
			}
		}
	}
	if (P > 1) {
		++_lateSeq;
		wr_tuner_flush(_tuner);                 /* the last part's demodulator + audio filter: nothing rides behind it */
		traceAdd(_source, 'S');
		return collectParts(P);
	}
	/* cheapest way to get the block to the GPU: (1) a device copy some other consumer of the
	 * source already made this block, (2) the source's raw bytes (2 per frame, converted in
	 * the kernel's load stage), (3) stage the float block once for everybody */
	wr_dev *sdev = NULL;
	int rc;
	const float *staged = _channels.empty() ? NULL : stagedBlock(_channels[0]->mixer, tunerBuffer, &sdev, true);
	const RawU8Block *raw = dynamic_cast<const RawU8Block *>(_source);
	size_t rawFrames = 0;
	const uint8_t *bytes = (raw && !staged) ? raw->rawU8(&rawFrames) : NULL;
	if (staged && sdev == _dev) {
		rc = wr_tuner_submit(_tuner, staged, nframes, WR_DEVICE);
	} else if (bytes && rawFrames == nframes && envUnsigned("WEBRADIO_NO_U8_STAGING", 0)) {
		rc = wr_tuner_submit_u8(_tuner, bytes, nframes, WR_HOST);
	} else {
		staged = _channels.empty() ? NULL : stagedBlock(_channels[0]->mixer, tunerBuffer, &sdev, false);
		if (!(staged && sdev == _dev) && !_source->hostBlockValid()) {
			LOG_ERROR("the source left its block on the device and the device copy is not there\n");
			return false;
		}
		rc = (staged && sdev == _dev) ? wr_tuner_submit(_tuner, staged, nframes, WR_DEVICE)
		                              : wr_tuner_submit(_tuner, tunerBuffer.data(), nframes, WR_HOST);
	}
	if (rc != WR_OK) {
		LOG_ERROR("wr_tuner_submit: %s\n", wr_last_error());
		return false;
	}
	return afterSubmit(pushed);
}

bool TunerBatch::streamInfo(bool *live, unsigned long long *launches, unsigned long long *blocks)
{
	std::lock_guard<std::mutex> g(_lock);
	int l = 0;
	unsigned long long a = 0, b = 0;
	if (!_tuner || wr_tuner_stream_info(_tuner, &l, &a, &b) != WR_OK)
		return false;
	if (live)
		*live = l != 0;
	if (launches)
		*launches = a;
	if (blocks)
		*blocks = b;
	return true;
}

/* do all receivers read the source block through windows of one shape -- the same decimation, the same channel-filter
 * length -- and sparsely enough for sparse staging to pay (a window at most every second filter length)? */
bool TunerBatch::sparseWindows(unsigned int *period, unsigned int *length)
{
	if (_channels.empty())
		return false;
	unsigned int d = 0, l = 0;
	for (size_t n = 0; n < _channels.size(); n++) {
		const LowPass *f = _channels[n]->chanFilter;
		if (!f->isRunning() || _channels[n]->slot < 0)
			return false;
		const unsigned int dn = f->decimation(), ln = f->firLength();
		if (n == 0) {
			d = dn;
			l = ln;
		} else if (dn != d || ln != l) {
			return false;
		}
	}
	if (l < 2 || d < 2 * l)
		return false;
	*period = d;
	*length = l;
	return true;
}

/* the block has been submitted (enqueued): bookkeeping, and this run()'s audio */
bool TunerBatch::afterSubmit(bool pushed)
{
	++_lateSeq;                            /* blocks submitted so far */
	if (pushed)                            /* a filter change may have moved a channel to another rate group */
		for (size_t n = 0; n < _channels.size(); n++) {
			const int before = _channels[n]->slot;
			if (wr_chan_slot(_tuner, _channels[n]->id, &_channels[n]->slot) != WR_OK)
				_channels[n]->slot = -1;
			if (_channels[n]->slot != before)
				_channels[n]->lateSeq = _lateSeq;          /* this block is the first in that slot: the ring entry
				                                              of the block before it is not this channel's */
		}
	/* the graph hands this block's audio on within this very run(): no waiting for the next launch.
	 * (WEBRADIO_AUDIO_LATE=2 hands out the audio of the block before the previous one: the demodulator and
	 * audio filter of a block then ride in the NEXT block's launch, as in bench.py -- one launch per block.) */
	if (!(_late && _lateDepth >= 2) && !_streaming)
		wr_tuner_flush(_tuner);                /* (a flush would close a streaming launch, every block: its post stage needs none) */
	traceAdd(_source, 'S');
	{
		int ready = 0;
		if (traceOn() && (!_late || _lateSeq > _lateDepth) && wr_tuner_audio_ring_ready(_tuner, &ready) == WR_OK)
			traceAdd(_source, ready ? 'A' : 'W');
	}
	if (_late) {
		/* WEBRADIO_AUDIO_LATE=1 (2): the sinks get every block's audio ONE run() (TWO) later (a block's worth
		 * of latency, 40 ms at C2).  Nothing in run() then waits for the GPU: this block's copy,
		 * kernels and audio transfer are merely enqueued, what is handed out is the previous
		 * block's audio, which arrived in the pinned ring while the host was busy elsewhere -- and
		 * the front ends of a Radio (radio.cxx:56-59 pumps them one after the other) keep all their
		 * GPUs busy at once.  The first block's run() hands out silence, the last block's audio is
		 * dropped at stop(). */
		_silence = _lateSeq <= _lateDepth;          /* (blocks submitted so far, this one included) */
		_lateQueued = true;
		_audioSlots = 0;
		if (!_silence) {
			const float *p = NULL;
			size_t stride = 0, frames = 0;
			unsigned int slots = 0;
			unsigned long long seq = 0;
			if (wr_tuner_audio_ring_acquire(_tuner, &p, &stride, &frames, &slots, &seq) == WR_OK) {
				_ringHeld = true;
				_audioPtr = p;
				_audioStride = stride;
				_audioFrames = frames;
				_audioSlots = slots;
			}
			/* else: nothing queued -- receivers in several rate groups are not ringed (see the C ABI);
			 * audio() then fetches per channel, which is this block's audio, on time */
		}
		_submitOk = true;
		return true;
	}
	/* one transfer brings back the audio of every channel: through the tuner's pinned ring
	 * (queued behind the kernels by the submit itself), read in place until the next block */
	size_t stride = 0, frames = 0;
	unsigned int slots = 0;
	{
		const float *p = NULL;
		unsigned long long seq = 0;
		if (wr_tuner_audio_ring_acquire(_tuner, &p, &stride, &frames, &slots, &seq) == WR_OK) {
			_ringHeld = true;
			_audioPtr = p;
			_audioStride = stride;
			_audioFrames = frames;
			_audioSlots = slots;
			_submitOk = true;
			return true;
		}
	}
	if (wr_tuner_fetch_audio_all(_tuner, NULL, 0, &stride, &frames, &slots) != WR_OK && frames * slots == 0) {
		/* several rate groups: fall back to per-channel fetches in audio() */
		_audioSlots = 0;
		_submitOk = true;
		return true;
	}
	_audio.resize((size_t)slots * frames + 1);
	if (frames != 0 && slots != 0 &&
	    wr_tuner_fetch_audio_all(_tuner, _audio.data(), _audio.size(), &stride, &frames, &slots) != WR_OK) {
		_audioSlots = 0;                /* per-channel fallback */
	} else {
		_audioPtr = _audio.data();
		_audioStride = stride;
		_audioFrames = frames;
		_audioSlots = slots;
	}
	_submitOk = true;
	return true;
}

/* lcm with a ceiling: anything this large stands for "do not split" */
static unsigned long lcmCapped(unsigned long a, unsigned long b)
{
	if (!a || !b)
		return 0;
	unsigned long x = a, y = b;
	while (y) {
		const unsigned long t = x % y;
		x = y;
		y = t;
	}
	const unsigned long long l = (unsigned long long)(a / x) * b;
	return l > (1ull << 40) ? 0 : (unsigned long)l;
}

/* in how many equal parts a block of `nframes` goes through (submitOnce): as many as WEBRADIO_PIECES allows (default 2: each further part costs a launch and a transfer of
 * its own, ~30 us, more than its overlap saves -- profiles/r04_host_pieces.txt;
 * 1 turns it off) such that every part is a whole number of audio frames of every receiver, all receivers decimate alike
 * (one rate group: the tuner's pinned audio ring then carries every part's audio) and a part stays large enough for the
 * link and the kernels to work at their pace (WEBRADIO_PIECE_MIN_FRAMES, default 500 000) */
unsigned int TunerBatch::piecesFor(unsigned int nframes)
{
	if (_pieces <= 1 || _late || _channels.empty() || !nframes)
		return 1;
	unsigned long q = 1;
	unsigned int d[3] = {0, 0, 0};
	for (size_t n = 0; n < _channels.size(); n++) {
		const Channel *c = _channels[n];
		if (c->slot < 0 || !c->chanFilter->isRunning() || !c->audioFilter->isRunning())
			return 1;
		const unsigned int e[3] = {c->chanFilter->decimation(), c->chanFilter2 ? c->chanFilter2->decimation() : 1u,
		                           c->audioFilter->decimation()};
		if (n == 0)
			memcpy(d, e, sizeof(d));
		else if (memcmp(d, e, sizeof(d)))
			return 1;
		q = lcmCapped(q, (unsigned long)e[0] * e[1] * e[2]);
		if (!q)
			return 1;
	}
	static const unsigned int minFrames = envUnsigned("WEBRADIO_PIECE_MIN_FRAMES", 500000);
	for (unsigned int p = _pieces; p > 1; p--)
		if (nframes % (p * q) == 0 && nframes / p >= minFrames)
			return p;
	return 1;
}

/* the audio of the P parts just submitted, in order: each part's rows are copied out of the tuner's pinned ring as soon as
 * they are there -- straight into the audio filters' output vectors where those already have the block's size (the steady
 * state: LowPass::process then has nothing left to copy), else into _audio for audio() to slice -- while the parts behind
 * it are still being worked on.  r05: a ring entry that is not the part expected -- a leftover of a run() that failed half
 * way -- is an error, not somebody else's audio.  (Helper threads for the copies were tried and taken out again: 256 rows of
 * 4 KB per part are 30-40 us of one core, but what an on-time block waits for is the GPU -- WEBRADIO_TIMES=1 says where the
 * time goes: profiles/r05_host_times.txt.) */
bool TunerBatch::collectParts(unsigned int P)
{
	size_t off = 0, total = 0;
	bool direct = true;
	for (unsigned int p = 0; p < P; p++) {
		const float *ptr = NULL;
		size_t stride = 0, frames = 0;
		unsigned int slots = 0;
		unsigned long long seq = 0;
		const double ta = g_times ? nowUs() : 0.0;
		if (wr_tuner_audio_ring_acquire(_tuner, &ptr, &stride, &frames, &slots, &seq) != WR_OK) {
			LOG_ERROR("audio of part %u of %u: %s\n", p, P, wr_last_error());
			drainRing();
			return false;
		}
		const double tb = g_times ? nowUs() : 0.0;
		if (g_times)
			g_tacc[p ? 5 : 3] += tb - ta;           /* waiting for the part's audio */
		if (p == 0) {
			total = frames * P;
			for (size_t n = 0; n < _channels.size() && direct; n++)
				direct = _channels[n]->slot >= 0 && (unsigned int)_channels[n]->slot < slots &&
				         _channels[n]->audioFilter->DspBlock::_out.size() == total;
			if (!direct)
				_audio.resize((size_t)slots * total + 1);
			_audioSlots = slots;
		}
		/* the tuner numbers its submits from 0: part p of this block is submit _partSeq0 + p */
		if (frames * P != total || slots != _audioSlots || seq != _partSeq0 + p) {
			wr_tuner_audio_ring_release(_tuner);
			LOG_ERROR("part %u of %u came back as submit %llu with %zu frames (expected submit %llu, %zu frames): stale ring entries\n",
			          p, P, seq, frames, _partSeq0 + p, total / P);
			drainRing();
			return false;
		}
		if (direct) {
			for (size_t n = 0; n < _channels.size(); n++)
				memcpy(_channels[n]->audioFilter->DspBlock::_out.data() + off, ptr + (size_t)_channels[n]->slot * stride,
				       frames * sizeof(float));
		} else {
			for (unsigned int sl = 0; sl < slots; sl++)
				memcpy(_audio.data() + (size_t)sl * total + off, ptr + (size_t)sl * stride, frames * sizeof(float));
		}
		wr_tuner_audio_ring_release(_tuner);
		off += frames;
		if (g_times)
			g_tacc[p ? 6 : 4] += nowUs() - tb;      /* copying it out */
	}
	if (g_times) {
		g_tlast = nowUs();
		++g_tn;
	}
	_ringHeld = false;
	_audioFrames = total;
	if (direct) {
		_delivered = true;
	} else {
		_audioPtr = _audio.data();
		_audioStride = total;
	}
	_submitOk = true;
	return true;
}

/* whatever is queued in the tuner's audio ring goes: after an error half way through a block its entries would be taken
 * for the next block's parts */
void TunerBatch::drainRing()
{
	wr_tuner_flush(_tuner);
	for (;;) {
		unsigned int queued = 0;
		if (wr_tuner_audio_ring_stats(_tuner, &queued, NULL) != WR_OK || !queued)
			break;
		const float *ptr = NULL;
		size_t stride = 0, frames = 0;
		unsigned int slots = 0;
		if (wr_tuner_audio_ring_acquire(_tuner, &ptr, &stride, &frames, &slots, NULL) != WR_OK)
			break;
		wr_tuner_audio_ring_release(_tuner);
	}
	_ringHeld = false;
}

bool TunerBatch::audio(const Channel *ch, vector<sample_t> &out)
{
	/* under the batch lock: withdraw() / unfuseChainOf() may run on an HTTP thread (a consumer connected
	 * inside a fused chain) while the run thread hands audio out; uncontended it costs 256 x ~20 ns a block */
	std::lock_guard<std::mutex> g(_lock);
	if (!_submitOk)
		return false;
	if (out.empty())
		return true;
	if (_delivered && out.size() == _audioFrames && out.data() == ch->audioFilter->DspBlock::_out.data())
		return true;                            /* collectParts put it there while the block's last parts were computing */
	if (_late && _silence) {
		memset(out.data(), 0, out.size() * sizeof(float));
		return true;
	}
	const int slot = ch->slot;
	if (_audioSlots && slot >= 0 && (unsigned int)slot < _audioSlots && _audioFrames == out.size() &&
	    (!_late || ch->lateSeq + (_lateDepth - 1u) < _lateSeq)) {
		memcpy(out.data(), _audioPtr + (size_t)slot * _audioStride, out.size() * sizeof(float));
		return true;
	}
	if (_late && _audioSlots) {
		/* One block late, and the ring entry is not this channel's previous block: another block length
		 * (a ragged block, a changed audio rate), or a channel enrolled (or moved to this slot) with the
		 * block just submitted, whose slot held somebody else's audio a block ago.  Fetching now would hand
		 * out the CURRENT block -- and the next run() the same block again from the ring.  The sinks get a
		 * block of silence instead, exactly as at the start of the stream. */
		memset(out.data(), 0, out.size() * sizeof(float));
		return true;
	}
	size_t got = 0;
	if (wr_chan_fetch(_tuner, ch->id, WR_STAGE_AUDIO, out.data(), out.size(), &got) != WR_OK) {
		LOG_ERROR("wr_chan_fetch: %s\n", wr_last_error());
		return false;
	}
	return got == out.size();
}

} // namespace wrhost
