/*
 * gpubatch.h -- host glue between the DspBlock graph and the C ABI (webradio_amd.h).
 *
 * The reference walks the graph depth first, one Receiver after the other
 * (dspblock.cxx:207-209), and every block materialises its output.  The GPU wants the
 * opposite: all Receivers of a tuner in one launch sequence, and none of the full-rate
 * intermediates.  TunerBatch reconciles the two without changing the graph API:
 *
 *   - when a DownConverter that hangs directly off a DspSource is started and the blocks
 *     after it form exactly the Receiver chain of radio.cxx:68-76
 *     (DownConverter -> LowPass -> Demodulator -> LowPass), the four blocks are enrolled
 *     as one CHANNEL of the source's TunerBatch (a wr_tuner on one GPU); so is the same chain
 *     with a second LowPass in front of the demodulator (narrow channels off a fast stream);
 *   - the first enrolled DownConverter::process() of a source block ("epoch") uploads the
 *     tuner buffer once and submits every channel; later process() calls of enrolled
 *     blocks in the same epoch are no-ops, except the audio LowPass, which copies its
 *     channel's audio out of the batch's single device-to-host transfer;
 *   - blocks that are not part of such a chain run one kernel each (wr_mix,
 *     wr_fir_decimate, wr_demod) on their own host buffers.
 */
#ifndef WRHOST_GPUBATCH_H_
#define WRHOST_GPUBATCH_H_

#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "dspblock.h"
#include "webradio_amd.h"

class DownConverter;
class LowPass;
class Demodulator;

namespace wrhost {

/* process-wide device contexts, one per GPU, opened on first use */
wr_dev *device(int index);
int deviceCount();
/* the device a block should use.  Every block below one DspSource shares that source's GPU:
 * the n-th source of the process that asks takes GPU n mod count (tuners shard one per GPU),
 * unless WEBRADIO_DEVICE pins all of them.  Blocks not below a source use WEBRADIO_DEVICE / 0. */
wr_dev *deviceFor(const DspBlock *block);
/* Device copy of the source's current block, uploaded once per source block and shared by
 * every GPU consumer of that source (the SpectrumSink and the tuner batch would otherwise
 * each push the same 32 MB over PCIe).  Returns NULL if `host` is not that source's current
 * output vector or the upload failed. */
const float *stagedBlock(const DspBlock *consumer, const vector<sample_t> &host, wr_dev **dev_out,
                         bool only_if_present = false);
/* false while the root source of `block` left its host vector unfilled for the current block (a byte-format
 * source all of whose consumers read on the device): a device path that fails must then fail loudly */
bool hostBlockValid(const DspBlock *block);
/* r06: did the tuner batch of `block`'s root source stream (a DeviceBlock source: wr_tuner_set_streaming)?  `live`: a launch is
 * open right now; `launches` / `blocks`: opened / taken so far.  false: no batch */
bool streamInfo(const DspBlock *block, bool *live, unsigned long long *launches, unsigned long long *blocks);
/* For a consumer of the source's block that is NOT part of the tuner batch (the SpectrumSink, which FrontEnd connects
 * first): have the batch submit this block now, before the consumer enqueues its own work -- the receivers' launches then
 * come first on the device's stream and the audio does not wait behind the spectrum's copy and transform (the batch
 * submits once per block whoever asks first).  No batch, or not the source's current block: nothing happens. */
void submitBatchFirst(const DspBlock *consumer, const vector<sample_t> &host);
/* For a consumer that reads only the END of the source's block (the SpectrumSink: the most recent complete frame and what
 * follows it): the block's device address with at least its last `tail_frames` frames staged -- the whole staged block
 * where somebody staged it, else just that tail, brought over by the sparse staging kernel (a block whose receivers all
 * read it through sparse windows never crosses PCIe whole).  NULL: not fed straight from a source, or it cannot be done
 * (the caller falls back to stagedBlock). */
const float *stagedTail(const DspBlock *consumer, const vector<sample_t> &host, wr_dev **dev_out, size_t tail_frames);


/* WEBRADIO_TRACE=1: what the tuner batches did, in order -- 'S' a block submitted (enqueued, nothing
 * waited for), 'A' audio taken from the pinned ring that was already there, 'W' audio that had to
 * be waited for.  (source, kind) pairs; tests read it to see that the front ends of a Radio have
 * their blocks in flight at the same time. */
struct TraceEvent { const void *source; char kind; };
const std::vector<TraceEvent> &trace();
void traceClear();

/* a resizable device buffer */
struct DevBuf {
	DevBuf() : dev(NULL), ptr(NULL), bytes(0) {}
	~DevBuf() { release(); }
	bool reserve(wr_dev *d, size_t nbytes);
	void release();
	wr_dev *dev;
	void *ptr;
	size_t bytes;
};

class TunerBatch;

struct Channel {
	TunerBatch *batch;
	int id;                       /* wr_tuner channel id */
	int slot;                     /* its row in the tuner's audio array (refreshed when parameters are pushed) */
	DownConverter *mixer;
	LowPass *chanFilter;
	LowPass *chanFilter2;         /* a second LowPass in front of the demodulator, or NULL */
	Demodulator *demod;
	LowPass *audioFilter;
	bool dirty;                   /* parameters changed since the last submit */
	unsigned long long lateSeq;   /* the first block (counted from 1) submitted with the channel in `slot`: an
	                                 older ring entry's row `slot` is not this channel's (WEBRADIO_AUDIO_LATE) */
};

class TunerBatch {
public:
	/* enrol the Receiver chain that starts at `mixer` if the graph has that shape;
	 * returns the channel or NULL (-> the blocks run stand-alone) */
	static Channel *enrol(DownConverter *mixer);
	static void withdraw(Channel *ch);
	static void unfuseChainOf(DspBlock *block);
	/* the root source's batch device, or NULL if the block is not below a batched source */
	static wr_dev *batchDeviceOf(const DspBlock *block);
	static DspSource *rootSource(const DspBlock *block);
	static void destroyFor(DspSource *src);

	/* called from DownConverter::process of an enrolled block */
	bool submitOnce(const vector<sample_t> &tunerBuffer, unsigned int nframes);
	/* called from the audio LowPass::process of an enrolled block */
	bool audio(const Channel *ch, vector<sample_t> &out);
	/* a setter changed a parameter of this channel (any thread): re-stage it at the next
	 * block boundary */
	static void markDirty(Channel *ch);

	bool streamInfo(bool *live, unsigned long long *launches, unsigned long long *blocks);
	bool streaming() const { return _streaming; }
	wr_dev *dev() const { return _dev; }
	DspSource *source() const { return _source; }
	size_t channels() const { return _channels.size(); }

private:
	TunerBatch(DspSource *source, wr_dev *dev);
	~TunerBatch();
	bool ensureTuner(unsigned int nframes);
	bool pushParams(Channel *ch);
	unsigned int piecesFor(unsigned int nframes);
	bool sparseWindows(unsigned int *period, unsigned int *length);
	bool afterSubmit(bool pushed);
	bool collectParts(unsigned int parts);
	void drainRing();

	DspSource *_source;
	wr_dev *_dev;
	wr_tuner *_tuner;
	unsigned int _rate;
	size_t _maxFrames;
	unsigned long _submittedEpoch;
	bool _submitOk;
	std::vector<Channel *> _channels;
	std::vector<float> _audio;        /* [slot][frames] of the last submit (fallback without the ring) */
	const float *_audioPtr;           /* where audio() reads: a pinned ring slot, or _audio */
	bool _ringHeld;                   /* a slot of the tuner's pinned audio ring is acquired */
	size_t _audioStride, _audioFrames;
	unsigned int _audioSlots;
	bool _late;                       /* WEBRADIO_AUDIO_LATE: hand out the PREVIOUS block's audio (see submitOnce) */
	unsigned int _lateDepth;          /* 1: the previous block's audio; 2: the one before (no flush: one launch per block) */
	bool _lateQueued;                 /* a block has been submitted whose audio has not been handed out yet */
	bool _silence;                    /* late mode, first block: nothing to hand out yet */
	unsigned long long _lateSeq;      /* blocks submitted so far */
	unsigned int _pieces;             /* WEBRADIO_PIECES: parts an on-time block is put through in (see submitOnce) */
	bool _delivered;                  /* this block's audio already lies in the audio filters' output vectors */
	std::mutex _lock;
	unsigned long long _partSeq0;     /* the number the tuner gave the first part of the block collectParts is about to collect
	                                     (wr_tuner_submit_count before the parts went out: the library's own count) */
	bool _streaming;                  /* the source produces its blocks in device memory (DeviceBlock): wr_tuner_set_streaming */
};

} // namespace wrhost

#endif /* WRHOST_GPUBATCH_H_ */
