/*
 * dspblock.h -- push-model operator graph of the MI355X webradio backend.
 *
 * Source-compatible with webradio's src/dsp/dspblock.h:50-140: same class names, same
 * public and protected members with the same meaning, so callers written against the
 * reference (radio.cxx, main.cxx, the web handlers, any DspBlock subclass) build
 * against this header unchanged.  The scheduling semantics are the reference's,
 * restated in dspblock.cxx and pinned to it by tests/test_dspblock_parity.py.
 *
 * What is new is private: each block knows its producer (`upstream`), and a block may
 * tell the runtime that its output buffer is never looked at (`elideOutput`) -- the
 * fused GPU path computes a whole Receiver chain in one launch sequence and only the
 * audio filter's output exists on the host.
 */
#ifndef DSPBLOCK_H_
#define DSPBLOCK_H_

// the reference hard-enables its profiler the same way (dspblock.h:29)
#define DSPBLOCK_PROFILE

#include <stdint.h>
#include <time.h>

#include <string>
#include <vector>

#define DEFAULT_SAMPLE_RATE		48000
#define DEFAULT_CHANNELS		2
#define DEFAULT_BLOCK_SIZE		16384

using namespace std;	// reference headers do this and its sources rely on it (radio.h:44)

typedef float sample_t;

class DspSource;
namespace wrhost { class TunerBatch; struct Channel; }

class DspBlock
{
	friend class DspSource;
	friend class wrhost::TunerBatch;
public:
	DspBlock(const string &name = "<undefined>", const string &type = "DspBlock");
	virtual ~DspBlock();

	/* graph construction (dspblock.cxx:57-92 in the reference) */
	void connect(DspBlock *block);
	void disconnect(DspBlock *block);

	/* negotiated stream format */
	unsigned int inputSampleRate() const { return _inRate; }
	unsigned int outputSampleRate() const { return _outputSampleRate; }
	unsigned int inputChannels() const { return _inChannels; }
	unsigned int outputChannels() const { return _outputChannels; }
	unsigned int decimation() const { return _decim; }
	unsigned int interpolation() const { return _interp; }

	/* per-block profiler (process-CPU nanoseconds around process()) */
	uint64_t nsPerFrameAll() const;
	uint64_t nsPerFrameOne() const { return _nsTotal / _framesIn; }
	uint64_t totalNanoseconds() const { return _nsTotal; }
	unsigned int totalIn() const { return _framesIn; }
	unsigned int totalOut() const { return _framesOut; }

	bool isRunning() const { return _running; }
	const string& name() const { return _name; }
	const string& type() const { return _type; }

protected:
	/* the operator interface every block implements */
	virtual bool init() { return false; }
	virtual void deinit() {}
	virtual bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer) { return false; }

	/* init() may override these two (rate change, channel change) */
	unsigned int	_outputSampleRate;
	unsigned int	_outputChannels;

	/* ---- additions of this backend (not in the reference) ---- */
	DspBlock* upstream() const { return _producer; }
	const vector<DspBlock*>& downstream() const { return _consumers; }
	/* A block whose output no consumer reads on the host (because the consumer gets its
	 * data from the tuner batch on the GPU) asks the runtime not to materialise it. */
	void elideOutput(bool on) { _elide = on; }
	bool outputElided() const { return _elide; }
	/* Device hand-over between blocks that run block by block on the GPU (a chain that is not
	 * the fused Receiver shape, e.g. several LowPass stages in a row): a block that leaves its
	 * output in device memory publishes the pointer, a consumer that can read device memory
	 * says so, and when every consumer can, the host copy of that output is never made. */
	virtual bool acceptsDeviceInput() const { return false; }
	/* a consumer of a DspSource that takes the source's block from the device copy the GPU glue stages once per
	 * block (wrhost::stagedBlock) and never looks at the host vector: when all of a source's consumers do, a
	 * source that holds the block in another form (FileTuner: the RTL-SDR bytes) need not fill the vector */
	virtual bool readsSourceOnDevice() const { return false; }
	bool consumersReadOnDevice() const;
	/* the tuner-batch channel this block is part of while its Receiver chain is fused (gpubatch.h) */
	virtual wrhost::Channel *gpuChannel() const { return NULL; }
	static void (*gpuUnfuse)(DspBlock *block);       /* takes the chain `block` is part of out of its tuner batch */
	void publishDeviceOutput(const void *devptr) { _devOut = devptr; }
	const void *upstreamDeviceOutput() const { return _producer ? _producer->_devOut : NULL; }
	bool hostOutputNeeded() const;
	/* frames of the block currently being pushed through this block */
	unsigned int currentInputFrames() const { return _curInFrames; }
	unsigned int currentOutputFrames() const { return _curOutFrames; }

private:
	bool start();
	void stop();
	bool run(const vector<sample_t> &inBuffer);
	bool runFrames(const vector<sample_t> &inBuffer, unsigned int inframes);
	void setSampleRate(unsigned int rate);
	void setChannels(unsigned int channels);

	const string		_name;
	const string		_type;
	unsigned int		_inRate;
	unsigned int		_inChannels;
	unsigned int		_decim;
	unsigned int		_interp;
	uint64_t			_nsTotal;
	uint64_t			_framesIn;
	uint64_t			_framesOut;
	uint64_t			_calls;			/* runFrames() calls: the profiler samples the clock (see runFrames) */
	bool				_running;
	bool				_elide;
	unsigned int		_curInFrames;
	unsigned int		_curOutFrames;
	DspBlock*			_producer;
	const void*			_devOut;		/* this block's output of the current block, on the device */
	vector<sample_t>	_out;
	vector<DspBlock*>	_consumers;
};

class DspSource : public DspBlock
{
public:
	DspSource(const string &name = "<undefined>", const string &type = "DspSource");
	virtual ~DspSource();

	unsigned int blockSize() const { return _blockSize; }

	bool start() { return DspBlock::start(); }
	void stop();
	bool run();
	void setSampleRate(unsigned int rate) { DspBlock::setSampleRate(rate); }
	void setChannels(unsigned int channels) { DspBlock::setChannels(channels); }
	void setBlockSize(unsigned int bytes);

	/* number of run() calls so far: the "epoch" the tuner batch keys its launches on */
	unsigned long epoch() const { return _epoch; }
	/* per-source GPU resources, owned by the GPU glue (gpubatch.cxx) and released through
	 * `_gpuCleanup` when the source dies: the tuner batch (created on demand by the first
	 * DownConverter), the device copy of the current block shared by every GPU consumer of
	 * this source, and the GPU the source's blocks run on (-1: not chosen yet) */
	wrhost::TunerBatch* batch() const { return _batch; }
	void setBatch(wrhost::TunerBatch *b) { _batch = b; }
	void* gpuStage() const { return _gpuStage; }
	void setGpuStage(void *p) { _gpuStage = p; }
	int gpuIndex() const { return _gpuIndex; }
	void setGpuIndex(int i) { _gpuIndex = i; }
	void setGpuCleanup(void (*fn)(DspSource*)) { _gpuCleanup = fn; }
	/* called at the top of run(), before the source's own process() may rewrite its block vector:
	 * the GPU glue waits there for an upload of the previous block that is still in flight */
	void setGpuBeforeRun(void (*fn)(DspSource*)) { _gpuBeforeRun = fn; }
	/* called at the top of stop(), before the block vector is released: the GPU glue lets go of
	 * whatever it holds on that memory (page locks) */
	void setGpuBeforeStop(void (*fn)(DspSource*)) { _gpuBeforeStop = fn; }
	/* the vector the source's process() filled for the current block */
	const vector<sample_t>& currentBlock() const { return _out; }
	/* false while the source left the vector unfilled for this block because every consumer reads the block on
	 * the device (consumersReadOnDevice): a consumer whose device path fails must then fail, not read it */
	bool hostBlockValid() const { return _hostBlockValid; }
protected:
	void setHostBlockValid(bool v) { _hostBlockValid = v; }

private:
	unsigned int		_blockSize;
	bool				_hostBlockValid;
	unsigned long		_epoch;
	vector<sample_t>	_pump;		/* the (zeroed) input vector handed to the source's own process() */
	wrhost::TunerBatch*	_batch;
	void*				_gpuStage;
	int					_gpuIndex;
	void				(*_gpuCleanup)(DspSource*);
	void				(*_gpuBeforeRun)(DspSource*);
	void				(*_gpuBeforeStop)(DspSource*);
};

#endif /* DSPBLOCK_H_ */
