/*
 * dspblock.cxx -- scheduling of the operator graph: connect / start / run / stop.
 *
 * Behaviour restated from webradio's src/dsp/dspblock.cxx (line numbers below refer to
 * it) and pinned by comparing scenario traces with the real thing
 * (tests/test_dspblock_parity.py):
 *   connect    :57-76   hot-connect starts the new consumer first, with whatever rates
 *                       it already had (quirk Q9); duplicates are refused
 *   disconnect :78-92   hot-disconnect stops the consumer
 *   start      :106-151 out := in, init(), integer rate ratio or fail+deinit, then
 *                       cascade rate/channels and start every consumer; any failure
 *                       stops this block's whole subtree
 *   stop       :153-167 consumers first, then deinit(), then drop the output buffer
 *   run        :169-212 size the output for in*interp/decim frames (truncating),
 *                       time process(), push the result to each consumer in order
 *
 * Attribution: connect/disconnect/start/stop/run follow mikestir/webradio's src/dsp/dspblock.cxx
 * (Copyright (C) Mike Stirling, AGPL-3.0) step for step, log lines included -- trace-identical
 * scheduling IS the contract of this row.  What is this backend's own: runFrames
 * and output elision, the device hand-over between blocks, the sampled profiler clock, the kept
 * pump vector and the GPU hooks of DspSource.
 */
#include <algorithm>
#include <inttypes.h>

#include <stdlib.h>

#include "debug.h"
#include "dspblock.h"

namespace {
/* WEBRADIO_WALL=1 (measurement only): every process() is bracketed -- not a sample of them -- with the MONOTONIC
 * clock, so that the profiler getters say where a run()'s wall time goes (the process-CPU clock also counts the HIP
 * runtime's own threads and does not see the time spent waiting for the GPU) */
const bool g_wall = getenv("WEBRADIO_WALL") && atoi(getenv("WEBRADIO_WALL")) != 0;

uint64_t cpuNanoseconds()
{
	/* the reference brackets process() with the process-CPU clock (dspblock.cxx:188) */
	timespec ts;
	clock_gettime(g_wall ? CLOCK_MONOTONIC : CLOCK_PROCESS_CPUTIME_ID, &ts);
	return (uint64_t)ts.tv_sec * 1000000000ULL + (uint64_t)ts.tv_nsec;
}
}

/* set by the GPU glue when the first chain is fused (a pointer, so that this file links without it) */
void (*DspBlock::gpuUnfuse)(DspBlock *) = NULL;

DspBlock::DspBlock(const string &name, const string &type)
	: _outputSampleRate(DEFAULT_SAMPLE_RATE), _outputChannels(DEFAULT_CHANNELS),
	  _name(name), _type(type),
	  _inRate(DEFAULT_SAMPLE_RATE), _inChannels(DEFAULT_CHANNELS),
	  _decim(1), _interp(1), _nsTotal(0), _framesIn(0), _framesOut(0), _calls(0),
	  _running(false), _elide(false), _curInFrames(0), _curOutFrames(0), _producer(NULL), _devOut(NULL)
{
}

DspBlock::~DspBlock()
{
	if (_running)
		stop();
}

void DspBlock::connect(DspBlock *block)
{
	/* a running producer starts the newcomer before it is hooked up (and before any rate
	 * has been propagated to it) */
	if (_running)
		block->start();

	const bool already = find(_consumers.begin(), _consumers.end(), block) != _consumers.end();
	if (already) {
		LOG_ERROR("Block %s:%s already connected to %s:%s\n", block->type().c_str(),
		          block->name().c_str(), type().c_str(), name().c_str());
		return;
	}
	/* a block inside a fused Receiver chain produces no host output: a second consumer ends the
	 * fusion of that chain (it goes on block by block, its filters from empty histories) */
	if (_elide && gpuUnfuse)
		gpuUnfuse(this);
	_consumers.push_back(block);
	block->_producer = this;
	LOG_DEBUG("Added block %s:%s as consumer of %s:%s\n", block->type().c_str(),
	          block->name().c_str(), type().c_str(), name().c_str());
}

void DspBlock::disconnect(DspBlock *block)
{
	if (_running)
		block->stop();

	_consumers.erase(remove(_consumers.begin(), _consumers.end(), block), _consumers.end());
	if (block->_producer == this)
		block->_producer = NULL;
	LOG_DEBUG("Removed block %s:%s as consumer of %s:%s\n", block->type().c_str(),
	          block->name().c_str(), type().c_str(), name().c_str());
}

uint64_t DspBlock::nsPerFrameAll() const
{
	uint64_t sum = nsPerFrameOne();
	LOG_DEBUG("%s:%s %" PRIu64 " ns/frame\n", type().c_str(), name().c_str(), sum);
	for (size_t n = 0; n < _consumers.size(); n++)
		sum += _consumers[n]->nsPerFrameAll();
	return sum;
}

bool DspBlock::start()
{
	/* pass-through defaults; init() may change the output side */
	_outputSampleRate = _inRate;
	_outputChannels = _inChannels;

	LOG_DEBUG("Starting block %s:%s\n", type().c_str(), name().c_str());
	if (!init()) {
		LOG_ERROR("Block %s:%s failed to initialise\n", type().c_str(), name().c_str());
		return false;
	}

	/* rates must be related by an integer factor, one way or the other */
	if (_inRate >= _outputSampleRate) {
		_decim = _inRate / _outputSampleRate;
		_interp = 1;
	} else {
		_decim = 1;
		_interp = _outputSampleRate / _inRate;
	}
	if (_inRate * _interp / _decim != _outputSampleRate) {
		LOG_ERROR("Sample rates must be integer related\n");
		deinit();
		return false;
	}

	_framesIn = _framesOut = 0;
	_nsTotal = 0;
	_running = true;

	for (size_t n = 0; n < _consumers.size(); n++) {
		DspBlock *c = _consumers[n];
		c->setSampleRate(_outputSampleRate);
		c->setChannels(_outputChannels);
		if (!c->start()) {
			LOG_ERROR("Downstream failed to start - aborting pipeline\n");
			stop();
			return false;
		}
	}
	return true;
}

void DspBlock::stop()
{
	for (size_t n = 0; n < _consumers.size(); n++)
		_consumers[n]->stop();

	if (_running) {
		LOG_DEBUG("Stopping block %s:%s\n", type().c_str(), name().c_str());
		_running = false;
		deinit();
	}
	vector<sample_t>().swap(_out);
}

bool DspBlock::run(const vector<sample_t> &inBuffer)
{
	return runFrames(inBuffer, _inChannels ? (unsigned int)(inBuffer.size() / _inChannels) : 0);
}

/* `inframes` is explicit because an upstream block with an elided output hands down an
 * empty vector that still stands for a block of that many frames. */
bool DspBlock::runFrames(const vector<sample_t> &inBuffer, unsigned int inframes)
{
	if (!_running) {
		LOG_ERROR("Pipeline not started\n");
		return false;
	}

	const unsigned int outframes = inframes * _interp / _decim;
	_curInFrames = inframes;
	_curOutFrames = outframes;
	const size_t want = _elide ? 0 : (size_t)outframes * _outputChannels;
	if (_out.size() != want) {
		LOG_DEBUG("Resizing %s:%s buffer to %u frames (%u channels)\n", type().c_str(),
		          name().c_str(), outframes, _outputChannels);
		_out.resize(want);
	}

	_devOut = NULL;
	/* The process-CPU clock the reference brackets every process() with (dspblock.cxx:186-204) is
	 * a system call: two per block and receiver, 2 560 per tuner block with 256 receivers -- half a
	 * millisecond, as much as the block's PCIe transfer.  A block whose output nobody looks at on
	 * the host does its work elsewhere (the tuner batch, or a kernel it only enqueues) and is
	 * booked as zero; the others are timed on their first 16 calls and on every 16th after that,
	 * weighted 16 -- the getters (nsPerFrame..., main.cxx:117-121) keep their meaning. */
	const bool timed = g_wall || (!_elide && (_calls < 16 || (_calls & 15) == 0));
	const uint64_t weight = (g_wall || _calls < 16) ? 1 : 16;
	++_calls;
	const uint64_t t0 = timed ? cpuNanoseconds() : 0;
	if (!process(inBuffer, _out)) {
		LOG_ERROR("Pipeline failed at block %s:%s\n", type().c_str(), name().c_str());
		return false;
	}
	if (timed)
		_nsTotal += (cpuNanoseconds() - t0) * weight;
	_framesIn += inframes;
	_framesOut += outframes;

	for (size_t n = 0; n < _consumers.size(); n++)
		if (!_consumers[n]->runFrames(_out, outframes))
			return false;
	return true;
}

/* do all consumers of this (source) block take its output from the staged device copy? */
bool DspBlock::consumersReadOnDevice() const
{
	if (_consumers.empty())
		return false;
	for (size_t n = 0; n < _consumers.size(); n++)
		if (!_consumers[n]->readsSourceOnDevice())
			return false;
	return true;
}

/* does anybody read this block's output on the host? */
bool DspBlock::hostOutputNeeded() const
{
	for (size_t n = 0; n < _consumers.size(); n++)
		if (!_consumers[n]->acceptsDeviceInput())
			return true;
	return false;
}

void DspBlock::setSampleRate(unsigned int rate)
{
	if (_running)
		return;
	LOG_DEBUG("Setting %s:%s input sample rate to %u\n", type().c_str(), name().c_str(), rate);
	_inRate = rate;
}

void DspBlock::setChannels(unsigned int channels)
{
	if (_running)
		return;
	LOG_DEBUG("Setting %s:%s input channel count to %u\n", type().c_str(), name().c_str(), channels);
	_inChannels = channels;
}

DspSource::DspSource(const string &name, const string &type)
	: DspBlock(name, type), _blockSize(DEFAULT_BLOCK_SIZE), _hostBlockValid(true), _epoch(0), _batch(NULL), _gpuStage(NULL),
	  _gpuIndex(-1), _gpuCleanup(NULL), _gpuBeforeRun(NULL), _gpuBeforeStop(NULL)
{
}

DspSource::~DspSource()
{
	/* consumers may still reference GPU resources while stopping: stop first */
	if (isRunning())
		stop();
	if (_gpuCleanup)
		_gpuCleanup(this);
}

/* The reference builds a fresh zero-filled vector of blockSize floats for every call
 * (dspblock.h:134); sources ignore its contents and only its length matters, so one
 * zeroed vector is kept instead of allocating 32 MB per block at 100 Msps. */
bool DspSource::run()
{
	if (_gpuBeforeRun)
		_gpuBeforeRun(this);
	if (_pump.size() != _blockSize)
		_pump.assign(_blockSize, 0.0f);
	++_epoch;
	return DspBlock::run(_pump);
}

void DspSource::stop()
{
	if (_gpuBeforeStop)
		_gpuBeforeStop(this);
	DspBlock::stop();
}

void DspSource::setBlockSize(unsigned int size)
{
	if (isRunning())
		return;
	LOG_DEBUG("Setting %s:%s source block size to %u\n", type().c_str(), name().c_str(), size);
	_blockSize = size;
}
