/*
 * lowpass.cxx -- host side of the decimating FIR block (webradio src/dsp/lowpass.cxx; 64 taps
 * unless setFirLength says otherwise).  Tap design is one-off host work (wr_lowpass_design_n
 * restates lowpass.cxx:164-197); the filtering runs on the GPU, fused into the tuner batch
 * (64 taps) or as wr_fir_decimate_n.
 */
#include "lowpass.h"

#include "debug.h"
#include "gpubatch.h"

LowPass::LowPass(const string &name)
	: DspBlock(name, "LowPass"), _firLength(WR_FIR_LENGTH), _passband(0), _reqDecimation(0), _reqOutputRate(DEFAULT_SAMPLE_RATE),
	  _channel(NULL), _stage(-1), _in(new wrhost::DevBuf()), _out(new wrhost::DevBuf()),
	  _history(new wrhost::DevBuf())
{
}

LowPass::~LowPass()
{
	delete _in;
	delete _out;
	delete _history;
}

void LowPass::setPassband(unsigned int hz)
{
	_passband = hz;
	if (isRunning())
		recalculate();
}

void LowPass::setFirLength(unsigned int n)
{
	if (isRunning())
		return;
	if (n < 2 || n > WR_FIR_MAX || (n & (n - 1)) != 0) {
		LOG_ERROR("FIR length %u is not a power of two in [2, %d]\n", n, WR_FIR_MAX);
		return;
	}
	_firLength = n;
}

/* asking for a decimation cancels a requested rate and vice versa; both are ignored while
 * running (lowpass.cxx:63-79) */
void LowPass::setDecimation(unsigned int n)
{
	if (isRunning())
		return;
	_reqDecimation = n;
	_reqOutputRate = 0;
}

void LowPass::setOutputSampleRate(unsigned int hz)
{
	if (isRunning())
		return;
	_reqOutputRate = hz;
	_reqDecimation = 0;
}

bool LowPass::init()
{
	if (_reqOutputRate > 0) {
		_outputSampleRate = _reqOutputRate;
	} else if (_reqDecimation > 0) {
		_outputSampleRate = inputSampleRate() / _reqDecimation;
	} else {
		LOG_ERROR("Must specify either decimation or output rate\n");
		return false;
	}
	_outputChannels = inputChannels();
	recalculate();
	if (!_channel) {
		/* stand-alone: an empty history, as a fresh LowPass::block (lowpass.cxx:138-139) */
		wr_dev *dev = wrhost::deviceFor(this);
		const size_t bytes = (size_t)(_firLength - 1) * inputChannels() * sizeof(float);
		if (!dev)
			return false;
		_history->release();             /* wr_dev_malloc zero-fills */
		if (!_history->reserve(dev, bytes))
			return false;
	}
	return true;
}

void LowPass::deinit()
{
	/* the chain's channel is withdrawn by its DownConverter */
	_in->release();
	_out->release();
	_history->release();
	vector<float>().swap(_coeff);
}

void LowPass::recalculate()
{
	/* designed aside and swapped in whole: the run thread never sees half-written taps */
	vector<float> fresh(_firLength);
	wr_lowpass_design_n(_firLength, _passband, inputSampleRate(), fresh.data(), NULL);
	{
		std::lock_guard<std::mutex> g(_coeffLock);
		_coeff.swap(fresh);
	}
	wrhost::TunerBatch::markDirty(_channel);
}

bool LowPass::process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer)
{
	if (_channel) {
		if (_stage != 1)
			return true;                 /* a channel filter (first or second stage): computed inside the batch */
		return _channel->batch->audio(_channel, outBuffer);   /* audio filter: one slice of the batch's transfer */
	}
	const unsigned int ch = inputChannels();
	const unsigned int nframes = currentInputFrames();
	const unsigned int outframes = currentOutputFrames();
	wr_dev *dev = wrhost::deviceFor(this);
	const size_t inBytes = (size_t)nframes * ch * sizeof(float);
	const size_t outBytes = (size_t)outframes * outputChannels() * sizeof(float);
	if (!dev || !_out->reserve(dev, outBytes ? outBytes : sizeof(float)))
		return false;
	if (!_history->ptr) {
		/* started as part of a fused chain that has since been taken apart (a second consumer was
		 * attached inside it, DspBlock::connect): an empty history, like a fresh LowPass::block */
		const size_t hbytes = (size_t)(_firLength - 1) * ch * sizeof(float);
		if (!_history->reserve(dev, hbytes ? hbytes : sizeof(float)))
			return false;
	}
	/* input: the producer's device output if there is one, else an upload (see DownConverter) */
	const float *din = (const float *)upstreamDeviceOutput();
	if (!din) {
		if (!_in->reserve(dev, inBytes ? inBytes : sizeof(float)) ||
		    wr_dev_upload(dev, _in->ptr, inBuffer.data(), inBytes) != WR_OK) {
			LOG_ERROR("LowPass: %s\n", wr_last_error());
			return false;
		}
		din = (const float *)_in->ptr;
	}
	vector<float> taps;
	{
		std::lock_guard<std::mutex> g(_coeffLock);
		taps = _coeff;
	}
	if (taps.size() != _firLength ||
	    wr_fir_decimate_n(dev, din, nframes, ch, decimation(), _firLength, taps.data(),
	                      (float *)_history->ptr, (float *)_out->ptr) != WR_OK) {
		LOG_ERROR("LowPass: %s\n", wr_last_error());
		return false;
	}
	publishDeviceOutput(_out->ptr);
	const bool onHost = hostOutputNeeded();
	elideOutput(!onHost);
	if (onHost) {
		outBuffer.resize((size_t)outframes * outputChannels());
		if (wr_dev_download(dev, outBuffer.data(), _out->ptr, outBytes) != WR_OK) {
			LOG_ERROR("LowPass: %s\n", wr_last_error());
			return false;
		}
	}
	return true;
}
