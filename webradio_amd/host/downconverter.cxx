/*
 * downconverter.cxx -- host side of the NCO/mixer block (webradio src/dsp/downconverter.cxx).
 * The arithmetic is in the HIP library: fused into the tuner batch when this block heads a
 * Receiver chain, otherwise wr_mix on this block's own buffers.
 */
#include "downconverter.h"

#include "debug.h"
#include "gpubatch.h"

DownConverter::DownConverter(const string &name)
	: DspBlock(name, "DownConverter"), _unusedFilter(new LowPass(name)), _ifHz(0), _phase(0),
	  _phaseStep(0), _channel(NULL), _in(new wrhost::DevBuf()), _out(new wrhost::DevBuf())
{
	/* the 65536-entry sine table upstream builds here (downconverter.cxx:49-51) lives once
	 * per GPU inside wr_dev */
}

DownConverter::~DownConverter()
{
	if (_channel)
		wrhost::TunerBatch::withdraw(_channel);
	delete _unusedFilter;
	delete _in;
	delete _out;
}

void DownConverter::setIF(int hz)
{
	_ifHz = hz;
	/* upstream only recomputes the step while running (downconverter.cxx:63-66); init()
	 * computes it otherwise */
	if (isRunning()) {
		wr_phase_step(hz, inputSampleRate(), &_phaseStep);
		wrhost::TunerBatch::markDirty(_channel);
	}
}

bool DownConverter::init()
{
	if (inputChannels() != 2) {
		LOG_ERROR("Expect IQ input\n");
		return false;
	}
	_outputSampleRate = inputSampleRate();
	_outputChannels = inputChannels();
	if (wr_phase_step(_ifHz, inputSampleRate(), &_phaseStep) != WR_OK)
		return false;
	LOG_DEBUG("phaseStep = %d for %d Hz\n", _phaseStep, _ifHz);

	_channel = wrhost::TunerBatch::enrol(this);
	if (!_channel && !wrhost::deviceFor(this))
		return false;                   /* no GPU, no CPU path */
	return true;
}

void DownConverter::deinit()
{
	if (_channel) {
		wrhost::TunerBatch::withdraw(_channel);
		_channel = NULL;
	}
	_in->release();
	_out->release();
}

bool DownConverter::process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer)
{
	const unsigned int nframes = currentInputFrames();
	if (_channel) {
		/* the whole tuner -- every enrolled receiver -- goes to the GPU on the first of
		 * these calls per source block */
		return _channel->batch->submitOnce(inBuffer, nframes);
	}
	/* stand-alone (not the fused Receiver shape): one kernel for this block.  The input is, in
	 * order of preference, the producer's output where it already lies in device memory, the
	 * device copy of the tuner block every GPU consumer of the source shares, or an upload;
	 * the output stays on the device and only reaches the host if a consumer needs it there. */
	wr_dev *dev = wrhost::deviceFor(this);
	const size_t bytes = (size_t)nframes * 2 * sizeof(float);
	if (!dev || !_out->reserve(dev, bytes))
		return false;
	wr_dev *sdev = NULL;
	const float *din = (const float *)upstreamDeviceOutput();
	if (!din) {
		din = wrhost::stagedBlock(this, inBuffer, &sdev);
		if (din && sdev != dev)
			din = NULL;
	}
	if (!din) {
		if (!wrhost::hostBlockValid(this)) {
			LOG_ERROR("DownConverter: the source left its block on the device and the device copy is not there\n");
			return false;
		}
		if (!_in->reserve(dev, bytes) || wr_dev_upload(dev, _in->ptr, inBuffer.data(), bytes) != WR_OK) {
			LOG_ERROR("DownConverter: %s\n", wr_last_error());
			return false;
		}
		din = (const float *)_in->ptr;
	}
	if (wr_mix(dev, din, (float *)_out->ptr, nframes, &_phase, _phaseStep) != WR_OK) {
		LOG_ERROR("DownConverter: %s\n", wr_last_error());
		return false;
	}
	publishDeviceOutput(_out->ptr);
	const bool onHost = hostOutputNeeded();
	elideOutput(!onHost);               /* from the next block on the runtime does not size it either */
	if (onHost) {
		outBuffer.resize((size_t)nframes * 2);
		if (wr_dev_download(dev, outBuffer.data(), _out->ptr, bytes) != WR_OK) {
			LOG_ERROR("DownConverter: %s\n", wr_last_error());
			return false;
		}
	}
	return true;
}
