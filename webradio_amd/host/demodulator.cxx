/*
 * demodulator.cxx -- host side of the AM/FM/USB/LSB detector (webradio
 * src/dsp/demodulator.cxx).  Fused into the tuner batch, or wr_demod stand-alone.
 */
#include "demodulator.h"

#include "debug.h"
#include "gpubatch.h"

Demodulator::Demodulator(const string &name)
	: DspBlock(name, "AMDemod"), _mode(AM), _channel(NULL), _in(new wrhost::DevBuf()),
	  _out(new wrhost::DevBuf())
{
	_prev[0] = _prev[1] = 0.0f;
	/* index = enum value (demodulator.cxx:37-41) */
	const char *names[] = { "AM", "FM", "USB", "LSB" };
	for (int n = 0; n < (int)MAX_MODE; n++)
		_modeNames.push_back(names[n]);
}

Demodulator::~Demodulator()
{
	delete _in;
	delete _out;
}

void Demodulator::setMode(const Mode mode)
{
	_mode = mode;
	wrhost::TunerBatch::markDirty(_channel);
}

bool Demodulator::setModeString(const string &mode)
{
	for (size_t n = 0; n < _modeNames.size(); n++)
		if (_modeNames[n] == mode) {
			setMode((Mode)n);
			return true;
		}
	return false;
}

bool Demodulator::init()
{
	if (inputChannels() != 2) {
		LOG_ERROR("Expect IQ input\n");
		return false;
	}
	_outputSampleRate = inputSampleRate();
	_outputChannels = 1;
	if (!_channel && !wrhost::deviceFor(this))
		return false;
	return true;
}

void Demodulator::deinit()
{
	_in->release();
	_out->release();
}

bool Demodulator::process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer)
{
	if (_channel)
		return true;                     /* computed inside the batch */
	if ((int)_mode < 0 || _mode >= MAX_MODE) {
		LOG_ERROR("Bad mode\n");
		return false;
	}
	const unsigned int nframes = currentInputFrames();
	wr_dev *dev = wrhost::deviceFor(this);
	const size_t inBytes = (size_t)nframes * 2 * sizeof(float), outBytes = (size_t)nframes * sizeof(float);
	if (!dev || !_out->reserve(dev, outBytes ? outBytes : sizeof(float)))
		return false;
	const float *din = (const float *)upstreamDeviceOutput();
	if (!din) {
		if (!_in->reserve(dev, inBytes ? inBytes : sizeof(float)) ||
		    wr_dev_upload(dev, _in->ptr, inBuffer.data(), inBytes) != WR_OK) {
			LOG_ERROR("Demodulator: %s\n", wr_last_error());
			return false;
		}
		din = (const float *)_in->ptr;
	}
	if (wr_demod(dev, (int)_mode, din, nframes, _prev, (float *)_out->ptr) != WR_OK) {
		LOG_ERROR("Demodulator: %s\n", wr_last_error());
		return false;
	}
	publishDeviceOutput(_out->ptr);
	const bool onHost = hostOutputNeeded();
	elideOutput(!onHost);
	if (onHost) {
		outBuffer.resize(nframes);
		if (wr_dev_download(dev, outBuffer.data(), _out->ptr, outBytes) != WR_OK) {
			LOG_ERROR("Demodulator: %s\n", wr_last_error());
			return false;
		}
	}
	return true;
}
