/*
 * spectrumsink.cxx -- host side of the spectrum sink (webradio src/io/spectrumsink.cxx):
 * frames accumulate on the GPU, the most recent complete frame is windowed and
 * transformed there, and getSpectrum() fetches dB values on demand.
 */
#include "spectrumsink.h"

#include "debug.h"
#include "gpubatch.h"

SpectrumSink::SpectrumSink(const string &name)
	: SampleSink(name, "SpectrumSink"), _fftSize(DEFAULT_FFT_SIZE), _hop(0), _spec(NULL), _dev(NULL)
{
}

SpectrumSink::~SpectrumSink()
{
	if (_spec)
		wr_spectrum_destroy(_spec);
}

void SpectrumSink::setFftSize(unsigned int size)
{
	if (isRunning())
		return;
	if (size & (size - 1)) {
		LOG_ERROR("size must be a power of 2\n");
		return;
	}
	_fftSize = size;
}

void SpectrumSink::setHop(unsigned int hop)
{
	if (!isRunning())
		_hop = hop;
}

bool SpectrumSink::init()
{
	std::lock_guard<std::mutex> g(_lock);
	wr_dev *dev = wrhost::deviceFor(this);
	if (!dev)
		return false;
	_dev = dev;
	if (wr_spectrum_create(&_spec, dev, _fftSize, _hop) != WR_OK) {
		LOG_ERROR("SpectrumSink: %s\n", wr_last_error());
		_spec = NULL;
		return false;
	}
	return true;
}

void SpectrumSink::deinit()
{
	std::lock_guard<std::mutex> g(_lock);
	if (_spec)
		wr_spectrum_destroy(_spec);
	_spec = NULL;
}

bool SpectrumSink::process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer)
{
	/* like upstream this assumes IQ input (spectrumsink.cxx:93) */
	std::lock_guard<std::mutex> g(_lock);
	if (!_spec)
		return false;
	/* the receivers of the same tuner go first on the device's stream: their audio is what run() waits for */
	wrhost::submitBatchFirst(this, inBuffer);
	/* fed straight from the tuner: use the device copy every GPU consumer of it shares */
	/* ... of which the transform reads the most recent complete frame and keeps what follows it (wr_spectrum_push): the
	 * last fftSize + hop frames are all that has to be there */
	wr_dev *sdev = NULL;
	const size_t nframes = inBuffer.size() / 2, tail = (size_t)_fftSize + (_hop ? _hop : _fftSize);
	const float *staged = nframes > 2 * tail ? wrhost::stagedTail(this, inBuffer, &sdev, tail) : NULL;
	if (!(staged && sdev == _dev))
		staged = wrhost::stagedBlock(this, inBuffer, &sdev);
	if (!(staged && sdev == _dev) && !wrhost::hostBlockValid(this)) {
		LOG_ERROR("SpectrumSink: the source left its block on the device and the device copy is not there\n");
		return false;
	}
	int rc = (staged && sdev == _dev) ? wr_spectrum_push(_spec, staged, inBuffer.size() / 2, WR_DEVICE)
	                                  : wr_spectrum_push(_spec, inBuffer.data(), inBuffer.size() / 2, WR_HOST);
	if (rc != WR_OK) {
		LOG_ERROR("SpectrumSink: %s\n", wr_last_error());
		return false;
	}
	return true;
}

/* Before the first complete frame upstream hands out uninitialised memory (quirk Q8);
 * here the caller's array is left untouched in that case. */
void SpectrumSink::getSpectrum(float *magnitudes)
{
	std::lock_guard<std::mutex> g(_lock);
	if (_spec)
		wr_spectrum_get_db(_spec, magnitudes);
}
