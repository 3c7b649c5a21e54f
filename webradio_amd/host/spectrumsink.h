/*
 * spectrumsink.h -- windowed-FFT spectrum sink.  Public surface of webradio's
 * src/io/spectrumsink.h:44-53 (fftSize/setFftSize/getSpectrum, default size 512).
 */
#ifndef SPECTRUMSINK_H_
#define SPECTRUMSINK_H_

#include <mutex>
#include <string>
#include <vector>

#include "samplesink.h"
#include "webradio_amd.h"

#define DEFAULT_FFT_SIZE		512

using namespace std;

class SpectrumSink : public SampleSink
{
public:
	SpectrumSink(const string &name = "<undefined>");
	virtual ~SpectrumSink();

	unsigned int fftSize() const { return _fftSize; }
	void setFftSize(unsigned int size);
	/* extension: frames start every `hop` input frames (0 = back to back, the upstream
	 * behaviour; fftSize/2 = the 50 % overlap waterfall) -- ignored while running */
	void setHop(unsigned int hop);

	void getSpectrum(float *magnitudes);

	bool readsSourceOnDevice() const { return true; }           /* the staged device copy of the tuner block */

private:
	bool init();
	void deinit();
	bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer);

	unsigned int	_fftSize;
	unsigned int	_hop;
	wr_spectrum*	_spec;
	wr_dev*			_dev;
	std::mutex		_lock;		/* getSpectrum comes from HTTP threads (waterfallhandler.cxx:56-57) */
};

#endif /* SPECTRUMSINK_H_ */
