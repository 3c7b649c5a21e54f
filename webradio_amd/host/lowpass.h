/*
 * lowpass.h -- 64-tap decimating FIR low-pass block.  Public surface of webradio's
 * src/dsp/lowpass.h:36-44.  (No <fftw3.h>: the taps come from wr_lowpass_design.)
 */
#ifndef FILTER_H_
#define FILTER_H_

#include <string>
#include <vector>

#include <mutex>

#include "dspblock.h"

using namespace std;

namespace wrhost { class TunerBatch; struct Channel; struct DevBuf; }

class LowPass : public DspBlock
{
	friend class wrhost::TunerBatch;
public:
	LowPass(const string &name = "<undefined>");
	virtual ~LowPass();

	unsigned int passband() const { return _passband; }
	void setPassband(unsigned int hz);
	void setDecimation(unsigned int n);
	void setOutputSampleRate(unsigned int hz);
	/* The reference compiles the FIR length in ("FIXME: Make runtime variable",
	 * dsp/lowpass.cxx:38-39) although init()/recalculate()/process() are written in terms of
	 * _firLength.  Extension: a power of two in [2, 1024], default 64, ignored while running.
	 * A chain whose filters are not both 64 taps long runs block by block instead of inside the
	 * fused tuner batch. */
	void setFirLength(unsigned int n);
	unsigned int firLength() const { return _firLength; }

private:
	bool init();
	void deinit();
	bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer);
	bool acceptsDeviceInput() const { return _channel == NULL; }   /* stand-alone: reads the producer's device output */
	wrhost::Channel *gpuChannel() const { return _channel; }
	void recalculate();

	unsigned int	_firLength;
	unsigned int	_passband;
	unsigned int	_reqDecimation;
	unsigned int	_reqOutputRate;
	vector<float>	_coeff;			/* _firLength taps, lowpass.cxx:183-189 */
	std::mutex		_coeffLock;		/* setPassband() arrives on HTTP threads (httpserver.cxx:262): a new
									 * design is swapped in under this lock, readers copy under it */
	wrhost::Channel*	_channel;
	int				_stage;			/* 0 channel filter, 1 audio filter of an enrolled chain */
	wrhost::DevBuf*	_in;
	wrhost::DevBuf*	_out;
	wrhost::DevBuf*	_history;
};

#endif /* FILTER_H_ */
