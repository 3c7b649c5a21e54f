/*
 * downconverter.h -- NCO + complex mixer block.  Public surface of webradio's
 * src/dsp/downconverter.h:36-46 (ctor, bandwidth/decimation forwarded to an embedded,
 * never-started LowPass, IF getter/setter).
 */
#ifndef DOWNCONVERTER_H_
#define DOWNCONVERTER_H_

#include <string>
#include <vector>

#include "dspblock.h"
#include "lowpass.h"

using namespace std;

namespace wrhost { class TunerBatch; struct Channel; struct DevBuf; }

class DownConverter : public DspBlock
{
	friend class wrhost::TunerBatch;
public:
	DownConverter(const string &name = "<undefined>");
	virtual ~DownConverter();

	/* upstream forwards these four to a LowPass member that is constructed but never
	 * connected, started or run (downconverter.cxx:44, downconverter.h:41-44) */
	unsigned int bandwidth() const { return _unusedFilter->passband(); }
	void setBandwidth(unsigned int hz) { _unusedFilter->setPassband(hz); }
	unsigned int decimation() const { return _unusedFilter->decimation(); }
	void setDecimation(unsigned int n) { _unusedFilter->setDecimation(n); }

	int IF() const { return _ifHz; }
	void setIF(int hz);

private:
	bool init();
	void deinit();
	bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer);
	bool acceptsDeviceInput() const { return _channel == NULL; }
	bool readsSourceOnDevice() const { return true; }            /* fused: the tuner batch; stand-alone: stagedBlock */   /* stand-alone: reads the producer's device output */
	wrhost::Channel *gpuChannel() const { return _channel; }

	LowPass*		_unusedFilter;
	int				_ifHz;
	unsigned int	_phase;			/* survives stop()/start(), like upstream's (quirk Q5) */
	int				_phaseStep;
	wrhost::Channel*	_channel;	/* non-NULL while enrolled in a TunerBatch */
	wrhost::DevBuf*	_in;
	wrhost::DevBuf*	_out;
};

#endif /* DOWNCONVERTER_H_ */
