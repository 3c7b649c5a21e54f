/*
 * demodulator.h -- IQ -> mono detector (AM / FM / USB / LSB).  Public surface of
 * webradio's src/dsp/demodulator.h:34-51, including the block type string "AMDemod".
 */
#ifndef DEMODULATOR_H_
#define DEMODULATOR_H_

#include <string>
#include <vector>

#include "dspblock.h"

using namespace std;

namespace wrhost { class TunerBatch; struct Channel; struct DevBuf; }

class Demodulator : public DspBlock
{
	friend class wrhost::TunerBatch;
public:
	Demodulator(const string &name = "<undefined>");
	virtual ~Demodulator();

	enum Mode {
		AM,
		FM,
		USB,
		LSB,
		MAX_MODE
	};

	const Mode mode() const { return _mode; }
	void setMode(const Mode mode);
	const string& modeString() const { return _modeNames[_mode]; }
	bool setModeString(const string &mode);

private:
	bool init();
	void deinit();
	bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer);
	bool acceptsDeviceInput() const { return _channel == NULL; }   /* stand-alone: reads the producer's device output */
	wrhost::Channel *gpuChannel() const { return _channel; }

	Mode			_mode;
	vector<string>	_modeNames;
	float			_prev[2];		/* prev_i, prev_q: survive stop()/start() (quirk Q5) */
	wrhost::Channel*	_channel;
	wrhost::DevBuf*	_in;
	wrhost::DevBuf*	_out;
};

#endif /* DEMODULATOR_H_ */
