/*
 * audiostream.h -- the terminal audio sink of a Receiver.
 *
 * Upstream's AudioStreamManager (src/web/audiostream.{h,cxx}) encodes MP3 with LAME and
 * feeds per-client pipes of the HTTP server; codec and sockets are outside the DSP hot
 * path and are not rebuilt here.  This class keeps the name, the base class and the
 * constructor radio.cxx uses (radio.cxx:72,82) and simply retains the audio it is
 * given, so a Receiver built from this tree ends in a sink a caller (a test, a
 * recorder, an encoder thread) can drain.  Link the real web/audiostream.cxx instead to
 * get upstream's MP3 streaming.
 */
#ifndef AUDIOSTREAM_H_
#define AUDIOSTREAM_H_

#include <string>
#include <vector>

#include "samplesink.h"

using namespace std;

class AudioStreamManager : public SampleSink
{
public:
	AudioStreamManager(const string &name = "<undefined>")
		: SampleSink(name, "AudioStreamManager"), _keep(1u << 20), _total(0) {}
	virtual ~AudioStreamManager() {}

	/* audio received since start (the last `capacity` samples are retained, at times up to twice as many) */
	const vector<float>& samples() const { return _samples; }
	unsigned long totalSamples() const { return _total; }
	void setCapacity(size_t samples) { _keep = samples; }
	void clear() { _samples.clear(); }

protected:
	bool init() { _samples.clear(); _total = 0; return true; }
	void deinit() {}
	bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer) {
		_total += inBuffer.size();
		if (inBuffer.size() >= _keep) {                 /* a block longer than what is kept: its tail is all there is to keep */
			_samples.assign(inBuffer.end() - _keep, inBuffer.end());
			return true;
		}
		_samples.insert(_samples.end(), inBuffer.begin(), inBuffer.end());
		/* trim in large steps: dropping the front of a vector moves everything behind it, and
		 * doing that for every block of a long stream would cost more than the DSP */
		if (_samples.size() > 2 * _keep)
			_samples.erase(_samples.begin(), _samples.begin() + (_samples.size() - _keep));
		return true;
	}

private:
	vector<float>	_samples;
	size_t			_keep;
	unsigned long	_total;
};

#endif /* AUDIOSTREAM_H_ */
