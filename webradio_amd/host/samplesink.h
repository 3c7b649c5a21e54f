/*
 * samplesink.h -- base class of terminal blocks (audio sinks, SpectrumSink).
 * Same public surface as webradio's src/io/samplesink.h:34-57: a DspBlock that carries
 * a `subdevice` string which can only be changed while stopped.
 */
#ifndef SAMPLESINK_H_
#define SAMPLESINK_H_

#include <string>
#include <vector>

#include "dspblock.h"

using namespace std;

class SampleSink : public DspBlock
{
public:
	SampleSink(const string &name = "<undefined>", const string &type = "SampleSink")
		: DspBlock(name, type) {}
	virtual ~SampleSink() {}

	const string& subdevice() const { return _chosen; }
	const vector<string>& subdevices() const { return _subdevices; }
	void setSubdevice(const string &subdevice) {
		if (!isRunning())
			_chosen = subdevice;
	}

protected:
	vector<string> _subdevices;		/* filled by subclasses that can enumerate */
private:
	string _chosen;
};

#endif /* SAMPLESINK_H_ */
