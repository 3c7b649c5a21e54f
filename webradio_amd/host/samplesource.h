/*
 * samplesource.h -- base class of pumping blocks (tuners, file and synthetic sources).
 * Same public surface as webradio's src/io/samplesource.h:40-64.
 */
#ifndef SAMPLESOURCE_H_
#define SAMPLESOURCE_H_

#include <string>
#include <vector>

#include "dspblock.h"

using namespace std;

class SampleSource : public DspSource
{
public:
	SampleSource(const string &name = "<undefined>", const string &type = "SampleSource")
		: DspSource(name, type) {}
	virtual ~SampleSource() {}

	const string& subdevice() const { return _chosen; }
	const vector<string>& subdevices() const { return _subdevices; }
	void setSubdevice(const string &subdevice) {
		if (!isRunning())
			_chosen = subdevice;
	}

protected:
	vector<string> _subdevices;
private:
	string _chosen;
};

#endif /* SAMPLESOURCE_H_ */
