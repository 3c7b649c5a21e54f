/*
 * tuner.h -- a SampleSource with a centre frequency, ppm offset, AGC and gain.
 * Same public/protected surface and defaults as webradio's src/io/tuner.h:36-77 (the web
 * handlers and FrontEnd call these; hardware tuners override the virtual setters).
 *
 * Attribution: the class DECLARATION below -- member names, signatures, default values -- is
 * that of mikestir/webradio's src/io/tuner.h (Copyright (C) Mike Stirling, AGPL-3.0), kept
 * token-compatible on purpose: the reference's radio.cxx, tunerhandler.cxx and
 * tunercontrolhandler.cxx must compile against this header unchanged.  Nothing but
 * the interface is taken; there is no implementation in it to take.
 */
#ifndef TUNER_H_
#define TUNER_H_

#include <string>

#include "samplesource.h"

using namespace std;

#define DEFAULT_TUNER_SAMPLE_RATE	1200000
#define DEFAULT_TUNER_CHANNELS		2

class Tuner : public SampleSource
{
public:
	Tuner(const string &name = "<undefined>", const string &type = "Tuner")
		: SampleSource(name, type),
		  _centreFrequency(100000000), _offsetPPM(0), _AGC(true), _gainDB(0) {}
	virtual ~Tuner() {}

	/* note: shadows DspBlock::name() with the device name, as upstream does */
	const string& name() const { return _name; }
	const string& manufacturer() const { return _manufacturer; }
	const string& product() const { return _product; }
	const string& serial() const { return _serial; }

	unsigned int centreFrequency() const { return _centreFrequency; }
	int offsetPPM() const { return _offsetPPM; }
	bool AGC() const { return _AGC; }
	virtual float gainDB() const { return _gainDB; }

	virtual void setCentreFrequency(unsigned int hz) {}
	virtual void setOffsetPPM(int ppm) {}
	virtual void setAGC(bool agc) {}
	virtual void setGainDB(float gain) {}

protected:
	string			_name;
	string			_manufacturer;
	string			_product;
	string			_serial;

	unsigned int	_centreFrequency;
	int				_offsetPPM;
	bool			_AGC;
	float			_gainDB;
};

typedef Tuner* (*TunerFactory)(const string&);

#endif /* TUNER_H_ */
