/*
 * radio.h -- application wiring: FrontEnd (tuner + spectrum) and Receiver
 * (DownConverter -> LowPass -> Demodulator -> LowPass -> AudioStreamManager), plus the
 * Radio namespace that pumps every front end.  Same public surface as webradio's
 * src/radio.h:42-102, so main.cxx and the web handlers build against it.
 * (The reference's own radio.cxx also compiles unchanged against the headers of this
 * directory -- tests/test_boundary.py proves it -- this is the backend's own copy of
 * the wiring so that it can be used without the reference tree.)
 *
 * Attribution: the class declarations (FrontEnd, Receiver, namespace Radio: member names,
 * signatures, order) are those of mikestir/webradio's src/radio.h (Copyright (C) Mike Stirling,
 * AGPL-3.0), kept source-compatible on purpose so that main.cxx and src/web compile against
 * this header unchanged.
 */
#ifndef RADIO_H_
#define RADIO_H_

#include <stdint.h>

#include <map>
#include <string>

#include "tuner.h"
#include "spectrumsink.h"
#include "lowpass.h"
#include "downconverter.h"
#include "demodulator.h"
#include "audiostream.h"

class FrontEnd;

class Receiver {
public:
	Receiver();
	~Receiver();

	FrontEnd* frontEnd() { return _frontEnd; }
	void setFrontEnd(FrontEnd *frontend);

	DspBlock* input() { return _downconverter; }

	DownConverter* downconverter() { return _downconverter; }
	LowPass* channelFilter() { return _channelFilter; }
	Demodulator* demodulator() { return _demodulator; }
	LowPass* audioFilter() { return _audioFilter; }
	AudioStreamManager* stream() { return _stream; }

	const string& uuid() { return _uuid; }

private:
	DownConverter*		_downconverter;
	LowPass*			_channelFilter;
	Demodulator*		_demodulator;
	LowPass*			_audioFilter;
	AudioStreamManager*	_stream;
	string				_uuid;
	FrontEnd*			_frontEnd;
};

class FrontEnd {
public:
	friend class Receiver;

	FrontEnd(TunerFactory factory);
	~FrontEnd();

	Tuner* tuner() { return _tuner; }
	SpectrumSink* spectrum() { return _spectrum; }

	const string& uuid() const { return _uuid; }
	const map<string, Receiver*>& receivers() const { return _receivers; }

private:
	void addReceiver(Receiver *rx);
	void removeReceiver(Receiver *rx);

	Tuner*					_tuner;
	SpectrumSink*			_spectrum;
	string					_uuid;
	map<string, Receiver*>	_receivers;
};

namespace Radio {
	const map<string, FrontEnd*>& frontEnds();
	const map<string, Receiver*>& receivers();
	void profile();
	void run();
}

#endif /* RADIO_H_ */
