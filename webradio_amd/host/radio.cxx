/*
 * radio.cxx -- the wiring behind radio.h (behaviour of webradio's src/radio.cxx:35-163).
 *
 *   ids        "%04X" of the number of objects of that kind alive at creation
 *   Receiver   five blocks chained in order; channel filter 80 kHz -> 240 kHz, audio filter
 *              8 kHz -> 48 kHz, AM, audio sink subdevice = the receiver id; attaching to a
 *              front end = connecting the mixer to that tuner
 *   FrontEnd   tuner from the factory + a SpectrumSink fed by it
 *   Radio      run() pumps each front end's tuner once; profile() logs ns/frame per block
 *
 * Attribution: this file restates, statement for statement where the contract fixes the behaviour
 * (object ids, default rates, log lines the handlers' users see), mikestir/webradio's
 * src/radio.cxx (Copyright (C) Mike Stirling, AGPL-3.0).  It exists only so that the backend can be
 * used without the reference tree; the reference's own radio.cxx runs unchanged on these blocks
 * (tests/test_gpu_host.py::test_reference_radio_cxx_on_our_blocks) and is the one to prefer.
 */
#include "radio.h"

#include <stdio.h>

#include "debug.h"

namespace {

map<string, FrontEnd*> &frontEndRegistry()
{
	static map<string, FrontEnd*> reg;
	return reg;
}

map<string, Receiver*> &receiverRegistry()
{
	static map<string, Receiver*> reg;
	return reg;
}

string hexId(size_t n)
{
	char text[8];
	snprintf(text, sizeof(text), "%04X", (unsigned int)(n & 0xFFFF));
	return text;
}

} // namespace

const map<string, FrontEnd*>& Radio::frontEnds() { return frontEndRegistry(); }
const map<string, Receiver*>& Radio::receivers() { return receiverRegistry(); }

void Radio::profile()
{
	map<string, FrontEnd*> &fes = frontEndRegistry();
	for (map<string, FrontEnd*>::iterator fe = fes.begin(); fe != fes.end(); ++fe)
		fe->second->tuner()->nsPerFrameAll();
}

void Radio::run()
{
	map<string, FrontEnd*> &fes = frontEndRegistry();
	for (map<string, FrontEnd*>::iterator fe = fes.begin(); fe != fes.end(); ++fe)
		fe->second->tuner()->run();		/* result ignored, as upstream does */
}

Receiver::Receiver() : _frontEnd(NULL)
{
	_uuid = hexId(receiverRegistry().size());

	_downconverter = new DownConverter(_uuid);
	_channelFilter = new LowPass(_uuid);
	_demodulator = new Demodulator(_uuid);
	_audioFilter = new LowPass(_uuid);
	_stream = new AudioStreamManager(_uuid);

	DspBlock *chain[] = { _downconverter, _channelFilter, _demodulator, _audioFilter, _stream };
	for (size_t n = 0; n + 1 < sizeof(chain) / sizeof(chain[0]); n++)
		chain[n]->connect(chain[n + 1]);

	_channelFilter->setPassband(80000);
	_channelFilter->setOutputSampleRate(240000);
	_audioFilter->setPassband(8000);
	_audioFilter->setOutputSampleRate(48000);
	_demodulator->setMode(Demodulator::AM);
	_stream->setSubdevice(_uuid);

	receiverRegistry()[_uuid] = this;
	LOG_DEBUG("receiver %s: chain built (DownConverter, channel LowPass, Demodulator, audio LowPass, sink)\n", _uuid.c_str());
}

Receiver::~Receiver()
{
	if (_frontEnd)
		setFrontEnd(NULL);
	receiverRegistry().erase(_uuid);
	delete _downconverter;
	delete _channelFilter;
	delete _demodulator;
	delete _audioFilter;
	delete _stream;
	LOG_DEBUG("receiver %s: chain released\n", _uuid.c_str());
}

void Receiver::setFrontEnd(FrontEnd *frontend)
{
	if (_frontEnd)
		_frontEnd->removeReceiver(this);
	_frontEnd = frontend;
	if (_frontEnd)
		_frontEnd->addReceiver(this);
}

FrontEnd::FrontEnd(TunerFactory factory)
{
	_uuid = hexId(frontEndRegistry().size());
	_tuner = factory(_uuid);
	_spectrum = new SpectrumSink(_uuid);
	_tuner->connect(_spectrum);
	frontEndRegistry()[_uuid] = this;
	LOG_DEBUG("front end %s: tuner and SpectrumSink wired\n", _uuid.c_str());
}

FrontEnd::~FrontEnd()
{
	for (map<string, Receiver*>::iterator rx = _receivers.begin(); rx != _receivers.end(); ++rx)
		_tuner->disconnect(rx->second->input());
	_receivers.clear();
	frontEndRegistry().erase(_uuid);
	delete _tuner;
	delete _spectrum;
	LOG_DEBUG("front end %s: released\n", _uuid.c_str());
}

void FrontEnd::addReceiver(Receiver *rx)
{
	_tuner->connect(rx->input());
	_receivers[rx->uuid()] = rx;
	LOG_DEBUG("receiver %s now listens to front end %s\n", rx->uuid().c_str(), _uuid.c_str());
}

void FrontEnd::removeReceiver(Receiver *rx)
{
	_receivers.erase(rx->uuid());
	_tuner->disconnect(rx->input());
	LOG_DEBUG("receiver %s detached from front end %s\n", rx->uuid().c_str(), _uuid.c_str());
}
