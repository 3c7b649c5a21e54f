/*
 * host_pipeline.cxx -- drives the host runtime the way webradio's main.cxx does
 * (main.cxx:71-122): a FrontEnd around a Tuner, Receivers attached to it, Radio::run()
 * in a loop -- but with a replaying tuner instead of an RTL-SDR stick and the audio
 * collected from each Receiver's terminal sink.  TEST INFRASTRUCTURE.
 *
 * Built twice against the SAME headers (webradio_amd/host):
 *   libwr_host_pipeline.so      with this repo's radio.cxx
 *   oracle/_ref/libwr_boundary.so   with the REFERENCE's radio.cxx, compiled unchanged
 *                                   from /root/reference/src/radio.cxx (the drop-in proof)
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>

#include <atomic>
#include <thread>

#include <vector>

#include "radio.h"
#include "filetuner.h"
#include "gpubatch.h"

namespace {

const float *g_iq = NULL;
size_t g_frames = 0, g_pos = 0;
unsigned int g_rate = 0, g_block = 0;

/* a tuner that replays a caller-supplied IQ recording block by block */
class ReplayTuner : public Tuner {
public:
	ReplayTuner(const string &name) : Tuner(name, "ReplayTuner") {
		_name = "replay";
		_manufacturer = "webradio_amd tests";
	}
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const vector<sample_t> &, vector<sample_t> &out) {
		size_t frames = out.size() / 2;
		if (g_pos + frames > g_frames)
			return false;
		memcpy(out.data(), g_iq + 2 * g_pos, out.size() * sizeof(float));
		g_pos += frames;
		return true;
	}
};

/* r06: the same replay with every block produced in DEVICE memory (DeviceBlock: three buffers in turn) -- the tuner batch
 * then streams the blocks (wr_tuner_set_streaming).  WR_TEST_DEVICE_SOURCE=1 selects it. */
class ReplayDeviceTuner : public Tuner, public DeviceBlock {
public:
	ReplayDeviceTuner(const string &name) : Tuner(name, "ReplayDeviceTuner"), _dev(NULL), _all(NULL), _at(0), _frames(0) {
		_name = "replay (device memory)";
		_manufacturer = "webradio_amd tests";
	}
	~ReplayDeviceTuner() { if (_all) wr_dev_free(_dev, _all); }
	const float *deviceBlock(wr_dev **dev, size_t *frames) const {
		if (dev) *dev = _dev;
		if (frames) *frames = _frames;
		return _all ? (const float *)_all + 2 * _at : NULL;
	}
protected:
	bool init() {
		/* the whole recording goes to the device once (a resident capture): the blocks are consecutive pieces of it */
		_dev = wrhost::deviceFor(this);
		if (!_dev)
			return false;
		if (!_all && (wr_dev_malloc(_dev, g_frames * 2 * sizeof(float), &_all) != WR_OK ||
		              wr_dev_upload(_dev, _all, g_iq, g_frames * 2 * sizeof(float)) != WR_OK))
			return false;
		return true;
	}
	void deinit() {}
	bool process(const vector<sample_t> &, vector<sample_t> &out) {
		size_t frames = out.size() / 2;
		if (g_pos + frames > g_frames)
			return false;
		_at = g_pos;
		_frames = frames;
		const bool skip = consumersReadOnDevice();
		if (!skip)
			memcpy(out.data(), g_iq + 2 * g_pos, out.size() * sizeof(float));
		setHostBlockValid(!skip);
		g_pos += frames;
		return true;
	}
	wr_dev *_dev;
	void *_all;
	size_t _at, _frames;
};

Tuner *makeTuner(const string &name)
{
	if (getenv("WR_TEST_DEVICE_SOURCE") && atoi(getenv("WR_TEST_DEVICE_SOURCE")))
		return new ReplayDeviceTuner(name);
	return new ReplayTuner(name);
}

/* a sink that keeps what it is handed (any channel count) */
class TapSink : public DspBlock {
public:
	TapSink() : DspBlock("tap", "TapSink") {}
	vector<float> got;
	vector<unsigned int> sizes;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const vector<sample_t> &in, vector<sample_t> &) {
		got.insert(got.end(), in.begin(), in.end());
		sizes.push_back((unsigned int)in.size());
		return true;
	}
};

} // namespace

extern "C" {

/* Runs `nrx` receivers over the recording.  modes: Demodulator::Mode per receiver.
 * audio_out receives nrx rows of `audio_cap` floats; *audio_len = samples per receiver.
 * With `retune_at` >= 0 receiver 0 is retuned to `retune_if` before that block (the
 * REST handlers do this from other threads, receiverhandler.cxx:130-137).
 * spectrum_out (optional) receives fft_size dB values from FrontEnd::spectrum().
 * Returns 0, or a negative stage code. */
/* what the tuner batch of the last wr_host_run streamed (wr_tuner_stream_info): launches opened, blocks taken */
static unsigned long long g_stream_launches, g_stream_blocks;
unsigned long long wr_host_stream_launches(void) { return g_stream_launches; }
unsigned long long wr_host_stream_blocks(void) { return g_stream_blocks; }

int wr_host_run(const float *iq, size_t nframes, unsigned int rate, unsigned int block_frames,
                unsigned int nrx, const int *if_hz, const int *modes,
                unsigned int chan_passband, unsigned int chan_rate,
                unsigned int audio_passband, unsigned int audio_rate,
                int retune_at, int retune_if,
                float *audio_out, size_t audio_cap, size_t *audio_len,
                unsigned int fft_size, float *spectrum_out)
{
	g_iq = iq;
	g_frames = nframes;
	g_pos = 0;
	g_rate = rate;
	g_block = block_frames;

	FrontEnd *fe = new FrontEnd(makeTuner);
	fe->tuner()->setSampleRate(rate);
	fe->tuner()->setChannels(2);
	fe->tuner()->setBlockSize(block_frames * 2);
	if (fft_size)
		fe->spectrum()->setFftSize(fft_size);

	std::vector<Receiver *> rx(nrx);
	for (unsigned int n = 0; n < nrx; n++) {
		rx[n] = new Receiver();
		rx[n]->downconverter()->setIF(if_hz[n]);
		rx[n]->channelFilter()->setPassband(chan_passband);
		rx[n]->channelFilter()->setOutputSampleRate(chan_rate);
		rx[n]->audioFilter()->setPassband(audio_passband);
		rx[n]->audioFilter()->setOutputSampleRate(audio_rate);
		rx[n]->demodulator()->setMode((Demodulator::Mode)modes[n]);
		rx[n]->stream()->setCapacity(audio_cap);
		/* test hook for LowPass::setFirLength (an extension) */
		if (getenv("WR_TEST_FIR_LENGTH")) {
			rx[n]->channelFilter()->setFirLength((unsigned int)atoi(getenv("WR_TEST_FIR_LENGTH")));
			rx[n]->audioFilter()->setFirLength((unsigned int)atoi(getenv("WR_TEST_FIR_LENGTH")));
		}
		if (getenv("WR_TEST_FIR_LENGTH_CHAN"))       /* the channel filter alone: such a Receiver stays in the tuner batch */
			rx[n]->channelFilter()->setFirLength((unsigned int)atoi(getenv("WR_TEST_FIR_LENGTH_CHAN")));
		if (getenv("WR_TEST_FIR_LENGTH_AUDIO"))      /* the audio filter alone (r05: that stays in the batch too) */
			rx[n]->audioFilter()->setFirLength((unsigned int)atoi(getenv("WR_TEST_FIR_LENGTH_AUDIO")));
		rx[n]->setFrontEnd(fe);
	}
	int rc = 0;
	if (!fe->tuner()->start()) {
		rc = -1;
	} else {
		size_t blocks = nframes / block_frames;
		for (size_t b = 0; b < blocks; b++) {
			if ((long)b == (long)retune_at && nrx)
				rx[0]->downconverter()->setIF(retune_if);
			Radio::run();
		}
		{
			bool live = false;
			g_stream_launches = g_stream_blocks = 0;
			if (nrx)
				(void)wrhost::streamInfo(rx[0]->downconverter(), &live, &g_stream_launches, &g_stream_blocks);
		}
		if (spectrum_out && fft_size)
			fe->spectrum()->getSpectrum(spectrum_out);
		size_t len = nrx ? rx[0]->stream()->samples().size() : 0;
		for (unsigned int n = 0; n < nrx && rc == 0; n++) {
			const vector<float> &a = rx[n]->stream()->samples();
			if (a.size() != len || len > audio_cap)
				rc = -2;
			else
				memcpy(audio_out + (size_t)n * audio_cap, a.data(), len * sizeof(float));
		}
		*audio_len = len;
		Radio::profile();
		fe->tuner()->stop();
	}
	for (unsigned int n = 0; n < nrx; n++)
		delete rx[n];
	delete fe;
	return rc;
}

/* stop()/start() in the middle of the stream (what a REST client does when it changes the
 * tuner's sample rate, tunercontrolhandler.cxx:88-107): LowPass histories are dropped
 * (lowpass.cxx:118-129), NCO phase and Demodulator prev_i/q survive (quirk Q5).
 * Two front ends are run side by side to exercise one TunerBatch per source.
 * audio_out: [2 tuners][nrx][audio_cap]. */
int wr_host_run_restart(const float *iq, size_t nframes, unsigned int rate, unsigned int block_frames,
                        unsigned int nrx, const int *if_hz, int mode,
                        unsigned int chan_passband, unsigned int chan_rate,
                        unsigned int audio_passband, unsigned int audio_rate, unsigned int restart_at,
                        float *audio_out, size_t audio_cap, size_t *audio_len)
{
	g_iq = iq;
	g_frames = nframes;
	g_pos = 0;
	/* both tuners replay the same recording; the second one starts one block later */
	struct Replay2 : public Tuner {
		Replay2(const string &n) : Tuner(n, "Replay2"), pos(0) {}
		size_t pos;
		bool init() { return true; }
		void deinit() {}
		bool process(const vector<sample_t> &, vector<sample_t> &out) {
			size_t frames = out.size() / 2;
			if (pos + frames > g_frames)
				return false;
			memcpy(out.data(), g_iq + 2 * pos, out.size() * sizeof(float));
			pos += frames;
			return true;
		}
	};
	struct F { static Tuner *make(const string &n) { return new Replay2(n); } };
	FrontEnd *fe[2] = { new FrontEnd(F::make), new FrontEnd(F::make) };
	std::vector<Receiver *> rx;
	for (int t = 0; t < 2; t++) {
		fe[t]->tuner()->setSampleRate(rate);
		fe[t]->tuner()->setChannels(2);
		fe[t]->tuner()->setBlockSize(block_frames * 2);
		static_cast<Replay2 *>(fe[t]->tuner())->pos = (size_t)t * block_frames;
		for (unsigned int n = 0; n < nrx; n++) {
			Receiver *r = new Receiver();
			r->downconverter()->setIF(if_hz[n]);
			r->channelFilter()->setPassband(chan_passband);
			r->channelFilter()->setOutputSampleRate(chan_rate);
			r->audioFilter()->setPassband(audio_passband);
			r->audioFilter()->setOutputSampleRate(audio_rate);
			r->demodulator()->setMode((Demodulator::Mode)mode);
			r->stream()->setCapacity(audio_cap);
			r->setFrontEnd(fe[t]);
			rx.push_back(r);
		}
	}
	int rc = 0;
	if (!fe[0]->tuner()->start() || !fe[1]->tuner()->start())
		rc = -1;
	size_t blocks = nframes / block_frames - 1;
	std::vector<std::vector<float> > keep(rx.size());
	for (size_t b = 0; b < blocks && rc == 0; b++) {
		if (b == restart_at) {
			for (size_t n = 0; n < rx.size(); n++)          /* the sinks clear themselves on init() */
				keep[n] = rx[n]->stream()->samples();
			for (int t = 0; t < 2; t++) {
				fe[t]->tuner()->stop();
				if (!fe[t]->tuner()->start())
					rc = -3;
			}
		}
		Radio::run();
	}
	size_t len = 0;
	for (size_t n = 0; n < rx.size() && rc == 0; n++) {
		std::vector<float> all = keep[n];
		const vector<float> &a = rx[n]->stream()->samples();
		all.insert(all.end(), a.begin(), a.end());
		if (n == 0)
			len = all.size();
		if (all.size() != len || len > audio_cap)
			rc = -2;
		else
			memcpy(audio_out + n * audio_cap, all.data(), len * sizeof(float));
	}
	*audio_len = len;
	for (int t = 0; t < 2; t++)
		fe[t]->tuner()->stop();
	for (size_t n = 0; n < rx.size(); n++)
		delete rx[n];
	delete fe[0];
	delete fe[1];
	return rc;
}

/* BASELINE config 1: one DownConverter + FM demodulator on a recorded RTL-SDR IQ file.
 * with_frontend = 1: FrontEnd(FileTuner::factory) + Receiver (a SpectrumSink is attached, so
 * the float block is staged on the GPU once and shared); 0: the bare graph
 * FileTuner -> Receiver chain, which ships the raw bytes (2 per frame) to the GPU. */
int wr_host_run_file(const char *path, unsigned int rate, unsigned int block_frames, unsigned int nblocks,
                     int if_hz, int mode, unsigned int chan_passband, unsigned int chan_rate,
                     unsigned int audio_passband, unsigned int audio_rate, int with_frontend,
                     float *audio_out, size_t audio_cap, size_t *audio_len)
{
	FrontEnd *fe = NULL;
	Tuner *tuner;
	if (with_frontend) {
		fe = new FrontEnd(FileTuner::factory);
		tuner = fe->tuner();
	} else {
		tuner = FileTuner::factory("file");
	}
	tuner->setSubdevice(path);
	tuner->setSampleRate(rate);
	tuner->setChannels(2);
	tuner->setBlockSize(block_frames * 2);
	Receiver *rx = new Receiver();
	rx->downconverter()->setIF(if_hz);
	rx->channelFilter()->setPassband(chan_passband);
	rx->channelFilter()->setOutputSampleRate(chan_rate);
	rx->audioFilter()->setPassband(audio_passband);
	rx->audioFilter()->setOutputSampleRate(audio_rate);
	rx->demodulator()->setMode((Demodulator::Mode)mode);
	rx->stream()->setCapacity(audio_cap);
	if (fe)
		rx->setFrontEnd(fe);
	else
		tuner->connect(rx->input());
	int rc = 0;
	if (!tuner->start()) {
		rc = -1;
	} else {
		for (unsigned int b = 0; b < nblocks; b++)
			if (!tuner->run())
				rc = -4;
		{
			bool live = false;
			g_stream_launches = g_stream_blocks = 0;
			(void)wrhost::streamInfo(rx->downconverter(), &live, &g_stream_launches, &g_stream_blocks);
		}
		/* one more block than the file holds: process() must fail, not crash */
		if (rc == 0 && tuner->run())
			rc = -5;
		const vector<float> &a = rx->stream()->samples();
		*audio_len = a.size();
		if (a.size() > audio_cap)
			rc = -2;
		else
			memcpy(audio_out, a.data(), a.size() * sizeof(float));
		tuner->stop();
	}
	if (!fe)
		tuner->disconnect(rx->input());
	delete rx;
	if (fe)
		delete fe;
	else
		delete tuner;
	return rc;
}

/* Control-plane stress (hard part H7): while the pipeline thread pumps Radio::run(), another
 * thread hammers the setters the REST handlers call from libmicrohttpd's connection threads
 * (receiverhandler.cxx:130-137): setIF, setPassband, setModeString.  The reference races here
 * (SURVEY 3.3); this runtime stages them and applies them at block boundaries.  Returns the
 * number of setter calls made, or < 0.  audio_out: [nrx][audio_cap] of the LAST block only. */
long wr_host_setter_stress(const float *iq, size_t nframes, unsigned int rate, unsigned int block_frames,
                           unsigned int nrx, unsigned int chan_rate, unsigned int audio_rate,
                           float *audio_out, size_t audio_cap, size_t *audio_len)
{
	g_iq = iq;
	g_frames = nframes;
	g_pos = 0;
	FrontEnd *fe = new FrontEnd(makeTuner);
	fe->tuner()->setSampleRate(rate);
	fe->tuner()->setChannels(2);
	fe->tuner()->setBlockSize(block_frames * 2);
	std::vector<Receiver *> rx(nrx);
	for (unsigned int n = 0; n < nrx; n++) {
		rx[n] = new Receiver();
		rx[n]->channelFilter()->setOutputSampleRate(chan_rate);
		rx[n]->audioFilter()->setOutputSampleRate(audio_rate);
		rx[n]->channelFilter()->setPassband(rate / 16);
		rx[n]->audioFilter()->setPassband(chan_rate / 8);
		rx[n]->stream()->setCapacity(audio_cap);
		rx[n]->setFrontEnd(fe);
	}
	if (!fe->tuner()->start())
		return -1;
	std::atomic<bool> quit(false);
	std::atomic<long> calls(0);
	std::thread ctl([&]() {
		unsigned int lcg = 1;
		const char *modes[] = { "AM", "FM", "USB", "LSB" };
		while (!quit.load()) {
			lcg = lcg * 1664525u + 1013904223u;
			Receiver *r = rx[(lcg >> 8) % nrx];
			switch ((lcg >> 4) & 3) {
			case 0: r->downconverter()->setIF((int)((lcg >> 10) % rate) - (int)(rate / 2)); break;
			case 1: r->channelFilter()->setPassband(rate / (8 + ((lcg >> 12) & 31))); break;
			case 2: r->demodulator()->setModeString(modes[(lcg >> 14) & 3]); break;
			default: r->audioFilter()->setPassband(chan_rate / (4 + ((lcg >> 16) & 15))); break;
			}
			calls++;
		}
	});
	long rc = 0;
	size_t blocks = nframes / block_frames;
	for (size_t b = 0; b < blocks; b++)
		Radio::run();
	quit.store(true);
	ctl.join();
	/* one quiet block after the storm: its audio must be what the final settings give */
	g_pos = 0;
	for (unsigned int n = 0; n < nrx; n++)
		rx[n]->stream()->clear();
	Radio::run();
	size_t len = rx[0]->stream()->samples().size();
	for (unsigned int n = 0; n < nrx && rc == 0; n++) {
		const vector<float> &a = rx[n]->stream()->samples();
		if (a.size() != len || len > audio_cap)
			rc = -2;
		else
			memcpy(audio_out + (size_t)n * audio_cap, a.data(), len * sizeof(float));
	}
	*audio_len = len;
	fe->tuner()->stop();
	for (unsigned int n = 0; n < nrx; n++)
		delete rx[n];
	delete fe;
	return rc ? rc : calls.load();
}

/* A graph that is NOT the fused Receiver shape, built with the public DspBlock API the way a
 * user of the reference would: tuner -> DownConverter -> LowPass x nstages -> Demodulator ->
 * LowPass -> AudioStreamManager.  Several decimating stages in a row are what makes a narrow
 * channel off a fast stream possible at all with 64-tap filters (one stage gives
 * maxbin = 0, all-zero taps).  Runs block by block on the GPU with the intermediates handed
 * over in device memory.  Returns the number of audio samples, or < 0. */
long wr_host_run_multistage(const float *iq, size_t nframes, unsigned int rate, unsigned int block_frames,
                            int if_hz, int mode, unsigned int nstages, const unsigned int *stage_rates,
                            const unsigned int *stage_passbands, unsigned int audio_passband,
                            unsigned int audio_rate, float *audio_out, size_t audio_cap)
{
	g_iq = iq;
	g_frames = nframes;
	g_pos = 0;
	g_rate = rate;
	g_block = block_frames;
	Tuner *tuner = makeTuner("multistage");
	tuner->setSampleRate(rate);
	tuner->setChannels(2);
	tuner->setBlockSize(block_frames * 2);
	DownConverter mixer("ms");
	Demodulator demod("ms");
	LowPass audio("ms-audio");
	AudioStreamManager sink("ms");
	std::vector<LowPass *> stages;
	mixer.setIF(if_hz);
	demod.setMode((Demodulator::Mode)mode);
	tuner->connect(&mixer);
	DspBlock *last = &mixer;
	for (unsigned int n = 0; n < nstages; n++) {
		LowPass *f = new LowPass("ms-stage");
		f->setPassband(stage_passbands[n]);
		f->setOutputSampleRate(stage_rates[n]);
		last->connect(f);
		last = f;
		stages.push_back(f);
	}
	last->connect(&demod);
	demod.connect(&audio);
	audio.setPassband(audio_passband);
	audio.setOutputSampleRate(audio_rate);
	audio.connect(&sink);
	sink.setCapacity(audio_cap);
	long rc = -1;
	if (tuner->start()) {
		for (size_t b = 0; b < nframes / block_frames; b++)
			tuner->run();
		const vector<float> &a = sink.samples();
		rc = (long)a.size();
		if (a.size() <= audio_cap)
			memcpy(audio_out, a.data(), a.size() * sizeof(float));
		else
			rc = -2;
		tuner->stop();
	}
	delete tuner;
	for (size_t n = 0; n < stages.size(); n++)
		delete stages[n];
	return rc;
}

/* events the tuner batches have traced so far (WEBRADIO_TRACE=1): non-zero = a chain was fused */
int wr_host_trace_count(void)
{
	return (int)wrhost::trace().size();
}

int wr_host_registry_sizes(void)
{
	return (int)(Radio::frontEnds().size() * 1000 + Radio::receivers().size());
}

/* stop() -> setSampleRate / setBlockSize -> start(): the source's tuner batch must follow the new
 * input rate (the NCO step of every channel, downconverter.cxx:80) and the new block size.  One
 * front end; `blocks1` blocks at (rate1, block1), then `blocks2` at (rate2, block2) from frame
 * blocks1*block1 of the same recording on.  audio_out: [2 phases][nrx][audio_cap]. */
int wr_host_run_rerate(const float *iq, size_t nframes, unsigned int rate1, unsigned int block1,
                       unsigned int rate2, unsigned int block2, unsigned int blocks1, unsigned int blocks2,
                       unsigned int nrx, const int *if_hz, int mode,
                       unsigned int chan_passband, unsigned int chan_rate,
                       unsigned int audio_passband, unsigned int audio_rate,
                       float *audio_out, size_t audio_cap, size_t *len1, size_t *len2)
{
	g_iq = iq;
	g_frames = nframes;
	g_pos = 0;
	FrontEnd *fe = new FrontEnd(makeTuner);
	fe->tuner()->setSampleRate(rate1);
	fe->tuner()->setChannels(2);
	fe->tuner()->setBlockSize(block1 * 2);
	std::vector<Receiver *> rx;
	for (unsigned int n = 0; n < nrx; n++) {
		Receiver *r = new Receiver();
		r->downconverter()->setIF(if_hz[n]);
		r->channelFilter()->setPassband(chan_passband);
		r->channelFilter()->setOutputSampleRate(chan_rate);
		r->audioFilter()->setPassband(audio_passband);
		r->audioFilter()->setOutputSampleRate(audio_rate);
		r->demodulator()->setMode((Demodulator::Mode)mode);
		r->stream()->setCapacity(audio_cap);
		r->setFrontEnd(fe);
		rx.push_back(r);
	}
	int rc = fe->tuner()->start() ? 0 : -1;
	for (unsigned int b = 0; b < blocks1 && rc == 0; b++)
		Radio::run();
	*len1 = *len2 = 0;
	for (size_t n = 0; n < rx.size() && rc == 0; n++) {
		const vector<float> &a = rx[n]->stream()->samples();
		if (a.size() > audio_cap) { rc = -2; break; }
		memcpy(audio_out + n * audio_cap, a.data(), a.size() * sizeof(float));
		*len1 = a.size();
	}
	fe->tuner()->stop();
	fe->tuner()->setSampleRate(rate2);
	fe->tuner()->setBlockSize(block2 * 2);
	if (rc == 0 && !fe->tuner()->start())
		rc = -3;
	for (unsigned int b = 0; b < blocks2 && rc == 0; b++)
		Radio::run();
	for (size_t n = 0; n < rx.size() && rc == 0; n++) {
		const vector<float> &a = rx[n]->stream()->samples();
		if (a.size() > audio_cap) { rc = -2; break; }
		memcpy(audio_out + (rx.size() + n) * audio_cap, a.data(), a.size() * sizeof(float));
		*len2 = a.size();
	}
	fe->tuner()->stop();
	for (size_t n = 0; n < rx.size(); n++)
		delete rx[n];
	delete fe;
	return rc;
}

/* a second consumer attached INSIDE a fused chain while it runs (here: a tap on receiver
 * 0's demodulator after `tap_at` blocks, the way a per-receiver scope would be).  The fusion had
 * elided that block's host output; DspBlock::connect takes the chain out of the tuner batch, it goes
 * on block by block (filters from empty histories), the other receivers stay fused.
 * tap_out receives what the tap was handed (tap_cap floats at most), *tap_len its length,
 * *tap_blocks the number of process() calls it saw. */
int wr_host_run_tap(const float *iq, size_t nframes, unsigned int rate, unsigned int block_frames,
                    unsigned int nrx, const int *if_hz, int mode,
                    unsigned int chan_passband, unsigned int chan_rate,
                    unsigned int audio_passband, unsigned int audio_rate, unsigned int tap_at,
                    float *audio_out, size_t audio_cap, size_t *audio_len,
                    float *tap_out, size_t tap_cap, size_t *tap_len, unsigned int *tap_blocks)
{
	g_iq = iq;
	g_frames = nframes;
	g_pos = 0;
	FrontEnd *fe = new FrontEnd(makeTuner);
	fe->tuner()->setSampleRate(rate);
	fe->tuner()->setChannels(2);
	fe->tuner()->setBlockSize(block_frames * 2);
	std::vector<Receiver *> rx;
	for (unsigned int n = 0; n < nrx; n++) {
		Receiver *r = new Receiver();
		r->downconverter()->setIF(if_hz[n]);
		r->channelFilter()->setPassband(chan_passband);
		r->channelFilter()->setOutputSampleRate(chan_rate);
		r->audioFilter()->setPassband(audio_passband);
		r->audioFilter()->setOutputSampleRate(audio_rate);
		r->demodulator()->setMode((Demodulator::Mode)mode);
		r->stream()->setCapacity(audio_cap);
		r->setFrontEnd(fe);
		rx.push_back(r);
	}
	TapSink *tap = new TapSink();
	int rc = fe->tuner()->start() ? 0 : -1;
	const size_t blocks = nframes / block_frames;
	for (size_t b = 0; b < blocks && rc == 0; b++) {
		if (b == tap_at && nrx)
			rx[0]->demodulator()->connect(tap);
		if (!fe->tuner()->run())
			rc = -4;
	}
	*audio_len = 0;
	for (size_t n = 0; n < rx.size() && rc == 0; n++) {
		const vector<float> &a = rx[n]->stream()->samples();
		if (a.size() > audio_cap) { rc = -2; break; }
		memcpy(audio_out + n * audio_cap, a.data(), a.size() * sizeof(float));
		*audio_len = a.size();
	}
	*tap_len = tap->got.size() < tap_cap ? tap->got.size() : tap_cap;
	memcpy(tap_out, tap->got.data(), *tap_len * sizeof(float));
	*tap_blocks = (unsigned int)tap->sizes.size();
	fe->tuner()->stop();
	if (nrx)
		rx[0]->demodulator()->disconnect(tap);
	for (size_t n = 0; n < rx.size(); n++)
		delete rx[n];
	delete fe;
	delete tap;
	return rc;
}

/* Two front ends pumped by Radio::run() (radio.cxx:56-59: one after the other), `blocks` runs with
 * a pause in between; returns every receiver's audio and the tuner batches' trace (WEBRADIO_TRACE=1:
 * "S0A0S1A1|..." -- S submit, A audio that was already in the ring, W audio waited for, digit = front
 * end, | = end of a Radio::run()).  With WEBRADIO_AUDIO_LATE=1 no run() waits for the GPU.
 * audio_out: [2][nrx][audio_cap]. */
int wr_host_run_two_traced(const float *iq, size_t nframes, unsigned int rate, unsigned int block_frames,
                           unsigned int nrx, const int *if_hz, int mode,
                           unsigned int chan_passband, unsigned int chan_rate,
                           unsigned int audio_passband, unsigned int audio_rate, unsigned int blocks,
                           float *audio_out, size_t audio_cap, size_t *audio_len, char *trace_out, size_t trace_cap)
{
	g_iq = iq;
	g_frames = nframes;
	struct Replay3 : public Tuner {
		Replay3(const string &n) : Tuner(n, "Replay3"), pos(0) {}
		size_t pos;
		bool init() { return true; }
		void deinit() {}
		bool process(const vector<sample_t> &, vector<sample_t> &out) {
			size_t frames = out.size() / 2;
			if (pos + frames > g_frames)
				return false;
			memcpy(out.data(), g_iq + 2 * pos, out.size() * sizeof(float));
			pos += frames;
			return true;
		}
	};
	struct F { static Tuner *make(const string &n) { return new Replay3(n); } };
	FrontEnd *fe[2] = { new FrontEnd(F::make), new FrontEnd(F::make) };
	std::vector<Receiver *> rx;
	for (int t = 0; t < 2; t++) {
		fe[t]->tuner()->setSampleRate(rate);
		fe[t]->tuner()->setChannels(2);
		fe[t]->tuner()->setBlockSize(block_frames * 2);
		static_cast<Replay3 *>(fe[t]->tuner())->pos = (size_t)t * block_frames;    /* second tuner: one block later */
		for (unsigned int n = 0; n < nrx; n++) {
			Receiver *r = new Receiver();
			r->downconverter()->setIF(if_hz[n]);
			r->channelFilter()->setPassband(chan_passband);
			r->channelFilter()->setOutputSampleRate(chan_rate);
			r->audioFilter()->setPassband(audio_passband);
			r->audioFilter()->setOutputSampleRate(audio_rate);
			r->demodulator()->setMode((Demodulator::Mode)mode);
			r->stream()->setCapacity(audio_cap);
			r->setFrontEnd(fe[t]);
			rx.push_back(r);
		}
	}
	int rc = (fe[0]->tuner()->start() && fe[1]->tuner()->start()) ? 0 : -1;
	wrhost::traceClear();
	std::string tr;
	size_t seen = 0;
	for (unsigned int b = 0; b < blocks && rc == 0; b++) {
		Radio::run();
		const std::vector<wrhost::TraceEvent> &ev = wrhost::trace();
		for (; seen < ev.size(); seen++) {
			tr += ev[seen].kind;
			tr += (ev[seen].source == (const void *)static_cast<DspSource *>(fe[0]->tuner())) ? '0' : '1';
		}
		tr += '|';
		timespec ts = { 0, 5000000 };          /* let the GPU finish what was enqueued */
		nanosleep(&ts, NULL);
	}
	size_t len = 0;
	for (size_t n = 0; n < rx.size() && rc == 0; n++) {
		const vector<float> &a = rx[n]->stream()->samples();
		if (n == 0)
			len = a.size();
		if (a.size() != len || len > audio_cap) { rc = -2; break; }
		memcpy(audio_out + n * audio_cap, a.data(), len * sizeof(float));
	}
	*audio_len = len;
	if (trace_out && trace_cap) {
		strncpy(trace_out, tr.c_str(), trace_cap - 1);
		trace_out[trace_cap - 1] = 0;
	}
	for (int t = 0; t < 2; t++)
		fe[t]->tuner()->stop();
	for (size_t n = 0; n < rx.size(); n++)
		delete rx[n];
	delete fe[0];
	delete fe[1];
	return rc;
}

} // extern "C"
