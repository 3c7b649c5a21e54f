/*
 * ref_chain.cxx -- runs the REAL reference blocks of the hot path (DownConverter, LowPass,
 * Demodulator, SpectrumSink: /root/reference/src/dsp/{dspblock,downconverter,lowpass,demodulator}.cxx
 * and io/spectrumsink.cxx, compiled where they lie) through their PUBLIC API and hands what they
 * produced to the tests.  TEST INFRASTRUCTURE ONLY -- nothing under webradio_amd/ links or loads it.
 *
 * Three of those files call FFTW (lowpass.cxx:98-100,180: a 64-point inverse complex DFT per
 * setPassband; spectrumsink.cxx:65-68,115: the N-point forward DFT of every frame).  FFTW3 itself is
 * not in the image; the image's own implementation of the FFTW3 API is -- hipFFTW
 * (/opt/rocm/include/hipfft/hipfftw.h, /opt/rocm/lib/libhipfftw.so: AMD's fftw3-compatible front
 * end of rocFFT).  oracle/Makefile (`make ref_chain`) points the name <fftw3.h> at that header with a
 * symlink under oracle/_ref/ and links libhipfftw.so: no line of ours stands between the reference's
 * FFTW calls and a real FFT library.  rocFFT executes on a GPU, so this library builds here and RUNS
 * only on the GPU box (tests/test_gpu_reference_pin.py); the vectors it produced there are committed
 * under tests/golden/ (tests/golden/make_reference_chain_golden.py) and checked on CPU too.
 *
 * What the reference's callers use, nothing else: construct, setters, connect, start, run
 * (radio.cxx:62-90), getSpectrum (waterfallhandler.cxx:56-61).  No private member is read: the taps
 * of a LowPass are observed as its impulse response.
 */
#include <cstring>
#include <string>
#include <vector>

#include "dspblock.h"
#include "downconverter.h"
#include "lowpass.h"
#include "demodulator.h"
#include "spectrumsink.h"

namespace {

/* a source that plays a caller-owned buffer block by block (the role of io/rtlsdrtuner.cxx / randsource.cxx) */
class PlaySource : public DspSource {
public:
	PlaySource() : DspSource("play", "PlaySource"), data(0), total(0), pos(0) {}
	const float *data;
	size_t total, pos;                /* floats */
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const std::vector<sample_t> &, std::vector<sample_t> &out) {
		size_t n = out.size();
		if (pos + n > total)
			return false;
		memcpy(&out[0], data + pos, n * sizeof(float));
		pos += n;
		return true;
	}
};

/* a sink that keeps what it is given (the role of web/audiostream.cxx) */
class Capture : public DspBlock {
public:
	Capture(const std::string &name) : DspBlock(name, "Capture") {}
	std::vector<float> got;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const std::vector<sample_t> &in, std::vector<sample_t> &) {
		got.insert(got.end(), in.begin(), in.end());
		return true;
	}
};

size_t give(const std::vector<float> &v, float *out, size_t cap)
{
	if (out && v.size() <= cap && !v.empty())
		memcpy(out, &v[0], v.size() * sizeof(float));
	return v.size();
}

}  // namespace

extern "C" {

/* LowPass::init + recalculate (lowpass.cxx:81-116,164-197) observed from outside: a unit impulse through a
 * one-channel LowPass at decimation 1 comes out as its coefficients, y[k] = h[k] (lowpass.cxx:145-159).
 * Returns the number of taps written (64), or a negative number when start() or run() failed. */
int ref_lowpass_impulse_response(unsigned int passband, unsigned int input_rate, float *taps, unsigned int ntaps)
{
	PlaySource src;
	LowPass lp("lp");
	Capture cap("cap");
	std::vector<float> x(2 * ntaps, 0.0f);
	x[0] = 1.0f;
	src.data = &x[0];
	src.total = x.size();
	src.setSampleRate(input_rate);
	src.setChannels(1);
	src.setBlockSize(x.size());
	lp.setPassband(passband);
	lp.setOutputSampleRate(input_rate);
	src.connect(&lp);
	lp.connect(&cap);
	if (!src.start())
		return -1;
	bool ok = src.run();
	src.stop();
	if (!ok || cap.got.size() < ntaps)
		return -2;
	memcpy(taps, &cap.got[0], ntaps * sizeof(float));
	return (int)ntaps;
}

/* One Receiver chain as radio.cxx:68-83 wires it -- tuner -> DownConverter -> LowPass -> Demodulator -> LowPass ->
 * sink -- fed `nframes` IQ frames in blocks of `block_frames`, with a capture on every stage.  Every *_cap is the
 * capacity in floats of the buffer before it; the counts actually produced come back in produced[0..2]
 * (channel IQ floats, demodulator floats, audio floats).  0, or negative when start()/run() failed. */
int ref_receiver_chain(unsigned int fs, int if_hz, unsigned int chan_passband, unsigned int chan_rate, int mode,
                       unsigned int audio_passband, unsigned int audio_rate, const float *iq, size_t nframes,
                       size_t block_frames, float *chan_iq, size_t chan_cap, float *demod, size_t demod_cap,
                       float *audio, size_t audio_cap, size_t *produced)
{
	PlaySource src;
	DownConverter dc("dc");
	LowPass chan("chan"), aud("audio");
	Demodulator dem("demod");
	Capture c_chan("c_chan"), c_dem("c_dem"), c_aud("c_aud");
	src.data = iq;
	src.total = 2 * nframes;
	src.setSampleRate(fs);
	src.setChannels(2);
	src.setBlockSize(2 * block_frames);          /* blockSize counts floats (dspblock.h:134) */
	dc.setIF(if_hz);
	chan.setPassband(chan_passband);
	chan.setOutputSampleRate(chan_rate);
	dem.setMode((Demodulator::Mode)mode);
	aud.setPassband(audio_passband);
	aud.setOutputSampleRate(audio_rate);
	src.connect(&dc);
	dc.connect(&chan);
	chan.connect(&dem);
	chan.connect(&c_chan);
	dem.connect(&aud);
	dem.connect(&c_dem);
	aud.connect(&c_aud);
	if (!src.start())
		return -1;
	int rc = 0;
	for (size_t b = 0; b + block_frames <= nframes; b += block_frames)
		if (!src.run()) {
			rc = -2;
			break;
		}
	src.stop();
	produced[0] = give(c_chan.got, chan_iq, chan_cap);
	produced[1] = give(c_dem.got, demod, demod_cap);
	produced[2] = give(c_aud.got, audio, audio_cap);
	return rc;
}

/* The mixer alone: DownConverter::process (downconverter.cxx:91-114) at full rate. */
int ref_downconverter(unsigned int fs, int if_hz, const float *iq, size_t nframes, size_t block_frames, float *mixed)
{
	PlaySource src;
	DownConverter dc("dc");
	Capture cap("cap");
	src.data = iq;
	src.total = 2 * nframes;
	src.setSampleRate(fs);
	src.setChannels(2);
	src.setBlockSize(2 * block_frames);
	dc.setIF(if_hz);
	src.connect(&dc);
	dc.connect(&cap);
	if (!src.start())
		return -1;
	int rc = 0;
	for (size_t b = 0; b + block_frames <= nframes; b += block_frames)
		if (!src.run()) {
			rc = -2;
			break;
		}
	src.stop();
	if (rc == 0)
		memcpy(mixed, &cap.got[0], cap.got.size() * sizeof(float));
	return rc;
}

/* SpectrumSink (spectrumsink.cxx:60-142): `nframes` IQ frames in blocks of `block_frames`, getSpectrum after the
 * last block: db[fft_size]. */
int ref_spectrum(unsigned int fs, unsigned int fft_size, const float *iq, size_t nframes, size_t block_frames, float *db)
{
	PlaySource src;
	SpectrumSink spec("spec");
	src.data = iq;
	src.total = 2 * nframes;
	src.setSampleRate(fs);
	src.setChannels(2);
	src.setBlockSize(2 * block_frames);
	spec.setFftSize(fft_size);
	src.connect(&spec);
	if (!src.start())
		return -1;
	int rc = 0;
	for (size_t b = 0; b + block_frames <= nframes; b += block_frames)
		if (!src.run()) {
			rc = -2;
			break;
		}
	if (rc == 0)
		spec.getSpectrum(db);
	src.stop();
	return rc;
}

}  // extern "C"

/* ---- the reference CPU path as a timed baseline (bench.py: cpu_baseline.kind = "reference") --------------------------
 * T pipeline threads, each with its own tuner and its own subset of the receivers (the reference pipeline itself is
 * single-threaded, radio.cxx:56-59: one run() pumps every receiver of a front end in turn), all fed the same block.
 * The tuner hands its block out the way RtlSdrTuner::process does (io/rtlsdrtuner.cxx:265-285): it SWAPS a filled vector
 * into the output, it does not copy 32 MB per block. */
#include <pthread.h>
#include <time.h>

namespace {

class SwapSource : public DspSource {
public:
	SwapSource() : DspSource("swap", "SwapSource") {}
	std::vector<float> spare;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const std::vector<sample_t> &, std::vector<sample_t> &out) {
		if (spare.size() != out.size())
			return false;
		out.swap(spare);              /* both vectors hold the block: the one swapped out is the one handed out before */
		return true;
	}
};

class SumSink : public DspBlock {
public:
	SumSink() : DspBlock("sum", "SumSink"), sum(0.0), frames(0) {}
	double sum;
	size_t frames;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const std::vector<sample_t> &in, std::vector<sample_t> &) {
		for (size_t n = 0; n < in.size(); n++)
			sum += in[n] < 0 ? -in[n] : in[n];
		frames += in.size();
		return true;
	}
};

struct BenchThread {
	pthread_t tid;
	unsigned int fs, cpb, crate, apb, arate, nblocks;
	int mode;
	std::vector<int> ifs;
	const float *iq;
	size_t nframes;
	pthread_barrier_t *ready, *go;
	double sum;
	size_t frames;
	int rc;
};

/* FFTW's planner is not thread-safe (and the reference never plans from two threads): start() and stop() -- LowPass::init's
 * fftwf_plan_dft_1d / fftwf_execute, deinit's fftwf_destroy_plan / fftwf_cleanup -- one thread at a time; both are untimed */
pthread_mutex_t g_plan_lock = PTHREAD_MUTEX_INITIALIZER;

void *bench_thread(void *p)
{
	BenchThread *b = (BenchThread *)p;
	b->rc = 0;
	SwapSource src;
	std::vector<DownConverter *> dc;
	std::vector<LowPass *> f1, f2;
	std::vector<Demodulator *> dm;
	std::vector<SumSink *> sk;
	src.setSampleRate(b->fs);
	src.setChannels(2);
	src.setBlockSize(2 * b->nframes);
	src.spare.assign(b->iq, b->iq + 2 * b->nframes);
	for (size_t c = 0; c < b->ifs.size(); c++) {          /* radio.cxx:68-83 */
		dc.push_back(new DownConverter("dc"));
		f1.push_back(new LowPass("chan"));
		dm.push_back(new Demodulator("demod"));
		f2.push_back(new LowPass("audio"));
		sk.push_back(new SumSink());
		dc[c]->setIF(b->ifs[c]);
		f1[c]->setPassband(b->cpb);
		f1[c]->setOutputSampleRate(b->crate);
		dm[c]->setMode((Demodulator::Mode)b->mode);
		f2[c]->setPassband(b->apb);
		f2[c]->setOutputSampleRate(b->arate);
		src.connect(dc[c]);
		dc[c]->connect(f1[c]);
		f1[c]->connect(dm[c]);
		dm[c]->connect(f2[c]);
		f2[c]->connect(sk[c]);
	}
	pthread_mutex_lock(&g_plan_lock);
	if (!src.start())
		b->rc = -1;
	pthread_mutex_unlock(&g_plan_lock);
	if (b->rc == 0 && !src.run())                          /* one untimed block: buffers sized, pages touched */
		b->rc = -2;
	if (b->rc == 0)                                        /* the vector swapped out on that first call was the runtime's own, */
		src.spare.assign(b->iq, b->iq + 2 * b->nframes);   /* zero-filled one: from now on both hold the block */
	pthread_barrier_wait(b->ready);
	pthread_barrier_wait(b->go);
	for (unsigned int n = 0; b->rc == 0 && n < b->nblocks; n++)
		if (!src.run())
			b->rc = -3;
	pthread_barrier_wait(b->ready);                        /* (reused: everybody has finished) */
	pthread_mutex_lock(&g_plan_lock);
	src.stop();
	pthread_mutex_unlock(&g_plan_lock);
	b->sum = 0;
	b->frames = 0;
	for (size_t c = 0; c < sk.size(); c++) {
		b->sum += sk[c]->sum;
		b->frames += sk[c]->frames;
		delete dc[c]; delete f1[c]; delete dm[c]; delete f2[c]; delete sk[c];
	}
	return 0;
}

double mono()
{
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

}  // namespace

extern "C" {

/* seconds of wall time `nthreads` pipeline threads take to put `nblocks` blocks of `nframes` frames through the `nch`
 * receivers dealt among them (receiver c on thread c mod nthreads); < 0: a start()/run() failed.  *audio_frames (optional):
 * audio frames produced in all, *abs_sum: their absolute sum (so that nothing is optimised away, and for a look). */
double ref_bench_receivers(unsigned int fs, const int *ifs, unsigned int nch, unsigned int chan_passband, unsigned int chan_rate,
                           int mode, unsigned int audio_passband, unsigned int audio_rate, const float *iq, size_t nframes,
                           unsigned int nblocks, unsigned int nthreads, size_t *audio_frames, double *abs_sum)
{
	if (!nthreads || !nch || nthreads > nch)
		return -1.0;
	pthread_barrier_t ready, go;
	pthread_barrier_init(&ready, 0, nthreads + 1);
	pthread_barrier_init(&go, 0, nthreads + 1);
	std::vector<BenchThread> th(nthreads);
	for (unsigned int t = 0; t < nthreads; t++) {
		BenchThread &b = th[t];
		b.fs = fs; b.cpb = chan_passband; b.crate = chan_rate; b.apb = audio_passband; b.arate = audio_rate;
		b.mode = mode; b.nblocks = nblocks; b.iq = iq; b.nframes = nframes; b.ready = &ready; b.go = &go;
		for (unsigned int c = t; c < nch; c += nthreads)
			b.ifs.push_back(ifs[c]);
		pthread_create(&b.tid, 0, bench_thread, &b);
	}
	pthread_barrier_wait(&ready);
	const double t0 = mono();
	pthread_barrier_wait(&go);
	pthread_barrier_wait(&ready);
	const double dt = mono() - t0;
	double sum = 0;
	size_t frames = 0;
	int rc = 0;
	for (unsigned int t = 0; t < nthreads; t++) {
		pthread_join(th[t].tid, 0);
		sum += th[t].sum;
		frames += th[t].frames;
		if (th[t].rc)
			rc = th[t].rc;
	}
	pthread_barrier_destroy(&ready);
	pthread_barrier_destroy(&go);
	if (audio_frames)
		*audio_frames = frames;
	if (abs_sum)
		*abs_sum = sum;
	return rc ? (double)rc : dt;
}

}  // extern "C"
