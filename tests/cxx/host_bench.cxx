/*
 * host_bench.cxx -- the drop-in path at BASELINE config-2 scale: one FrontEnd, 256 Receivers
 * wired exactly as radio.cxx wires them, Radio::run() pumping 4 000 000-frame blocks.  What a
 * webradio process would see: the tuner block lives in HOST memory (PCIe is inside the
 * timing), every Receiver's AudioStreamManager gets its audio.  TEST/MEASUREMENT DRIVER.
 *
 *   host_bench [receivers=256] [blocks=10] [block_frames=4000000] [f32|u8|dev]
 *
 * u8: the source holds its block in the RTL-SDR byte format (RawU8Block, like FileTuner: what an RTL-SDR or a
 * recording delivers, io/rtlsdrtuner.cxx:86-117) -- 8 MB per 4 M-frame block over PCIe instead of 32 MB.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "filetuner.h"
#include "gpubatch.h"
#include "radio.h"

namespace {

std::vector<float> g_block;

/* a tuner that hands out the same synthetic block every time.  Like RtlSdrTuner::process
 * (io/rtlsdrtuner.cxx:265-285) it SWAPS a filled buffer into the output vector instead of
 * copying: what is timed is the framework, not a 32 MB memcpy of the source's own. */
class SynthTuner : public Tuner {
public:
	SynthTuner(const string &n) : Tuner(n, "SynthTuner") {}
protected:
	bool init() { spare = g_block; primed = false; return true; }
	void deinit() {}
	bool process(const vector<sample_t> &, vector<sample_t> &out) {
		if (spare.size() != out.size())
			return false;
		out.swap(spare);
		if (!primed) {                     /* the vector swapped out was the runtime's empty one */
			spare = g_block;
			primed = true;
		}
		return true;
	}
	vector<float> spare;
	bool primed;
};
Tuner *make(const string &n) { return new SynthTuner(n); }

std::vector<uint8_t> g_bytes;

/* the same block quantised to the RTL-SDR byte format; like FileTuner it leaves the float vector unfilled when
 * every consumer reads the block on the device */
class SynthU8Tuner : public Tuner, public RawU8Block {
public:
	SynthU8Tuner(const string &n) : Tuner(n, "SynthU8Tuner") {}
	const uint8_t *rawU8(size_t *frames) const { if (frames) *frames = g_bytes.size() / 2; return buf[cur].data(); }
	unsigned int rawU8Buffers() const { return 2; }        /* like FileTuner: blocks alternate between two buffers */
protected:
	bool init() { buf[0] = g_bytes; buf[1] = g_bytes; cur = 0; return true; }
	void deinit() {}
	bool process(const vector<sample_t> &, vector<sample_t> &out) {
		if (out.size() != g_bytes.size())
			return false;
		cur ^= 1u;
		const bool skip = consumersReadOnDevice();
		if (!skip)
			for (size_t n = 0; n < out.size(); n++)
				out[n] = ((float)g_bytes[n] - 128.0) / 128.0;
		setHostBlockValid(!skip);
		return true;
	}
	std::vector<uint8_t> buf[2];
	unsigned int cur;
};
Tuner *make_u8(const string &n) { return new SynthU8Tuner(n); }

/* r06: the same block RESIDENT in GPU memory (three copies taken in turn, as a capture card with a DMA ring would leave
 * them): a DeviceBlock source -- nothing crosses PCIe, and the tuner batch streams it (wr_tuner_set_streaming: the kernel
 * bench.py's headline times, through Radio::run() and 256 Receivers) */
class SynthDeviceTuner : public Tuner, public DeviceBlock {
public:
	SynthDeviceTuner(const string &n) : Tuner(n, "SynthDeviceTuner"), dev(NULL), cur(0) { memset(copy, 0, sizeof(copy)); }
	~SynthDeviceTuner() { for (int i = 0; i < 3; i++) if (copy[i]) wr_dev_free(dev, copy[i]); }
	const float *deviceBlock(wr_dev **d, size_t *frames) const {
		if (d) *d = dev;
		if (frames) *frames = g_block.size() / 2;
		return (const float *)copy[cur];
	}
protected:
	bool init() {
		dev = wrhost::deviceFor(this);
		if (!dev)
			return false;
		for (int i = 0; i < 3; i++)
			if (!copy[i] && (wr_dev_malloc(dev, g_block.size() * sizeof(float), &copy[i]) != WR_OK ||
			                 wr_dev_upload(dev, copy[i], g_block.data(), g_block.size() * sizeof(float)) != WR_OK))
				return false;
		return true;
	}
	void deinit() {}
	bool process(const vector<sample_t> &, vector<sample_t> &out) {
		if (out.size() != g_block.size())
			return false;
		cur = (cur + 1u) % 3u;
		const bool skip = consumersReadOnDevice();
		if (!skip)
			out = g_block;
		setHostBlockValid(!skip);
		return true;
	}
	wr_dev *dev;
	void *copy[3];
	unsigned int cur;
};
Tuner *make_dev(const string &n) { return new SynthDeviceTuner(n); }

double now()
{
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

} // namespace

int main(int argc, char **argv)
{
	const unsigned int nrx = argc > 1 ? atoi(argv[1]) : 256;
	const unsigned int blocks = argc > 2 ? atoi(argv[2]) : 10;
	const unsigned int frames = argc > 3 ? atoi(argv[3]) : 4000000;
	const bool u8 = argc > 4 && !strcmp(argv[4], "u8");
	const bool resident = argc > 4 && !strcmp(argv[4], "dev");
	const unsigned int fs = 100000000;

	g_block.resize((size_t)frames * 2);
	unsigned int lcg = 12345;
	for (size_t n = 0; n < g_block.size(); n++) {
		lcg = lcg * 1664525u + 1013904223u;
		g_block[n] = 0.01f * (float)((int)(lcg >> 8) - (1 << 23)) / (float)(1 << 23);
	}
	for (unsigned int c = 0; c < nrx; c += 4) {              /* carriers on every 4th channel */
		double f = -39843750.0 + 312500.0 * c, a = 0.5 / 64;
		for (unsigned int n = 0; n < frames; n++) {
			double ph = 2.0 * M_PI * fmod(f * (double)n / fs, 1.0);
			g_block[2 * n] += (float)(a * cos(ph));
			g_block[2 * n + 1] += (float)(a * sin(ph));
		}
		if (c >= 16)
			break;                                           /* a few carriers are enough */
	}

	if (u8) {
		g_bytes.resize(g_block.size());
		for (size_t n = 0; n < g_block.size(); n++) {
			double q = floor(127.5 + 127.0 * 16.0 * g_block[n] + 0.5);        /* (x16: the synthetic block is quiet) */
			g_bytes[n] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
		}
	}
	FrontEnd *fe = new FrontEnd(u8 ? make_u8 : resident ? make_dev : make);
	fe->tuner()->setSampleRate(fs);
	fe->tuner()->setChannels(2);
	fe->tuner()->setBlockSize(frames * 2);
	std::vector<Receiver *> rx;
	for (unsigned int c = 0; c < nrx; c++) {
		Receiver *r = new Receiver();
		r->downconverter()->setIF(-39843750 + 312500 * (int)c);
		r->channelFilter()->setPassband(6400000);
		r->channelFilter()->setOutputSampleRate(250000);
		r->audioFilter()->setPassband(8000);
		r->audioFilter()->setOutputSampleRate(50000);
		r->demodulator()->setMode(Demodulator::FM);
		r->stream()->setCapacity(256);          /* the sink stands for the MP3 encoder: keep a tail for the checksum */
		r->setFrontEnd(fe);
		rx.push_back(r);
	}
	if (!fe->tuner()->start()) {
		fprintf(stderr, "start failed\n");
		return 1;
	}
	for (int w = 0; w < 8; w++)                              /* warm-up (allocations, page locks, the clocks' ramp) */
		Radio::run();
	std::vector<double> per;                                 /* WR_HOST_BENCH_TIMES=1: every run()'s own duration */
	const bool times = getenv("WR_HOST_BENCH_TIMES") != NULL;
	const double t0 = now();
	for (unsigned int b = 0; b < blocks; b++) {
		const double a = times ? now() : 0.0;
		Radio::run();
		if (times)
			per.push_back(now() - a);
	}
	const double dt = now() - t0;
	if (times) {
		fprintf(stderr, "run() us:");
		for (size_t n = 0; n < per.size() && n < 48; n++)
			fprintf(stderr, " %.0f", per[n] * 1e6);
		fprintf(stderr, "\n");
	}
	double sum = 0;
	unsigned long total = 0;
	for (size_t n = 0; n < rx.size(); n++) {
		total += rx[n]->stream()->totalSamples();
		const vector<float> &a = rx[n]->stream()->samples();
		for (size_t i = 0; i < a.size(); i++)
			sum += fabs(a[i]);
	}
	bool live = false;
	unsigned long long launches = 0, streamed = 0;
	(void)wrhost::streamInfo(rx.empty() ? NULL : rx[0]->downconverter(), &live, &launches, &streamed);
	printf("{\"source\": \"%s\", \"audio\": \"%s\", \"receivers\": %u, \"blocks\": %u, \"block_frames\": %u, \"ms_per_block\": %.3f, "
	       "\"msps_tuner_input\": %.1f, \"audio_samples_per_receiver\": %lu, \"audio_abs_sum\": %.3f, "
	       "\"stream_info\": {\"live\": %s, \"launches\": %llu, \"blocks\": %llu}}\n",
	       u8 ? "u8 (RTL-SDR byte format, 2 B per frame over PCIe)" : resident ? "device (the block resident in GPU memory: DeviceBlock)" : "f32 (8 B per frame over PCIe)",
	       (getenv("WEBRADIO_AUDIO_LATE") && atoi(getenv("WEBRADIO_AUDIO_LATE")) >= 2) ? "two blocks late (WEBRADIO_AUDIO_LATE=2: one launch per block)"
	       : (getenv("WEBRADIO_AUDIO_LATE") && atoi(getenv("WEBRADIO_AUDIO_LATE"))) ? "one block late (WEBRADIO_AUDIO_LATE=1)" : "on time",
	       nrx, blocks, frames, dt / blocks * 1e3, (double)frames * blocks / dt / 1e6,
	       total / (unsigned long)rx.size(), sum, live ? "true" : "false", launches, streamed);
	if (getenv("WR_HOST_BENCH_PROFILE")) {
		/* DspBlock's own profiler (process-CPU ns inside process(), dspblock.h:69-75), summed per block type */
		uint64_t mix = 0, f1 = 0, dem = 0, f2 = 0, snk = 0;
		for (size_t n = 0; n < rx.size(); n++) {
			mix += rx[n]->downconverter()->totalNanoseconds();
			f1 += rx[n]->channelFilter()->totalNanoseconds();
			dem += rx[n]->demodulator()->totalNanoseconds();
			f2 += rx[n]->audioFilter()->totalNanoseconds();
			snk += rx[n]->stream()->totalNanoseconds();
		}
		const double per = 1e-3 / (blocks + 2);
		fprintf(stderr, "process() CPU us per block, all receivers: mixer %.1f  chan filter %.1f  demod %.1f  audio filter %.1f  "
		        "sink %.1f  tuner %.1f  spectrum %.1f\n", mix * per, f1 * per, dem * per, f2 * per, snk * per,
		        fe->tuner()->totalNanoseconds() * per, fe->spectrum()->totalNanoseconds() * per);
	}
	fe->tuner()->stop();
	for (size_t n = 0; n < rx.size(); n++)
		delete rx[n];
	delete fe;
	return 0;
}
