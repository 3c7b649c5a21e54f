/*
 * cpu_host_checks.cxx -- host-only logic of the runtime that needs no GPU: FileTuner
 * (read, (u8-128)/128 conversion, loop, end of file, raw byte access) driven through
 * DspSource::run() into a capture sink.  Built from dspblock.cxx + filetuner.cxx only.
 */
#include <string.h>

#include <vector>

#include "filetuner.h"

namespace {
class Capture : public DspBlock {
public:
	Capture() : DspBlock("cap", "Capture") {}
	std::vector<float> got;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const vector<sample_t> &in, vector<sample_t> &) {
		got.insert(got.end(), in.begin(), in.end());
		return true;
	}
};
}

extern "C" long wr_filetuner_play(const char *path, unsigned int block_frames, unsigned int nruns, int loop,
                                  float *out, size_t cap, int *ok_runs, unsigned char *last_raw, size_t *raw_frames)
{
	FileTuner t("f");
	Capture c;
	t.setSubdevice(path);
	t.setSampleRate(2048000);
	t.setChannels(2);
	t.setBlockSize(block_frames * 2);
	t.setLoop(loop != 0);
	t.connect(&c);
	*ok_runs = 0;
	if (!t.start())
		return -1;
	for (unsigned int n = 0; n < nruns; n++)
		if (t.run())
			(*ok_runs)++;
	size_t fr = 0;
	const uint8_t *raw = t.rawU8(&fr);
	*raw_frames = fr;
	if (raw && last_raw)
		memcpy(last_raw, raw, fr * 2);
	t.stop();
	size_t n = c.got.size() < cap ? c.got.size() : cap;
	memcpy(out, c.got.data(), n * sizeof(float));
	return (long)c.got.size();
}

/* RawU8Block::rawU8Buffers() = 2 is a promise: the bytes handed out for one block stay untouched until the source starts
 * producing the block after the next (the GPU runtime lets a transfer out of them run that long).  Returns 0 when, over
 * `nruns` blocks, consecutive blocks come out of different buffers and block r's bytes are still there after run r + 1. */
extern "C" int wr_filetuner_two_buffers(const char *path, unsigned int block_frames, unsigned int nruns)
{
	FileTuner t("f");
	Capture c;
	t.setSubdevice(path);
	t.setSampleRate(2048000);
	t.setChannels(2);
	t.setBlockSize(block_frames * 2);
	t.setLoop(true);
	t.connect(&c);
	if (t.rawU8Buffers() != 2)
		return 1;
	if (!t.start())
		return 2;
	const uint8_t *prev = NULL;
	std::vector<uint8_t> prev_copy;
	int rc = 0;
	for (unsigned int n = 0; n < nruns && !rc; n++) {
		if (!t.run()) {
			rc = 3;
			break;
		}
		size_t fr = 0;
		const uint8_t *raw = t.rawU8(&fr);
		if (!raw || fr != block_frames)
			rc = 4;
		else if (prev && raw == prev)
			rc = 5;                                        /* the same buffer twice in a row */
		else if (prev && memcmp(prev, prev_copy.data(), prev_copy.size()))
			rc = 6;                                        /* the block before was overwritten */
		if (!rc) {
			prev = raw;
			prev_copy.assign(raw, raw + fr * 2);
		}
	}
	t.stop();
	return rc;
}

/* ---- device hand-over bookkeeping of DspBlock (no GPU involved: pointers are just tokens) ---- */
namespace {
struct Src : public DspSource {
	Src() : DspSource("s", "Src") {}
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const vector<sample_t> &, vector<sample_t> &out) {
		for (size_t n = 0; n < out.size(); n++)
			out[n] = (float)n;
		return true;
	}
};
/* pretends to compute on a device: publishes a token, fills the host vector only if asked to */
struct DevStage : public DspBlock {
	DevStage(const char *n, bool takesDevice) : DspBlock(n, "DevStage"), takes(takesDevice), hostFills(0),
	                                            sawUpstream(NULL), sawHostFrames(0) {}
	bool takes;
	int token;
	int hostFills;
	const void *sawUpstream;
	size_t sawHostFrames;
protected:
	bool init() { _outputSampleRate = inputSampleRate(); _outputChannels = inputChannels(); return true; }
	void deinit() {}
	bool acceptsDeviceInput() const { return takes; }
	bool process(const vector<sample_t> &in, vector<sample_t> &out) {
		sawUpstream = upstreamDeviceOutput();
		sawHostFrames = in.size() / 2;
		publishDeviceOutput(&token);
		const bool onHost = hostOutputNeeded();
		elideOutput(!onHost);
		if (onHost) {
			out.assign((size_t)currentOutputFrames() * 2, 1.0f);
			hostFills++;
		}
		return true;
	}
};
}

/* returns a bit mask of failed expectations (0 = all good) */
extern "C" int wr_handover_checks(void)
{
	int bad = 0;
	Src src;
	src.setSampleRate(48000);
	src.setChannels(2);
	src.setBlockSize(64);
	DevStage a("a", true), b("b", true), hostsink("h", false);
	src.connect(&a);
	a.connect(&b);
	if (!src.start())
		return -1;
	src.run();
	src.run();
	/* a is fed by the source: no device input, full host block; its only consumer takes device
	 * input, so a never fills its host output, and b sees a's token and an EMPTY host vector
	 * that still stands for 32 frames (b has no consumer: nobody needs its output on the host) */
	if (a.sawUpstream != NULL || a.sawHostFrames != 32) bad |= 1;
	if (a.hostFills != 0) bad |= 2;
	if (b.sawUpstream != (const void *)&a.token || b.sawHostFrames != 0) bad |= 4;
	if (b.hostFills != 0) bad |= 8;
	/* a host consumer joins b while running (hot-connect, quirk Q9): from then on b fills its host output */
	b.connect(&hostsink);
	src.run();
	if (b.hostFills != 1 || hostsink.sawHostFrames != 32) bad |= 16;
	if (hostsink.sawUpstream != (const void *)&b.token) bad |= 32;       /* published all the same */
	/* and a second, host-only consumer of a makes a fill its output too, while b keeps reading the token */
	DevStage tap("t", false);
	a.connect(&tap);
	src.run();
	if (a.hostFills != 1 || tap.sawHostFrames != 32 || b.sawUpstream != (const void *)&a.token) bad |= 64;
	src.stop();
	return bad;
}
