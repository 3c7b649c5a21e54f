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
