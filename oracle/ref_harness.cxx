/*
 * ref_harness.cxx -- drives a DspBlock implementation through its PUBLIC API and
 * records what happened, as text.  TEST INFRASTRUCTURE ONLY.
 *
 * This one source is compiled twice, unchanged:
 *   1. against the real reference (/root/reference/src/dsp/{dspblock,demodulator}.cxx,
 *      which build from their own sources) -> oracle/_ref/libwr_ref.so
 *   2. against this repo's host runtime (webradio_amd/host/dspblock.{h,cxx})
 *      -> webradio_amd/host/libwr_host_harness.so   (WR_HARNESS_NO_DEMOD: the
 *      product Demodulator needs a GPU and is checked by the -m gpu tests)
 * tests/test_dspblock_parity.py requires the two traces to be identical, which
 * pins row a0 of SURVEY.md section 8 (DspBlock::connect/start/run/stop,
 * dspblock.cxx:57-76,106-151,153-167,169-212) to the reference's behaviour.
 *
 * It only uses what the reference's callers use: subclassing DspBlock/DspSource
 * with init/deinit/process (dspblock.h:82-84) and the public members
 * (dspblock.h:60-79,130-137).
 */
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>

#include "dspblock.h"
#ifndef WR_HARNESS_NO_DEMOD
#include "demodulator.h"
#endif

namespace {

std::string g_trace;

void tr(const char *fmt, ...)
{
	char line[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(line, sizeof(line), fmt, ap);
	va_end(ap);
	g_trace += line;
}

/* A block whose behaviour is scripted: requested output rate / channels,
 * whether init or the n-th process call fails.  It logs every callback. */
class ScriptBlock : public DspBlock {
public:
	ScriptBlock(const std::string &name, unsigned outRate = 0, unsigned outCh = 0)
		: DspBlock(name, "Script"), reqRate(outRate), reqCh(outCh),
		  failInit(false), failProcessAt(-1), nproc(0) {}
	unsigned reqRate, reqCh;
	bool failInit;
	int failProcessAt;
	int nproc;
protected:
	bool init() {
		tr("%s.init in=%u/%u\n", name().c_str(), inputSampleRate(), inputChannels());
		if (failInit)
			return false;
		if (reqRate)
			_outputSampleRate = reqRate;
		if (reqCh)
			_outputChannels = reqCh;
		return true;
	}
	void deinit() { tr("%s.deinit\n", name().c_str()); }
	bool process(const std::vector<sample_t> &in, std::vector<sample_t> &out) {
		double s = 0;
		for (size_t n = 0; n < in.size(); n++)
			s += in[n];
		tr("%s.process in=%zu out=%zu sum=%.1f dec=%u int=%u\n", name().c_str(),
		   in.size(), out.size(), s, decimation(), interpolation());
		for (size_t n = 0; n < out.size(); n++)
			out[n] = (float)(n % 7);
		return nproc++ != failProcessAt;
	}
};

class ScriptSource : public DspSource {
public:
	ScriptSource(const std::string &name) : DspSource(name, "ScriptSource"), counter(0) {}
	int counter;
protected:
	bool init() { tr("%s.init in=%u/%u bs=%u\n", name().c_str(), inputSampleRate(), inputChannels(), blockSize()); return true; }
	void deinit() { tr("%s.deinit\n", name().c_str()); }
	bool process(const std::vector<sample_t> &in, std::vector<sample_t> &out) {
		tr("%s.process in=%zu out=%zu\n", name().c_str(), in.size(), out.size());
		for (size_t n = 0; n < out.size(); n++)
			out[n] = (float)((counter + n) % 5);
		counter++;
		return true;
	}
};

void state(const char *tag, const DspBlock &b)
{
	tr("%s: run=%d in=%u/%u out=%u/%u dec=%u int=%u tin=%u tout=%u\n", tag, (int)b.isRunning(),
	   b.inputSampleRate(), b.inputChannels(), b.outputSampleRate(), b.outputChannels(),
	   b.decimation(), b.interpolation(), b.totalIn(), b.totalOut());
}

/* 0: plain decimating chain with fan-out, three blocks */
void scenario_chain()
{
	ScriptSource src("src");
	ScriptBlock a("a", 12000, 0), b("b", 0, 1), c("c", 3000, 0), d("d");
	src.setSampleRate(48000);
	src.setChannels(2);
	src.setBlockSize(64);
	src.connect(&a);
	a.connect(&b);
	a.connect(&c);
	c.connect(&d);
	tr("start=%d\n", (int)src.start());
	for (int n = 0; n < 3; n++)
		tr("run=%d\n", (int)src.run());
	state("src", src); state("a", a); state("b", b); state("c", c); state("d", d);
	src.stop();
	state("a-stopped", a);
}

/* 1: non-integer rate ratio: start fails, block is deinit'ed, nothing runs */
void scenario_bad_ratio()
{
	ScriptSource src("src");
	ScriptBlock a("a", 7000, 0), b("b");
	src.setSampleRate(48000);
	src.setChannels(2);
	src.setBlockSize(32);
	src.connect(&a);
	a.connect(&b);
	tr("start=%d\n", (int)src.start());
	state("src", src); state("a", a); state("b", b);
	tr("run=%d\n", (int)src.run());
	src.stop();
}

/* 2: downstream init failure tears the whole started subtree down */
void scenario_downstream_fail()
{
	ScriptSource src("src");
	ScriptBlock a("a", 24000, 0), b("b"), c("c"), d("d");
	c.failInit = true;
	src.setSampleRate(48000);
	src.setChannels(2);
	src.setBlockSize(32);
	src.connect(&a);
	a.connect(&b);
	a.connect(&c);
	src.connect(&d);
	tr("start=%d\n", (int)src.start());
	state("src", src); state("a", a); state("b", b); state("c", c); state("d", d);
	src.stop();
}

/* 3: a process() failure stops the walk and propagates false */
void scenario_process_fail()
{
	ScriptSource src("src");
	ScriptBlock a("a"), b("b"), c("c");
	a.failProcessAt = 1;
	src.setSampleRate(8000);
	src.setChannels(1);
	src.setBlockSize(16);
	src.connect(&a);
	a.connect(&b);
	src.connect(&c);
	tr("start=%d\n", (int)src.start());
	for (int n = 0; n < 3; n++)
		tr("run=%d\n", (int)src.run());
	state("a", a); state("b", b); state("c", c);
	src.stop();
}

/* 4: connect on a running block (Q9), duplicate connect, hot disconnect */
void scenario_hot_connect()
{
	ScriptSource src("src");
	ScriptBlock a("a", 24000, 0), late("late"), late2("late2", 12000, 0);
	src.setSampleRate(96000);
	src.setChannels(2);
	src.setBlockSize(64);
	src.connect(&a);
	src.connect(&a);                      /* duplicate: ignored */
	tr("start=%d\n", (int)src.start());
	a.connect(&late);                     /* started with stale (default) rates */
	a.connect(&late2);
	state("late", late); state("late2", late2);
	tr("run=%d\n", (int)src.run());
	a.disconnect(&late);
	state("late-disc", late);
	tr("run=%d\n", (int)src.run());
	src.setSampleRate(1000);              /* ignored while running */
	src.setBlockSize(8);                  /* ignored while running */
	tr("run=%d\n", (int)src.run());
	state("src", src);
	src.stop();
	src.setBlockSize(8);
	tr("bs=%u\n", src.blockSize());
}

/* 5: interpolation and odd block sizes (truncating frame arithmetic) */
void scenario_interp()
{
	ScriptSource src("src");
	ScriptBlock up("up", 32000, 0), down("down", 6400, 0), odd("odd", 2000, 1);
	src.setSampleRate(8000);
	src.setChannels(2);
	src.setBlockSize(30);
	src.connect(&up);
	up.connect(&down);
	src.connect(&odd);
	tr("start=%d\n", (int)src.start());
	tr("run=%d\n", (int)src.run());
	tr("run=%d\n", (int)src.run());
	state("up", up); state("down", down); state("odd", odd);
	src.stop();
}

/* 6: not started; restart keeps working; destructor stops a running block */
void scenario_restart()
{
	ScriptSource src("src");
	ScriptBlock a("a", 4000, 0);
	src.setSampleRate(8000);
	src.setChannels(2);
	src.setBlockSize(16);
	src.connect(&a);
	tr("run-before-start=%d\n", (int)src.run());
	tr("start=%d\n", (int)src.start());
	tr("run=%d\n", (int)src.run());
	src.stop();
	src.stop();
	tr("run-after-stop=%d\n", (int)src.run());
	tr("start=%d\n", (int)src.start());
	tr("run=%d\n", (int)src.run());
	state("a", a);
	{
		ScriptBlock tmp("tmp");
		src.connect(&tmp);
		state("tmp", tmp);
		src.disconnect(&tmp);
	}
	tr("run=%d\n", (int)src.run());
	src.stop();
}

typedef void (*scenario_fn)();
scenario_fn g_scenarios[] = {
	scenario_chain, scenario_bad_ratio, scenario_downstream_fail, scenario_process_fail,
	scenario_hot_connect, scenario_interp, scenario_restart,
};

#ifndef WR_HARNESS_NO_DEMOD
class VecSource : public DspSource {
public:
	VecSource(const float *d, size_t n) : DspSource("vec", "VecSource"), data(d), total(n), pos(0) {}
	const float *data;
	size_t total, pos;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const std::vector<sample_t> &, std::vector<sample_t> &out) {
		if (pos + out.size() > total)
			return false;
		memcpy(out.data(), data + pos, out.size() * sizeof(float));
		pos += out.size();
		return true;
	}
};

class Capture : public DspBlock {
public:
	Capture() : DspBlock("cap", "Capture") {}
	std::vector<float> got;
protected:
	bool init() { return true; }
	void deinit() {}
	bool process(const std::vector<sample_t> &in, std::vector<sample_t> &) {
		got.insert(got.end(), in.begin(), in.end());
		return true;
	}
};
#endif

} // namespace

extern "C" {

int wr_harness_scenarios(void)
{
	return (int)(sizeof(g_scenarios) / sizeof(g_scenarios[0]));
}

/* Runs scenario `idx`; copies the trace (NUL terminated) into buf.  Returns the
 * trace length, or -1 for a bad index. */
long wr_harness_run(int idx, char *buf, size_t buflen)
{
	if (idx < 0 || idx >= wr_harness_scenarios())
		return -1;
	g_trace.clear();
	g_scenarios[idx]();
	if (buf && buflen) {
		size_t n = g_trace.size() < buflen - 1 ? g_trace.size() : buflen - 1;
		memcpy(buf, g_trace.data(), n);
		buf[n] = 0;
	}
	return (long)g_trace.size();
}

#ifndef WR_HARNESS_NO_DEMOD
/* Reference Demodulator (dsp/demodulator.cxx) driven through DspSource::run in
 * blocks of `block_frames` IQ frames; `mode_switch_at` >= 0 switches to
 * `mode2` before that block index (setMode is legal while running,
 * demodulator.h:49).  Returns the number of output samples written. */
long wr_ref_demod(const char *mode, const float *iq, size_t nframes, size_t block_frames,
                  int mode_switch_at, const char *mode2, float *out, size_t out_cap)
{
	VecSource src(iq, nframes * 2);
	Demodulator dem("dem");
	Capture cap;
	if (!dem.setModeString(mode))
		return -1;
	src.setSampleRate(240000);
	src.setChannels(2);
	src.setBlockSize((unsigned)(block_frames * 2));
	src.connect(&dem);
	dem.connect(&cap);
	if (!src.start())
		return -2;
	size_t nblocks = nframes / block_frames;
	for (size_t b = 0; b < nblocks; b++) {
		if ((long)b == (long)mode_switch_at && mode2)
			dem.setModeString(mode2);
		if (!src.run())
			return -3;
	}
	src.stop();
	size_t n = cap.got.size() < out_cap ? cap.got.size() : out_cap;
	memcpy(out, cap.got.data(), n * sizeof(float));
	return (long)cap.got.size();
}
#endif

} // extern "C"
