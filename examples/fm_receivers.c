/*
 * fm_receivers.c -- the C ABI from plain C: no HIP, no C++, no Python in the client.
 *
 * Eight narrow-band FM receivers off one synthetic 2 Msps tuner stream (one carrier each, a
 * different audio tone per carrier), three blocks of 40 000 frames handed over in HOST memory the
 * way io/rtlsdrtuner.cxx hands blocks to DspSource consumers; the audio of every receiver comes
 * back and the tone each one recovers is measured with a single-bin DFT.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/fm_receivers.c -Lwebradio_amd/lib -lwebradio_amd -lm \
 *       -Wl,-rpath,$PWD/webradio_amd/lib -o /tmp/fm_receivers && /tmp/fm_receivers
 *
 * Exit code 0 and "ok" when every receiver hears its own tone (tests/test_gpu_c_client.py).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "webradio_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != WR_OK) { \
	fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, wr_last_error()); return 1; } } while (0)

enum { NRX = 8, FS = 2000000, BLOCK = 40000, NBLOCKS = 3, CHAN_RATE = 50000, AUDIO_RATE = 10000 };

int main(void)
{
	static float iq[2 * BLOCK];
	static float audio[NRX][NBLOCKS * BLOCK / (FS / AUDIO_RATE)];
	const double PI = 3.14159265358979323846;
	int if_hz[NRX], chan[NRX];
	double tone[NRX];
	wr_dev *dev = NULL;
	wr_tuner *tuner = NULL;

	CHECK(wr_dev_open(&dev, 0, NULL));
	CHECK(wr_tuner_create(&tuner, dev, FS, NRX, BLOCK, WR_NCO_ROTATE));
	for (int r = 0; r < NRX; r++) {
		if_hz[r] = -350000 + r * 100000;
		tone[r] = 300.0 + 100.0 * r;                     /* Hz */
		CHECK(wr_chan_add(tuner, &chan[r]));
		CHECK(wr_chan_set_if(tuner, chan[r], if_hz[r]));
		/* (the reference's design rule: bins below 64 * passband / rate / 2 pass, lowpass.cxx:167 -- a
		 * passband under rate / 32 gives an all-zero filter) */
		CHECK(wr_chan_set_filter(tuner, chan[r], WR_FILTER_CHANNEL, 100000, CHAN_RATE));
		CHECK(wr_chan_set_filter(tuner, chan[r], WR_FILTER_AUDIO, 4000, AUDIO_RATE));
		CHECK(wr_chan_set_mode(tuner, chan[r], WR_FM));
	}
	size_t have = 0;
	for (int b = 0; b < NBLOCKS; b++) {
		for (int n = 0; n < BLOCK; n++) {
			const double t = (double)(b * BLOCK + n) / FS;
			double re = 0.0, im = 0.0;
			for (int r = 0; r < NRX; r++) {            /* FM: 2 kHz deviation around each carrier */
				const double ph = 2.0 * PI * if_hz[r] * t + (2000.0 / tone[r]) * sin(2.0 * PI * tone[r] * t);
				re += 0.1 * cos(ph);
				im += 0.1 * sin(ph);
			}
			iq[2 * n] = (float)re;
			iq[2 * n + 1] = (float)im;
		}
		CHECK(wr_tuner_submit(tuner, iq, BLOCK, WR_HOST));
		size_t got = 0;
		for (int r = 0; r < NRX; r++) {
			CHECK(wr_chan_fetch(tuner, chan[r], WR_STAGE_AUDIO, audio[r] + have, BLOCK, &got));
			if (got != BLOCK / (FS / AUDIO_RATE)) {
				fprintf(stderr, "receiver %d: %zu audio frames\n", r, got);
				return 1;
			}
		}
		have += got;
	}
	int bad = 0;
	for (int r = 0; r < NRX; r++) {
		/* skip the filters' start-up, then compare the power at this receiver's tone with the power
		 * at its neighbours' tones */
		double best = 0.0;
		int best_r = -1;
		for (int q = 0; q < NRX; q++) {
			double sr = 0.0, si = 0.0;
			for (size_t n = 40; n < have; n++) {
				const double a = 2.0 * PI * tone[q] * (double)n / AUDIO_RATE;
				sr += audio[r][n] * cos(a);
				si += audio[r][n] * sin(a);
			}
			const double p = sr * sr + si * si;
			if (p > best) {
				best = p;
				best_r = q;
			}
		}
		printf("receiver %d at %+7d Hz hears %4.0f Hz%s\n", r, if_hz[r], tone[best_r], best_r == r ? "" : "  <-- not its own tone");
		bad += best_r != r;
	}
	CHECK(wr_tuner_destroy(tuner));
	CHECK(wr_dev_close(dev));
	puts(bad ? "MISMATCH" : "ok");
	return bad ? 1 : 0;
}
