/*
 * wr_internal.h -- internal C++ interface between the C ABI (wr_capi.hip), the host
 * design math (wr_design.cpp) and the kernels (wr_kernels.hip, wr_fft.hip).
 * Not installed; the public boundary is include/webradio_amd.h.
 */
#ifndef WR_INTERNAL_H_
#define WR_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "webradio_amd.h"

#define WR_HIST      (WR_FIR_LENGTH - 1)   /* 63 frames of FIR history, lowpass.cxx:133 */
#define WR_LANES     64                    /* wavefront width on gfx950 */
#define WR_SPLIT_N   256                   /* entries of the coarse and of the fine NCO table */
#define WR_MAX_DEVICES 64                  /* device indices the per-device tables of this library hold */
#define WR_TAPSETS   4                     /* distinct channel filters a lane group may mix in the fast DDC kernel:
                                              receivers of one tuner mostly share a passband (radio.cxx:78-79),
                                              the UI lets each choose its own (receiverhandler.cxx:130-137) */

/* ---- what other translation units need of a wr_dev (wr_capi.hip) ---- */
int         wrc_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));   /* sets wr_last_error() */
int         wrc_dev_index(const wr_dev *dev);
hipStream_t wrc_dev_stream(const wr_dev *dev);
/* the event the tuner's last launch stamped on completion (wr_tuner_mark_launches), or NULL (wr_ring.hip) */
int         wrc_tuner_launch_mark(wr_tuner *t, hipEvent_t *ev);

/* ---- host design math (wr_design.cpp) ---- */
void     wrd_sin_table(float *table);
void     wrd_sin_table_rounded(float *table);
int      wrd_phase_step(int if_hz, unsigned int input_rate);
unsigned wrd_lowpass_maxbin(unsigned int fir_length, unsigned int passband, unsigned int input_rate);
void     wrd_lowpass_design(unsigned int fir_length, unsigned int passband, unsigned int input_rate, float *coeff);
void     wrd_spectrum_window(unsigned int n, float *window);
void     wrd_split_tables(float *hi_cs /* [256][2] cos,sin */, float *lo_cs /* [256][2] */);
void     wrd_twiddles(unsigned int n, float *tw /* [n/2][2] cos,-sin of 2*pi*k/n */);

/* ---- per-slot parameter block of one rate group of a tuner, device SoA ---- */
struct WrGroupDev {
	/* all arrays have `slots` entries (slots is a multiple of 64) unless noted */
	/* the arrays a block CHANGES come as two sets: a launch reads set `sp` and writes set
	 * `sp ^ 1`, so consecutive blocks' DDC launches depend only on each other */
	unsigned int *phase[2];     /* left-aligned phase at block start: DownConverter::phase << 1 */
	unsigned int *step;         /* phaseStep << 1 (two's complement) for this block */
	float        *hist_cs[2];   /* [63][slots][2] LO (cos, sin) each of the last 63 input frames was mixed with;
	                               WR_NCO_ROTATE: the TURN (cos, sin) into frame r - 62 of the next block */
	float        *hist_lo[2];   /* WR_NCO_ROTATE only: [63][slots][2] the LO values (segment anchors) */
	int          *flags;        /* bit0: slot active (host-written only) */
	int          *mode;         /* wr_mode, or -1 for an idle slot (what the post-DDC kernels test) */
	float        *taps1;        /* [64][slots] channel-filter taps, taps1[j*slots+s] = coeff[j] */
	float        *taps2;        /* [64][slots] audio-filter taps */
	float        *taps2u;       /* [lane groups][64] the group's one audio filter, where its channels share one
	                               (WrTunerLaunch::uniform2_mask) */
	/* what a DDC wave needs before its first tap, laid out so that it is ONE coalesced load each (a
	 * wave that gathers them -- 64 tap rows, four table entries per lane -- queues 320 cache-line
	 * requests behind the other 31 waves of its CU, and every wave of the launch does so at once) */
	float        *rot;          /* WR_NCO_ROTATE: [slots][4] the two turns of the slot's step:
	                               cis(2 pi S / 65536), cis(2 pi (S+1) / 65536), S = step >> 16 */
	float        *taps1u;       /* [groups][WR_TAPSETS][64] the distinct channel filters of lane group g (when
	                               there are at most WR_TAPSETS): taps1u[(g*WR_TAPSETS + q)*64 + j] =
	                               coeff_q[63 - j], the tap of window sample j */
	int          *tapsel;       /* [slots] which of its group's filters the slot's channel uses */
	/* optional SECOND channel-filter stage between the DDC and the demodulator (one
	 * 64-tap stage cannot cut a 12.5 kHz channel out of a 100 Msps stream, lowpass.cxx:167 gives bin 0) */
	float        *taps1b;       /* [64][slots] its taps */
	float        *iq2_hist[2];  /* [63][slots][2] its history: the last 63 first-stage frames; ping-pong */
	float        *chan_iq2[2];  /* [k1max / d1b][slots][2] its output = the demodulator's input */
	/* receiver controls the reference only stubs (receiverhandler.cxx:112,118-119,127) */
	/* a channel filter of MORE than 64 taps (128 or 256; the rate group is keyed by the length too): its taps
	 * and, as LowPass::block does (lowpass.cxx:138-142), the last L - 1 MIXED frames of every channel */
	float        *taps1L;       /* [L][slots] */
	float        *mixhist[2];   /* [L - 1][slots][2], ping-pong with the state set */
	float        *gain;         /* [slots] af_gain as a factor (1 = 0 dB) */
	float        *squelch;      /* [slots] squelch threshold as a power (0 = open) */
	float        *prev_iq[2];   /* [slots][2] Demodulator::prev_i/prev_q, ping-pong by block parity */
	float        *chan_iq[2];   /* [k1max][slots][2] channel-filter output, time major; double buffered so
	                               that block b+1's DDC can run while block b is being demodulated */
	float        *dem[2];       /* [63 + k1max][slots] demod output, 63 history rows in front; ping-pong */
	float        *audio;        /* [slots][k2max] audio, channel major */
	float        *audio_set[4]; /* four arrays of that shape, `audio` one of them: inside a streaming launch the blocks' post stages store
	                               by turns (block j into set member (first + j) mod 4), so that blocks whose post-stage tasks run side by
	                               side never store into the same array; the stream's close makes the array its last block wrote
	                               `audio` (wr_capi.hip: stream_close) */
	/* taps of the group's audio filter and of its second channel stage: 64, or 128 / 256 (the rate group is keyed by
	 * them; dem then carries l2 - 1 history rows, iq2_hist l1b - 1, taps2 / taps1b l2 / l1b rows) */
	unsigned int  l2 = 64, l1b = 64;
};

struct WrTunerLaunch {
	const float *cur;           /* this block's IQ, nframes frames (device), or NULL when ... */
	const uint8_t *cur_u8;      /* ... the block is in the RTL-SDR byte format (2 bytes per frame) */
	const float *hist;          /* last 63 IQ frames of the previous block (device) */
	float       *hist_next;     /* receives the history for the next block */
	int          parity;        /* which of the group's prev_iq / dem ping-pong buffers is current */
	int          sp;            /* which state set (phase, LO history) this block reads */
	int          cb;            /* which chan_iq buffer this block writes */
	size_t       nframes;
	unsigned int d1, d2;
	unsigned int slots;         /* row stride of every per-slot array */
	unsigned int slots_used;    /* slots [0, slots_used) hold channels (multiple of 64) */
	size_t       k1, k2;        /* frames per channel at channel / audio rate for this block */
	size_t       k2max;         /* channel stride of audio */
	int          nco_mode;
	float        audio_scale;   /* multiplies the audio on store (1 = as the reference) */
	int          use_gain, use_squelch;   /* some channel has an af_gain / a squelch threshold set */
	unsigned long long uniform_mask;   /* bit g: all channels of lane group g share ONE channel filter */
	unsigned long long uniform2_mask;  /* bit g: ... and ONE audio filter (WrGroupDev::taps2u) */
	unsigned long long fewsets_mask;   /* bit g: lane group g has at most WR_TAPSETS distinct channel filters */
	unsigned char nsets[64];           /* distinct channel filters of lane group g (valid where fewsets_mask says) */
	int          one_filter;           /* every channel of the rate group uses ONE and the same channel filter */
	void        *ev_start, *ev_stop;   /* hipEvent_t pair the DDC launch itself stamps (profiling), or NULL */
	bool         seeking = false;      /* the first block after wr_tuner_seek: phase = frame * step in closed form, the state sets
	                                      handed to the launch are all-zero ones (wr_capi.hip: seek_pending) */
	unsigned int seek_lo = 0;          /* the frame's low 32 bits */
};

/* ---- kernel launchers (wr_kernels.hip); all return hipError_t ---- */
hipError_t wrk_mix(hipStream_t st, const float *in, float *out, size_t nframes,
                   unsigned int phase, int step, const float *table_dev);
hipError_t wrk_fir(hipStream_t st, const float *in, size_t nframes, unsigned int channels,
                   unsigned int decim, unsigned int fir_length, const float *coeff_dev, const float *hist_dev,
                   float *out);
hipError_t wrk_hist_update(hipStream_t st, const float *in, size_t nframes, unsigned int channels,
                           unsigned int fir_length, float *hist_dev, float *scratch_dev);
hipError_t wrk_demod(hipStream_t st, int mode, const float *in, size_t nframes, float prev_i,
                     float prev_q, float *out);
hipError_t wrk_u8_to_f32(hipStream_t st, const uint8_t *in, float *out, size_t count);
hipError_t wrk_stage_windows(hipStream_t st, const void *src_mapped, bool u8, float *dst, size_t nframes, unsigned int period,
                             unsigned int len, size_t tail_frames);

/* what one launch of the post stage (demodulator + audio filter of ONE block) works on */
struct WrPostArgs {
	const float *chan_iq;        /* [k1][slots][2] the block's channel IQ */
	unsigned int k1, slots;
	const int   *mode;
	const float *prev_iq;        /* [slots][2] frame before the block's first */
	float       *prev_next;
	const float *dem_hist;       /* [63][slots] audio filter history ([64 nseg - 1] rows) */
	float       *dem_hist_next;
	size_t       k2;
	unsigned int tiles;          /* tiles of POST_TK audio frames per lane group */
	unsigned int run;            /* consecutive tiles one workgroup takes */
	unsigned int ntiles;         /* workgroups per lane group = ceil(tiles / run), plus the one that writes the state */
	const float *taps2;
	const float *taps2u;         /* [groups][64], read through the scalar cache */
	unsigned long long uni2;     /* bit g: lane group g's channels share one audio filter (taps2u) */
	float       *audio;
	size_t       k2max;
	float        scale;
	unsigned int d2;
	unsigned int groups;         /* lane groups in use */
	const float *gain;           /* [slots] af_gain factors, or NULL: all 1 */
	const float *squelch;        /* [slots] squelch thresholds (power of the demodulator's input, mean over the
	                                d2 frames behind an audio frame), or NULL: all open */
	float       *audio_host;     /* r04: page-locked HOST memory (a slot of the tuner's audio ring, mapped) that gets every
	                                audio sample too, rows `host_stride` floats apart -- the block's audio is in the ring when
	                                the launch has run, no device-to-host copy behind it; NULL: device memory only */
	size_t       host_stride;
	unsigned int nseg = 1;       /* the audio filter's taps / 64: 1, or 2 / 4 (dem_hist then has 127 / 255 rows, taps2 128 / 256,
	                                taps2u [groups][taps]) */
	const float *chan_prev = nullptr;   /* r05 (the streaming launch): the channel IQ of the block BEFORE this one, [k1][slots][2], k1 >= 64 --
	                                the 63 rows of audio-filter history and the demodulator's previous frame are then made from
	                                ITS last 64 rows (the same operations on the same frames: the same bits) instead of read
	                                from dem_hist / prev_iq, so that a block's post stage does not wait for the one before */
};
/* `post` (optional): the post stage of the PREVIOUS block, run by extra workgroups of the same
 * launch beside this block's DDC (only taken up by the ROTATE / uniform-taps kernel; *post_taken
 * says whether it was) */
/* WR_TUNE_DDC_NG2_MIN_PASSES: set (value < 0: default again) and/or read; returns the value in force before */
long wrk_tune_ng2_min_passes(long value, bool set);
hipError_t wrk_tuner_ddc(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G,
                         const float *table_dev, const float *hi_dev, const float *lo_dev,
                         int num_cus, const WrPostArgs *post = nullptr, bool *post_taken = nullptr);
/* the same for a channel filter of `len` = 128 or 256 taps (k_tuner_ddc_long: the reference's arithmetic in every
 * nco mode); also rolls the group's state (phase, mixed history) into the other set */
hipError_t wrk_tuner_ddc_long(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G, unsigned int len,
                              const float *table_dev, int num_cus, bool rotate, bool rotate_one_filter, const float *hi_dev,
                              const float *lo_dev, const WrPostArgs *post = nullptr, bool *post_taken = nullptr);
hipError_t wrk_tuner_demod(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G);
/* second channel-filter stage: k1a first-stage frames of chan_iq[cb] -> k1a / d1b frames of chan_iq2[cb];
 * history from iq2_hist[p2], next history into iq2_hist[p2 ^ 1] */
hipError_t wrk_tuner_iq2(hipStream_t st, const WrGroupDev &G, unsigned int slots, unsigned int slots_used,
                         size_t k1a, unsigned int d1b, int cb, int p2);
hipError_t wrk_tuner_audio(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G);
WrPostArgs wrk_post_args(const WrTunerLaunch &L, const WrGroupDev &G);
hipError_t wrk_tuner_post(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G);
hipError_t wrk_tuner_post_args(hipStream_t st, const WrPostArgs &A);
bool wrk_tuner_post_supported(unsigned int d2);
hipError_t wrk_input_hist(hipStream_t st, const float *cur, const uint8_t *cur_u8, size_t nframes,
                          const float *hist, float *hist_next);
hipError_t wrk_seek(hipStream_t st, const WrGroupDev &G, unsigned int slots, int sp, int parity, int p2,
                    unsigned long long frame);
hipError_t wrk_gather_rows(hipStream_t st, const float *src, size_t rows, size_t row_stride_floats,
                           size_t col_offset_floats, unsigned int width_floats, float *dst);

/* ---- the streaming launch (r05; wr_tuner_set_streaming): ONE persistent launch of the fused path that takes the
 * tuner's blocks as they are submitted -- the host rings a doorbell in page-locked memory instead of launching a
 * kernel per block, so a launch's ramp and tail (about 4.6 us at C2: every wave's state, the LDS tables, the first
 * windows; the ragged end) are paid once per STREAM, not once per block, and a block's post stage starts the moment
 * its last channel-IQ row is out, whether or not another block follows (dsp/dspblock.cxx:169-212: a block's output
 * leaves within its own run()).  See k_tuner_stream in wr_kernels.hip. ---- */
#define WR_STREAM_MAXJ   WR_STREAM_MAX_BLOCKS             /* blocks one streaming launch takes at most (the host then opens the next) */
#ifndef WR_STREAM_RING
#define WR_STREAM_RING   6u                /* blocks of channel IQ the ring between the DDC waves and the post stage holds.  r06: 6,
                                              not 8 -- the launch runs at the package's power limit (profiles/r06_power.txt) and the
                                              channel IQ's way to memory and back is a seventh of what it spends; 123 MB of ring
                                              (C2) beside the input windows stay in the 256 MB Infinity Cache where 164 MB did not
                                              quite: 29.8-30.0 against 30.5-30.7 us per block, 33.0-33.9 against 34.2-34.4 at the
                                              driver's 20 steps, two boxes; 5 and 4 leave the DDC too little room to run ahead */
#endif
struct WrStreamDesc {                      /* one submitted block */
	unsigned long long cur;                /* device address of its frames (float pairs, or byte pairs) */
	unsigned long long audio_host;         /* mapped page-locked ring slot that takes its audio too, or 0 */
	unsigned int count;                    /* r06: blocks rung with this one = its index + 1: the 4 bytes a device-side doorbell copies
	                                          into WrStreamDev::ready_up (they must lie in page-locked memory until the copy has run) */
	unsigned int pad;
};
struct WrStreamCtl {                       /* page-locked host memory, mapped: the doorbell and the way back */
	volatile unsigned int ready;           /* host -> device: blocks submitted so far (monotonic) */
	volatile unsigned int stop;            /* host -> device: 1 = `ready` is final */
	volatile unsigned int pad0[14];
	volatile unsigned int done;            /* device -> host: blocks whose audio is complete */
	volatile unsigned int final_blocks;    /* device -> host, at exit: blocks the launch processed */
	volatile unsigned int err;             /* device -> host: non-zero = a wait ran into its deadline (WR_STREAM_ERR_*) */
	volatile unsigned int self_closed;     /* device -> host: the doorbell was silent for `idle_ticks`: closed on its own */
	volatile unsigned int long_blocks;     /* device -> host (r06): blocks whose post stage went in the LONG runs of tiles (the host was ahead) */
	volatile unsigned int pad1[11];
};
#define WR_STREAM_SHARDS 16u               /* counters a block's DDC completions are spread over, a cache line each */
struct WrStreamDev {                       /* device memory: what the bell wave republishes, and the hand-over counters */
	unsigned int ready, stop;
	unsigned long long beat;                   /* the bell's heartbeat: the 100 MHz clock, every time round its loop */
	unsigned int ready_up;                     /* r06: the doorbell's device-side twin, written by a copy on the upload stream behind a
	                                              block's own copy (blocks out of page-locked host memory); the bell takes the larger */
	unsigned int pad0[27];
	/* what the waiting waves poll: one word each, written once per block (never the counters the arrivals land on:
	 * 4 000 waves arriving on ONE word queue up at 11-13 ns each -- MI355X_MICROARCH.md "fanin" -- and every wave
	 * that has an arrival in flight waits for it the next time it waits for vector memory) */
	unsigned long long progress;               /* low word: blocks whose channel IQ is complete; high word: blocks whose post
	                                              stage is complete, in order.  ONE writer (the watcher wave), one 8-byte store:
	                                              a DDC wave entering a block learns both with one load */
	unsigned int drain_from;                   /* 0, or 1 + the first block whose post-stage tasks are handed out by TICKET (post_ticket)
	                                              instead of by workgroup index: the watcher sets it when it learns that the stream is
	                                              closed, to a block nobody can have begun (its channel IQ is not complete yet) -- from
	                                              there on the DDC workgroups, which have nothing left to do, take post tasks too */
	unsigned int pad1[29];
	unsigned int post_done[WR_STREAM_MAXJ];    /* post-stage TASKS of block j that are finished (their stores released) */
	unsigned int post_fine[WR_STREAM_MAXJ];    /* r06: 1 = block j's post stage goes in the SHORT runs of tiles (WrStreamArgs::drain_run): the watcher
	                                              decides when the block's channel IQ is complete -- short unless two further blocks are rung
	                                              already (the host is ahead: what counts is what a block costs, not how soon its audio is out) */
	unsigned int post_ticket[WR_STREAM_MAXJ];  /* the next task of block j to hand out (blocks from drain_from on) */
	unsigned int ddc_done[WR_STREAM_MAXJ][WR_STREAM_SHARDS][32];   /* lane-group units of block j whose channel IQ is in memory:
	                                              the sum over the shards' first words (wave w arrives on shard w mod SHARDS) */
	WrStreamDesc desc[WR_STREAM_MAXJ];
	unsigned long long cur[WR_STREAM_MAXJ + 1u];   /* cur[j + 1] = where block j lies, cur[0] = 0: a block's address and its
	                                              predecessor's side by side */
};
struct WrStreamArgs {
	/* control */
	WrStreamCtl        *ctl;               /* mapped host memory */
	const WrStreamDesc *desc_host;         /* mapped host memory, [WR_STREAM_MAXJ] */
	WrStreamDev        *sdev;
	unsigned long long  idle_ticks;        /* 100 MHz ticks without a bell after which the launch closes itself */
	unsigned long long  wait_ticks;        /* ... any other wait may take before it gives up (an error) */
	unsigned int        n_ddc, n_post;     /* workgroups per role; one more rings the bell */
	unsigned int        dbg;               /* development switches (WR_STREAM_DBG): results are wrong with any of them set */
	unsigned long long *tl;                /* WR_STREAM_DBG & 16: [wave][8] cycle counts of the DDC waves (development aid) */
	unsigned long long  cur0, audio0;      /* block 0's descriptor (the launch exists because of it): here, not copied to
	                                          device memory ahead of the launch -- two small copies on the stream cost the
	                                          open 25 us; the bell wave puts it into WrStreamDev for whoever needs it later */
	/* the block shape (every block of a stream has it) */
	unsigned long long  nframes;           /* input frames per block, = k1 * d1 */
	unsigned int        k1, d1, is_u8;
	unsigned int        ext;               /* r06: blocks may be written by the DMA engine while the launch runs: window loads at agent scope */
	unsigned int        slots, groups;     /* row stride of the per-slot arrays; lane groups in use */
	unsigned long long  gmap0, gmap1;      /* which lane groups (as k_tuner_ddc) */
	unsigned int        kslow;             /* output frames of a block whose window reaches into the block before */
	/* the DDC's state (WrGroupDev), set `sp` read, set `sp ^ 1` written at exit */
	const float        *hist;              /* tuner input history [63][2] before block 0 */
	float              *hist_next;
	const unsigned int *phase, *step;
	unsigned int       *phase_next;
	const float        *hist_cs, *hist_lo;
	float              *hist_cs_next, *hist_lo_next;
	const int          *flags;
	const float        *taps1, *rot, *taps1u;
	const int          *tapsel;
	const float        *table, *hi_cs, *lo_cs;
	float              *ring;              /* [WR_STREAM_RING * k1][slots][2] channel IQ */
	float              *audio_bufs[4];     /* WrGroupDev::audio_set, from the current one on: block j's device audio goes to audio_bufs[j & 3] */
	/* the post stage: what wrk_post_args gives for ONE block, and the two ping-pong state sets */
	WrPostArgs          post;
	unsigned int        drain_run, drain_ntiles;   /* r06: `post.run` / `post.ntiles` for the blocks that go in short runs (WrStreamDev::post_fine:
	                                          a host-paced stream, and a closed stream's last blocks) -- there a block's end is latency, not energy */
	const float        *prev_iq[2];
	float              *dem[2];
	int                 parity0;           /* set block 0 reads */
};
#define WR_STREAM_ERR_WAIT   1u            /* a wait inside the launch ran into `wait_ticks` */
/* workgroups the launch needs co-resident (it sizes its roles to them); 0 = this launch shape cannot stream */
hipError_t wrk_stream_geometry(unsigned int d2, unsigned int groups, int num_cus, unsigned int *n_ddc, unsigned int *n_post);
hipError_t wrk_tuner_stream(hipStream_t st, const WrStreamArgs &A, void *ev_start, void *ev_stop);

/* ---- FFT (wr_fft.hip) ---- */
struct WrFftPlan {
	unsigned int n;             /* transform size */
	unsigned int n1, n2;        /* n = n1*n2; n2 == 1 for single-pass */
	float *tw_n;                /* [n/2][2] twiddles of the full size (device) */
	float *tw_sub;              /* [max(n1,n2)/2][2] twiddles of the LDS sub-transforms (device) */
	float *window;              /* [n] (device) */
	float *window_p1;           /* 65536-point path: the window in the order pass 1's threads take it (wr_fft.hip), else NULL */
	float *work;                /* [n][2] intermediate per frame in flight (device), batch-sized */
	size_t work_frames;
};
hipError_t wrk_fft_frames(hipStream_t st, const WrFftPlan &P, const float *iq, size_t hop,
                          size_t nframes_fft, float *bins_out /* or NULL */, float *db_out /* or NULL */);

hipError_t wrk_bins_to_db(hipStream_t st, const float *bins, unsigned int n, float *db);
hipError_t wrk_waterfall_row(hipStream_t st, const float *bins, unsigned int n, unsigned int width, int hold,
                             float *db_row, uint8_t *palette);

#endif /* WR_INTERNAL_H_ */
