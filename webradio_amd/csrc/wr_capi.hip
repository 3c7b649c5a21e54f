/*
 * wr_capi.hip -- the extern "C" boundary declared in include/webradio_amd.h: handle
 * bookkeeping, staged parameter updates, launch sequencing.  No DSP arithmetic lives
 * here (design math: wr_design.cpp; kernels: wr_kernels.hip, wr_fft.hip).
 *
 * There is deliberately no CPU code path: every data-path entry point needs a
 * wr_dev, and wr_dev_open fails with WR_ERR_NODEV when HIP reports no device.
 */
#include "wr_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

/* ------------------------------------------------------------------ errors -- */

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

int wrc_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                                    \
	do {                                                                                 \
		hipError_t e_ = (expr);                                                          \
		if (e_ != hipSuccess)                                                            \
			return fail(WR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
			            __FILE__, __LINE__);                                             \
	} while (0)

/* ------------------------------------------------------------------ structs -- */

#define WR_UPLOAD_RING 4

struct wr_dev {
	int device;
	hipStream_t stream;
	bool own_stream;
	int num_cus;
	float *table;              /* [65536] reference sine table */
	float *table_turn;         /* [65536] correctly rounded sin(2 pi i / 65536): WR_NCO_ROTATE's turns */
	float *turn_host;          /* the same table on the host (per-slot turns are looked up at upload time) */
	float *hi_cs, *lo_cs;      /* [256][2] split NCO tables */
	float *scratch;            /* growable scratch */
	size_t scratch_floats;
	float *coeff;              /* [WR_FIR_MAX] staging for wr_fir_decimate */
	/* `scratch` and `coeff` are one buffer each per device, used by calls that may come from
	 * different threads (SpectrumSink::getSpectrum on HTTP threads while Radio::run() pumps,
	 * httpserver.cxx:262; two front ends on one GPU).  Work is stream-ordered once enqueued, so the
	 * lock covers a call's ENQUEUE sequence (grow -> producer kernel -> consuming copy): nobody
	 * else's kernel can land between a producer and its consumer, and nobody frees the buffer
	 * under an enqueue in progress. */
	std::mutex *scratch_lock;
	/* behind each of the last WR_UPLOAD_RING async uploads (wr_dev_upload_async, wr_u8_to_f32_from_host): upload number n
	 * (from 1) owns event n % WR_UPLOAD_RING.  `uploads_issued` - `uploads_done` are possibly still in flight. */
	hipEvent_t upload_ev[WR_UPLOAD_RING];
	std::atomic<unsigned long long> uploads_issued, uploads_done;   /* uploads and waits may come from different threads */
	std::mutex *upload_lock;
	/* wr_u8_to_f32_from_host runs on a stream of its own so that a block's bytes cross PCIe while the block before it
	 * is being worked on: up_tail[i] marks what the device's stream held when call i was made (the last readers of the
	 * buffer the call AFTER it may overwrite), up_out[i] is the buffer call i wrote */
	hipStream_t up_stream;
	hipStream_t lazy_stream = nullptr;   /* r06: the copies a SpectrumSink keeps its newest frame with beside an open streaming launch -- device to
	                                        device, i.e. copy KERNELS on this runtime, which do not start before the launch has closed: on a
	                                        stream of their own, where nothing that must not wait (the upload stream's blocks and
	                                        doorbells) can queue behind them */
	hipEvent_t up_tail[WR_UPLOAD_RING], up_done;
	void *up_out[WR_UPLOAD_RING];
	unsigned long long up_calls;
	uint8_t *up_raw[2];         /* where the DMA engine puts the bytes of a block (alternating) before the conversion kernel */
	size_t up_raw_cap[2];
	std::map<void *, size_t> *registered;   /* host ranges THIS library page-locked (wr_dev_host_register), under scratch_lock */
	/* r05: the tuner whose streaming launch (wr_tuner_set_streaming) is running on `stream` right now, or NULL.  That
	 * kernel ends when it is told to, not by itself: whoever is about to wait for the stream -- or to put work on it
	 * that must not wait for the stream's idle deadline -- closes it first (dev_stream_sync, dev_settle_stream) */
	wr_tuner *streaming = nullptr;
	unsigned long long reg_gen = 1;    /* bumped by every wr_dev_host_register / _unregister */
};
#define SCRATCH_GUARD(d) std::lock_guard<std::mutex> scratch_guard_(*(d)->scratch_lock)

struct Chan {
	bool in_use;
	int if_hz;
	unsigned int stepL;        /* phaseStep << 1 */
	unsigned int phaseL;       /* host mirror of DownConverter::phase << 1 */
	int mode;
	bool have[3];              /* [0] channel filter, [1] audio filter, [2] optional second channel filter */
	float taps[3][WR_FIR_LENGTH];
	unsigned int len1;         /* taps of the channel filter: 64 (shorter ones are zero-extended to it) or 128 / 256 */
	float taps_long[WR_FIR_FUSED_MAX];   /* ... and, when len1 > 64, the taps themselves (taps[0] is not used then) */
	unsigned int len2, len1b;  /* the same for the audio filter (stage 1) and the second channel stage (stage 2): 64, 128 or 256 */
	float taps_long2[WR_FIR_FUSED_MAX], taps_long1b[WR_FIR_FUSED_MAX];
	unsigned int decim[3];
	float gain;                /* af_gain as a factor (1 = 0 dB) */
	float squelch;             /* squelch threshold as a power (0 = open) */
	bool cs_hist_reset;        /* channel filter history (LO rows of this slot) must be zeroed */
	int group;                 /* index into wr_tuner::groups, -1 while unconfigured */
	int slot;
	float prev_iq[2];          /* only meaningful while group < 0 (parked state) */
	bool prev_dirty;           /* prev_iq must be uploaded to the slot */
	bool phase_dirty;
	bool dem_hist_reset;       /* audio filter history must be zeroed */
};

struct Group {
	unsigned int d1, d2;
	unsigned int d1b = 0;      /* decimation of the second channel-filter stage, 0 = there is none */
	unsigned int l1 = WR_FIR_LENGTH;   /* taps of the group's channel filters: 64, or 128 / 256 (k_tuner_ddc_long) */
	unsigned int l2 = WR_FIR_LENGTH;   /* ... of its audio filters: 64, or 128 / 256 (k_tuner_demod + k_tuner_audio with 127 / 255 history rows) */
	unsigned int l1b = WR_FIR_LENGTH;  /* ... of its second channel stage (k_tuner_iq2) */
	int p2 = 0;                /* which iq2_hist set the next block reads */
	bool use_gain = false, use_squelch = false;
	unsigned int slots;
	size_t k1max, k2max;
	WrGroupDev dev;
	int audio_cur = 0;         /* which member of dev.audio_set is dev.audio */
	int parity;                /* index of the current ping-pong buffers (prev_iq, dem) */
	int last_parity;           /* parity the last submit wrote its demod rows with */
	bool last_demod_kept = false; /* the last submit left its demod rows in HBM */
	/* The post stage (demodulator + audio filter) of the last block, not launched yet: it goes
	 * out with the NEXT block's DDC launch (extra workgroups of the same kernel, see
	 * k_tuner_ddc) or, when somebody needs the results first, on its own (tuner_flush). */
	bool post_pending = false;
	WrPostArgs post_args;
	/* wr_tuner_seek, lazily: the next submit's launch takes the phase in closed form (WrTunerLaunch::seeking) and reads
	 * all-zero state sets (z_hist: LO rows; z_prev; z_dem), so a seek costs no launch of its own and the post stage of the
	 * chunk before may still ride in that launch.  Anything else that touches the group's state first makes it real
	 * (seek_materialize: k_seek on the actual sets, as before r03). */
	bool seek_pending = false;
	unsigned long long seek_frame = 0;
	float *z_hist = nullptr, *z_prev = nullptr, *z_dem = nullptr;
	bool pend_direct = false;          /* that post stage writes its audio into the ring slot itself (ring_reserve) */
	unsigned long long pend_seq = 0;   /* ring bookkeeping of that block */
	size_t pend_k2 = 0;
	unsigned int pend_slots = 0;
	int sp;                    /* state set (phase, LO history) the next block reads */
	int cb;                    /* chan_iq buffer the next block writes */
	int last_cb;               /* chan_iq buffer the last submit wrote */
	std::vector<int> owner;    /* slot -> chan or -1 */
	bool dirty;                /* parameters must be uploaded before the next launch */
	bool uniform_taps;         /* each 64-slot lane group uses one channel-filter tap set */
	unsigned long long uniform_mask = 0; /* bit i: lane group i does */
	unsigned long long uniform2_mask = 0; /* bit i: the channels of lane group i share one AUDIO filter */
	unsigned long long fewsets_mask = 0; /* bit i: lane group i has at most WR_TAPSETS distinct channel filters
	                                        (the others take the per-lane-taps kernel) */
	unsigned char nsets[64] = {0};       /* how many */
	bool one_filter = false;             /* ONE channel filter for every channel of the group */
	bool long_uniform = false;           /* l1 > 64: every lane group's channels share one long filter (k_tuner_ddc_long_rot) */
	bool long_one = false;               /* ... and it is the same filter in every lane group (two lane groups per wave) */
	size_t last_k1, last_k2;
	int active;
};

struct wr_tuner {
	wr_dev *dev;
	unsigned int input_rate;
	unsigned int max_channels;
	size_t max_block_frames;
	int nco_mode;
	unsigned int keep_mask;
	std::vector<Chan> chans;
	std::vector<Group *> groups;
	float *in_stage;           /* [max_block_frames][2] for WR_HOST submits */
	int last_staging = 0;      /* how the last WR_HOST block travelled: 0 none yet, 1 copied whole, 2 staged sparsely */
	const void *probe_ptr = nullptr;   /* the last WR_HOST block found to be PAGEABLE (not asked about again while ... */
	unsigned long long probe_gen = 0;  /* ... the device's registrations stay as they were: wr_dev::reg_gen) */
	unsigned int probe_left = 0;
	float *in_hist[2];         /* [63][2] ping-pong: last 63 IQ frames of the previous block */
	int in_par;
	bool submitted;
	float audio_scale;
	bool profiling;
	unsigned int prof_stride;  /* 1: every submit's launch stamps its own start/stop; n > 1: one event pair
	                              around every n consecutive submits (see wr_tuner_profile) */
	unsigned int prof_tick;
	std::vector<unsigned int> ev_span;   /* launches between the events of pair i */
	/* wr_tuner_mark_launches: every launch stamps one of these on completion (its own dispatch signal, no packet of
	 * its own on the stream), so that another stream can wait for "the tuner's work so far" -- wr_ring_exchange_after */
	bool mark_launches = false;
	hipEvent_t launch_ev[4] = {nullptr, nullptr, nullptr, nullptr};
	unsigned long long launches_marked = 0;
	std::vector<hipEvent_t> ev;    /* start/stop pairs */
	size_t ev_used;                /* events recorded and not yet read */
	double prof_ms;
	unsigned int prof_n;
	/* pinned audio ring (wr_tuner_audio_ring) */
	struct RingSlot {
		float *host = nullptr;         /* pinned, [max slots][k2max] floats */
		size_t cap = 0;                /* floats */
		hipEvent_t done = nullptr;     /* the copy into this slot */
		size_t stride = 0, frames = 0;
		unsigned int slots = 0;
		unsigned long long seq = 0;
		/* a block of a streaming launch: its audio is in the slot when WrStreamCtl::done of stream number `stream_gen`
		 * has reached `stream_wait` (0: `done` is the event to wait for) */
		unsigned int stream_wait = 0;
		unsigned long long stream_gen = 0;
	};
	std::vector<RingSlot> ring;
	unsigned int ring_head = 0, ring_count = 0;    /* next slot to fill, slots queued */
	bool ring_held = false;                        /* oldest slot handed out, not yet released */
	unsigned long long ring_overruns = 0;
	unsigned long long submit_seq = 0;             /* blocks submitted so far */
	bool defer_post = true;                        /* see Group::post_pending; WR_DEFER_POST=0 turns it off */
	/* wr_tuner_set_blocks_per_launch: consecutive WR_DEVICE blocks that lie back to back in memory are
	 * held (pointer and length only) and launched as ONE block -- the bits do not depend on how the
	 * stream is cut into blocks, the fixed cost of a launch (~4.5 us at C2) is paid once per group */
	unsigned int coalesce = 1;
	const float *held_base = nullptr;
	size_t held_frames = 0, held_each = 0;
	unsigned int held_count = 0;
	std::mutex ring_lock;                          /* producer (submit) vs consumer thread */
	/* r05, wr_tuner_set_streaming: blocks in device memory go to ONE persistent launch (k_tuner_stream) through a
	 * doorbell instead of a launch each; see wr_internal.h and stream_open / stream_bell / stream_close below */
	struct Stream {
		bool enabled = false;
		bool host_bytes = false;           /* wr_tuner_set_streaming(t, 2): byte blocks out of page-locked HOST memory stream too */
		bool live = false;                 /* a launch is running and takes blocks */
		bool unchecked = false;            /* a closed launch whose outcome (WrStreamCtl::err, final_blocks) has not been read yet */
		WrStreamCtl *ctl = nullptr;        /* page-locked, mapped */
		WrStreamDesc *desc = nullptr;      /* page-locked, mapped, [WR_STREAM_MAXJ] */
		WrStreamDev *sdev = nullptr;
		float *ring = nullptr;             /* channel IQ between the DDC waves and the post stage */
		size_t ring_floats = 0;
		Group *g = nullptr;
		unsigned int count = 0;            /* blocks rung into the live (or last) launch */
		size_t nframes = 0;
		bool u8 = false;
		int parity0 = 0;
		size_t k1 = 0, k2 = 0;
		unsigned long long gen = 0;        /* launches opened so far */
		unsigned long long blocks = 0;     /* blocks streamed so far, all launches */
		std::chrono::steady_clock::time_point last_bell;
		const float *last_iq = nullptr;    /* channel IQ of the last block of the last launch (wr_chan_fetch) */
		/* r06: blocks in the RTL-SDR byte format out of PAGE-LOCKED host memory stream too (wr_tuner_submit_u8(..., WR_HOST)):
		 * the bytes cross PCIe as a DMA copy on the upload stream into one of four device buffers taken in turn, and the
		 * doorbell is rung by a 4-byte copy queued behind it (WrStreamDev::ready_up) -- the host waits for neither */
		uint8_t *raw[4] = {nullptr, nullptr, nullptr, nullptr};
		size_t raw_cap = 0;                /* bytes each */
		unsigned long long raw_next = 0;   /* submits that took a raw buffer so far */
		unsigned long long raw_gen[4] = {0, 0, 0, 0};      /* launch (gen) and block index of the buffer's last tenant */
		unsigned int raw_idx[4] = {0, 0, 0, 0};
		hipEvent_t raw_ev = nullptr;       /* behind the copy of a block that OPENS a launch: the launch waits for it */
		bool ext = false;                  /* the live launch takes such blocks (WrStreamArgs::ext) */
		bool up_pending = false;           /* doorbells of the live launch are queued on the upload stream */
		unsigned long long host_blocks = 0;    /* blocks streamed out of host memory so far */
		unsigned long long long_blocks = 0;    /* blocks of checked launches whose post stage went in long runs (WrStreamCtl::long_blocks) */
	} stream;
};

struct wr_spectrum {
	wr_dev *dev;
	unsigned int n, hop;
	WrFftPlan plan;
	float *stage;              /* device stream buffer */
	size_t stage_cap;          /* frames */
	size_t pending;            /* frames buffered in stage, not yet consumed */
	float *bins;               /* [n][2] most recent transform */
	unsigned long frames_done;
	/* r06: a frame pushed while a streaming launch is open on the device is NOT transformed then and there (the transform
	 * would have to close the launch, every block): its frames -- from the frame's first to the block's last -- are copied
	 * into `stage` on the upload stream (the DMA engine: no wave slot needed beside the launch) and transformed when somebody
	 * asks for the spectrum (spectrum_resolve), which is what the reference's own FIXME asks for (io/spectrumsink.cxx:93-94:
	 * only the most recent frame is observable, waterfallhandler.cxx:56-61 reads it at 5 Hz) */
	bool deferred = false;
	float *keep_host = nullptr;    /* page-locked: the kept frames (from the frame's first to the block's last, padded at the front) */
	size_t keep_cap = 0;           /* frames */
	size_t def_off = 0, def_keep = 0;   /* the frame begins `def_off` frames into keep_host; `def_keep` frames from there are kept */
	size_t def_rest = 0;       /* frames behind the deferred frame's hop that belong to the NEXT frame (at stage + 2 * hop) */
	hipEvent_t def_ev = nullptr;   /* behind that copy */
	unsigned long long deferred_pushes = 0, resolves = 0;
};

/* ------------------------------------------------------------------ helpers -- */

/* r05: a streaming launch (wr_tuner_set_streaming) runs until it is told to stop.  Everything that waits for the
 * device's stream, frees device memory (hipFree waits for the device) or must not sit behind an idle launch closes it
 * first; the close is a store to page-locked memory, the launch then finishes the blocks it has and ends. */
static int stream_close(wr_tuner *t);
static int stream_check(wr_tuner *t);
static void stream_free(wr_tuner *t);
static int dev_settle_stream(wr_dev *d)
{
	return d->streaming ? stream_close(d->streaming) : WR_OK;
}
static hipError_t dev_stream_sync(wr_dev *d)
{
	if (dev_settle_stream(d))
		return hipErrorUnknown;
	return hipStreamSynchronize(d->stream);
}
/* ... and, on a tuner's own paths, read what a launch that was closed on the way left behind (WrStreamCtl::err,
 * final_blocks) before the caller is handed results: a launch that ran into a deadline or closed itself early is an
 * error here, not stale audio with WR_OK */
/* an entry point that puts work on the device's stream (or touches the null stream) while a streaming launch is open on it:
 * close the launch first -- the work would otherwise sit behind a kernel that ends only when it is told to, or after
 * WR_STREAM_IDLE_MS of silence (include/webradio_amd.h: "calls on the same wr_dev close it first") */
#define DEV_SETTLE(d_) do { if (int rc_ = dev_settle_stream(d_)) return rc_; } while (0)
#define TUNER_SYNC_CHECKED(t_) do { HIP_TRY(dev_stream_sync((t_)->dev)); if (int rc_ = stream_check(t_)) return rc_; } while (0)


static int dev_bind(wr_dev *d)
{
	HIP_TRY(hipSetDevice(d->device));
	return WR_OK;
}

static int dev_scratch(wr_dev *d, size_t floats)
{
	if (d->scratch_floats >= floats)
		return WR_OK;
	if (d->scratch) {
		HIP_TRY(dev_stream_sync(d));
		HIP_TRY(hipFree(d->scratch));
		d->scratch = nullptr;
		d->scratch_floats = 0;
	}
	HIP_TRY(hipMalloc((void **)&d->scratch, floats * sizeof(float)));
	d->scratch_floats = floats;
	return WR_OK;
}

template <typename T>
static int dev_alloc_zero(T **p, size_t count)
{
	HIP_TRY(hipMalloc((void **)p, (count ? count : 1) * sizeof(T)));
	HIP_TRY(hipMemset(*p, 0, (count ? count : 1) * sizeof(T)));
	/* (the fill is ordered on the null stream; a context on a non-blocking stream of its own -- WR_STREAM_PRIVATE --
	 * is not ordered behind it: wait here, this is set-up code) */
	HIP_TRY(hipStreamSynchronize(nullptr));
	return WR_OK;
}

/* ------------------------------------------------------------------ misc -- */

extern "C" int wr_abi_version(void) { return WR_ABI_VERSION; }
extern "C" const char *wr_last_error(void) { return g_err; }

extern "C" int wr_tune(int key, long value, long *previous)
{
	if (key != WR_TUNE_DDC_NG2_MIN_PASSES)
		return fail(WR_ERR_ARG, "wr_tune: unknown key %d", key);
	const long before = wrk_tune_ng2_min_passes(value, true);
	if (previous)
		*previous = before;
	return WR_OK;
}

extern "C" int wr_device_count(int *count)
{
	if (!count)
		return fail(WR_ERR_ARG, "count is NULL");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) {
		*count = 0;
		return fail(WR_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
	}
	*count = n;
	return WR_OK;
}

extern "C" int wr_phase_step(int if_hz, unsigned int input_rate, int *phase_step)
{
	if (!phase_step || !input_rate)
		return fail(WR_ERR_ARG, "wr_phase_step: bad argument");
	*phase_step = wrd_phase_step(if_hz, input_rate);
	return WR_OK;
}

extern "C" int wr_sin_table(float *table_host)
{
	if (!table_host)
		return fail(WR_ERR_ARG, "table is NULL");
	wrd_sin_table(table_host);
	return WR_OK;
}

extern "C" int wr_lowpass_design(unsigned int passband, unsigned int input_rate, float *coeff_host,
                                 unsigned int *maxbin_out)
{
	if (!coeff_host || !input_rate)
		return fail(WR_ERR_ARG, "wr_lowpass_design: bad argument");
	wrd_lowpass_design(WR_FIR_LENGTH, passband, input_rate, coeff_host);
	if (maxbin_out)
		*maxbin_out = wrd_lowpass_maxbin(WR_FIR_LENGTH, passband, input_rate);
	return WR_OK;
}

static bool fir_length_ok(unsigned int n)
{
	return n >= 2 && n <= WR_FIR_MAX && (n & (n - 1)) == 0;
}

extern "C" int wr_lowpass_design_n(unsigned int fir_length, unsigned int passband, unsigned int input_rate,
                                   float *coeff_host, unsigned int *maxbin_out)
{
	if (!coeff_host || !input_rate)
		return fail(WR_ERR_ARG, "wr_lowpass_design_n: bad argument");
	if (!fir_length_ok(fir_length))
		return fail(WR_ERR_ARG, "wr_lowpass_design_n: fir_length %u is not a power of two in [2, %d]", fir_length,
		            WR_FIR_MAX);
	wrd_lowpass_design(fir_length, passband, input_rate, coeff_host);
	if (maxbin_out)
		*maxbin_out = wrd_lowpass_maxbin(fir_length, passband, input_rate);
	return WR_OK;
}

extern "C" int wr_spectrum_window(unsigned int fft_size, float *window_host)
{
	if (!window_host || !fft_size)
		return fail(WR_ERR_ARG, "wr_spectrum_window: bad argument");
	wrd_spectrum_window(fft_size, window_host);
	return WR_OK;
}

/* ------------------------------------------------------------------ device -- */

extern "C" int wr_dev_open(wr_dev **dev, int device_index, void *hip_stream)
{
	if (!dev)
		return fail(WR_ERR_ARG, "dev is NULL");
	*dev = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail(WR_ERR_NODEV, "no HIP device (%s): this backend has no CPU path",
		            e == hipSuccess ? "count = 0" : hipGetErrorString(e));
	if (device_index < 0 || device_index >= n)
		return fail(WR_ERR_ARG, "device %d out of range (%d devices)", device_index, n);
	HIP_TRY(hipSetDevice(device_index));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device_index));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(WR_ERR_NODEV, "device %d is %s; this library is built for gfx950 only",
		            device_index, prop.gcnArchName);

	wr_dev *d = new (std::nothrow) wr_dev();
	if (!d)
		return fail(WR_ERR_NOMEM, "out of memory");
	/* (value-initialised: every member zero) */
	d->device = device_index;
	d->num_cus = prop.multiProcessorCount;
	/* NULL selects HIP's default (null) stream -- which is also what
	 * torch.cuda.current_stream().cuda_stream is unless the caller switched streams */
	d->stream = (hipStream_t)hip_stream;
	d->own_stream = false;


	std::vector<float> table(WR_TABLE_SIZE), turn(WR_TABLE_SIZE), hi(2 * WR_SPLIT_N), lo(2 * WR_SPLIT_N);
	wrd_sin_table(table.data());
	wrd_sin_table_rounded(turn.data());
	wrd_split_tables(hi.data(), lo.data());
	int rc = WR_OK;
	d->turn_host = (float *)malloc(WR_TABLE_SIZE * sizeof(float));
	d->scratch_lock = new (std::nothrow) std::mutex();
	d->registered = new (std::nothrow) std::map<void *, size_t>();
	d->upload_lock = new (std::nothrow) std::mutex();
	if (!d->turn_host || !d->scratch_lock || !d->registered || !d->upload_lock) {
		free(d->turn_host);
		delete d->scratch_lock;
		delete d->registered;
		delete d->upload_lock;
		delete d;
		return fail(WR_ERR_NOMEM, "out of memory");
	}
	memcpy(d->turn_host, turn.data(), WR_TABLE_SIZE * sizeof(float));
	do {
		if ((e = hipMalloc((void **)&d->table, WR_TABLE_SIZE * sizeof(float))) != hipSuccess) break;
		if ((e = hipMalloc((void **)&d->table_turn, WR_TABLE_SIZE * sizeof(float))) != hipSuccess) break;
		if ((e = hipMemcpy(d->table_turn, turn.data(), WR_TABLE_SIZE * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		if ((e = hipMalloc((void **)&d->hi_cs, 2 * WR_SPLIT_N * sizeof(float))) != hipSuccess) break;
		if ((e = hipMalloc((void **)&d->lo_cs, 2 * WR_SPLIT_N * sizeof(float))) != hipSuccess) break;
		if ((e = hipMalloc((void **)&d->coeff, WR_FIR_MAX * sizeof(float))) != hipSuccess) break;
		if ((e = hipMemcpy(d->table, table.data(), WR_TABLE_SIZE * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		if ((e = hipMemcpy(d->hi_cs, hi.data(), 2 * WR_SPLIT_N * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		if ((e = hipMemcpy(d->lo_cs, lo.data(), 2 * WR_SPLIT_N * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
	} while (0);
	if (e != hipSuccess) {
		rc = fail(WR_ERR_HIP, "wr_dev_open: %s", hipGetErrorString(e));
		wr_dev_close(d);
		return rc;
	}
	*dev = d;
	return WR_OK;
}

extern "C" int wr_dev_close(wr_dev *d)
{
	if (!d)
		return WR_OK;
	(void)hipSetDevice(d->device);
	(void)dev_stream_sync(d);
	(void)hipFree(d->table);
	(void)hipFree(d->table_turn);
	(void)hipFree(d->hi_cs);
	(void)hipFree(d->lo_cs);
	(void)hipFree(d->coeff);
	(void)hipFree(d->scratch);
	free(d->turn_host);
	if (d->registered)
		for (auto &r : *d->registered)              /* what the caller forgot to release */
			(void)hipHostUnregister(r.first);
	delete d->registered;
	delete d->scratch_lock;
	for (int i = 0; i < WR_UPLOAD_RING; ++i)
		if (d->upload_ev[i])
			(void)hipEventDestroy(d->upload_ev[i]);
	delete d->upload_lock;
	if (d->up_stream) {
		(void)hipStreamSynchronize(d->up_stream);
		(void)hipStreamDestroy(d->up_stream);
	}
	for (int i = 0; i < WR_UPLOAD_RING; ++i)
		if (d->up_tail[i])
			(void)hipEventDestroy(d->up_tail[i]);
	if (d->up_done)
		(void)hipEventDestroy(d->up_done);
	(void)hipFree(d->up_raw[0]);
	(void)hipFree(d->up_raw[1]);
	if (d->own_stream)
		(void)hipStreamDestroy(d->stream);
	delete d;
	return WR_OK;
}

extern "C" int wr_dev_sync(wr_dev *d)
{
	if (!d)
		return fail(WR_ERR_ARG, "dev is NULL");
	wr_tuner *live = d->streaming;                       /* (closed by the sync: its outcome is this call's) */
	HIP_TRY(dev_stream_sync(d));
	return live ? stream_check(live) : WR_OK;
}

extern "C" void *wr_dev_stream(wr_dev *d) { return d ? (void *)d->stream : nullptr; }
int wrc_dev_index(const wr_dev *d) { return d->device; }
hipStream_t wrc_dev_stream(const wr_dev *d) { return d->stream; }

extern "C" int wr_dev_malloc(wr_dev *d, size_t bytes, void **ptr_dev)
{
	if (!d || !ptr_dev)
		return fail(WR_ERR_ARG, "wr_dev_malloc: bad argument");
	if (dev_bind(d))
		return WR_ERR_HIP;
	HIP_TRY(hipMalloc(ptr_dev, bytes ? bytes : 4));
	HIP_TRY(hipMemsetAsync(*ptr_dev, 0, bytes ? bytes : 4, d->stream));
	return WR_OK;
}

extern "C" int wr_dev_free(wr_dev *d, void *ptr_dev)
{
	if (!d)
		return fail(WR_ERR_ARG, "dev is NULL");
	if (!ptr_dev)
		return WR_OK;
	HIP_TRY(dev_stream_sync(d));
	HIP_TRY(hipFree(ptr_dev));
	return WR_OK;
}

extern "C" int wr_dev_upload(wr_dev *d, void *dst_dev, const void *src_host, size_t bytes)
{
	if (!d || (!dst_dev && bytes) || (!src_host && bytes))
		return fail(WR_ERR_ARG, "wr_dev_upload: bad argument");
	if (!bytes)
		return WR_OK;
	HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, d->stream));
	HIP_TRY(dev_stream_sync(d));
	return WR_OK;
}

/* Page-lock a host buffer the caller will upload from repeatedly (a source's block vector): a copy
 * out of pageable memory is staged by the runtime and holds the calling thread for its whole
 * duration; out of registered memory it is one DMA the thread does not wait for. */
extern "C" int wr_dev_host_register(wr_dev *d, void *host, size_t bytes)
{
	if (!d || !host || !bytes)
		return fail(WR_ERR_ARG, "wr_dev_host_register: bad argument");
	if (dev_bind(d))
		return WR_ERR_HIP;
	SCRATCH_GUARD(d);
	++d->reg_gen;
	hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
	if (e == hipErrorHostMemoryAlreadyRegistered) {
		(void)hipGetLastError();
		auto own = d->registered->find(host);
		if (own == d->registered->end())
			/* page-locked by somebody else (the application, torch, another library): that is all
			 * this call is for -- it is neither undone nor recorded, and wr_dev_host_unregister
			 * will leave it alone */
			return WR_OK;
		/* one of OURS whose memory was freed and handed out again by the allocator, perhaps with
		 * another length: register the range as it is now */
		(void)hipHostUnregister(host);
		d->registered->erase(own);
		e = hipHostRegister(host, bytes, hipHostRegisterDefault);
	}
	if (e != hipSuccess) {
		(void)hipGetLastError();                    /* not sticky: a later launch check must not trip over it */
		return fail(WR_ERR_HIP, "wr_dev_host_register: %s", hipGetErrorString(e));
	}
	(*d->registered)[host] = bytes;
	return WR_OK;
}

extern "C" int wr_dev_host_unregister(wr_dev *d, void *host)
{
	if (!d || !host)
		return fail(WR_ERR_ARG, "wr_dev_host_unregister: bad argument");
	if (dev_bind(d))
		return WR_ERR_HIP;
	SCRATCH_GUARD(d);
	auto own = d->registered->find(host);
	if (own == d->registered->end())
		return WR_OK;                               /* not page-locked by this library: not ours to release */
	d->registered->erase(own);
	++d->reg_gen;
	hipError_t e = hipHostUnregister(host);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return fail(WR_ERR_HIP, "wr_dev_host_unregister: %s", hipGetErrorString(e));
	}
	return WR_OK;
}

/* Enqueue a host-to-device copy on the device's stream and return; the host buffer must stay
 * untouched until wr_dev_wait_uploads (or wr_dev_sync) returns. */
/* marks "the upload just enqueued on the stream ends here"; the event it reuses belonged to the upload WR_UPLOAD_RING
 * before it, which is waited for first if nobody has yet */
static int upload_mark_locked(wr_dev *d, hipStream_t st)
{
	const unsigned long long n = d->uploads_issued + 1;
	hipEvent_t &ev = d->upload_ev[n % WR_UPLOAD_RING];
	if (!ev)
		HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	else if (n > WR_UPLOAD_RING && d->uploads_done < n - WR_UPLOAD_RING) {
		HIP_TRY(hipEventSynchronize(ev));
		d->uploads_done = n - WR_UPLOAD_RING;
	}
	HIP_TRY(hipEventRecord(ev, st));
	d->uploads_issued = n;
	return WR_OK;
}

static int upload_mark(wr_dev *d, hipStream_t st)
{
	std::lock_guard<std::mutex> g(*d->upload_lock);
	return upload_mark_locked(d, st);
}

extern "C" int wr_dev_upload_async(wr_dev *d, void *dst_dev, const void *src_host, size_t bytes)
{
	if (!d || (bytes && (!dst_dev || !src_host)))
		return fail(WR_ERR_ARG, "wr_dev_upload_async: bad argument");
	if (dev_bind(d))
		return WR_ERR_HIP;
	if (bytes)
		HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, d->stream));
	return upload_mark(d, d->stream);
}

extern "C" int wr_dev_wait_uploads_but(wr_dev *d, unsigned int newest)
{
	if (!d)
		return fail(WR_ERR_ARG, "dev is NULL");
	if (newest >= WR_UPLOAD_RING)
		return fail(WR_ERR_ARG, "wr_dev_wait_uploads_but: at most %d uploads can be left in flight", WR_UPLOAD_RING - 1);
	if (d->uploads_issued <= d->uploads_done + newest)
		return WR_OK;
	if (dev_bind(d))
		return WR_ERR_HIP;
	std::lock_guard<std::mutex> g(*d->upload_lock);
	const unsigned long long issued = d->uploads_issued;
	if (issued <= d->uploads_done + newest)
		return WR_OK;
	/* uploads 1..upto must have completed.  They are not all on one stream (wr_dev_upload_async: the device's;
	 * wr_dev_upload_ahead / wr_u8_to_f32_from_host: the upload stream), so the event of upload `upto` does not speak
	 * for the ones before it: every event in (done, upto] is waited for -- at most WR_UPLOAD_RING - 1 of them, the
	 * issuing side never lets more stay open */
	const unsigned long long upto = issued - newest;
	for (unsigned long long i = d->uploads_done + 1; i <= upto; ++i)
		HIP_TRY(hipEventSynchronize(d->upload_ev[i % WR_UPLOAD_RING]));
	d->uploads_done = upto;
	return WR_OK;
}

extern "C" int wr_dev_wait_uploads(wr_dev *d)
{
	return wr_dev_wait_uploads_but(d, 0);
}

extern "C" int wr_dev_download(wr_dev *d, void *dst_host, const void *src_dev, size_t bytes)
{
	if (!d || (!dst_host && bytes) || (!src_dev && bytes))
		return fail(WR_ERR_ARG, "wr_dev_download: bad argument");
	if (!bytes)
		return WR_OK;
	wr_tuner *live = d->streaming;
	HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(dev_stream_sync(d));
	return live ? stream_check(live) : WR_OK;
}

/* --------------------------------------------------- one kernel per block -- */

/* how often this process has run one of the reference's blocks as a stand-alone kernel (wr_mix, wr_fir_decimate(_n),
 * wr_demod): what a Receiver that is NOT in a tuner batch costs per block -- a test that expects a chain to stay in the
 * batch reads 0 here */
static std::atomic<unsigned long long> g_block_kernel_calls{0};
extern "C" unsigned long long wr_block_kernel_calls(void)
{
	return g_block_kernel_calls.load(std::memory_order_relaxed);
}

extern "C" int wr_mix(wr_dev *d, const float *in_dev, float *out_dev, size_t nframes,
                      unsigned int *phase_io, int phase_step)
{
	if (!d || !phase_io || (nframes && (!in_dev || !out_dev)))
		return fail(WR_ERR_ARG, "wr_mix: bad argument");
	DEV_SETTLE(d);
	g_block_kernel_calls.fetch_add(1, std::memory_order_relaxed);
	HIP_TRY(wrk_mix(d->stream, in_dev, out_dev, nframes, *phase_io, phase_step, d->table));
	/* DownConverter::phase after nframes increments (downconverter.cxx:103) */
	*phase_io = (*phase_io + (unsigned int)nframes * (unsigned int)phase_step) & 0x7FFFFFFFu;
	return WR_OK;
}

extern "C" int wr_fir_decimate_n(wr_dev *d, const float *in_dev, size_t nframes, unsigned int channels,
                                 unsigned int decimation, unsigned int fir_length, const float *coeff_host,
                                 float *history_dev, float *out_dev)
{
	if (!d || !coeff_host || !history_dev || !channels || !decimation ||
	    (nframes && (!in_dev || !out_dev)))
		return fail(WR_ERR_ARG, "wr_fir_decimate: bad argument");
	if (!fir_length_ok(fir_length))
		return fail(WR_ERR_ARG, "wr_fir_decimate: fir_length %u is not a power of two in [2, %d]", fir_length,
		            WR_FIR_MAX);
	DEV_SETTLE(d);
	SCRATCH_GUARD(d);
	int rc = dev_scratch(d, (size_t)(fir_length - 1) * channels);
	if (rc)
		return rc;
	HIP_TRY(hipMemcpyAsync(d->coeff, coeff_host, fir_length * sizeof(float), hipMemcpyHostToDevice, d->stream));
	g_block_kernel_calls.fetch_add(1, std::memory_order_relaxed);
	HIP_TRY(wrk_fir(d->stream, in_dev, nframes, channels, decimation, fir_length, d->coeff, history_dev, out_dev));
	HIP_TRY(wrk_hist_update(d->stream, in_dev, nframes, channels, fir_length, history_dev, d->scratch));
	return WR_OK;
}

extern "C" int wr_fir_decimate(wr_dev *d, const float *in_dev, size_t nframes, unsigned int channels,
                               unsigned int decimation, const float *coeff_host, float *history_dev,
                               float *out_dev)
{
	return wr_fir_decimate_n(d, in_dev, nframes, channels, decimation, WR_FIR_LENGTH, coeff_host, history_dev,
	                         out_dev);
}

extern "C" int wr_demod(wr_dev *d, int mode, const float *in_dev, size_t nframes, float *prev_io,
                        float *out_dev)
{
	if (!d || !prev_io || (nframes && (!in_dev || !out_dev)))
		return fail(WR_ERR_ARG, "wr_demod: bad argument");
	if (mode < WR_AM || mode > WR_LSB)
		return fail(WR_ERR_ARG, "wr_demod: bad mode %d", mode);   /* demodulator.cxx:105-107 */
	g_block_kernel_calls.fetch_add(1, std::memory_order_relaxed);
	DEV_SETTLE(d);
	HIP_TRY(wrk_demod(d->stream, mode, in_dev, nframes, prev_io[0], prev_io[1], out_dev));
	if (nframes) {
		/* prev_i/q = last input frame (demodulator.cxx:110-111) */
		HIP_TRY(hipMemcpyAsync(prev_io, in_dev + 2 * (nframes - 1), 2 * sizeof(float),
		                       hipMemcpyDeviceToHost, d->stream));
		HIP_TRY(dev_stream_sync(d));
	}
	return WR_OK;
}

extern "C" int wr_u8_to_f32(wr_dev *d, const uint8_t *in_dev, float *out_dev, size_t count)
{
	if (!d || (count && (!in_dev || !out_dev)))
		return fail(WR_ERR_ARG, "wr_u8_to_f32: bad argument");
	DEV_SETTLE(d);
	HIP_TRY(wrk_u8_to_f32(d->stream, in_dev, out_dev, count));
	return WR_OK;
}

/* The upload stream (wr_dev_upload_ahead, wr_u8_to_f32_from_host): a block crosses PCIe beside whatever the device's stream
 * is doing with the block before (8 MB take 150 us at 55 GB/s -- more than all the kernels of a C2 block together).
 * upload_ahead_begin makes the upload stream wait for the last readers of `out_dev`: those were enqueued before the call
 * that followed the last one to write `out_dev` (a caller alternating between two buffers: before the previous call), or --
 * the same buffer twice in a row, or one not seen lately -- by now.  upload_ahead_end makes the device's stream wait for
 * what was put on the upload stream in between and marks the upload (wr_dev_wait_uploads).  Under d->upload_lock. */
static int dev_up_stream(wr_dev *d)
{
	if (!d->up_stream) {
		int prio_low = 0, prio_high = 0;                     /* lowest priority: the kernels of the block before go first */
		HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
		HIP_TRY(hipStreamCreateWithPriority(&d->up_stream, hipStreamNonBlocking, prio_low));
		/* up_done (recorded on the upload stream: nothing of the device's stream pays for it) publishes the copied block and keeps
		 * the default fence; up_tail only keeps the copy from overwriting what earlier work still reads: a device-scope release */
		HIP_TRY(hipEventCreateWithFlags(&d->up_done, hipEventDisableTiming));
		for (int i = 0; i < WR_UPLOAD_RING; ++i)
			HIP_TRY(hipEventCreateWithFlags(&d->up_tail[i], hipEventDisableTiming | hipEventReleaseToDevice));
	}
	return WR_OK;
}

static int upload_ahead_begin(wr_dev *d, void *out_dev, unsigned long long *call)
{
	if (int rc = dev_up_stream(d))
		return rc;
	const unsigned long long n = d->up_calls;
	HIP_TRY(hipEventRecord(d->up_tail[n % WR_UPLOAD_RING], d->stream));
	unsigned long long after = n;                      /* wait for the tail recorded by call `after` */
	for (unsigned long long back = 1; back < WR_UPLOAD_RING && back <= n; ++back)
		if (d->up_out[(n - back) % WR_UPLOAD_RING] == out_dev) {
			after = n - back + 1;
			break;
		}
	HIP_TRY(hipStreamWaitEvent(d->up_stream, d->up_tail[after % WR_UPLOAD_RING], 0));
	d->up_out[n % WR_UPLOAD_RING] = out_dev;
	d->up_calls = n + 1;
	*call = n;
	return WR_OK;
}

static int upload_ahead_end(wr_dev *d)
{
	HIP_TRY(hipEventRecord(d->up_done, d->up_stream));
	HIP_TRY(hipStreamWaitEvent(d->stream, d->up_done, 0));
	return upload_mark_locked(d, d->up_stream);
}

extern "C" int wr_dev_upload_ahead(wr_dev *d, void *dst_dev, const void *src_host, size_t bytes)
{
	if (!d || (bytes && (!dst_dev || !src_host)))
		return fail(WR_ERR_ARG, "wr_dev_upload_ahead: bad argument");
	if (dev_bind(d))
		return WR_ERR_HIP;
	std::lock_guard<std::mutex> up_guard(*d->upload_lock);
	unsigned long long n = 0;
	if (int rc = upload_ahead_begin(d, dst_dev, &n))
		return rc;
	if (bytes)
		HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, d->up_stream));
	return upload_ahead_end(d);
}

extern "C" int wr_u8_to_f32_from_host(wr_dev *d, const uint8_t *in_host, float *out_dev, size_t count)
{
	if (!d || (count && (!in_host || !out_dev)))
		return fail(WR_ERR_ARG, "wr_u8_to_f32_from_host: bad argument");
	if (dev_bind(d))
		return WR_ERR_HIP;
	void *mapped = nullptr;
	hipError_t e = hipHostGetDevicePointer(&mapped, const_cast<uint8_t *>(in_host), 0);
	if (e != hipSuccess || !mapped) {
		(void)hipGetLastError();
		return fail(WR_ERR_ARG, "wr_u8_to_f32_from_host: the buffer is not page-locked (wr_dev_host_register): %s",
		            hipGetErrorString(e));
	}
	std::lock_guard<std::mutex> up_guard(*d->upload_lock);         /* the ring of tails: calls may come from several threads */
	/* The bytes come over with the DMA engine and are converted out of device memory.  (r03 tried the kernel reading
	 * host memory itself, WR_U8_ZEROCOPY=1: one launch and 47 GB/s -- but kernels running beside it take up to ten times
	 * as long, k_tuner_post 14 -> 107 us, a 32 MB device copy 13 -> 146 us: its reads, microseconds each, sit in the
	 * same L2 / fabric queues as everybody's HBM traffic.  A DMA copy does not go through them.) */
	static const bool zerocopy = getenv("WR_U8_ZEROCOPY") && atoi(getenv("WR_U8_ZEROCOPY")) != 0;
	if (zerocopy) {
		unsigned long long n = 0;
		if (int rc = upload_ahead_begin(d, (void *)out_dev, &n))
			return rc;
		HIP_TRY(wrk_u8_to_f32(d->up_stream, (const uint8_t *)mapped, out_dev, count));
		return upload_ahead_end(d);
	}
	/* Only the COPY runs on the upload stream, into one of two raw buffers in turn; the conversion follows on the device's
	 * stream once the copy's event has fired.  So the copies of consecutive blocks follow each other on the link without
	 * a kernel in between (the upload stream waits for nothing but the conversion that last read the raw buffer it is about
	 * to overwrite -- two calls ago), and `out_dev` is written in stream order like any kernel's output. */
	const unsigned int rb = (unsigned int)(d->up_calls & 1u);
	if (d->up_raw_cap[rb] < count) {
		if (d->up_stream)
			HIP_TRY(hipStreamSynchronize(d->up_stream));
		HIP_TRY(dev_stream_sync(d));
		(void)hipFree(d->up_raw[rb]);
		d->up_raw[rb] = nullptr;
		d->up_raw_cap[rb] = 0;
		HIP_TRY(hipMalloc((void **)&d->up_raw[rb], count));
		d->up_raw_cap[rb] = count;
	}
	unsigned long long n = 0;
	if (int rc = upload_ahead_begin(d, (void *)d->up_raw[rb], &n))
		return rc;
	HIP_TRY(hipMemcpyAsync(d->up_raw[rb], in_host, count, hipMemcpyHostToDevice, d->up_stream));
	if (int rc = upload_ahead_end(d))                          /* the host buffer is free again when the COPY is done */
		return rc;
	HIP_TRY(wrk_u8_to_f32(d->stream, d->up_raw[rb], out_dev, count));
	return WR_OK;
}

extern "C" int wr_stage_windows_from_host(wr_dev *d, const void *in_host, int is_u8, float *out_dev, size_t nframes,
                                          unsigned int period, unsigned int length, size_t tail_frames)
{
	if (!d || (nframes && (!in_host || !out_dev)) || !period || !length)
		return fail(WR_ERR_ARG, "wr_stage_windows_from_host: bad argument");
	if (length > 4096u)
		return fail(WR_ERR_ARG, "wr_stage_windows_from_host: windows of %u frames", length);
	if (((uintptr_t)in_host | (uintptr_t)out_dev) & 15u)
		return fail(WR_ERR_ARG, "wr_stage_windows_from_host: both buffers must be 16-byte aligned");
	if (dev_bind(d))
		return WR_ERR_HIP;
	void *mapped = nullptr;
	hipError_t e = hipHostGetDevicePointer(&mapped, const_cast<void *>(in_host), 0);
	if (e != hipSuccess || !mapped) {
		(void)hipGetLastError();
		return fail(WR_ERR_ARG, "wr_stage_windows_from_host: the buffer is not page-locked (wr_dev_host_register): %s",
		            hipGetErrorString(e));
	}
	HIP_TRY(wrk_stage_windows(d->stream, mapped, is_u8 != 0, out_dev, nframes, period, length, tail_frames));
	return upload_mark(d, d->stream);                  /* the host buffer is free again when the kernel has read it */
}

static bool host_sparse_enabled()
{
	static const bool on = !(getenv("WR_HOST_SPARSE") && atoi(getenv("WR_HOST_SPARSE")) == 0);
	return on;
}

static bool lazy_seek_enabled()
{
	static const bool on = !(getenv("WR_LAZY_SEEK") && atoi(getenv("WR_LAZY_SEEK")) == 0);
	return on;
}

/* WR_LONG_ROTATE=0: channel filters of 128 / 256 taps take the reference's arithmetic in every nco mode (r03's first version) */
static bool long_rot_enabled()
{
	static const bool on = !(getenv("WR_LONG_ROTATE") && atoi(getenv("WR_LONG_ROTATE")) == 0);
	return on;
}

/* ------------------------------------------------------------------ tuner -- */

/* lane groups that hold at least one channel: [0, used) in units of slots */
static unsigned int group_slots_used(const Group *g)
{
	unsigned int hi = 0;
	for (unsigned int s = 0; s < g->slots; ++s)
		if (g->owner[s] >= 0)
			hi = s + 1;
	return ((hi + WR_LANES - 1) / WR_LANES) * WR_LANES;
}

static void group_free(Group *g)
{
	if (!g)
		return;
	(void)hipFree(g->dev.phase[0]);
	(void)hipFree(g->dev.phase[1]);
	(void)hipFree(g->dev.step);
	(void)hipFree(g->dev.hist_cs[0]);
	(void)hipFree(g->dev.hist_cs[1]);
	(void)hipFree(g->dev.hist_lo[0]);
	(void)hipFree(g->dev.hist_lo[1]);
	(void)hipFree(g->dev.flags);
	(void)hipFree(g->dev.mode);
	(void)hipFree(g->dev.taps1);
	(void)hipFree(g->dev.taps2);
	(void)hipFree(g->dev.rot);
	(void)hipFree(g->dev.taps1u);
	(void)hipFree(g->dev.taps2u);
	(void)hipFree(g->dev.tapsel);
	(void)hipFree(g->dev.taps1b);
	(void)hipFree(g->dev.iq2_hist[0]);
	(void)hipFree(g->dev.iq2_hist[1]);
	(void)hipFree(g->dev.chan_iq2[0]);
	(void)hipFree(g->dev.chan_iq2[1]);
	(void)hipFree(g->dev.taps1L);
	(void)hipFree(g->dev.mixhist[0]);
	(void)hipFree(g->dev.mixhist[1]);
	(void)hipFree(g->dev.gain);
	(void)hipFree(g->dev.squelch);
	(void)hipFree(g->z_hist);
	(void)hipFree(g->z_prev);
	(void)hipFree(g->z_dem);
	(void)hipFree(g->dev.prev_iq[0]);
	(void)hipFree(g->dev.prev_iq[1]);
	(void)hipFree(g->dev.chan_iq[0]);
	(void)hipFree(g->dev.chan_iq[1]);
	(void)hipFree(g->dev.dem[0]);
	(void)hipFree(g->dev.dem[1]);
	for (float *a : g->dev.audio_set)
		(void)hipFree(a);                               /* (`audio` is one of them) */
	delete g;
}

static int group_create(wr_tuner *t, unsigned int d1, unsigned int d1b, unsigned int d2, unsigned int l1, unsigned int l1b,
                        unsigned int l2, Group **out)
{
	Group *g = new (std::nothrow) Group();
	if (!g)
		return fail(WR_ERR_NOMEM, "out of memory");
	memset(&g->dev, 0, sizeof(g->dev));
	g->parity = g->last_parity = 0;
	g->sp = g->cb = g->last_cb = 0;
	g->d1 = d1;
	g->d1b = d1b;
	g->d2 = d2;
	g->l1 = l1;
	g->l2 = l2;
	g->l1b = d1b ? l1b : (unsigned int)WR_FIR_LENGTH;
	g->slots = ((t->max_channels + WR_LANES - 1) / WR_LANES) * WR_LANES;
	g->k1max = t->max_block_frames / d1;       /* first-stage frames; the later stages need no more */
	g->k2max = g->k1max / (d1b ? d1b : 1u) / d2;
	if (g->k2max == 0)
		g->k2max = 1;
	g->owner.assign(g->slots, -1);
	g->dirty = true;
	g->uniform_taps = false;
	g->last_k1 = g->last_k2 = 0;
	g->active = 0;
	const size_t S = g->slots;
	g->dev.l2 = g->l2;
	g->dev.l1b = g->l1b;
	int rc = WR_OK;
	if (!rc) rc = dev_alloc_zero(&g->dev.phase[0], S);
	if (!rc) rc = dev_alloc_zero(&g->dev.phase[1], S);
	if (!rc) rc = dev_alloc_zero(&g->dev.step, S);
	if (!rc) rc = dev_alloc_zero(&g->dev.hist_cs[0], (size_t)WR_HIST * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.hist_cs[1], (size_t)WR_HIST * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.hist_lo[0], (size_t)WR_HIST * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.hist_lo[1], (size_t)WR_HIST * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.flags, S);
	if (!rc) rc = dev_alloc_zero(&g->dev.mode, S);
	if (!rc) rc = dev_alloc_zero(&g->dev.taps1, S * WR_FIR_LENGTH);
	if (!rc) rc = dev_alloc_zero(&g->dev.taps2, S * g->l2);
	if (!rc) rc = dev_alloc_zero(&g->dev.rot, S * 4);
	if (!rc) rc = dev_alloc_zero(&g->dev.taps1u, S * WR_TAPSETS);
	if (!rc) rc = dev_alloc_zero(&g->dev.taps2u, S / WR_LANES * g->l2);      /* [lane groups][l2] */
	if (!rc) rc = dev_alloc_zero(&g->dev.tapsel, S);
	if (l1 > WR_FIR_LENGTH) {
		if (!rc) rc = dev_alloc_zero(&g->dev.taps1L, S * l1);
		if (!rc) rc = dev_alloc_zero(&g->dev.mixhist[0], (size_t)(l1 - 1) * S * 2);
		if (!rc) rc = dev_alloc_zero(&g->dev.mixhist[1], (size_t)(l1 - 1) * S * 2);
	}
	if (!rc) rc = dev_alloc_zero(&g->dev.gain, S);
	if (!rc) rc = dev_alloc_zero(&g->dev.squelch, S);
	if (!rc) rc = dev_alloc_zero(&g->dev.iq2_hist[0], (size_t)(g->l1b - 1) * S * 2);      /* wr_tuner_seek clears it */
	if (!rc) rc = dev_alloc_zero(&g->dev.iq2_hist[1], (size_t)(g->l1b - 1) * S * 2);
	if (d1b) {
		if (!rc) rc = dev_alloc_zero(&g->dev.taps1b, S * g->l1b);
		if (!rc) rc = dev_alloc_zero(&g->dev.chan_iq2[0], (g->k1max / d1b + 1) * S * 2);
		if (!rc) rc = dev_alloc_zero(&g->dev.chan_iq2[1], (g->k1max / d1b + 1) * S * 2);
	}
	if (!rc) rc = dev_alloc_zero(&g->z_hist, (size_t)WR_HIST * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->z_prev, S * 2);
	if (!rc) rc = dev_alloc_zero(&g->z_dem, (size_t)(g->l2 - 1) * S);
	if (!rc) rc = dev_alloc_zero(&g->dev.prev_iq[0], S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.prev_iq[1], S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.chan_iq[0], (g->k1max ? g->k1max : 1) * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.chan_iq[1], (g->k1max ? g->k1max : 1) * S * 2);
	if (!rc) rc = dev_alloc_zero(&g->dev.dem[0], ((size_t)g->l2 - 1 + g->k1max) * S);
	if (!rc) rc = dev_alloc_zero(&g->dev.dem[1], ((size_t)g->l2 - 1 + g->k1max) * S);
	for (int i = 0; i < 4 && !rc; ++i)
		rc = dev_alloc_zero(&g->dev.audio_set[i], g->k2max * S);
	g->dev.audio = g->dev.audio_set[0];
	g->audio_cur = 0;
	if (rc) {
		group_free(g);
		return rc;
	}
	*out = g;
	return WR_OK;
}

extern "C" int wr_tuner_create(wr_tuner **tuner, wr_dev *dev, unsigned int input_rate,
                               unsigned int max_channels, size_t max_block_frames, int nco_mode)
{
	if (!tuner || !dev || !input_rate || !max_channels || !max_block_frames)
		return fail(WR_ERR_ARG, "wr_tuner_create: bad argument");
	if (nco_mode != WR_NCO_SPLIT && nco_mode != WR_NCO_EXACT && nco_mode != WR_NCO_ROTATE)
		return fail(WR_ERR_ARG, "wr_tuner_create: bad nco_mode %d", nco_mode);
	if (max_channels > WR_MAX_CHANNELS)
		return fail(WR_ERR_ARG, "wr_tuner_create: %u channels, at most %u per tuner (64 lane groups of 64)",
		            max_channels, (unsigned)WR_MAX_CHANNELS);
	*tuner = nullptr;
	if (dev_bind(dev))
		return WR_ERR_HIP;
	DEV_SETTLE(dev);                                        /* (the allocations below fill through the null stream and wait for it) */
	wr_tuner *t = new (std::nothrow) wr_tuner();
	if (!t)
		return fail(WR_ERR_NOMEM, "out of memory");
	t->dev = dev;
	t->input_rate = input_rate;
	t->max_channels = max_channels;
	t->max_block_frames = max_block_frames;
	t->nco_mode = nco_mode;
	t->keep_mask = 0;
	t->in_stage = nullptr;
	t->in_hist[0] = t->in_hist[1] = nullptr;
	t->in_par = 0;
	t->submitted = false;
	t->audio_scale = 1.0f;
	{
		const char *e = getenv("WR_DEFER_POST");
		t->defer_post = !(e && *e == '0');
	}
	t->profiling = false;
	t->prof_stride = 1;
	t->prof_tick = 0;
	t->ev_used = 0;
	t->prof_ms = 0.0;
	t->prof_n = 0;
	int rc = dev_alloc_zero(&t->in_hist[0], (size_t)WR_HIST * 2);
	if (!rc)
		rc = dev_alloc_zero(&t->in_hist[1], (size_t)WR_HIST * 2);
	if (rc) {
		wr_tuner_destroy(t);
		return rc;
	}
	*tuner = t;
	return WR_OK;
}

extern "C" int wr_tuner_destroy(wr_tuner *t)
{
	if (!t)
		return WR_OK;
	(void)hipSetDevice(t->dev->device);
	(void)stream_close(t);
	(void)dev_stream_sync(t->dev);
	stream_free(t);
	for (Group *g : t->groups)
		group_free(g);
	for (hipEvent_t e : t->ev)
		(void)hipEventDestroy(e);
	for (hipEvent_t e : t->launch_ev)
		if (e)
			(void)hipEventDestroy(e);
	for (wr_tuner::RingSlot &r : t->ring) {
		if (r.done)
			(void)hipEventDestroy(r.done);
		(void)hipHostFree(r.host);
	}
	(void)hipFree(t->in_stage);
	(void)hipFree(t->in_hist[0]);
	(void)hipFree(t->in_hist[1]);
	delete t;
	return WR_OK;
}

static int tuner_quiesce(wr_tuner *t);
static int tuner_flush(wr_tuner *t);
static void stream_free(wr_tuner *t);
static int ring_push(wr_tuner *t, Group *g, unsigned long long seq, size_t k2, unsigned int used, bool direct);
static float *ring_reserve(wr_tuner *t, Group *g, size_t k2, unsigned int used);

static int tuner_launch_held(wr_tuner *t);

/* Blocks waiting for their successors (wr_tuner_set_blocks_per_launch) go out before anything is
 * staged or read: a setter takes effect at the boundary after the last block SUBMITTED, and a
 * getter sees the state after it. */
static int settle_held(wr_tuner *t)
{
	if (!t || (!t->held_count && !t->stream.live))
		return WR_OK;
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	return tuner_launch_held(t);
}

/* why the last chan_get() of this thread returned no channel: the error of sending the held blocks
 * out (already in wr_last_error()), or WR_OK when there simply is no such channel */
static thread_local int g_settle_rc = WR_OK;

static Chan *chan_get(wr_tuner *t, int chan)
{
	g_settle_rc = settle_held(t);
	if (g_settle_rc)
		return nullptr;
	if (!t || chan < 0 || (size_t)chan >= t->chans.size() || !t->chans[chan].in_use)
		return nullptr;
	return &t->chans[chan];
}

extern "C" int wr_chan_add(wr_tuner *t, int *chan)
{
	if (!t || !chan)
		return fail(WR_ERR_ARG, "wr_chan_add: bad argument");
	if (int rc = settle_held(t))
		return rc;
	int live = 0;
	for (const Chan &c : t->chans)
		live += c.in_use ? 1 : 0;
	if ((unsigned int)live >= t->max_channels)
		return fail(WR_ERR_STATE, "wr_chan_add: tuner already has %u channels", t->max_channels);
	int idx = -1;
	for (size_t i = 0; i < t->chans.size(); ++i)
		if (!t->chans[i].in_use) {
			idx = (int)i;
			break;
		}
	if (idx < 0) {
		t->chans.push_back(Chan());
		idx = (int)t->chans.size() - 1;
	}
	Chan &c = t->chans[idx];
	memset(&c, 0, sizeof(c));
	c.in_use = true;
	c.len1 = c.len2 = c.len1b = WR_FIR_LENGTH;
	c.mode = WR_AM;                /* Demodulator ctor, demodulator.cxx:34 */
	c.gain = 1.0f;                 /* what the reference reports: af_gain 0, squelch_threshold 0 (receiverhandler.cxx:118-119) */
	c.squelch = 0.0f;
	c.group = -1;
	c.slot = -1;
	*chan = idx;
	return WR_OK;
}

static int chan_unseat(wr_tuner *t, Chan &c, bool keep_state);

extern "C" int wr_chan_remove(wr_tuner *t, int chan)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_remove: no channel %d", chan);
	int rc = chan_unseat(t, *c, false);
	c->in_use = false;
	return rc;
}

extern "C" int wr_chan_count(wr_tuner *t, int *count)
{
	if (!t || !count)
		return fail(WR_ERR_ARG, "wr_chan_count: bad argument");
	int live = 0;
	for (const Chan &c : t->chans)
		live += c.in_use ? 1 : 0;
	*count = live;
	return WR_OK;
}

/* take a channel out of its group slot; optionally keep Demodulator prev_i/q
 * (they survive stop()/start() in the reference, quirk Q5) */
static int chan_unseat(wr_tuner *t, Chan &c, bool keep_state)
{
	if (c.group < 0)
		return WR_OK;
	Group *g = t->groups[c.group];
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	{
		int rc = tuner_quiesce(t);
		if (rc)
			return rc;
	}
	if (keep_state)
		HIP_TRY(hipMemcpy(c.prev_iq, g->dev.prev_iq[g->parity] + 2 * c.slot, 2 * sizeof(float),
		                  hipMemcpyDeviceToHost));
	g->owner[c.slot] = -1;
	g->active--;
	g->dirty = true;
	c.group = -1;
	c.slot = -1;
	c.cs_hist_reset = true;
	c.prev_dirty = keep_state;
	c.phase_dirty = true;
	return WR_OK;
}

/* seat a fully configured channel into the group matching its decimations */
static int chan_seat(wr_tuner *t, int idx)
{
	Chan &c = t->chans[idx];
	if (!c.have[0] || !c.have[1])
		return WR_OK;
	if (c.group >= 0) {
		Group *g = t->groups[c.group];
		if (g->d1 == c.decim[0] && g->d2 == c.decim[1] && g->d1b == (c.have[2] ? c.decim[2] : 0u) &&
		    g->l1 == c.len1 && g->l2 == c.len2 && g->l1b == (c.have[2] ? c.len1b : (unsigned int)WR_FIR_LENGTH)) {
			g->dirty = true;
			return WR_OK;
		}
		int rc = chan_unseat(t, c, true);
		if (rc)
			return rc;
	}
	int gi = -1;
	for (size_t i = 0; i < t->groups.size(); ++i)
		if (t->groups[i]->d1 == c.decim[0] && t->groups[i]->d2 == c.decim[1] &&
		    t->groups[i]->d1b == (c.have[2] ? c.decim[2] : 0u) &&
		    t->groups[i]->l1 == c.len1 && t->groups[i]->l2 == c.len2 &&
		    t->groups[i]->l1b == (c.have[2] ? c.len1b : (unsigned int)WR_FIR_LENGTH)) {
			gi = (int)i;
			break;
		}
	if (gi < 0) {
		if (dev_bind(t->dev))
			return WR_ERR_HIP;
		Group *g = nullptr;
		int rc = group_create(t, c.decim[0], c.have[2] ? c.decim[2] : 0u, c.decim[1], c.len1, c.len1b, c.len2, &g);
		if (rc)
			return rc;
		t->groups.push_back(g);
		gi = (int)t->groups.size() - 1;
	}
	Group *g = t->groups[gi];
	int slot = -1;
	for (unsigned int s = 0; s < g->slots; ++s)
		if (g->owner[s] < 0) {
			slot = (int)s;
			break;
		}
	if (slot < 0)
		return fail(WR_ERR_STATE, "no free slot in rate group %u/%u", g->d1, g->d2);
	g->owner[slot] = idx;
	g->active++;
	g->dirty = true;
	c.group = gi;
	c.slot = slot;
	c.cs_hist_reset = true;        /* fresh LowPass::block: zero history (lowpass.cxx:138-139) */
	c.dem_hist_reset = true;
	c.prev_dirty = true;
	c.phase_dirty = true;
	return WR_OK;
}

extern "C" int wr_chan_set_if(wr_tuner *t, int chan, int if_hz)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_set_if: no channel %d", chan);
	c->if_hz = if_hz;
	c->stepL = (unsigned int)wrd_phase_step(if_hz, t->input_rate) << 1;
	if (c->group >= 0)
		t->groups[c->group]->dirty = true;
	return WR_OK;
}

/* `coeff`: 64 taps (a shorter filter zero-extended), or `len` = 128 or 256 of them */
static int set_taps_common(wr_tuner *t, int chan, int stage, const float *coeff, unsigned int decim,
                           unsigned int len = WR_FIR_LENGTH)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "no channel %d", chan);
	if (stage < 0 || stage > 2)
		return fail(WR_ERR_ARG, "stage must be 0 (channel), 1 (audio) or 2 (second channel filter)");
	if (!decim)
		return fail(WR_ERR_ARG, "decimation must be >= 1");
	unsigned int *lens[3] = {&c->len1, &c->len2, &c->len1b};
	float *longs[3] = {c->taps_long, c->taps_long2, c->taps_long1b};
	*lens[stage] = len;
	if (len > WR_FIR_LENGTH)
		memcpy(longs[stage], coeff, sizeof(float) * len);
	if (len <= WR_FIR_LENGTH)
		memcpy(c->taps[stage], coeff, sizeof(float) * WR_FIR_LENGTH);
	c->decim[stage] = decim;
	c->have[stage] = true;
	return chan_seat(t, chan);
}

/* LowPass::_firLength as a run-time value in the fused path (lowpass.cxx:38-39 "FIXME: Make runtime
 * variable"): a filter of L <= 64 taps IS the 64-tap filter whose taps L..63 -- the ones that meet
 * the oldest samples -- are zero.  lowpass.cxx:150-158 adds the products oldest sample first, so the
 * padded filter starts with 64 - L products that are +-0 and then runs through exactly the additions
 * of the short one: the same bits (finite input). */
static bool fused_fir_length_ok(unsigned int n)
{
	/* 128 or 256 taps: k_tuner_ddc_long (channel filter), k_tuner_iq2 (second channel stage), k_tuner_demod +
	 * k_tuner_audio (audio filter) with 127 / 255 rows of history -- every stage of the tuner's own launch sequence */
	return n >= 2 && n <= (unsigned int)WR_FIR_FUSED_MAX && (n & (n - 1)) == 0;
}

extern "C" int wr_chan_set_taps_n(wr_tuner *t, int chan, int stage, const float *coeff_host,
                                  unsigned int fir_length, unsigned int decimation)
{
	if (!t || !coeff_host)
		return fail(WR_ERR_ARG, "wr_chan_set_taps: bad argument");
	if (!fused_fir_length_ok(fir_length))
		return fail(WR_ERR_ARG, "wr_chan_set_taps_n: fir_length %u is not a power of two in [2, %d] (longer filters run "
		                        "block by block: wr_fir_decimate_n)", fir_length, WR_FIR_FUSED_MAX);
	float coeff[WR_FIR_FUSED_MAX] = {0.0f};
	memcpy(coeff, coeff_host, sizeof(float) * fir_length);
	return set_taps_common(t, chan, stage, coeff, decimation, fir_length > WR_FIR_LENGTH ? fir_length : (unsigned int)WR_FIR_LENGTH);
}

extern "C" int wr_chan_set_taps(wr_tuner *t, int chan, int stage, const float *coeff_host,
                                unsigned int decimation)
{
	return wr_chan_set_taps_n(t, chan, stage, coeff_host, WR_FIR_LENGTH, decimation);
}

extern "C" int wr_chan_set_filter_n(wr_tuner *t, int chan, int stage, unsigned int fir_length,
                                    unsigned int passband, unsigned int out_rate)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_set_filter: no channel %d", chan);
	if (stage < 0 || stage > 2)
		return fail(WR_ERR_ARG, "stage must be 0 (channel), 1 (audio) or 2 (second channel filter)");
	if (!fused_fir_length_ok(fir_length))
		return fail(WR_ERR_ARG, "wr_chan_set_filter_n: fir_length %u is not a power of two in [2, %d]", fir_length,
		            WR_FIR_FUSED_MAX);
	unsigned int in_rate;
	if (stage == 0) {
		in_rate = t->input_rate;
	} else {
		if (!c->have[0])
			return fail(WR_ERR_STATE, "set the channel filter (stage 0) before the %s", stage == 1 ? "audio filter"
			            : "second channel filter");
		in_rate = t->input_rate / c->decim[0];
		if (stage == 1 && c->have[2])
			in_rate /= c->decim[2];               /* the audio filter follows the LAST channel stage */
	}
	if (!out_rate || out_rate > in_rate)
		return fail(WR_ERR_RATE, "output rate %u not a decimation of %u", out_rate, in_rate);
	unsigned int decim = in_rate / out_rate;           /* dspblock.cxx:119-121 */
	if (in_rate / decim != out_rate || in_rate % out_rate)
		return fail(WR_ERR_RATE, "Sample rates must be integer related (%u -> %u)", in_rate, out_rate);
	float coeff[WR_FIR_FUSED_MAX] = {0.0f};
	wrd_lowpass_design(fir_length, passband, in_rate, coeff);
	return set_taps_common(t, chan, stage, coeff, decim, fir_length > WR_FIR_LENGTH ? fir_length : (unsigned int)WR_FIR_LENGTH);
}

extern "C" int wr_chan_set_filter(wr_tuner *t, int chan, int stage, unsigned int passband,
                                  unsigned int out_rate)
{
	return wr_chan_set_filter_n(t, chan, stage, WR_FIR_LENGTH, passband, out_rate);
}

extern "C" int wr_chan_set_mode(wr_tuner *t, int chan, int mode)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_set_mode: no channel %d", chan);
	if (mode < WR_AM || mode > WR_LSB)
		return fail(WR_ERR_ARG, "wr_chan_set_mode: bad mode %d", mode);
	c->mode = mode;
	if (c->group >= 0)
		t->groups[c->group]->dirty = true;
	return WR_OK;
}

/* The two receiver controls the reference's REST interface names and never implements ("FIXME:
 * af_gain, squelch", receiverhandler.cxx:112,127; both reported as 0, :118-119).  Staged like every
 * other setter: they take effect at the next block boundary. */
extern "C" int wr_chan_set_af_gain(wr_tuner *t, int chan, float gain_db)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_set_af_gain: no channel %d", chan);
	if (!(gain_db == gain_db) || gain_db < -200.0f || gain_db > 200.0f)
		return fail(WR_ERR_ARG, "wr_chan_set_af_gain: %g dB", (double)gain_db);
	c->gain = (float)pow(10.0, (double)gain_db / 20.0);
	if (c->group >= 0)
		t->groups[c->group]->dirty = true;
	return WR_OK;
}

extern "C" int wr_chan_set_squelch(wr_tuner *t, int chan, float threshold_dbfs, int enable)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_set_squelch: no channel %d", chan);
	if (enable && (!(threshold_dbfs == threshold_dbfs) || threshold_dbfs < -300.0f || threshold_dbfs > 100.0f))
		return fail(WR_ERR_ARG, "wr_chan_set_squelch: %g dBFS", (double)threshold_dbfs);
	c->squelch = enable ? (float)pow(10.0, (double)threshold_dbfs / 10.0) : 0.0f;
	if (c->group >= 0)
		t->groups[c->group]->dirty = true;
	return WR_OK;
}

extern "C" int wr_tuner_keep_stages(wr_tuner *t, unsigned int stage_mask)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (int rc = settle_held(t))
		return rc;
	t->keep_mask = stage_mask;
	return WR_OK;
}

extern "C" int wr_chan_get_state(wr_tuner *t, int chan, unsigned int *phase, float *prev_iq)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_get_state: no channel %d", chan);
	if (phase)
		*phase = c->phaseL >> 1;
	if (prev_iq) {
		if (c->group >= 0 && !c->prev_dirty) {
			Group *g = t->groups[c->group];
			if (dev_bind(t->dev))
				return WR_ERR_HIP;
			{
				int rc = tuner_quiesce(t);
				if (rc)
					return rc;
			}
			HIP_TRY(hipMemcpy(prev_iq, g->dev.prev_iq[g->parity] + 2 * c->slot, 2 * sizeof(float),
			                  hipMemcpyDeviceToHost));
		} else {
			prev_iq[0] = c->prev_iq[0];
			prev_iq[1] = c->prev_iq[1];
		}
	}
	return WR_OK;
}

extern "C" int wr_chan_set_state(wr_tuner *t, int chan, unsigned int phase, const float *prev_iq)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_set_state: no channel %d", chan);
	c->phaseL = phase << 1;
	c->phase_dirty = true;
	if (prev_iq) {
		c->prev_iq[0] = prev_iq[0];
		c->prev_iq[1] = prev_iq[1];
		c->prev_dirty = true;
	}
	if (c->group >= 0)
		t->groups[c->group]->dirty = true;
	return WR_OK;
}

extern "C" int wr_chan_slot(wr_tuner *t, int chan, int *slot)
{
	Chan *c = chan_get(t, chan);
	if (!c || !slot)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_slot: bad argument");
	if (c->group < 0)
		return fail(WR_ERR_STATE, "channel %d has no filters yet", chan);
	*slot = c->slot;
	return WR_OK;
}

/* launch whatever post stage is still pending (results of the last submit wanted now) */
static int tuner_launch_held(wr_tuner *t);

static int tuner_flush(wr_tuner *t)
{
	{
		int rc = tuner_launch_held(t);
		if (rc)
			return rc;
	}
	for (Group *g : t->groups) {
		if (!g->post_pending)
			continue;
		HIP_TRY(wrk_tuner_post_args(t->dev->stream, g->post_args));
		g->post_pending = false;
		int rc = ring_push(t, g, g->pend_seq, g->pend_k2, g->pend_slots, g->pend_direct);
		if (rc)
			return rc;
	}
	return WR_OK;
}

/* host is about to touch device arrays the kernels of the last submit read or write: get the
 * pending post stage out, then drain the stream */
static int seek_materialize(wr_tuner *t, Group *g)
{
	if (!g->seek_pending)
		return WR_OK;
	/* (a post stage still waiting for the next submit would write ITS end-of-block state over the seek's: the caller
	 * has flushed) */
	g->seek_pending = false;
	HIP_TRY(wrk_seek(t->dev->stream, g->dev, (unsigned int)g->slots, g->sp, g->parity, g->p2, g->seek_frame));
	return WR_OK;
}

static int tuner_quiesce(wr_tuner *t)
{
	int rc = tuner_flush(t);
	if (rc)
		return rc;
	for (Group *g : t->groups)
		if ((rc = seek_materialize(t, g)) != WR_OK)
			return rc;
	HIP_TRY(dev_stream_sync(t->dev));
	return stream_check(t);
}

/* push the host shadow of one group's parameters to its device arrays */
static int group_upload(wr_tuner *t, Group *g)
{
	const size_t S = g->slots;
	hipStream_t st = t->dev->stream;
	{
		int rc = tuner_quiesce(t);
		if (rc)
			return rc;
	}
	std::vector<unsigned int> step(S, 0);
	std::vector<int> flags(S, 0), mode(S, -1);      /* mode < 0 marks an idle slot */
	std::vector<float> taps1(S * WR_FIR_LENGTH, 0.0f), taps2(S * g->l2, 0.0f);
	std::vector<float> taps1b(g->d1b ? S * g->l1b : 0, 0.0f), gain(S, 1.0f), squelch(S, 0.0f);
	std::vector<float> taps1L(g->l1 > WR_FIR_LENGTH ? S * g->l1 : 0, 0.0f);
	g->use_gain = g->use_squelch = false;
	for (size_t s = 0; s < S; ++s) {
		int ci = g->owner[s];
		if (ci < 0)
			continue;
		Chan &c = t->chans[ci];
		step[s] = c.stepL;
		mode[s] = c.mode;
		flags[s] = 1;
		for (int j = 0; j < WR_FIR_LENGTH; ++j)
			taps1[(size_t)j * S + s] = c.taps[0][j];
		for (unsigned int j = 0; j < g->l2; ++j)
			taps2[(size_t)j * S + s] = g->l2 > WR_FIR_LENGTH ? c.taps_long2[j] : c.taps[1][j];
		if (g->d1b)
			for (unsigned int j = 0; j < g->l1b; ++j)
				taps1b[(size_t)j * S + s] = g->l1b > WR_FIR_LENGTH ? c.taps_long1b[j] : c.taps[2][j];
		if (g->l1 > WR_FIR_LENGTH)
			for (unsigned int j = 0; j < g->l1; ++j)
				taps1L[(size_t)j * S + s] = c.taps_long[j];
		gain[s] = c.gain;
		squelch[s] = c.squelch;
		g->use_gain = g->use_gain || c.gain != 1.0f;
		g->use_squelch = g->use_squelch || c.squelch > 0.0f;
	}
	/* Receivers of a tuner nearly always share one channel filter (radio.cxx:78-79 sets the same
	 * passband/rate for all), and when they do not (receiverhandler.cxx:130-137: every receiver has
	 * its own passband control) a lane group still holds only a few distinct ones.  The fast kernel
	 * folds the taps into the shared sample window, one copy of the window per distinct filter (up
	 * to WR_TAPSETS); a lane group with more than that takes the per-lane-taps kernel. */
	bool uniform = true;
	int first_rep = -1;                  /* a channel of the first lane group that has any */
	bool one_filter = true;
	unsigned long long umask = 0, fmask = 0;
	std::vector<float> taps1u(S * WR_TAPSETS, 0.0f);
	std::vector<int> tapsel(S, 0);
	memset(g->nsets, 0, sizeof(g->nsets));
	for (size_t base = 0; base < S; base += WR_LANES) {
		const size_t grp = base / WR_LANES;
		int reps[WR_TAPSETS];
		unsigned int nrep = 0;
		bool few = true;
		for (size_t s = base; s < base + WR_LANES; ++s) {
			int ci = g->owner[s];
			if (ci < 0)
				continue;
			unsigned int q = 0;
			for (; q < nrep; ++q)
				if (!memcmp(t->chans[ci].taps[0], t->chans[reps[q]].taps[0], sizeof(float) * WR_FIR_LENGTH))
					break;
			if (q == nrep) {
				if (nrep == WR_TAPSETS) {
					few = false;
					break;
				}
				reps[nrep++] = ci;
			}
			tapsel[s] = (int)q;
		}
		if (nrep) {
			if (first_rep < 0)
				first_rep = reps[0];
			if (nrep > 1 || !few ||
			    memcmp(t->chans[reps[0]].taps[0], t->chans[first_rep].taps[0], sizeof(float) * WR_FIR_LENGTH))
				one_filter = false;
		}
		if (!few || grp >= 64) {
			for (size_t s = base; s < base + WR_LANES; ++s)
				tapsel[s] = 0;
			if (nrep)
				uniform = false;
			continue;
		}
		for (unsigned int q = 0; q < nrep; ++q)
			for (size_t j = 0; j < WR_LANES; ++j)               /* window order: sample j meets coeff[63 - j] */
				taps1u[(grp * WR_TAPSETS + q) * WR_LANES + j] = t->chans[reps[q]].taps[0][WR_FIR_LENGTH - 1 - j];
		g->nsets[grp] = (unsigned char)(nrep ? nrep : 1);
		fmask |= 1ull << grp;
		if (nrep <= 1)
			umask |= 1ull << grp;
		else
			uniform = false;
	}
	/* the audio filter likewise (one per lane group or the per-lane path; its taps go through the
	 * scalar cache, see post_role) */
	const unsigned int l2 = g->l2;
	std::vector<float> taps2u(S / WR_LANES * l2, 0.0f);         /* [lane groups][l2] */
	unsigned long long u2mask = 0;
	for (size_t base = 0; base < S && base / WR_LANES < 64; base += WR_LANES) {
		int rep = -1;
		bool same = true;
		for (size_t s = base; s < base + WR_LANES && same; ++s) {
			int ci = g->owner[s];
			if (ci < 0)
				continue;
			if (rep < 0)
				rep = ci;
			else
				same = l2 > WR_FIR_LENGTH ? !memcmp(t->chans[ci].taps_long2, t->chans[rep].taps_long2, sizeof(float) * l2)
				                          : !memcmp(t->chans[ci].taps[1], t->chans[rep].taps[1], sizeof(float) * WR_FIR_LENGTH);
		}
		if (!same || rep < 0)
			continue;
		memcpy(&taps2u[base / WR_LANES * l2], l2 > WR_FIR_LENGTH ? t->chans[rep].taps_long2 : t->chans[rep].taps[1], sizeof(float) * l2);
		u2mask |= 1ull << (base / WR_LANES);
	}
	g->uniform2_mask = u2mask;
	g->uniform_taps = uniform;
	g->uniform_mask = umask;
	g->fewsets_mask = fmask;
	g->one_filter = one_filter && first_rep >= 0;
	g->long_uniform = g->long_one = false;
	if (g->l1 > WR_FIR_LENGTH) {
		g->long_uniform = g->long_one = true;
		int rep_all = -1;
		for (size_t base = 0; base < S && g->long_uniform; base += WR_LANES) {
			int rep = -1;
			for (size_t s = base; s < base + WR_LANES; ++s) {
				const int ci = g->owner[s];
				if (ci < 0)
					continue;
				if (rep < 0)
					rep = ci;
				else if (memcmp(t->chans[ci].taps_long, t->chans[rep].taps_long, sizeof(float) * g->l1))
					g->long_uniform = false;
			}
			if (rep >= 0) {
				if (rep_all < 0)
					rep_all = rep;
				else if (memcmp(t->chans[rep].taps_long, t->chans[rep_all].taps_long, sizeof(float) * g->l1))
					g->long_one = false;
			}
		}
		g->long_one = g->long_one && g->long_uniform;
	}
	if (g->one_filter)
		/* two lane groups that share a wave take the window from the first one's entry: an emptied lane
		 * group in between holds the common filter too */
		for (size_t grp = 0; grp < S / WR_LANES; ++grp)
			for (size_t j = 0; j < WR_LANES; ++j)
				taps1u[(grp * WR_TAPSETS) * WR_LANES + j] = t->chans[first_rep].taps[0][WR_FIR_LENGTH - 1 - j];
	/* per-slot turns of the ROTATE NCO (see WrGroupDev) */
	std::vector<float> rot(S * 4, 0.0f);
	{
		const float *turn = t->dev->turn_host;
		for (size_t s = 0; s < S; ++s) {
			const unsigned int Sx = step[s] >> 16;
			rot[4 * s + 0] = turn[(Sx + 16384u) & 0xFFFFu];
			rot[4 * s + 1] = turn[Sx & 0xFFFFu];
			rot[4 * s + 2] = turn[(Sx + 16385u) & 0xFFFFu];
			rot[4 * s + 3] = turn[(Sx + 1u) & 0xFFFFu];
		}
	}
	/* pageable sources: hipMemcpyAsync stages them before returning */
	HIP_TRY(hipMemcpyAsync(g->dev.step, step.data(), S * sizeof(unsigned int), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.flags, flags.data(), S * sizeof(int), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.mode, mode.data(), S * sizeof(int), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.taps1, taps1.data(), taps1.size() * sizeof(float), hipMemcpyHostToDevice, st));
	if (g->l1 > WR_FIR_LENGTH)
		HIP_TRY(hipMemcpyAsync(g->dev.taps1L, taps1L.data(), taps1L.size() * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.taps2, taps2.data(), taps2.size() * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.rot, rot.data(), rot.size() * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.taps1u, taps1u.data(), taps1u.size() * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.taps2u, taps2u.data(), taps2u.size() * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.tapsel, tapsel.data(), tapsel.size() * sizeof(int), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.gain, gain.data(), S * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(g->dev.squelch, squelch.data(), S * sizeof(float), hipMemcpyHostToDevice, st));
	if (g->d1b)
		HIP_TRY(hipMemcpyAsync(g->dev.taps1b, taps1b.data(), taps1b.size() * sizeof(float), hipMemcpyHostToDevice, st));
	for (size_t s = 0; s < S; ++s) {
		int ci = g->owner[s];
		if (ci < 0)
			continue;
		Chan &c = t->chans[ci];
		if (c.phase_dirty) {
			HIP_TRY(hipMemcpyAsync(g->dev.phase[g->sp] + s, &c.phaseL, sizeof(unsigned int), hipMemcpyHostToDevice, st));
			c.phase_dirty = false;
			/* ROTATE keeps the turn INTO the next frame; a phase set from outside breaks that
			 * chain, so the channel filter starts from an empty history (see wr_chan_set_state) */
			if (t->nco_mode == WR_NCO_ROTATE)
				c.cs_hist_reset = true;
		}
		if (c.prev_dirty) {
			HIP_TRY(hipMemcpyAsync(g->dev.prev_iq[g->parity] + 2 * s, c.prev_iq, 2 * sizeof(float), hipMemcpyHostToDevice, st));
			c.prev_dirty = false;
		}
		if (c.cs_hist_reset) {
			/* 63 LO rows of this slot: one float2 per row, stride S float2 */
			HIP_TRY(hipMemset2DAsync(g->dev.hist_cs[g->sp] + 2 * s, S * 2 * sizeof(float), 0, 2 * sizeof(float),
			                         WR_HIST, st));
			HIP_TRY(hipMemset2DAsync(g->dev.hist_lo[g->sp] + 2 * s, S * 2 * sizeof(float), 0, 2 * sizeof(float),
			                         WR_HIST, st));
			HIP_TRY(hipMemset2DAsync(g->dev.iq2_hist[g->p2] + 2 * s, S * 2 * sizeof(float), 0, 2 * sizeof(float),
			                         g->l1b - 1, st));
			if (g->l1 > WR_FIR_LENGTH)        /* the L - 1 mixed frames of this slot */
				HIP_TRY(hipMemset2DAsync(g->dev.mixhist[g->sp] + 2 * s, S * 2 * sizeof(float), 0, 2 * sizeof(float),
				                         g->l1 - 1, st));
			c.cs_hist_reset = false;
		}
		if (c.dem_hist_reset) {
			/* 63 history rows of this slot: one float per row, stride S */
			HIP_TRY(hipMemset2DAsync(g->dev.dem[g->parity] + s, S * sizeof(float), 0, sizeof(float), g->l2 - 1, st));
			c.dem_hist_reset = false;
		}
	}
	HIP_TRY(hipStreamSynchronize(st));     /* host vectors go out of scope */
	g->dirty = false;
	return WR_OK;
}

/* fold recorded event pairs into the running mean once more than `keep` pairs are pending */
static int prof_drain(wr_tuner *t, size_t keep)
{
	if (t->ev_used / 2 <= keep)
		return WR_OK;
	HIP_TRY(dev_stream_sync(t->dev));
	for (size_t i = 0; i + 1 < t->ev_used; i += 2) {
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]));
		const unsigned int span = (i / 2 < t->ev_span.size() && t->ev_span[i / 2]) ? t->ev_span[i / 2] : 1u;
		t->prof_ms += ms;                           /* prof_ms / prof_n = mean per launch */
		t->prof_n += span;
	}
	t->ev_used = 0;
	t->ev_span.clear();
	return WR_OK;
}

extern "C" int wr_tuner_profile(wr_tuner *t, int enable)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (int rc = settle_held(t))                    /* (a streaming launch stamps its events when it is opened) */
		return rc;
	if (t->ev_used & 1)
		t->ev_used--;                               /* a group left open: its start event is dropped */
	t->profiling = enable != 0;
	t->prof_stride = enable > 1 ? (unsigned int)enable : 1u;
	t->prof_tick = 0;
	return WR_OK;
}

extern "C" int wr_tuner_profile_read(wr_tuner *t, unsigned int *launches, double *mean_ms)
{
	if (!t || !launches || !mean_ms)
		return fail(WR_ERR_ARG, "wr_tuner_profile_read: bad argument");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	if (int rc = settle_held(t))
		return rc;
	int rc = prof_drain(t, 0);
	if (rc)
		return rc;
	*launches = t->prof_n;
	*mean_ms = t->prof_n ? t->prof_ms / t->prof_n : 0.0;
	t->prof_ms = 0.0;
	t->prof_n = 0;
	return WR_OK;
}

static int tuner_submit(wr_tuner *t, const void *iq, size_t nframes, int where, bool u8);

extern "C" int wr_tuner_submit(wr_tuner *t, const float *iq, size_t nframes, int where)
{
	return tuner_submit(t, iq, nframes, where, false);
}

extern "C" int wr_tuner_submit_u8(wr_tuner *t, const uint8_t *iq_u8, size_t nframes, int where)
{
	return tuner_submit(t, iq_u8, nframes, where, true);
}

static int tuner_submit_now(wr_tuner *t, const void *iq, size_t nframes, int where, bool u8);
static bool stream_follows(const wr_tuner *t, size_t nframes, int where, bool u8);
static int stream_bell(wr_tuner *t, const void *iq, bool behind_upload);
static int stream_open(wr_tuner *t, const void *iq, size_t nframes, bool u8, bool *took);
static int stream_host_u8(wr_tuner *t, const uint8_t *bytes, size_t nframes, bool *took);
static Group *single_group(wr_tuner *t);

static int tuner_launch_held(wr_tuner *t)
{
	if (t->stream.live)                                 /* (a streaming launch holds no blocks back: it is told that none follows) */
		return stream_close(t);
	if (!t->held_count)
		return WR_OK;
	const float *base = t->held_base;
	const size_t frames = t->held_frames;
	t->held_count = 0;
	t->held_base = nullptr;
	t->held_frames = 0;
	return tuner_submit_now(t, base, frames, WR_DEVICE, false);
}

/* may this block wait for its successor?  Only whole audio frames: a block that is not a multiple
 * of every rate group's decimations restarts the decimation phase at its start (dspblock.cxx:177-178
 * truncates per block), which a merged block would not */
static bool block_can_be_held(const wr_tuner *t, size_t nframes)
{
	for (const Group *g : t->groups) {
		if (g->active <= 0)
			continue;
		const size_t q = (size_t)g->d1 * (g->d1b ? g->d1b : 1u) * g->d2;
		if (!q || nframes % q)
			return false;
	}
	return true;
}

/* what a submit can be refused for before anything is enqueued -- checked before a block is HELD too,
 * so that a held wr_tuner_submit never returns WR_OK for a block a later call would have to refuse */
static int submit_precheck(const wr_tuner *t, size_t nframes, int where)
{
	if (nframes > t->max_block_frames)
		return fail(WR_ERR_ARG, "wr_tuner_submit: %zu frames exceeds max_block_frames %zu", nframes,
		            t->max_block_frames);
	if (where != WR_HOST && where != WR_DEVICE)
		return fail(WR_ERR_ARG, "wr_tuner_submit: bad `where`");
	for (const Chan &c : t->chans)
		if (c.in_use && c.group < 0)
			return fail(WR_ERR_STATE, "a channel has no filters (LowPass::init: \"Must specify either "
			                          "decimation or output rate\")");
	return WR_OK;
}

static int tuner_submit(wr_tuner *t, const void *iq, size_t nframes, int where, bool u8)
{
	if (!t || (nframes && !iq))
		return fail(WR_ERR_ARG, "wr_tuner_submit: bad argument");
	if (int rc = submit_precheck(t, nframes, where))
		return rc;
	if (t->stream.live) {
		if (stream_follows(t, nframes, where, u8))
			return stream_bell(t, iq, false);
		if (!(t->stream.ext && where == WR_HOST && u8 && stream_follows(t, nframes, WR_DEVICE, true)))
			if (int rc = stream_close(t))
				return rc;
	}
	if (t->stream.enabled && where == WR_DEVICE && nframes) {
		bool took = false;
		t->stream.ext = false;
		const int rc = stream_open(t, iq, nframes, u8, &took);
		if (rc || took)
			return rc;
	}
	if (t->stream.enabled && t->stream.host_bytes && where == WR_HOST && u8 && nframes) {
		bool took = false;
		const int rc = stream_host_u8(t, (const uint8_t *)iq, nframes, &took);
		if (rc || took)
			return rc;
	}
	if (t->coalesce > 1 && where == WR_DEVICE && !u8 && nframes && block_can_be_held(t, nframes)) {
		const float *p = (const float *)iq;
		/* (a setter called since the last submit has sent the held blocks out already: settle_held) */
		const bool follows = t->held_count && p == t->held_base + 2 * t->held_frames &&
		                     nframes == t->held_each && t->held_frames + nframes <= t->max_block_frames;
		if (!follows) {
			int rc = tuner_launch_held(t);
			if (rc)
				return rc;
			if (nframes * 2 > t->max_block_frames)          /* no room for a second one: nothing to wait for */
				return tuner_submit_now(t, iq, nframes, where, u8);
			t->held_base = p;
			t->held_frames = 0;
			t->held_each = nframes;
		}
		t->held_frames += nframes;
		t->held_count++;
		if (t->held_count >= t->coalesce || t->held_frames + nframes > t->max_block_frames)
			return tuner_launch_held(t);
		return WR_OK;
	}
	{
		int rc = tuner_launch_held(t);                      /* keep the stream in order */
		if (rc)
			return rc;
	}
	return tuner_submit_now(t, iq, nframes, where, u8);
}

static int tuner_submit_now(wr_tuner *t, const void *iq, size_t nframes, int where, bool u8)
{
	if (!t || (nframes && !iq))
		return fail(WR_ERR_ARG, "wr_tuner_submit: bad argument");
	if (int rc = submit_precheck(t, nframes, where))
		return rc;
	wr_dev *d = t->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	hipStream_t st = d->stream;

	const void *src = iq;
	if (where == WR_HOST) {
		if (!t->in_stage)
			HIP_TRY(hipMalloc((void **)&t->in_stage, t->max_block_frames * 2 * sizeof(float)));
		/* r04: a block in PAGE-LOCKED host memory whose receivers all read it through sparse windows of one shape -- the same
		 * decimation, the same channel-filter length, a window at most every second filter length -- never crosses PCIe
		 * whole: the staging kernel brings over the frames under the taps and the tail (wr_stage_windows_from_host says
		 * which), as floats whatever the source format.  Pageable memory, mixed shapes, dense windows: the copy, as before. */
		bool sparse = false;
		if (nframes && host_sparse_enabled() && !(((uintptr_t)iq | (uintptr_t)t->in_stage) & 15u)) {
			unsigned int sd = 0, sl = 0;
			bool same = true;
			for (const Group *g : t->groups) {
				if (g->active <= 0)
					continue;
				if (!sd) {
					sd = g->d1;
					sl = g->l1;
				} else if (g->d1 != sd || g->l1 != sl) {
					same = false;
				}
			}
			void *mapped = nullptr;
			if (same && sd && sd >= 2u * sl && nframes >= 4u * (size_t)sd) {
				/* is the block page-locked?  Asked once per buffer: the same source comes back block after block, and for a
				 * pageable one the question is a failing runtime call every time.  Only the probe's OWN error is cleared
				 * (a blanket hipGetLastError() here could swallow the pending error of an earlier asynchronous launch) */
				if (iq != t->probe_ptr || t->probe_gen != d->reg_gen || !t->probe_left--) {
					if (hipHostGetDevicePointer(&mapped, const_cast<void *>(iq), 0) != hipSuccess || !mapped) {
						(void)hipGetLastError();
						mapped = nullptr;
						t->probe_ptr = iq;                  /* pageable: remembered until something is page-locked or released
						                                       through this library (a "yes" is asked again every time: it is cheap,
						                                       and an address that is no longer mapped must never be used) */
						t->probe_gen = d->reg_gen;
						t->probe_left = 64;                 /* (... or until 64 blocks later: the application may page-lock it itself) */
					}
				}
			}
			if (mapped) {
				HIP_TRY(wrk_stage_windows(st, mapped, u8, t->in_stage, nframes, sd, sl, sl - 1u));
				if (int rc = upload_mark(d, st))            /* (wr_dev_wait_uploads: the kernel has read the host block) */
					return rc;
				sparse = true;
			}
		}
		if (nframes && !sparse)
			HIP_TRY(hipMemcpyAsync(t->in_stage, iq, nframes * 2 * (u8 ? sizeof(uint8_t) : sizeof(float)),
			                       hipMemcpyHostToDevice, st));
		src = t->in_stage;
		t->last_staging = sparse ? 2 : 1;
		if (sparse)
			u8 = false;                          /* the staged block holds floats */
	}
	const float *cur = u8 ? nullptr : (const float *)src;
	const uint8_t *cur_u8 = u8 ? (const uint8_t *)src : nullptr;

	t->stream.last_iq = nullptr;             /* (the last block's channel IQ is in the group's own buffers again) */
	bool hist_written = false;
	bool marked = false, unmarked = false;   /* wr_tuner_mark_launches: launches that stamped the submit's event / that could not */
	const unsigned long long seq = t->submit_seq++;
	/* stride 1: this submit's launch stamps its own start and stop (hipExtLaunchKernelGGL).  Stride
	 * n > 1: an event before the launch of the group's first submit and one after the launch of its
	 * last -- n launches and the gaps between them per pair, at 1/n of the events' own cost. */
	const unsigned int prof_pos = t->profiling ? t->prof_tick++ % t->prof_stride : 0u;
	const bool prof_now = t->profiling && t->prof_stride == 1;
	const bool group_first = t->profiling && t->prof_stride > 1 && prof_pos == 0;
	const bool group_last = t->profiling && t->prof_stride > 1 && prof_pos == t->prof_stride - 1 && (t->ev_used & 1);
	for (Group *g : t->groups) {
		if (g->active <= 0) {
			g->last_k1 = g->last_k2 = 0;
			continue;
		}
		if (g->dirty) {
			int rc = group_upload(t, g);
			if (rc)
				return rc;
		}
		WrTunerLaunch L;
		L.cur = cur;
		L.cur_u8 = cur_u8;
		L.hist = t->in_hist[t->in_par];
		L.hist_next = t->in_hist[t->in_par ^ 1];
		L.parity = g->parity;
		L.sp = g->sp;
		L.cb = g->cb;
		L.nframes = nframes;
		L.d1 = g->d1;
		L.d2 = g->d2;
		L.slots = g->slots;
		L.slots_used = group_slots_used(g);
		L.k1 = nframes / g->d1;                 /* dspblock.cxx:177-178 */
		L.k2 = L.k1 / g->d2;
		L.k2max = g->k2max;
		L.nco_mode = t->nco_mode;
		L.uniform_mask = g->uniform_mask;
		L.uniform2_mask = g->uniform2_mask;
		L.fewsets_mask = g->fewsets_mask;
		L.one_filter = g->one_filter ? 1 : 0;
		memcpy(L.nsets, g->nsets, sizeof(L.nsets));
		L.audio_scale = t->audio_scale;
		L.use_gain = g->use_gain ? 1 : 0;
		L.use_squelch = g->use_squelch ? 1 : 0;
		L.ev_start = L.ev_stop = nullptr;
		if (prof_now || group_first) {
			int rc = (t->ev_used & 1) ? WR_OK : prof_drain(t, 64);
			if (rc)
				return rc;
			while (t->ev.size() < t->ev_used + 2) {
				hipEvent_t e;
				HIP_TRY(hipEventCreate(&e));
				t->ev.push_back(e);
			}
		}
		if (prof_now) {
			L.ev_start = t->ev[t->ev_used];     /* stamped by the launch itself (wrk_tuner_ddc) */
			L.ev_stop = t->ev[t->ev_used + 1];
			if (t->mark_launches)
				unmarked = true;                /* the launch's stop event is the profiler's: the completion mark is an
				                                   ordinary record behind it (below) */
		} else if (t->mark_launches && g->l1 <= WR_FIR_LENGTH) {
			L.ev_start = nullptr;               /* completion only: the slot of this submit (every rate group's launch
			                                       stamps it in turn: the last one stands) */
			L.ev_stop = t->launch_ev[(t->launches_marked + 1) % 4];
			marked = true;
		} else if (t->mark_launches) {
			unmarked = true;
		}
		if (group_first && !(t->ev_used & 1)) {
			/* (once per submit: the first rate group's launch is the first of the bracket) */
			HIP_TRY(hipEventRecord(t->ev[t->ev_used], st));
			t->ev_used += 1;
		}
		/* a seek that is still pending (wr_tuner_seek): this launch takes the phase in closed form and reads the
		 * all-zero state sets; with the demodulator rows kept (two kernels) it is made real first */
		const bool two_kernels = (t->keep_mask & (1u << WR_STAGE_DEMOD)) != 0 || !wrk_tuner_post_supported(L.d2);
		WrGroupDev Gs = g->dev;
		if (g->seek_pending) {
			if (two_kernels || g->d1b || g->l1 > WR_FIR_LENGTH) {
				if (g->post_pending) {
					HIP_TRY(wrk_tuner_post_args(st, g->post_args));
					g->post_pending = false;
					int rc = ring_push(t, g, g->pend_seq, g->pend_k2, g->pend_slots, g->pend_direct);
					if (rc)
						return rc;
				}
				int rc = seek_materialize(t, g);
				if (rc)
					return rc;
			} else {
				L.seeking = true;
				L.seek_lo = (unsigned int)g->seek_frame;
				Gs.hist_cs[g->sp] = g->z_hist;
				Gs.hist_lo[g->sp] = g->z_hist;
				Gs.prev_iq[g->parity] = g->z_prev;
				Gs.dem[g->parity] = g->z_dem;
				g->seek_pending = false;
			}
		}
		/* The previous block's post stage rides along with this block's DDC where the kernel
		 * variant can take it (wrk_tuner_ddc says); otherwise it goes out on its own first. */
		bool rode = false;
		if (g->l1 > WR_FIR_LENGTH)
			/* a channel filter of 128 or 256 taps: the plain kernel with the reference's arithmetic (wr_kernels.hip:
			 * k_tuner_ddc_long), which also rolls phase and mixed history; r04: its ROTATE form carries the previous block's post stage like the 64-tap kernel */
			HIP_TRY(wrk_tuner_ddc_long(st, L, g->dev, g->l1, d->table, d->num_cus,
			                           t->nco_mode != WR_NCO_EXACT && g->long_uniform && long_rot_enabled(), g->long_one, d->hi_cs, d->lo_cs,
			                           g->post_pending ? &g->post_args : nullptr, &rode));
		else
			HIP_TRY(wrk_tuner_ddc(st, L, Gs, t->nco_mode == WR_NCO_ROTATE ? d->table_turn : d->table, d->hi_cs, d->lo_cs,
			                      d->num_cus, g->post_pending ? &g->post_args : nullptr, &rode));
		if (prof_now) {
			t->ev_span.resize(t->ev_used / 2 + 1, 1u);
			t->ev_used += 2;
		}
		if (g->post_pending) {
			if (!rode)
				HIP_TRY(wrk_tuner_post_args(st, g->post_args));
			g->post_pending = false;
			int rc = ring_push(t, g, g->pend_seq, g->pend_k2, g->pend_slots, g->pend_direct);
			if (rc)
				return rc;
		}
		/* A second channel-filter stage sits between the DDC and the demodulator: its kernel runs
		 * here, and everything after it works on ITS output (chan_iq2) at ITS rate. */
		WrTunerLaunch Lp = L;
		WrGroupDev Gp = Gs;
		if (g->d1b) {
			HIP_TRY(wrk_tuner_iq2(st, g->dev, g->slots, L.slots_used, L.k1, g->d1b, g->cb, g->p2));
			g->p2 ^= 1;
			Lp.k1 = L.k1 / g->d1b;
			Lp.k2 = Lp.k1 / g->d2;
			Gp.chan_iq[0] = g->dev.chan_iq2[0];
			Gp.chan_iq[1] = g->dev.chan_iq2[1];
		}
		/* demodulator output wanted (wr_tuner_keep_stages) or an unusual audio decimation: demod
		 * and audio filter as two kernels with the demod rows in HBM, at once.  Otherwise one
		 * fused pass -- deferred to the next launch where that launch can carry it. */
		/* (r04: a group with a long channel filter defers too where its launch can carry a post stage: the ROTATE kernel,
		 * every lane group on one long filter) */
		const bool long_rides = g->l1 > WR_FIR_LENGTH && g->long_uniform && long_rot_enabled();
		/* (r05: an audio filter of 128 / 256 taps -- k_tuner_post<D2, 2 | 4> -- goes out on its own behind the DDC: the
		 * workgroups that ride are compiled for 64 taps) */
		/* (r05: a group with a second channel stage defers as well -- its post stage reads chan_iq2, which the NEXT block's
		 * k_tuner_iq2 does not touch: that one writes the other buffer, behind the launch the post stage rides in) */
		const bool defer = !two_kernels && (g->l1 <= WR_FIR_LENGTH || long_rides) && Lp.k1 && t->defer_post &&
		                   t->nco_mode == WR_NCO_ROTATE && g->l2 == WR_FIR_LENGTH;
		if (two_kernels) {
			HIP_TRY(wrk_tuner_demod(st, Lp, Gp));
			HIP_TRY(wrk_tuner_audio(st, Lp, Gp));
		} else if (defer) {
			g->post_pending = true;
			g->post_args = wrk_post_args(Lp, Gp);
			g->pend_seq = seq;
			g->pend_k2 = Lp.k2;
			g->pend_slots = L.slots_used;
			g->post_args.audio_host = ring_reserve(t, g, Lp.k2, L.slots_used);
			g->post_args.host_stride = Lp.k2;           /* the ring's rows lie back to back (RingSlot::stride = frames) */
			g->pend_direct = g->post_args.audio_host != nullptr;
		} else {
			HIP_TRY(wrk_tuner_post(st, Lp, Gp));
		}
		if (!defer) {
			int rc = ring_push(t, g, seq, Lp.k2, L.slots_used, false);
			if (rc)
				return rc;
		}
		g->last_demod_kept = two_kernels;
		g->last_parity = g->parity;
		g->last_cb = g->cb;
		g->sp ^= 1;                    /* the kernels wrote the other state set */
		if (g->l1 <= WR_FIR_LENGTH)
			hist_written = true;       /* k_tuner_ddc stored the next input history (k_tuner_ddc_long keeps its own) */
		if (Lp.k1)
			g->parity ^= 1;            /* k_tuner_demod filled the other prev_iq / dem history */
		if (L.k1)
			g->cb ^= 1;
		g->last_k1 = Lp.k1;            /* frames at the demodulator's input */
		g->last_k2 = Lp.k2;
	}
	/* the bracket of a profiling stride closes behind the LAST rate group's launches of its last submit
	 * (it used to close behind the first group's: the others' launches were credited and not timed) */
	if (group_last && (t->ev_used & 1)) {
		HIP_TRY(hipEventRecord(t->ev[t->ev_used], st));
		t->ev_span.resize(t->ev_used / 2 + 1, 1u);
		t->ev_span[t->ev_used / 2] = t->prof_stride;
		t->ev_used += 1;
	}
	if (!hist_written)
		HIP_TRY(wrk_input_hist(st, cur, cur_u8, nframes, t->in_hist[t->in_par], t->in_hist[t->in_par ^ 1]));
	t->in_par ^= 1;
	if (t->mark_launches && !marked && !unmarked && nframes)
		unmarked = true;                            /* (no rate group launched anything, but k_input_hist above reads the block: the
		                                               completion mark must not be the block-before's -- a halo exchange ordered
		                                               behind it could overwrite what that kernel is still reading) */
	if (t->mark_launches && (marked || unmarked)) {
		/* (a launch that could not stamp it -- profiling, a long channel filter -- or a history kernel behind the DDC:
		 * an ordinary record, at an ordinary record's price) */
		if (unmarked || !hist_written)
			HIP_TRY(hipEventRecord(t->launch_ev[(t->launches_marked + 1) % 4], st));
		++t->launches_marked;
	}

	/* host mirror of the phase advance k_tuner_ddc wrote into the other state set */
	for (Chan &c : t->chans) {
		if (!c.in_use || c.group < 0)
			continue;
		c.phaseL += (unsigned int)nframes * c.stepL;
	}
	t->submitted = true;
	return WR_OK;
}

/* ------------------------------------------------------------------ the streaming launch -- */
/*
 * wr_tuner_set_streaming(tuner, 1): blocks submitted from DEVICE memory no longer cost a kernel launch each.  The
 * first one opens a persistent launch (k_tuner_stream, wr_stream_kernel.inc), every following block of the same size
 * rings its doorbell -- a descriptor and a counter in page-locked host memory -- and the launch takes it up where it
 * stands: no ramp, no tail, no kernel boundary between blocks, and a block's demodulator + audio filter start the
 * moment its last channel-IQ row is out (dsp/dspblock.cxx:169-212: a block's output leaves within its own run()).
 * Anything else that touches the tuner -- a setter, a getter, a flush, a block of another size or from host memory, a
 * wait for the device's stream through this library -- CLOSES the launch first: the host writes `stop`, the launch
 * finishes the blocks it was given, rolls the state as one launch over all of them would have, and ends.  The host's
 * own state is advanced by as many blocks at the close.  Same bits as one launch per block (tests/test_gpu_stream.py).
 *
 * Requirements, checked at the open (a submit that does not meet them goes the ordinary way): WR_NCO_ROTATE; one
 * rate group; at most 1024 channels, all on ONE 64-tap channel filter; no second channel stage; an audio decimation
 * the fused post stage has (1..6, 8, 10); whole audio frames per block; no kept demodulator rows, no seek pending, no
 * launch marks.  A block's memory must stay untouched until the NEXT block's audio is complete (the first frames of
 * a block read the last 63 of the one before in place).
 *
 * A caller that waits for the device's stream by other means (hipStreamSynchronize on a stream it handed to
 * wr_dev_open, torch.cuda.synchronize()) should call wr_tuner_flush first: an open launch that nobody rings ends by
 * itself only after WR_STREAM_IDLE_MS.
 */
#define WR_STREAM_IDLE_MS   500           /* the launch closes itself when the doorbell has been silent this long */
#define WR_STREAM_STALE_MS  100           /* ... and the host does not ring a launch it has left alone this long: it opens a new one */
#define WR_STREAM_WAIT_MS  2000           /* deadline of every other wait inside the launch (an error) */

static unsigned long long *g_stream_tl = nullptr;
extern "C" int wr_debug_stream_tl(unsigned long long *out, size_t n)
{
	if (!g_stream_tl)
		return 1;
	memcpy(out, g_stream_tl, n * sizeof(unsigned long long));
	return 0;
}

/* ONE streaming launch per GPU and process: every workgroup of such a launch must be resident (they wait for one another),
 * and two of them on one device -- two wr_dev contexts, two front ends sharing a GPU -- could each hold CUs the other is
 * waiting for.  A tuner that finds the device taken submits the ordinary way.  (Other PROCESSES on the same GPU are not
 * seen from here: one process per GPU is the deployment; the launch's own deadlines end a stand-off as an error.) */
static std::mutex g_stream_mu;
static wr_tuner *g_stream_on[WR_MAX_DEVICES];

static void stream_free(wr_tuner *t)
{
	wr_tuner::Stream &s = t->stream;
	(void)hipHostFree(s.ctl);
	(void)hipHostFree(s.desc);
	(void)hipFree(s.sdev);
	(void)hipFree(s.ring);
	for (uint8_t *&r : s.raw) {
		(void)hipFree(r);
		r = nullptr;
	}
	s.raw_cap = 0;
	if (s.raw_ev)
		(void)hipEventDestroy(s.raw_ev);
	s.raw_ev = nullptr;
	s.ctl = nullptr;
	s.desc = nullptr;
	s.sdev = nullptr;
	s.ring = nullptr;
	s.ring_floats = 0;
}

/* the outcome of a closed launch, once the device's stream has been waited for */
static int stream_check(wr_tuner *t)
{
	wr_tuner::Stream &s = t->stream;
	if (!s.unchecked)
		return WR_OK;
	s.unchecked = false;
	s.long_blocks += s.ctl->long_blocks;
	s.ctl->long_blocks = 0;
	if (s.ctl->err)
		return fail(WR_ERR_HIP, "streaming launch: a wait inside the launch ran into its deadline (code %u)", s.ctl->err);
	if (s.ctl->final_blocks != s.count)
		return fail(WR_ERR_STATE, "streaming launch: %u blocks rung, %u processed%s", s.count, s.ctl->final_blocks,
		            s.ctl->self_closed ? " (the launch had closed itself: the doorbell was silent too long)" : "");
	return WR_OK;
}

/* a ring entry for block `idx` (from 0) of the live launch: its audio is complete when WrStreamCtl::done > idx */
static float *stream_ring_entry(wr_tuner *t, Group *g, unsigned long long seq, size_t k2, unsigned int used, unsigned int idx)
{
	if (t->ring.empty())
		return nullptr;
	float *mapped = ring_reserve(t, g, k2, used);
	std::lock_guard<std::mutex> lk(t->ring_lock);
	if (!mapped) {
		if (single_group(t) == g && t->ring_count == t->ring.size())
			++t->ring_overruns;                         /* io/rtlsdrtuner.cxx:100-117: the new block is dropped */
		return nullptr;
	}
	wr_tuner::RingSlot &r = t->ring[t->ring_head];
	r.stride = r.frames = k2;
	r.slots = used;
	r.seq = seq;
	r.stream_wait = idx + 1u;
	r.stream_gen = t->stream.gen;
	t->ring_head = (t->ring_head + 1) % (unsigned int)t->ring.size();
	++t->ring_count;
	return mapped;
}

/* may the live launch take this block? */
static bool stream_follows(const wr_tuner *t, size_t nframes, int where, bool u8)
{
	const wr_tuner::Stream &s = t->stream;
	if (!s.live || !s.enabled || where != WR_DEVICE || nframes != s.nframes || u8 != s.u8 || s.count >= WR_STREAM_MAXJ)
		return false;
	if (s.g->dirty)
		return false;
	const auto idle = std::chrono::steady_clock::now() - s.last_bell;
	return idle < std::chrono::milliseconds(WR_STREAM_STALE_MS);
}

static int stream_bell(wr_tuner *t, const void *iq, bool behind_upload = false)
{
	wr_tuner::Stream &s = t->stream;
	const unsigned int j = s.count;
	const unsigned long long seq = t->submit_seq++;
	float *slot = stream_ring_entry(t, s.g, seq, s.k2, group_slots_used(s.g), j);
	s.desc[j].cur = (unsigned long long)(uintptr_t)iq;
	s.desc[j].audio_host = (unsigned long long)(uintptr_t)slot;
	s.desc[j].count = j + 1u;
	std::atomic_thread_fence(std::memory_order_release);      /* the descriptor before the count */
	if (behind_upload) {
		/* (r06) the block is still crossing PCIe on the upload stream: the count follows it there, into the doorbell's
		 * device-side twin -- in order with the data, nobody waits.  A stream memory operation (the command processor writes the
		 * word): a 4-byte hipMemcpyAsync is a copy KERNEL on this runtime (GPU_FORCE_BLIT_COPY_SIZE), and a kernel does not find
		 * a free wave slot with enough registers beside the launch it is supposed to ring -- measured: it ran 0.5 s later, when
		 * the launch had closed itself */
		HIP_TRY(hipStreamWriteValue32(t->dev->up_stream, &s.sdev->ready_up, j + 1u, 0));
		s.up_pending = true;
	} else {
		if (s.up_pending) {
			/* (doorbells of earlier blocks are still queued behind their copies on the upload stream: this count would overtake
			 * them -- the bell takes the larger of the two -- and ring blocks whose bytes have not landed) */
			HIP_TRY(hipStreamSynchronize(t->dev->up_stream));
			s.up_pending = false;
		}
		s.ctl->ready = j + 1u;
	}
	std::atomic_thread_fence(std::memory_order_seq_cst);
	s.count = j + 1u;
	++s.blocks;
	s.last_bell = std::chrono::steady_clock::now();
	return WR_OK;
}

static int stream_open(wr_tuner *t, const void *iq, size_t nframes, bool u8, bool *took)
{
	*took = false;
	wr_tuner::Stream &s = t->stream;
	wr_dev *d = t->dev;
	if (t->nco_mode != WR_NCO_ROTATE || !t->defer_post || t->mark_launches || (t->keep_mask & (1u << WR_STAGE_DEMOD)))
		return WR_OK;
	Group *g = single_group(t);
	if (!g || g->l1 != WR_FIR_LENGTH || g->l2 != WR_FIR_LENGTH || g->d1b || g->seek_pending || !wrk_tuner_post_supported(g->d2))
		return WR_OK;
	/* (64 channel-rate frames per block at least: a block's post stage takes its history from the block before) */
	if (nframes < (size_t)WR_FIR_LENGTH * g->d1 || nframes % ((size_t)g->d1 * g->d2) || nframes / g->d1 > 0x3FFFFFFFu)
		return WR_OK;
	if (dev_bind(d))
		return WR_ERR_HIP;
	if (int rc = tuner_launch_held(t))
		return rc;
	if (g->dirty)
		if (int rc = group_upload(t, g))
			return rc;
	const unsigned int used = group_slots_used(g), groups = used / 64u;
	if (!g->one_filter || !groups || groups > 16u)
		return WR_OK;
	if (int rc = dev_settle_stream(d))                      /* another tuner's launch in this context */
		return rc;
	{
		std::lock_guard<std::mutex> lk(g_stream_mu);
		if (d->device < 0 || d->device >= WR_MAX_DEVICES || (g_stream_on[d->device] && g_stream_on[d->device] != t))
			return WR_OK;                                   /* another context's launch holds this GPU: the ordinary way */
	}
	if (s.unchecked) {
		/* the launch before this one must be over before its doorbell is reused */
		HIP_TRY(dev_stream_sync(d));
		if (int rc = stream_check(t))
			return rc;
	}
	if (g->post_pending)                                    /* the post stage of a block that went the ordinary way */
		if (int rc = tuner_flush(t))
			return rc;
	const size_t k1 = nframes / g->d1, k2 = k1 / g->d2;
	unsigned int n_ddc = 0, n_post = 0;
	HIP_TRY(wrk_stream_geometry(g->d2, groups, d->num_cus, &n_ddc, &n_post));
	if (!n_ddc || !n_post)
		return WR_OK;
	if (getenv("WR_STREAM_NPOST")) {                        /* (development: another split of the resident workgroups) */
		const unsigned int np = (unsigned int)atoi(getenv("WR_STREAM_NPOST"));
		if (np >= 1u && np < n_ddc + n_post - 8u) {
			n_ddc = n_ddc + n_post - np;
			n_post = np;
		}
	}
	/* (each on its own: an allocation that failed last time is tried again, never skipped because an earlier one stands) */
	if (!s.ctl)
		HIP_TRY(hipHostMalloc((void **)&s.ctl, sizeof(WrStreamCtl), hipHostMallocMapped | hipHostMallocCoherent));
	if (!s.desc)
		HIP_TRY(hipHostMalloc((void **)&s.desc, sizeof(WrStreamDesc) * WR_STREAM_MAXJ, hipHostMallocMapped | hipHostMallocCoherent));
	if (!s.sdev)
		HIP_TRY(hipMalloc((void **)&s.sdev, sizeof(WrStreamDev)));
	const size_t ring_floats = (size_t)WR_STREAM_RING * k1 * g->slots * 2u;
	if (ring_floats > s.ring_floats) {
		HIP_TRY(dev_stream_sync(d));
		(void)hipFree(s.ring);
		s.ring = nullptr;
		s.ring_floats = 0;
		HIP_TRY(hipMalloc((void **)&s.ring, ring_floats * sizeof(float)));
		s.ring_floats = ring_floats;
	}
	void *ctl_dev = nullptr, *desc_dev = nullptr;
	HIP_TRY(hipHostGetDevicePointer(&ctl_dev, s.ctl, 0));
	HIP_TRY(hipHostGetDevicePointer(&desc_dev, s.desc, 0));
	hipStream_t st = d->stream;

	/* the launch's view of the group: what tuner_submit_now hands k_tuner_ddc and the post stage for one block */
	WrTunerLaunch L;
	L.cur = u8 ? nullptr : (const float *)iq;
	L.cur_u8 = u8 ? (const uint8_t *)iq : nullptr;
	L.hist = t->in_hist[t->in_par];
	L.hist_next = t->in_hist[t->in_par ^ 1];
	L.parity = g->parity;
	L.sp = g->sp;
	L.cb = g->cb;
	L.nframes = nframes;
	L.d1 = g->d1;
	L.d2 = g->d2;
	L.slots = g->slots;
	L.slots_used = used;
	L.k1 = k1;
	L.k2 = k2;
	L.k2max = g->k2max;
	L.nco_mode = t->nco_mode;
	L.uniform_mask = g->uniform_mask;
	L.uniform2_mask = g->uniform2_mask;
	L.fewsets_mask = g->fewsets_mask;
	L.one_filter = 1;
	memcpy(L.nsets, g->nsets, sizeof(L.nsets));
	L.audio_scale = t->audio_scale;
	L.use_gain = g->use_gain ? 1 : 0;
	L.use_squelch = g->use_squelch ? 1 : 0;
	L.ev_start = L.ev_stop = nullptr;

	WrStreamArgs A;
	memset(&A, 0, sizeof(A));
	A.ctl = (WrStreamCtl *)ctl_dev;
	A.desc_host = (const WrStreamDesc *)desc_dev;
	A.sdev = s.sdev;
	A.idle_ticks = (unsigned long long)WR_STREAM_IDLE_MS * 100000ull;      /* the constant clock runs at 100 MHz */
	A.wait_ticks = (unsigned long long)WR_STREAM_WAIT_MS * 100000ull;
	A.n_ddc = n_ddc;
	A.n_post = n_post;
	A.dbg = getenv("WR_STREAM_DBG") ? (unsigned int)atoi(getenv("WR_STREAM_DBG")) : 0u;
	if (A.dbg & 16u) {
		static unsigned long long *tlbuf = nullptr;
		if (!tlbuf)
			HIP_TRY(hipHostMalloc((void **)&tlbuf, 8192 * 8 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
		memset(tlbuf, 0, 8192 * 8 * sizeof(unsigned long long));
		void *tld = nullptr;
		HIP_TRY(hipHostGetDevicePointer(&tld, tlbuf, 0));
		A.tl = (unsigned long long *)tld;
		g_stream_tl = tlbuf;
	}
	A.nframes = nframes;
	A.k1 = (unsigned int)k1;
	A.d1 = g->d1;
	A.is_u8 = u8 ? 1u : 0u;
	A.ext = (s.ext || (getenv("WR_STREAM_EXT") && atoi(getenv("WR_STREAM_EXT")))) ? 1u : 0u;
	A.slots = g->slots;
	A.groups = groups;
	for (unsigned int gi = 0; gi < groups; ++gi)
		(gi < 8u ? A.gmap0 : A.gmap1) |= (unsigned long long)gi << ((gi & 7u) * 8u);
	A.kslow = (WR_HIST + g->d1 - 1u) / g->d1;
	if (A.kslow > k1)
		A.kslow = (unsigned int)k1;
	A.hist = L.hist;
	A.hist_next = L.hist_next;
	A.phase = g->dev.phase[g->sp];
	A.step = g->dev.step;
	A.phase_next = g->dev.phase[g->sp ^ 1];
	A.hist_cs = g->dev.hist_cs[g->sp];
	A.hist_lo = g->dev.hist_lo[g->sp];
	A.hist_cs_next = g->dev.hist_cs[g->sp ^ 1];
	A.hist_lo_next = g->dev.hist_lo[g->sp ^ 1];
	A.flags = g->dev.flags;
	A.taps1 = g->dev.taps1;
	A.rot = g->dev.rot;
	A.taps1u = g->dev.taps1u;
	A.tapsel = g->dev.tapsel;
	A.table = d->table_turn;
	A.hi_cs = d->hi_cs;
	A.lo_cs = d->lo_cs;
	A.ring = s.ring;
	for (int i = 0; i < 4; ++i)
		A.audio_bufs[i] = g->dev.audio_set[(g->audio_cur + i) & 3];
	A.post = wrk_post_args(L, g->dev);
	A.post.host_stride = k2;                                /* the ring's rows lie back to back (RingSlot::stride = frames) */
	A.prev_iq[0] = g->dev.prev_iq[0];
	A.prev_iq[1] = g->dev.prev_iq[1];
	A.dem[0] = g->dev.dem[0];
	A.dem[1] = g->dev.dem[1];
	A.parity0 = g->parity;

	/* the doorbell as the launch finds it: block 0 rung */
	++s.gen;
	s.g = g;
	s.count = 0;
	s.nframes = nframes;
	s.u8 = u8;
	s.parity0 = g->parity;
	s.k1 = k1;
	s.k2 = k2;
	s.last_iq = nullptr;
	const unsigned long long seq = t->submit_seq++;
	float *slot = stream_ring_entry(t, g, seq, k2, used, 0);
	s.ctl->ready = 1;
	s.ctl->stop = 0;
	s.ctl->done = 0;
	s.ctl->final_blocks = 0;
	s.ctl->long_blocks = 0;
	s.ctl->err = 0;
	s.ctl->self_closed = 0;
	s.desc[0].cur = (unsigned long long)(uintptr_t)iq;
	s.desc[0].audio_host = (unsigned long long)(uintptr_t)slot;
	std::atomic_thread_fence(std::memory_order_seq_cst);
	/* the counters, the progress words and the doorbell's device-side copy start from zero: one fill.  (Block 0's
	 * descriptor travels in the kernel's arguments; WrStreamDev::ready stays 0 until the bell republishes a count of 2
	 * or more -- it is NOT copied from the doorbell here: that copy would run when the stream gets to it, by which time
	 * the host may have rung again, and the bell, finding the higher count already in device memory, would never bring
	 * the descriptors over that go with it) */
	HIP_TRY(hipMemsetAsync(s.sdev, 0, offsetof(WrStreamDev, desc), st));
	A.cur0 = s.desc[0].cur;
	A.audio0 = s.desc[0].audio_host;

	void *ev0 = nullptr, *ev1 = nullptr;
	if (t->profiling) {
		/* one event pair for the whole launch, stamped by the dispatch itself */
		if (int rc = (t->ev_used & 1) ? WR_OK : prof_drain(t, 64))
			return rc;
		if (!(t->ev_used & 1)) {
			while (t->ev.size() < t->ev_used + 2) {
				hipEvent_t e;
				HIP_TRY(hipEventCreate(&e));
				t->ev.push_back(e);
			}
			ev0 = t->ev[t->ev_used];
			ev1 = t->ev[t->ev_used + 1];
		}
	}
	HIP_TRY(wrk_tuner_stream(st, A, ev0, ev1));
	if (ev0) {
		t->ev_span.resize(t->ev_used / 2 + 1, 1u);
		t->ev_span[t->ev_used / 2] = 1u;
		t->ev_used += 2;
	}
	s.live = true;
	s.count = 1;
	++s.blocks;
	s.last_bell = std::chrono::steady_clock::now();
	d->streaming = t;
	{
		std::lock_guard<std::mutex> lk(g_stream_mu);
		g_stream_on[d->device] = t;
	}
	*took = true;
	return WR_OK;
}

/* r06: a block in the RTL-SDR byte format out of PAGE-LOCKED host memory, streamed (see wr_tuner::Stream::raw).  `*took` false:
 * not this way (pageable memory, a shape the launch cannot take, no memory) -- the caller goes the ordinary way. */
static int stream_host_u8(wr_tuner *t, const uint8_t *bytes, size_t nframes, bool *took)
{
	*took = false;
	wr_tuner::Stream &s = t->stream;
	wr_dev *d = t->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	void *mapped = nullptr;
	if (hipHostGetDevicePointer(&mapped, const_cast<uint8_t *>(bytes), 0) != hipSuccess || !mapped) {
		(void)hipGetLastError();
		return WR_OK;                                       /* pageable: the copy would be staged by the runtime, synchronously */
	}
	const size_t need = nframes * 2;
	const bool follows = s.live && s.ext && stream_follows(t, nframes, WR_DEVICE, true);
	if (need > s.raw_cap) {
		if (s.live)
			if (int rc = stream_close(t))
				return rc;
		HIP_TRY(dev_stream_sync(d));
		if (d->up_stream)
			HIP_TRY(hipStreamSynchronize(d->up_stream));
		for (uint8_t *&r : s.raw) {
			(void)hipFree(r);
			r = nullptr;
		}
		s.raw_cap = 0;
		for (uint8_t *&r : s.raw)
			HIP_TRY(hipMalloc((void **)&r, need));
		s.raw_cap = need;
	}
	const unsigned int slot = (unsigned int)(s.raw_next & 3u);
	if (s.live && s.raw_gen[slot] == s.gen && s.ctl->done <= s.raw_idx[slot] + 1u) {
		/* the buffer's last tenant -- four blocks back in this very launch -- and the block behind it, whose first windows reach
		 * into its tail, are not through yet (a caller far ahead of the GPU): wait for their audio, which comes behind the
		 * last read of their frames */
		const auto t0 = std::chrono::steady_clock::now();
		while (s.ctl->done <= s.raw_idx[slot] + 1u && !s.ctl->err)
			if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4 * WR_STREAM_WAIT_MS))
				return fail(WR_ERR_HIP, "streaming launch: block %u did not finish", s.raw_idx[slot] + 1u);
	}
	{
		std::lock_guard<std::mutex> up_guard(*d->upload_lock);
		if (int rc = dev_up_stream(d))
			return rc;
		HIP_TRY(hipMemcpyAsync(s.raw[slot], bytes, need, hipMemcpyHostToDevice, d->up_stream));
		if (int rc = upload_mark_locked(d, d->up_stream))   /* (wr_dev_wait_uploads: the host block is free again when the copy has run) */
			return rc;
	}
	if (follows) {
		s.raw_gen[slot] = s.gen;
		s.raw_idx[slot] = s.count;
		++s.raw_next;
		++s.host_blocks;
		t->last_staging = 3;
		*took = true;
		return stream_bell(t, s.raw[slot], true);
	}
	if (s.live)
		if (int rc = stream_close(t))
			return rc;
	/* this block opens a launch (or, if it cannot, goes the ordinary way out of the device copy): either waits for the copy */
	if (!s.raw_ev)
		HIP_TRY(hipEventCreateWithFlags(&s.raw_ev, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(s.raw_ev, d->up_stream));
	HIP_TRY(hipStreamWaitEvent(d->stream, s.raw_ev, 0));
	s.ext = true;
	bool opened = false;
	if (int rc = stream_open(t, s.raw[slot], nframes, true, &opened))
		return rc;
	s.raw_gen[slot] = opened ? s.gen : 0;
	s.raw_idx[slot] = 0;
	++s.raw_next;
	*took = true;
	if (opened) {
		++s.host_blocks;
		t->last_staging = 3;
		return WR_OK;
	}
	s.ext = false;
	t->last_staging = 1;
	return tuner_submit_now(t, s.raw[slot], nframes, WR_DEVICE, true);
}

/* close the live launch: tell it that no block follows, and advance the host's picture of the tuner by the blocks
 * it was given.  Does not wait: the launch finishes them and ends, stream-ordered work queues behind it. */
static int stream_close(wr_tuner *t)
{
	wr_tuner::Stream &s = t->stream;
	if (!s.live)
		return WR_OK;
	if (s.up_pending) {
		/* doorbells still queued behind their blocks on the upload stream: the count must be final before the stop */
		HIP_TRY(hipStreamSynchronize(t->dev->up_stream));
		s.ctl->ready = s.count;
		s.up_pending = false;
	}
	std::atomic_thread_fence(std::memory_order_seq_cst);
	s.ctl->stop = 1;
	std::atomic_thread_fence(std::memory_order_seq_cst);
	s.live = false;
	s.unchecked = true;
	if (t->dev->streaming == t)
		t->dev->streaming = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_stream_mu);
		if (g_stream_on[t->dev->device] == t)
			g_stream_on[t->dev->device] = nullptr;
	}
	Group *g = s.g;
	const unsigned int J = s.count;
	for (Group *x : t->groups)
		if (x != g)
			x->last_k1 = x->last_k2 = 0;
	g->sp ^= 1;                        /* the launch rolls the DDC's state into the other set ONCE, behind its last block */
	g->last_parity = s.parity0;
	g->parity = s.parity0 ^ 1;         /* ... and so it does with the post stage's */
	g->last_k1 = s.k1;
	g->last_k2 = s.k2;
	g->last_demod_kept = false;
	s.last_iq = s.ring + (size_t)((J - 1u) % WR_STREAM_RING) * s.k1 * g->slots * 2u;
	/* the blocks of a stream store their audio into the group's four device arrays by turns (blocks' post stages
	 * may run side by side); the one the LAST block wrote is the group's audio from here on */
	g->audio_cur = (int)((g->audio_cur + (J - 1u)) & 3u);
	g->dev.audio = g->dev.audio_set[g->audio_cur];
	t->in_par ^= 1;
	for (Chan &c : t->chans) {
		if (!c.in_use || c.group < 0)
			continue;
		c.phaseL += (unsigned int)((unsigned long long)s.nframes * J) * c.stepL;
	}
	t->submitted = true;
	return WR_OK;
}

extern "C" int wr_tuner_set_streaming(wr_tuner *t, int enable)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	if (int rc = tuner_launch_held(t))
		return rc;
	t->stream.enabled = enable != 0;
	t->stream.host_bytes = enable >= 2;
	return WR_OK;
}

extern "C" int wr_tuner_stream_info(wr_tuner *t, int *live, unsigned long long *launches, unsigned long long *blocks)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (live)
		*live = t->stream.live ? 1 : 0;
	if (launches)
		*launches = t->stream.gen;
	if (blocks)
		*blocks = t->stream.blocks;
	return WR_OK;
}

extern "C" int wr_tuner_stream_long_blocks(wr_tuner *t, unsigned long long *blocks)
{
	if (!t || !blocks)
		return fail(WR_ERR_ARG, "wr_tuner_stream_long_blocks: bad argument");
	*blocks = t->stream.long_blocks;
	return WR_OK;
}

extern "C" int wr_tuner_stream_host_blocks(wr_tuner *t, unsigned long long *blocks)
{
	if (!t || !blocks)
		return fail(WR_ERR_ARG, "wr_tuner_stream_host_blocks: bad argument");
	*blocks = t->stream.host_blocks;
	return WR_OK;
}

extern "C" int wr_tuner_last_staging(wr_tuner *t, int *how)
{
	if (!t || !how)
		return fail(WR_ERR_ARG, "wr_tuner_last_staging: bad argument");
	*how = t->last_staging;
	return WR_OK;
}

extern "C" int wr_tuner_set_blocks_per_launch(wr_tuner *t, unsigned int nblocks)
{
	if (!t || !nblocks)
		return fail(WR_ERR_ARG, "wr_tuner_set_blocks_per_launch: bad argument");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	int rc = tuner_launch_held(t);
	if (rc)
		return rc;
	t->coalesce = nblocks;
	return WR_OK;
}

extern "C" int wr_tuner_flush(wr_tuner *t)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	return tuner_flush(t);
}

extern "C" int wr_chan_fetch(wr_tuner *t, int chan, int stage, float *out_host, size_t out_capacity,
                             size_t *count)
{
	Chan *c = chan_get(t, chan);
	if (!c || !count)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_fetch: bad argument");
	if (c->group < 0 || !t->submitted)
		return fail(WR_ERR_STATE, "wr_chan_fetch: nothing submitted yet");
	Group *g = t->groups[c->group];
	wr_dev *d = t->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	{
		int rc = tuner_flush(t);        /* the block's demod/audio may still be waiting for the next launch */
		if (rc)
			return rc;
	}
	size_t n = 0;
	switch (stage) {
	case WR_STAGE_CHAN_IQ: n = g->last_k1 * 2; break;
	case WR_STAGE_DEMOD:   n = g->last_k1; break;
	case WR_STAGE_AUDIO:   n = g->last_k2; break;
	default: return fail(WR_ERR_ARG, "wr_chan_fetch: bad stage %d", stage);
	}
	*count = n;
	if (!n)
		return WR_OK;
	if (!out_host || out_capacity < n)
		return fail(WR_ERR_ARG, "wr_chan_fetch: need room for %zu floats", n);
	const size_t S = g->slots;
	if (stage == WR_STAGE_AUDIO) {
		HIP_TRY(hipMemcpyAsync(out_host, g->dev.audio + (size_t)c->slot * g->k2max, n * sizeof(float),
		                       hipMemcpyDeviceToHost, d->stream));
	} else {
		SCRATCH_GUARD(d);
		int rc = dev_scratch(d, n);
		if (rc)
			return rc;
		if (stage == WR_STAGE_CHAN_IQ)
			HIP_TRY(wrk_gather_rows(d->stream, t->stream.last_iq ? t->stream.last_iq
			                                   : g->d1b ? g->dev.chan_iq2[g->last_cb] : g->dev.chan_iq[g->last_cb],
			                        g->last_k1, S * 2, (size_t)c->slot * 2, 2, d->scratch));
		else if (!g->last_demod_kept)
			return fail(WR_ERR_STATE, "wr_chan_fetch: the demodulator output was not kept "
			            "(call wr_tuner_keep_stages(tuner, 1u << WR_STAGE_DEMOD) before submitting)");
		else
			HIP_TRY(wrk_gather_rows(d->stream, g->dev.dem[g->last_parity] + (size_t)(g->l2 - 1) * S, g->last_k1, S,
			                        (size_t)c->slot, 1, d->scratch));
		HIP_TRY(hipMemcpyAsync(out_host, d->scratch, n * sizeof(float), hipMemcpyDeviceToHost, d->stream));
	}
	TUNER_SYNC_CHECKED(t);
	return WR_OK;
}

extern "C" int wr_tuner_audio_dev(wr_tuner *t, const float **audio_dev, size_t *chan_stride,
                                  size_t *frames)
{
	if (!t || !audio_dev || !chan_stride || !frames)
		return fail(WR_ERR_ARG, "wr_tuner_audio_dev: bad argument");
	if (int rc = settle_held(t))
		return rc;
	Group *g = nullptr;
	for (Group *x : t->groups)
		if (x->active > 0) {
			if (g)
				return fail(WR_ERR_STATE, "tuner has several rate groups; fetch per channel instead");
			g = x;
		}
	if (!g)
		return fail(WR_ERR_STATE, "tuner has no configured channel");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	{
		int rc = tuner_flush(t);        /* the caller is about to read the last block's audio */
		if (rc)
			return rc;
	}
	*audio_dev = g->dev.audio;
	*chan_stride = g->k2max;
	*frames = g->last_k2;
	return WR_OK;
}

extern "C" int wr_tuner_fetch_audio_all(wr_tuner *t, float *out_host, size_t out_capacity,
                                        size_t *chan_stride, size_t *frames, unsigned int *slots_used)
{
	if (!t || !chan_stride || !frames || !slots_used)
		return fail(WR_ERR_ARG, "wr_tuner_fetch_audio_all: bad argument");
	if (int rc = settle_held(t))
		return rc;
	Group *g = nullptr;
	for (Group *x : t->groups)
		if (x->active > 0) {
			if (g)
				return fail(WR_ERR_STATE, "tuner has several rate groups; fetch per channel instead");
			g = x;
		}
	if (!g || !t->submitted)
		return fail(WR_ERR_STATE, "nothing submitted yet");
	unsigned int used = group_slots_used(g);
	*chan_stride = g->last_k2;
	*frames = g->last_k2;
	*slots_used = used;
	const size_t need = (size_t)used * g->last_k2;
	if (!need)
		return WR_OK;
	if (!out_host || out_capacity < need)
		return fail(WR_ERR_ARG, "wr_tuner_fetch_audio_all: need room for %zu floats", need);
	wr_dev *d = t->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	{
		int rc = tuner_flush(t);
		if (rc)
			return rc;
	}
	HIP_TRY(hipMemcpy2DAsync(out_host, g->last_k2 * sizeof(float), g->dev.audio, g->k2max * sizeof(float),
	                         g->last_k2 * sizeof(float), used, hipMemcpyDeviceToHost, d->stream));
	TUNER_SYNC_CHECKED(t);
	return WR_OK;
}

/* ---- pinned audio ring ---- */

static Group *single_group(wr_tuner *t)
{
	Group *g = nullptr;
	for (Group *x : t->groups)
		if (x->active > 0) {
			if (g)
				return nullptr;
			g = x;
		}
	return g;
}

/* queue the copy of one block's audio, right behind the kernel that produces it */
static bool ring_direct_enabled()
{
	static const bool on = !(getenv("WR_RING_DIRECT") && atoi(getenv("WR_RING_DIRECT")) == 0);
	return on;
}

/* r04: the post stage whose arguments are being put together (a deferred one: it is launched later, riding in the next
 * block's launch or by a flush) may write its audio straight into the ring slot its block will be queued in -- the next
 * one to fill, which stays the next one until that block's own ring_push: pushes come in block order and this group is
 * the only one that pushes.  The audio is then in the ring when the launch has run; no device-to-host copy is enqueued
 * behind it (50 us for the 2 MB of a C2 block, in series with everything else of an on-time block).  Returns the slot's
 * device-side address, or NULL: no ring, several rate groups, the ring full right now (the block may still be queued by
 * copy if a slot is free by then), or page-locked memory that cannot be had. */
static float *ring_reserve(wr_tuner *t, Group *g, size_t k2, unsigned int used)
{
	if (t->ring.empty() || !ring_direct_enabled())
		return nullptr;
	std::lock_guard<std::mutex> lk(t->ring_lock);
	if (single_group(t) != g || t->ring_count == t->ring.size())
		return nullptr;
	wr_tuner::RingSlot &r = t->ring[t->ring_head];
	const size_t need = (size_t)used * k2;
	if (!need)
		return nullptr;
	if (need > r.cap) {
		(void)hipHostFree(r.host);
		r.host = nullptr;
		r.cap = 0;
		const size_t want = (size_t)g->slots * g->k2max;
		if (hipHostMalloc((void **)&r.host, (want ? want : 1) * sizeof(float), hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			r.host = nullptr;
			return nullptr;
		}
		r.cap = want;
	}
	void *mapped = nullptr;
	if (hipHostGetDevicePointer(&mapped, r.host, 0) != hipSuccess || !mapped) {
		(void)hipGetLastError();
		return nullptr;
	}
	return (float *)mapped;
}

static int ring_push(wr_tuner *t, Group *g, unsigned long long seq, size_t k2, unsigned int used, bool direct)
{
	if (t->ring.empty())
		return WR_OK;
	std::lock_guard<std::mutex> lk(t->ring_lock);
	if (single_group(t) != g)
		return WR_OK;                           /* no or several rate groups: not queued (see the header) */
	if (t->ring_count == t->ring.size()) {
		++t->ring_overruns;                     /* io/rtlsdrtuner.cxx:100-117: the new block is dropped */
		return WR_OK;
	}
	wr_tuner::RingSlot &r = t->ring[t->ring_head];
	const size_t need = (size_t)used * k2;
	if (need > r.cap) {
		(void)hipHostFree(r.host);
		r.host = nullptr;
		r.cap = 0;
		const size_t want = (size_t)g->slots * g->k2max;
		HIP_TRY(hipHostMalloc((void **)&r.host, (want ? want : 1) * sizeof(float), hipHostMallocDefault));
		r.cap = want;
		direct = false;                         /* (cannot happen: ring_reserve sized the slot) */
	}
	hipStream_t st = t->dev->stream;
	if (need && !direct) {
		if (k2 == g->k2max)                     /* rows back to back (the usual block size): one linear copy */
			HIP_TRY(hipMemcpyAsync(r.host, g->dev.audio, need * sizeof(float), hipMemcpyDeviceToHost, st));
		else
			HIP_TRY(hipMemcpy2DAsync(r.host, k2 * sizeof(float), g->dev.audio, g->k2max * sizeof(float),
			                         k2 * sizeof(float), used, hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(hipEventRecord(r.done, st));
	r.stream_wait = 0;
	r.stride = r.frames = k2;
	r.slots = used;
	r.seq = seq;
	t->ring_head = (t->ring_head + 1) % (unsigned int)t->ring.size();
	++t->ring_count;
	return WR_OK;
}

extern "C" int wr_tuner_audio_ring(wr_tuner *t, unsigned int depth)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (depth > 1024)
		return fail(WR_ERR_ARG, "wr_tuner_audio_ring: depth %u", depth);
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	if (int rc = settle_held(t))
		return rc;
	{
		int rc = tuner_quiesce(t);               /* no copy may be in flight into a slot we free */
		if (rc)
			return rc;
	}
	std::lock_guard<std::mutex> lk(t->ring_lock);
	if (t->ring_held)
		return fail(WR_ERR_STATE, "wr_tuner_audio_ring: a slot is still acquired");
	for (wr_tuner::RingSlot &r : t->ring) {
		if (r.done)
			(void)hipEventDestroy(r.done);
		(void)hipHostFree(r.host);
	}
	t->ring.clear();
	t->ring.resize(depth);
	for (wr_tuner::RingSlot &r : t->ring)
		HIP_TRY(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
	t->ring_head = t->ring_count = 0;
	t->ring_overruns = 0;
	return WR_OK;
}

extern "C" int wr_tuner_audio_ring_acquire(wr_tuner *t, const float **audio_host, size_t *chan_stride,
                                           size_t *frames, unsigned int *slots_used, unsigned long long *seq)
{
	if (!t || !audio_host || !chan_stride || !frames || !slots_used)
		return fail(WR_ERR_ARG, "wr_tuner_audio_ring_acquire: bad argument");
	hipEvent_t ev;
	wr_tuner::RingSlot *r;
	unsigned int swait = 0;
	unsigned long long sgen = 0;
	{
		std::lock_guard<std::mutex> lk(t->ring_lock);
		if (t->ring.empty())
			return fail(WR_ERR_STATE, "wr_tuner_audio_ring_acquire: no ring (wr_tuner_audio_ring)");
		if (t->ring_held)
			return fail(WR_ERR_STATE, "wr_tuner_audio_ring_acquire: release the previous slot first");
		if (!t->ring_count)
			return fail(WR_ERR_STATE, "wr_tuner_audio_ring_acquire: nothing queued");
		const unsigned int n = (unsigned int)t->ring.size();
		r = &t->ring[(t->ring_head + n - t->ring_count) % n];
		ev = r->done;
		swait = r->stream_wait;
		sgen = r->stream_gen;
		t->ring_held = true;
	}
	/* wait outside the lock: the producer may queue further blocks meanwhile */
	hipError_t e = hipSuccess;
	if (swait) {
		/* a block of a streaming launch: the launch's post stage wrote the slot itself and counts the blocks it has
		 * finished in page-locked memory (a launch older than the tuner's current one has ended: stream_open waits) */
		const auto t0 = std::chrono::steady_clock::now();
		unsigned int spins = 0;
		while (sgen == t->stream.gen && t->stream.ctl->done < swait && !t->stream.ctl->err) {
			if ((++spins & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4 * WR_STREAM_WAIT_MS)) {
				std::lock_guard<std::mutex> lk(t->ring_lock);
				t->ring_held = false;
				return fail(WR_ERR_HIP, "wr_tuner_audio_ring_acquire: the streaming launch did not deliver block %u", swait - 1u);
			}
#if defined(__x86_64__)
			__builtin_ia32_pause();
#endif
		}
		std::atomic_thread_fence(std::memory_order_acquire);
		if (sgen == t->stream.gen && t->stream.ctl->err) {
			std::lock_guard<std::mutex> lk(t->ring_lock);
			t->ring_held = false;
			return fail(WR_ERR_HIP, "wr_tuner_audio_ring_acquire: the streaming launch reported error %u", t->stream.ctl->err);
		}
	} else {
		e = hipEventSynchronize(ev);        /* (looking at the event ourselves first -- hipEventQuery in a loop -- changes nothing:
		                                       what an on-time caller waits for here is the GPU, not the wake-up; r05) */
	}
	if (e != hipSuccess) {
		std::lock_guard<std::mutex> lk(t->ring_lock);
		t->ring_held = false;
		return fail(WR_ERR_HIP, "wr_tuner_audio_ring_acquire: %s", hipGetErrorString(e));
	}
	*audio_host = r->host;
	*chan_stride = r->stride;
	*frames = r->frames;
	*slots_used = r->slots;
	if (seq)
		*seq = r->seq;
	return WR_OK;
}

/* has the copy of the oldest queued block landed (would wr_tuner_audio_ring_acquire return at once)? */
extern "C" int wr_tuner_audio_ring_ready(wr_tuner *t, int *ready)
{
	if (!t || !ready)
		return fail(WR_ERR_ARG, "wr_tuner_audio_ring_ready: bad argument");
	*ready = 0;
	hipEvent_t ev;
	{
		std::lock_guard<std::mutex> lk(t->ring_lock);
		if (t->ring.empty() || !t->ring_count || t->ring_held)
			return WR_OK;
		const unsigned int n = (unsigned int)t->ring.size();
		const wr_tuner::RingSlot &r = t->ring[(t->ring_head + n - t->ring_count) % n];
		ev = r.done;
		if (r.stream_wait) {
			*ready = (r.stream_gen != t->stream.gen || t->stream.ctl->done >= r.stream_wait) ? 1 : 0;
			return WR_OK;
		}
	}
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	const hipError_t e = hipEventQuery(ev);
	if (e == hipSuccess)
		*ready = 1;
	else if (e != hipErrorNotReady)
		return fail(WR_ERR_HIP, "wr_tuner_audio_ring_ready: %s", hipGetErrorString(e));
	return WR_OK;
}

extern "C" int wr_tuner_audio_ring_release(wr_tuner *t)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	std::lock_guard<std::mutex> lk(t->ring_lock);
	if (!t->ring_held)
		return fail(WR_ERR_STATE, "wr_tuner_audio_ring_release: nothing acquired");
	t->ring_held = false;
	--t->ring_count;
	return WR_OK;
}

extern "C" int wr_tuner_audio_ring_stats(wr_tuner *t, unsigned int *queued, unsigned long long *overruns)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	std::lock_guard<std::mutex> lk(t->ring_lock);
	if (queued)
		*queued = t->ring_count;
	if (overruns)
		*overruns = t->ring_overruns;
	return WR_OK;
}

extern "C" int wr_tuner_submit_count(wr_tuner *t, unsigned long long *submits)
{
	if (!t || !submits)
		return fail(WR_ERR_ARG, "tuner or submits is NULL");
	*submits = t->submit_seq;                /* (written by submits only: call from the thread that submits) */
	return WR_OK;
}

extern "C" int wr_tuner_set_audio_scale(wr_tuner *t, float scale)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (int rc = settle_held(t))
		return rc;
	t->audio_scale = scale;
	return WR_OK;
}

extern "C" int wr_chan_reset_history(wr_tuner *t, int chan)
{
	Chan *c = chan_get(t, chan);
	if (!c)
		return g_settle_rc ? g_settle_rc : fail(WR_ERR_ARG, "wr_chan_reset_history: no channel %d", chan);
	c->cs_hist_reset = true;
	c->dem_hist_reset = true;
	if (c->group >= 0)
		t->groups[c->group]->dirty = true;
	return WR_OK;
}

/* Time sharding of one stream (BASELINE config 5): every channel of the tuner as if the
 * stream began at `frame` -- both filter histories empty, Demodulator::prev_i/q zero -- except the
 * NCO, whose phase takes the closed-form value it has after `frame` input frames from phase 0
 * (downconverter.cxx:103: phase = frame * phaseStep mod 2^31).  One call for the whole tuner, a
 * handful of stream-ordered fills: no per-channel round trips. */
extern "C" int wr_tuner_seek(wr_tuner *t, unsigned long long frame)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	wr_dev *d = t->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	hipStream_t st = d->stream;
	int rc = tuner_launch_held(t);
	if (rc)
		return rc;
	for (Group *g : t->groups) {
		if (g->dirty) {
			rc = group_upload(t, g);
			if (rc)
				return rc;
		}
		/* Lazily where the group's launch can take it (one channel-filter stage of up to 64 taps): nothing is launched
		 * here, the next submit's DDC computes the phase in closed form and reads all-zero state sets, and the post
		 * stage of the chunk before, if it is still waiting, rides in that launch as usual -- a time-sharded stream
		 * (BASELINE config 5) then costs ONE launch per chunk instead of three (seek, DDC, post stage).
		 * WR_LAZY_SEEK=0: as before. */
		const bool lazy = !g->d1b && g->l1 <= WR_FIR_LENGTH && lazy_seek_enabled();
		if (!lazy && g->post_pending) {
			/* a post stage still waiting for the next submit would write ITS end-of-block state over ours */
			HIP_TRY(wrk_tuner_post_args(st, g->post_args));
			g->post_pending = false;
			if ((rc = ring_push(t, g, g->pend_seq, g->pend_k2, g->pend_slots, g->pend_direct)) != WR_OK)
				return rc;
		}
		const size_t S = g->slots;
		for (size_t s = 0; s < S; ++s) {
			const int ci = g->owner[s];
			if (ci < 0)
				continue;
			Chan &c = t->chans[ci];
			c.phaseL = (unsigned int)((unsigned long long)c.stepL * frame);     /* host mirror; mod 2^32, left-aligned */
			c.phase_dirty = c.prev_dirty = c.cs_hist_reset = c.dem_hist_reset = false;
			c.prev_iq[0] = c.prev_iq[1] = 0.0f;
		}
		if (lazy) {
			g->seek_pending = true;
			g->seek_frame = frame;
			continue;
		}
		g->seek_pending = false;
		/* on the device from the step array itself: no host data in flight, nothing to wait for */
		HIP_TRY(wrk_seek(st, g->dev, (unsigned int)S, g->sp, g->parity, g->p2, frame));
		if (g->l1 > WR_FIR_LENGTH)
			HIP_TRY(hipMemsetAsync(g->dev.mixhist[g->sp], 0, (size_t)(g->l1 - 1) * S * 2 * sizeof(float), st));
	}
	for (Chan &c : t->chans)
		if (c.in_use && c.group < 0) {
			c.phaseL = (unsigned int)((unsigned long long)c.stepL * frame);
			c.prev_iq[0] = c.prev_iq[1] = 0.0f;
		}
	return WR_OK;
}

extern "C" int wr_tuner_mark_launches(wr_tuner *t, int enable)
{
	if (!t)
		return fail(WR_ERR_ARG, "tuner is NULL");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	if (enable)
		for (int i = 0; i < 4; ++i)
			if (!t->launch_ev[i])
				HIP_TRY(hipEventCreateWithFlags(&t->launch_ev[i], hipEventDisableTiming | hipEventReleaseToDevice));
	if (enable && !t->mark_launches) {
		/* blocks launched before marking was on carry no mark: one ordinary record behind them (and behind the held
		 * ones, which go out now) stands for all of them */
		if (int rc = tuner_launch_held(t))
			return rc;
		if (t->submitted) {
			HIP_TRY(hipEventRecord(t->launch_ev[(t->launches_marked + 1) % 4], t->dev->stream));
			++t->launches_marked;
		}
	}
	t->mark_launches = enable != 0;
	return WR_OK;
}

/* wr_ring_exchange_after: the event that fires when every block submitted to `t` so far has been read (*ev = nullptr:
 * the tuner has launched nothing yet, nothing to wait for).  Blocks still held by wr_tuner_set_blocks_per_launch are
 * launched first; a tuner that does not mark its launches is an error, not a silent "no ordering". */
int wrc_tuner_launch_mark(wr_tuner *t, hipEvent_t *ev)
{
	*ev = nullptr;
	if (!t->mark_launches)
		return fail(WR_ERR_STATE, "wr_ring_exchange_after: wr_tuner_mark_launches(tuner, 1) first");
	if (dev_bind(t->dev))
		return WR_ERR_HIP;
	if (int rc = tuner_launch_held(t))
		return rc;
	if (t->launches_marked)
		*ev = t->launch_ev[t->launches_marked % 4];
	return WR_OK;
}

/* --------------------------------------------------------------- spectrum -- */

static void plan_free(WrFftPlan &p)
{
	(void)hipFree(p.tw_n);
	(void)hipFree(p.tw_sub);
	(void)hipFree(p.window);
	(void)hipFree(p.window_p1);
	(void)hipFree(p.work);
	memset(&p, 0, sizeof(p));
}

extern "C" int wr_spectrum_create(wr_spectrum **spec, wr_dev *dev, unsigned int fft_size, unsigned int hop)
{
	if (!spec || !dev)
		return fail(WR_ERR_ARG, "wr_spectrum_create: bad argument");
	*spec = nullptr;
	if (fft_size < 8 || fft_size > (1u << 20) || (fft_size & (fft_size - 1)))
		return fail(WR_ERR_ARG, "size must be a power of 2 in [8, 1048576]");   /* spectrumsink.cxx:53-56 */
	if (hop == 0)
		hop = fft_size;
	if (hop > fft_size)
		return fail(WR_ERR_ARG, "hop must not exceed fft_size");
	if (dev_bind(dev))
		return WR_ERR_HIP;
	DEV_SETTLE(dev);
	wr_spectrum *s = new (std::nothrow) wr_spectrum();
	if (!s)
		return fail(WR_ERR_NOMEM, "out of memory");
	memset(&s->plan, 0, sizeof(s->plan));
	s->dev = dev;
	s->n = fft_size;
	s->hop = hop;
	s->stage = nullptr;
	s->stage_cap = 0;
	s->pending = 0;
	s->bins = nullptr;
	s->frames_done = 0;

	WrFftPlan &p = s->plan;
	p.n = fft_size;
	if (fft_size <= 8192) {
		p.n1 = fft_size;
		p.n2 = 1;
	} else {
		unsigned int bits = 0;
		while ((1u << bits) < fft_size)
			bits++;
		p.n1 = 1u << ((bits + 1) / 2);
		p.n2 = fft_size / p.n1;
	}
	const unsigned int sub = (p.n2 == 1) ? 0 : (p.n1 > p.n2 ? p.n1 : p.n2);
	std::vector<float> tw(fft_size), win(fft_size), tws(sub ? sub : 2);
	wrd_twiddles(fft_size, tw.data());
	wrd_spectrum_window(fft_size, win.data());
	if (sub)
		wrd_twiddles(sub, tws.data());
	p.work_frames = (p.n2 == 1) ? 0 : 1;        /* grown on demand by wr_spectrum_batch_db */
	hipError_t e = hipSuccess;
	do {
		if ((e = hipMalloc((void **)&p.tw_n, fft_size * sizeof(float))) != hipSuccess) break;
		if ((e = hipMalloc((void **)&p.window, fft_size * sizeof(float))) != hipSuccess) break;
		if ((e = hipMalloc((void **)&p.tw_sub, (sub ? sub : 2) * sizeof(float))) != hipSuccess) break;
		if (p.work_frames &&
		    (e = hipMalloc((void **)&p.work, p.work_frames * fft_size * 2 * sizeof(float))) != hipSuccess) break;
		if ((e = hipMalloc((void **)&s->bins, (size_t)fft_size * 2 * sizeof(float))) != hipSuccess) break;
		if ((e = hipMemcpy(p.tw_n, tw.data(), fft_size * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		if ((e = hipMemcpy(p.window, win.data(), fft_size * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		if (p.n1 == 256 && p.n2 == 256) {
			/* pass 1's thread (t, c) of column tile T multiplies rows a*16 + t, a = 0..15, of column T*16 + c: stored as
			 * [T][a / 4][thread][a % 4], a thread takes its 16 values with four 16-byte loads that are contiguous across
			 * the threads of a wave (they were 16 four-byte loads in 64-byte runs: as many memory instructions as the samples) */
			std::vector<float> wp(fft_size);
			for (unsigned int T = 0; T < 16; ++T)
				for (unsigned int tid = 0; tid < 256; ++tid)
					for (unsigned int a = 0; a < 16; ++a)
						wp[((T * 4 + a / 4) * 256 + tid) * 4 + a % 4] = win[(a * 16 + (tid >> 4)) * 256 + T * 16 + (tid & 15)];
			if ((e = hipMalloc((void **)&p.window_p1, fft_size * sizeof(float))) != hipSuccess) break;
			if ((e = hipMemcpy(p.window_p1, wp.data(), fft_size * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		}
		if (sub && (e = hipMemcpy(p.tw_sub, tws.data(), sub * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
	} while (0);
	if (e != hipSuccess) {
		int rc = fail(WR_ERR_HIP, "wr_spectrum_create: %s", hipGetErrorString(e));
		wr_spectrum_destroy(s);
		return rc;
	}
	*spec = s;
	return WR_OK;
}

extern "C" int wr_spectrum_destroy(wr_spectrum *s)
{
	if (!s)
		return WR_OK;
	(void)hipSetDevice(s->dev->device);
	(void)dev_stream_sync(s->dev);
	if (s->dev->lazy_stream)
		(void)hipStreamSynchronize(s->dev->lazy_stream);    /* (a deferred frame's copy may be on its way) */
	if (s->def_ev)
		(void)hipEventDestroy(s->def_ev);
	(void)hipHostFree(s->keep_host);
	plan_free(s->plan);
	(void)hipFree(s->stage);
	(void)hipFree(s->bins);
	delete s;
	return WR_OK;
}

/* the deferred frame, if there is one: transformed now, and what follows its hop moved to the stage's front (where the
 * frames carried over to the next push live).  Closes an open streaming launch: once per poll, not once per block. */
static int spectrum_resolve(wr_spectrum *s)
{
	if (!s->deferred)
		return WR_OK;
	wr_dev *d = s->dev;
	DEV_SETTLE(d);
	hipStream_t st = d->stream;
	if (s->def_keep > s->stage_cap) {
		float *nb = nullptr;
		const size_t cap = s->def_keep + s->n;
		HIP_TRY(hipMalloc((void **)&nb, cap * 2 * sizeof(float)));
		HIP_TRY(hipStreamSynchronize(st));                  /* (whatever still reads the old stage) */
		if (s->stage)
			HIP_TRY(hipFree(s->stage));
		s->stage = nb;
		s->stage_cap = cap;
	}
	HIP_TRY(hipStreamWaitEvent(st, s->def_ev, 0));
	HIP_TRY(hipMemcpyAsync(s->stage, s->keep_host + 2 * s->def_off, s->def_keep * 2 * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(wrk_fft_frames(st, s->plan, s->stage, s->hop, 1, s->bins, nullptr));
	const size_t rest = s->def_rest;
	if (rest) {
		if (rest <= s->hop) {
			HIP_TRY(hipMemcpyAsync(s->stage, s->stage + 2 * (size_t)s->hop, rest * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
		} else {
			SCRATCH_GUARD(d);
			if (int rc = dev_scratch(d, rest * 2))
				return rc;
			HIP_TRY(hipMemcpyAsync(d->scratch, s->stage + 2 * (size_t)s->hop, rest * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
			HIP_TRY(hipMemcpyAsync(s->stage, d->scratch, rest * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
		}
	}
	s->pending = rest;
	s->deferred = false;
	++s->resolves;
	return WR_OK;
}

extern "C" int wr_spectrum_push(wr_spectrum *s, const float *iq, size_t nframes, int where)
{
	if (!s || (nframes && !iq))
		return fail(WR_ERR_ARG, "wr_spectrum_push: bad argument");
	if (where != WR_HOST && where != WR_DEVICE)
		return fail(WR_ERR_ARG, "wr_spectrum_push: bad `where`");
	wr_dev *d = s->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	const bool inside = where == WR_DEVICE && s->pending + nframes >= s->n &&
	                    ((s->pending + nframes - s->n) / s->hop) * s->hop >= s->pending;   /* the newest frame starts inside this block */
	if (inside && d->streaming) {
		/* (r06) a streaming launch is open: keep the frame, transform it when somebody asks (see wr_spectrum::deferred).  A
		 * deferred frame of an earlier push that nobody asked for is simply superseded: nobody can observe it any more. */
		const size_t have = s->pending + nframes;
		const size_t nfft = (have - s->n) / s->hop + 1;
		const size_t first = (nfft - 1) * s->hop - s->pending;  /* where that frame starts in THIS block */
		const size_t keep = nframes - first;                    /* = n + what follows the frame's hop ... */
		const size_t rest = have - nfft * s->hop;               /* ... of which this much belongs to the next frame */
		/* WHERE it is kept: in page-locked HOST memory, brought there by the DMA engine.  A device-to-device copy is a copy
		 * KERNEL on this runtime, and a kernel queued beside an open launch is trouble: dispatched while the launch still
		 * fills the chip it holds workgroup slots the launch's own workgroups wait for, and its waves, behind spinning waves
		 * of a higher priority, may never finish -- the launch then runs into its deadline (measured: tools/scratch/fe_loop.py,
		 * 12 of 12 at the device stream's priority, 2 of 8 at the lowest).  Copies of 16 KB or less are kernels too
		 * (GPU_FORCE_BLIT_COPY_SIZE), so at least 4096 frames (32 KB) of the block's end travel. */
		const size_t MINF = 4096;
		const size_t copyf = keep >= MINF ? keep : (nframes >= MINF ? MINF : nframes);
		const size_t off = copyf - keep;
		if (copyf * 2 * sizeof(float) <= 16384u) {
			/* (a block of under 2 K frames: nothing the DMA engine would copy -- the ordinary way, which closes the launch) */
			if (int rc = spectrum_resolve(s))
				return rc;
			goto eager;
		}
		if (copyf > s->keep_cap) {
			if (s->keep_host) {
				if (d->lazy_stream)
					HIP_TRY(hipStreamSynchronize(d->lazy_stream));
				(void)hipHostFree(s->keep_host);
				s->keep_host = nullptr;
				s->keep_cap = 0;
			}
			HIP_TRY(hipHostMalloc((void **)&s->keep_host, (copyf + s->n) * 2 * sizeof(float), hipHostMallocDefault));
			s->keep_cap = copyf + s->n;
		}
		{
			std::lock_guard<std::mutex> up_guard(*d->upload_lock);
			if (!d->lazy_stream) {
				int prio_low = 0, prio_high = 0;
				HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
				HIP_TRY(hipStreamCreateWithPriority(&d->lazy_stream, hipStreamNonBlocking, prio_low));
			}
			if (!s->def_ev)
				HIP_TRY(hipEventCreateWithFlags(&s->def_ev, hipEventDisableTiming));
			if (d->up_stream) {
				/* (the block may itself be on its way on the upload stream: behind it) */
				HIP_TRY(hipEventRecord(s->def_ev, d->up_stream));
				HIP_TRY(hipStreamWaitEvent(d->lazy_stream, s->def_ev, 0));
			}
			HIP_TRY(hipMemcpyAsync(s->keep_host, iq + 2 * (first - off), copyf * 2 * sizeof(float), hipMemcpyDeviceToHost, d->lazy_stream));
			HIP_TRY(hipEventRecord(s->def_ev, d->lazy_stream));
		}
		s->def_off = off;
		s->def_keep = keep;
		s->frames_done += nfft;
		s->def_rest = rest;
		s->pending = rest;                                  /* (logically; physically at stage + 2 * hop until resolved) */
		s->deferred = true;
		++s->deferred_pushes;
		return WR_OK;
	}
eager:
	if (s->deferred) {
		if (inside) {
			/* this block's newest frame supersedes the deferred one, and reads nothing carried over: drop it */
			s->deferred = false;
		} else if (int rc = spectrum_resolve(s)) {
			return rc;
		}
	}
	DEV_SETTLE(d);
	hipStream_t st = d->stream;
	if (inside) {
		/* A block that already lies in device memory and whose most recent complete frame starts INSIDE it (any block of
		 * fftSize + hop frames or more; whatever was carried over belongs to frames nobody can observe,
		 * spectrumsink.cxx:114-116,136-141): that frame is transformed where it lies and only the tail that belongs to
		 * the NEXT frame is kept -- not the whole block copied into the stage first (32 MB device to device per 4 M-frame
		 * block, and three more enqueues on the host).  Nothing before that frame is read: the caller may have staged the
		 * block's tail only (r04: the host runtime's stagedTail). */
		const size_t have = s->pending + nframes;
		const size_t nfft = (have - s->n) / s->hop + 1;
		const size_t first = (nfft - 1) * s->hop - s->pending;  /* where that frame starts in THIS block */
		HIP_TRY(wrk_fft_frames(st, s->plan, iq + 2 * first, s->hop, 1, s->bins, nullptr));
		s->frames_done += nfft;
		const size_t rest = have - nfft * s->hop;
		if (rest) {
			if (rest > s->stage_cap) {
				float *nb = nullptr;
				const size_t cap = rest + s->n;
				HIP_TRY(hipMalloc((void **)&nb, cap * 2 * sizeof(float)));
				HIP_TRY(dev_stream_sync(s->dev));
				if (s->stage)
					HIP_TRY(hipFree(s->stage));
				s->stage = nb;
				s->stage_cap = cap;
			}
			HIP_TRY(hipMemcpyAsync(s->stage, iq + 2 * (first + s->hop), rest * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
		}
		s->pending = rest;
		return WR_OK;
	}
	const size_t have = s->pending + nframes;
	if (have > s->stage_cap) {
		float *nb = nullptr;
		size_t cap = have + s->n;
		HIP_TRY(hipMalloc((void **)&nb, cap * 2 * sizeof(float)));
		if (s->pending)
			HIP_TRY(hipMemcpyAsync(nb, s->stage, s->pending * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
		HIP_TRY(dev_stream_sync(s->dev));
		if (s->stage)
			HIP_TRY(hipFree(s->stage));
		s->stage = nb;
		s->stage_cap = cap;
	}
	if (nframes)
		HIP_TRY(hipMemcpyAsync(s->stage + 2 * s->pending, iq, nframes * 2 * sizeof(float),
		                       where == WR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
	/* frames start every `hop`; the reference transforms each one but only the most
	 * recent is observable through getSpectrum (spectrumsink.cxx:114-116,136-141) */
	size_t nfft = 0;
	if (have >= s->n)
		nfft = (have - s->n) / s->hop + 1;
	if (nfft) {
		const float *last = s->stage + 2 * (nfft - 1) * s->hop;
		HIP_TRY(wrk_fft_frames(st, s->plan, last, s->hop, 1, s->bins, nullptr));
		s->frames_done += nfft;
		const size_t consumed = nfft * s->hop;
		const size_t rest = have - consumed;
		if (rest) {
			/* compact the tail to the front (ranges may overlap: go through the work area) */
			if (rest <= consumed) {
				HIP_TRY(hipMemcpyAsync(s->stage, s->stage + 2 * consumed, rest * 2 * sizeof(float),
				                       hipMemcpyDeviceToDevice, st));
			} else {
				SCRATCH_GUARD(d);
				int rc = dev_scratch(d, rest * 2);
				if (rc)
					return rc;
				HIP_TRY(hipMemcpyAsync(d->scratch, s->stage + 2 * consumed, rest * 2 * sizeof(float),
				                       hipMemcpyDeviceToDevice, st));
				HIP_TRY(hipMemcpyAsync(s->stage, d->scratch, rest * 2 * sizeof(float),
				                       hipMemcpyDeviceToDevice, st));
			}
		}
		s->pending = rest;
	} else {
		s->pending = have;
	}
	if (where == WR_HOST)
		HIP_TRY(dev_stream_sync(s->dev));
	return WR_OK;
}

extern "C" int wr_spectrum_lazy_info(wr_spectrum *s, unsigned long long *deferred_pushes, unsigned long long *resolves)
{
	if (!s)
		return fail(WR_ERR_ARG, "spectrum is NULL");
	if (deferred_pushes)
		*deferred_pushes = s->deferred_pushes;
	if (resolves)
		*resolves = s->resolves;
	return WR_OK;
}

extern "C" int wr_spectrum_get_bins(wr_spectrum *s, float *bins_host)
{
	if (!s || !bins_host)
		return fail(WR_ERR_ARG, "wr_spectrum_get_bins: bad argument");
	if (!s->frames_done)
		return fail(WR_ERR_STATE, "no complete frame yet");
	if (dev_bind(s->dev))
		return WR_ERR_HIP;
	if (int rc = spectrum_resolve(s))
		return rc;
	HIP_TRY(hipMemcpyAsync(bins_host, s->bins, (size_t)s->n * 2 * sizeof(float), hipMemcpyDeviceToHost,
	                       s->dev->stream));
	HIP_TRY(dev_stream_sync(s->dev));
	return WR_OK;
}

extern "C" int wr_spectrum_get_db(wr_spectrum *s, float *magnitudes_host)
{
	if (!s || !magnitudes_host)
		return fail(WR_ERR_ARG, "wr_spectrum_get_db: bad argument");
	if (!s->frames_done)
		return fail(WR_ERR_STATE, "no complete frame yet");
	wr_dev *d = s->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	if (int rc = spectrum_resolve(s))
		return rc;
	SCRATCH_GUARD(d);
	int rc = dev_scratch(d, s->n);
	if (rc)
		return rc;
	/* dB + fftshift of the stored bins */
	HIP_TRY(wrk_bins_to_db(d->stream, s->bins, s->n, d->scratch));
	HIP_TRY(hipMemcpyAsync(magnitudes_host, d->scratch, (size_t)s->n * sizeof(float), hipMemcpyDeviceToHost,
	                       d->stream));
	HIP_TRY(dev_stream_sync(d));
	return WR_OK;
}

extern "C" int wr_spectrum_get_waterfall_row(wr_spectrum *s, unsigned int width, int hold, float *db_row_host,
                                             uint8_t *palette_host)
{
	if (!s || !width || width > s->n || s->n % width)
		return fail(WR_ERR_ARG, "wr_spectrum_get_waterfall_row: width must divide fft_size");
	if (!s->frames_done)
		return fail(WR_ERR_STATE, "no complete frame yet");
	wr_dev *d = s->dev;
	if (dev_bind(d))
		return WR_ERR_HIP;
	if (int rc = spectrum_resolve(s))
		return rc;
	SCRATCH_GUARD(d);
	int rc = dev_scratch(d, (size_t)width * 2);
	if (rc)
		return rc;
	float *db_dev = d->scratch;
	uint8_t *pal_dev = (uint8_t *)(d->scratch + width);
	HIP_TRY(wrk_waterfall_row(d->stream, s->bins, s->n, width, hold, db_dev, pal_dev));
	if (db_row_host)
		HIP_TRY(hipMemcpyAsync(db_row_host, db_dev, (size_t)width * sizeof(float), hipMemcpyDeviceToHost, d->stream));
	if (palette_host)
		HIP_TRY(hipMemcpyAsync(palette_host, pal_dev, width, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(dev_stream_sync(d));
	return WR_OK;
}

extern "C" int wr_spectrum_frames_done(wr_spectrum *s, unsigned long *frames)
{
	if (!s || !frames)
		return fail(WR_ERR_ARG, "wr_spectrum_frames_done: bad argument");
	*frames = s->frames_done;
	return WR_OK;
}

extern "C" int wr_spectrum_batch_db(wr_spectrum *s, const float *iq_dev, size_t nframes_fft, float *db_dev)
{
	if (!s || (nframes_fft && (!iq_dev || !db_dev)))
		return fail(WR_ERR_ARG, "wr_spectrum_batch_db: bad argument");
	if (dev_bind(s->dev))
		return WR_ERR_HIP;
	DEV_SETTLE(s->dev);
	/* the two-pass transforms keep their intermediate in `work`: room for the whole batch, up
	 * to 128 MB (it then still sits in the 256 MB Infinity Cache between the passes), means one
	 * pair of launches per call instead of one per 64 frames */
	WrFftPlan &p = s->plan;
	if (p.n2 != 1 && p.work_frames < nframes_fft) {
		const size_t frame_bytes = (size_t)p.n * 2 * sizeof(float);
		size_t want = ((size_t)128 << 20) / frame_bytes;
		if (want < 1)
			want = 1;
		if (want > nframes_fft)
			want = nframes_fft;
		if (want > p.work_frames) {
			HIP_TRY(dev_stream_sync(s->dev));
			(void)hipFree(p.work);
			p.work = nullptr;
			p.work_frames = 0;
			HIP_TRY(hipMalloc((void **)&p.work, want * frame_bytes));
			p.work_frames = want;
		}
	}
	HIP_TRY(wrk_fft_frames(s->dev->stream, s->plan, iq_dev, s->hop, nframes_fft, nullptr, db_dev));
	return WR_OK;
}
